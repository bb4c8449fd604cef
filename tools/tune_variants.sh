# Build-knob A/B harness for the kernels (DFX_NT, DFX_ZB, DFX_WAVES, DFX_MINWAVES, DFX_ADAM_ZB, DFX_BCAST_INFLIGHT).
#
#   here (no GPU):   bash tools/tune_variants.sh build zb1:-DDFX_ZB=1 zb2:-DDFX_ZB=2 zb4:-DDFX_ZB=4
#   on the GPU box:  gpurun -- 'bash tools/tune_variants.sh run zb1 zb2 zb4 -- --zc 8,4 --steps 64'
#
# "build" compiles build/libdfx_<tag>.so for every tag:flags pair (build/ is git-ignored but travels to the box);
# "run" times sweep / bcast / project / adam of each variant with tools/tune_kernels.py (arguments after --) and
# appends one line per configuration to gpurun_out/tune_variants.txt.
set -e
mode=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build gpurun_out
if [ "$mode" = build ]; then
  for spec in "$@"; do
    tag=${spec%%:*}; flags=${spec#*:}
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC $flags -o build/libdfx_$tag.so datafusion_amd/csrc/dfx_kernels.hip
    echo "built build/libdfx_$tag.so ($flags)"
  done
elif [ "$mode" = run ]; then
  tags=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
  [ "$1" = "--" ] && shift
  for tag in "${tags[@]}"; do
    DFX_LIB=$PWD/build/libdfx_$tag.so timeout -k 10 280 python -u tools/tune_kernels.py "$@" >> gpurun_out/tune_variants.txt 2>&1
  done
  grep "^libdfx" gpurun_out/tune_variants.txt | tail -n 40
else
  echo "usage: tune_variants.sh build tag:flags... | run tag... -- <tune_kernels.py args>"; exit 2
fi

# The per-GPU slab of BASELINE configs[2] at G = 8 (t512 x l32 x z4 x 256 x 256) through a ONE-rank RCCL group, sweep cut
# into 1 / 3 / 6 launches: what slicing and the issued collectives cost a rank before any real exchange
# (profiles/r03_c3g8_one_rank_rccl_slices.txt).  Usage on a GPU box: bash tools/bench_c3g8_slices.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/c3g8_slices.txt
: > $OUT
echo "## unsharded: python bench.py --workload c3g8 --single --no-cpu --steps 30" >> $OUT
timeout -k 10 200 python bench.py --workload c3g8 --single --no-cpu --steps 30 2>gpurun_out/c3g8_slices.err | python tools/_bench_line.py >> $OUT
for K in 1 3 6 auto; do
  echo "## DFX_FORCE_COMM=1 python bench.py --gpus 1 --workload c3g8 --single --no-cpu --steps 30 --comm-slices $K" >> $OUT
  DFX_FORCE_COMM=1 timeout -k 10 200 python bench.py --gpus 1 --workload c3g8 --single --no-cpu --steps 30 --comm-slices $K 2>>gpurun_out/c3g8_slices.err | python tools/_bench_line.py >> $OUT
done
cat $OUT

# Other geometries / methods on one GPU, one JSON line each -> gpurun_out/bench_matrix_<tag>.txt.  bash tools/bench_matrix.sh <tag>
TAG=${1:-x}; OUT=gpurun_out/bench_matrix_$TAG.txt; : > $OUT
run() { echo "## bench.py $*" >> $OUT; timeout -k 10 400 python3 bench.py --no-cpu "$@" >> $OUT 2>> gpurun_out/bench_matrix_$TAG.err || echo "FAILED rc=$?" >> $OUT; }
run --workload c2r2 && run --workload c2r8 && run --workload cells && run --workload cells --method adam && run --workload cells --method adam-autograd --steps 20 &&
run --workload c3g8 --steps 30 && run --workload c4g8 --steps 30 && run --workload c3z16 --steps 10 --warmup 2 && run --workload c3 --single --steps 10 --warmup 2 && run --workload c4 --single --steps 10 --warmup 2 &&
run --method adam --moment1 explicit
python3 - $OUT <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("##"): print(ln.strip()); continue
    if not ln.startswith("{"): print(ln.strip()); continue
    d = json.loads(ln); o = d.get("other_residual_mode") or {}
    r = d.get("roofline") or {}
    print("   it/s %.1f  ms %.3f  frac %s  %s %s" % (d["value"], d["ms_per_step"], ("%.3f" % r["frac"]) if r else "-", ("| explicit %.1f it/s frac %.3f" % (o["value"], o["roofline"]["frac"])) if o else "", d["config"].get("placement_probe_GBps") or ""))
PY

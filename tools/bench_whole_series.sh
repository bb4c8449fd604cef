# configs[2] / configs[3] whole on one GPU, process after process (a series of well over 50 s of allocate / solve /
# free on one box): the sweep's fraction of the HBM peak must not drift with the history of the device memory.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/whole_series.txt; : > $OUT
for W in c3 c4 c3 c4 c3 c3; do
  echo "## bench.py --workload $W --single --no-cpu --steps 10 --warmup 2  (t = $(date +%s))" >> $OUT
  timeout -k 10 300 python3 bench.py --workload $W --single --no-cpu --steps 10 --warmup 2 >> $OUT 2>> gpurun_out/whole_series.err || echo "FAILED rc=$?" >> $OUT
done
python3 - $OUT <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if not ln.startswith("{"): print(ln.strip()); continue
    d = json.loads(ln); r = d["roofline"]; p = d["config"].get("placement") or {}
    print("   it/s %.2f  ms/it %.2f (blocks %s)  sweep %.2f ms  frac %.3f  placement: conflicts %s, set-up %.1f s, pair sweep %.0f GB/s" % (
        d["value"], d["ms_per_step"], ", ".join("%.2f" % v for v in d["timed_blocks"]["ms_per_step"]), r["avg_launch_ms"], r["frac"],
        p.get("positions_sharing_a_class"), p.get("seconds", 0), p.get("pair_sweep_GBps", 0)))
PY

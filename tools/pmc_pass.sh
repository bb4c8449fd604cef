# Two separate PMC passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2: they do not fit together).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --no-cpu --steps 4 --warmup 1 > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            agg[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if max(v) > 1000:
            print(c, k, "n=%d" % len(v), "avg=%.1f" % (sum(v) / len(v)), "max=%.1f" % max(v))
PY

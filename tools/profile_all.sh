# rocprofv3 kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, they do not fit one pass)
# for the default bench (C2 CG, both residual modes) and for the fused Adam bench.  Usage: bash tools/profile_all.sh <tag>
# Writes gpurun_out/prof_<tag>_*; copy the summaries to profiles/ (see the end of this script for the file names).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for M in cg adam; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$M -- python3 bench.py --method $M --no-cpu --steps 20 > gpurun_out/prof_${TAG}_${M}_bench.json 2> gpurun_out/prof_${TAG}_$M.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${M}_$c -- python3 bench.py --method $M --no-cpu --steps 4 --warmup 1 > gpurun_out/pmc_${TAG}_${M}_$c.json 2> gpurun_out/pmc_${TAG}_${M}_$c.err
  done
  f=$(find gpurun_out/prof_${TAG}_$M -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/prof_${TAG}_${M}_kernel_stats.csv
done
python3 - $TAG <<'PY'
import csv, glob, collections, hashlib, json, re, sys
tag = sys.argv[1]
res = {}
for m in ("cg", "adam"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmc_{tag}_{m}_{c}/**/*counter_collection.csv", recursive=True)
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f[0])):
            if row["Counter_Name"] == c:
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            if max(v) > 100000:      # KiB: the 5-D kernels only
                short = k.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in k else k[:50]
                # steady state: drop the first launch of each kernel (cold caches, page faults of fresh allocations)
                vv = v[1:] if len(v) > 2 else v
                print(m, c, short, "n=%d" % len(v), "avg_KiB=%.1f" % (sum(vv) / len(vv)))
                res.setdefault(short, {})[c] = sum(vv) / len(vv)
MODES = {"0": "PROJECT", "1": "XP", "2": "P", "3": "PINIT"}
def canon(k):
    """k_sweep_fast<1, 8, 4, true> -> k_sweep_fast<XP,8,r4,factored> (the key bench.py looks up)."""
    m = re.match(r"k_sweep_fast<(\d), (\d+), (\d+), (true|false)>", k)
    if m:
        return f"k_sweep_fast<{MODES[m.group(1)]},{m.group(2)},r{m.group(3)},{'factored' if m.group(4) == 'true' else 'explicit'}>"
    return k
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[canon(k)] = {"FETCH_SIZE_KiB": d["FETCH_SIZE"], "WRITE_SIZE_KiB": d["WRITE_SIZE"],
                         "traffic_bytes": int(d["FETCH_SIZE"] * 2 * 1024 + d["WRITE_SIZE"] * 1024)}
N = 2 ** 30
doc = {"workload": "c2",
       "kernel_source_sha256": hashlib.sha256(open("datafusion_amd/csrc/dfx_kernels.hip", "rb").read()).hexdigest(),
       "source": f"tools/profile_all.sh {tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over "
                 "`python3 bench.py --no-cpu --steps 4 --warmup 1` (CG: both residual modes run in that command) and "
                 "`... --method adam`; per-launch averages without each kernel's first launch",
       "correction": "MI355X_MICROARCH.md HBM section: counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of a 16 B/lane "
                     "coalesced streaming read -> doubled; WRITE_SIZE exact",
       "kernels": out,
       "algorithmic_bytes": {f"k_sweep_fast<XP,{zc},r4,{res}>": w * N for zc in (4, 8) for res, w in (("factored", 16), ("explicit", 20))}}
doc["algorithmic_bytes"].update({f"k_bcast_fast<{zc}, 4, true, false>": 8 * N for zc in (4, 8)})
doc["algorithmic_bytes"].update({f"k_adam_fast<{zc}, 4, false, true, false>": 16 * N for zc in (4, 8)})
for k, b in doc["algorithmic_bytes"].items():
    if k in out:
        out[k]["traffic_over_algorithmic"] = out[k]["traffic_bytes"] / b
json.dump(doc, open(f"gpurun_out/pmc_{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
for M in cg adam; do head -12 gpurun_out/prof_${TAG}_${M}_kernel_stats.csv | cut -c1-160; done
# -> profiles/<tag>_c2_kernel_stats.csv, <tag>_c2_adam_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_c2_bench_under_rocprof.json

# Investigate the bimodal sweep time on the nZ=4 slabs: same command in several processes under PMC.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPE=${1:-256,16,4,512,512}
rm -rf gpurun_out/slow; mkdir -p gpurun_out/slow
for i in 1 2 3 4 5; do
  timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/slow/p$i -- python3 tools/tune_kernels.py --shape $SHAPE --zc 4 --steps 64 --rounds 2 > gpurun_out/slow/p$i.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for i in range(1, 6):
    f = glob.glob(f"gpurun_out/slow/p{i}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_sweep_fast<1" in k or "k_bcast_fast" in k:
            agg[k[:45]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "Start_Timestamp" in row:
                dur[k[:45]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for k, d in agg.items():
        print(f"p{i}", k, "ms=%.2f" % (sorted(dur[k])[len(dur[k]) // 2] if dur[k] else -1),
              " ".join(f"{c.replace('_sum','')}={sum(v)/len(v):.3g}" for c, v in sorted(d.items())))
PY

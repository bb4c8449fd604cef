"""Summarise the passes of tools/placement_counters.sh: per offset of p (regime), per counter, the mean over the
launches of k_sweep_fast, and the ratio slow / fast.  Per-instance counters (pass p5, JSON) are listed per TCC
channel when the JSON carries dimensions."""
import collections
import csv
import glob
import json
import sys

out, offs = sys.argv[1], sys.argv[2].split(",")
reps = 3


def sweeps(rows):
    """rows of one csv -> dispatch id -> {counter: value}, duration; only k_sweep_fast, in dispatch order."""
    d = collections.OrderedDict()
    for r in rows:
        if "k_sweep_fast" not in r["Kernel_Name"]:
            continue
        e = d.setdefault(int(r["Dispatch_Id"]), {"_ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [d[k] for k in sorted(d)]


for p in sorted(glob.glob(f"{out}/p*/")):
    fs = glob.glob(p + "**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counter csv")
        continue
    sw = sweeps(csv.DictReader(open(fs[0])))
    if len(sw) != reps * len(offs):
        print(p, f"unexpected number of sweep dispatches: {len(sw)}")
    per = {o: sw[i * reps:(i + 1) * reps] for i, o in enumerate(offs)}
    names = sorted({c for e in sw for c in e})
    print(f"== {p.rstrip('/').split('/')[-1]}   (mean of {reps} launches; p at +{' / +'.join(offs)} GiB; last column = first / second)")
    for c in names:
        m = [sum(e.get(c, 0.0) for e in per[o]) / max(len(per[o]), 1) for o in offs]
        ratio = m[0] / m[1] if len(m) > 1 and m[1] else float("nan")
        print(f"  {c:44s} " + " ".join(f"{v:14.6g}" for v in m) + f"   {ratio:6.3f}")

# per-instance pass
for f in glob.glob(f"{out}/p5/**/*results.json", recursive=True):
    try:
        js = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print("p5 json unreadable:", e)
        continue
    tool = js["rocprofiler-sdk-tool"][0]
    print("p5 json keys:", list(tool.keys()))
    cc = tool.get("callback_records", {}).get("counter_collection") or tool.get("counter_collection") or []
    print("p5 counter_collection records:", len(cc))
    if cc:
        print("p5 first record:", json.dumps(cc[0])[:1500])

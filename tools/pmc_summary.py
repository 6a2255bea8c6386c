"""Summarise rocprofv3 --pmc counter_collection CSVs per nf:: kernel (mean over the full-size launches).
    python tools/pmc_summary.py <dir> [<dir> ...] [--min-ms 1.0] [--out profiles/x.json] [--note "..."]"""
import argparse, csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csv.field_size_limit(1 << 30)
ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--min-ms", type=float, default=1.0)
ap.add_argument("--out", default=None)
ap.add_argument("--note", default="")
a = ap.parse_args()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = {}
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "nf::" not in name:
                continue
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
            if dur < a.min_ms:
                continue
            short = name.split("(")[0].replace("void ", "")
            key = (f, row["Dispatch_Id"], row["Counter_Name"])
            if key in seen:
                continue
            seen[key] = 1
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[short]["_ms"].append(dur)
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["_launches"] = len(cs["_ms"])
import bench
res = {"kernel_src_sha": bench.kernel_src_sha(), "note": a.note, "kernels": out}
txt = json.dumps(res, indent=1)
print(txt)
if a.out:
    open(a.out, "w").write(txt)

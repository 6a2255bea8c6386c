"""Condense the counter passes of tools/collect_train_pmc.sh into profiles/r03_train_pmc.csv: one row per (workload, kernel): launches,
mean duration, FETCH_SIZE / WRITE_SIZE bytes per launch (gfx950 corrections of MI355X_MICROARCH.md), matrix-pipe busy fraction, clock."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "r03ptrain")
csv.field_size_limit(1 << 30)


def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(P, d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "nf::" not in r["Kernel_Name"]:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            if key in seen:
                continue
            seen.add(key)
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == list(acc[k].keys())[0]:
                acc[k]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    return acc


rows = []
for tag, pre in (("32^4 2 spline layers B=16", ""), ("16^3 8 spline layers B=256", "c3")):
    fe, wr, sq = collect(pre + "FETCH_SIZE"), collect(pre + "WRITE_SIZE"), collect(pre + "sq")
    for k in sorted(set(fe) | set(wr) | set(sq)):
        mean = lambda v: sum(v) / len(v) if v else 0.0
        ms = fe.get(k, sq.get(k, {})).get("_ms", [])
        s = sq.get(k, {})
        cyc = mean(s.get("GRBM_GUI_ACTIVE", [])) / 8.0
        rows.append([tag, k, len(ms), round(mean(ms), 4), round(mean(fe.get(k, {}).get("FETCH_SIZE", [])) * 2048 / 1e6, 2),
                     round(mean(wr.get(k, {}).get("WRITE_SIZE", [])) * 1024 / 1e6, 2),
                     round(mean(s.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / 1024.0 / cyc, 3) if cyc else "",
                     round(cyc / (mean(s.get("_ms", [])) * 1e6), 2) if cyc and s.get("_ms") else "",
                     int(mean(s.get("SQ_INSTS_VALU", []))), int(mean(s.get("SQ_INSTS_SALU", []))), int(mean(s.get("SQ_INSTS_LDS", [])))])
rows.sort(key=lambda r: (r[0], -r[2] * r[3]))
out = os.path.join(ROOT, "profiles", "r03_train_pmc.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["workload", "kernel", "launches", "mean_ms", "fetch_MB_per_launch", "write_MB_per_launch", "mfma_busy_all_simds", "clock_GHz",
                "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"])
    w.writerows(rows)
print(out, len(rows), "rows")
for r in rows[:14]:
    print(r)

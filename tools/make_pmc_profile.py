"""Turn rocprofv3 --pmc counter_collection CSVs (separate FETCH_SIZE / WRITE_SIZE / SQ passes over tools/kbench.py) into the
profile JSON bench.py reads its `roofline.traffic` from.  FETCH_SIZE is in KiB-ish units of 64-B requests tallied at half
their size on gfx950 (MI355X_MICROARCH.md, HBM): bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is exact for 16-B stores.
    python tools/make_pmc_profile.py --fetch DIR --write DIR [--sq DIR] --slab 256 --k2-slab 33 --out profiles/r02_pmc_kernels.json"""
import argparse, csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
csv.field_size_limit(1 << 30)
ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--sq", default=None)
ap.add_argument("--slab", type=int, default=256)
ap.add_argument("--k2-slab", type=int, default=33)
ap.add_argument("--min-ms", type=float, default=0.3)
ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
ap.add_argument("--lattice", default="32,32,32,32")
a = ap.parse_args()


def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "nf::" not in name:
                continue
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
            if dur < a.min_ms:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            if key in seen:
                continue
            seen.add(key)
            short = name.split("(")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == list(acc[short].keys())[0]:
                acc[short]["_ms"].append(dur)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_launches": len(cs["_ms"])} for k, cs in acc.items()}


fetch, write = collect(a.fetch), collect(a.write)
sq = collect(a.sq) if a.sq else {}
kernels = {}
for k in sorted(set(fetch) | set(write)):
    fb = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    wb = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    e = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
         "launch_ms_under_profiler": fetch.get(k, write.get(k, {})).get("_ms"),
         "slab_batch": a.k2_slab if "rqs_kernel" in k else a.slab}
    if k in sq:
        s = sq[k]
        e["counters"] = {c: v for c, v in s.items() if not c.startswith("_")}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s:
            cycles = s["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma_busy_fraction_of_all_simds"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles
            e["clock_ghz_under_profiler"] = cycles / (s["_ms"] * 1e6)
        if "SQ_LDS_IDX_ACTIVE" in s and "GRBM_GUI_ACTIVE" in s:
            e["lds_busy_fraction"] = s["SQ_LDS_IDX_ACTIVE"] / 256.0 / (s["GRBM_GUI_ACTIVE"] / 8.0)
    kernels[k] = e
res = {"kernel_src_sha": bench.kernel_src_sha(), "lattice": [int(t) for t in a.lattice.split(",")], "knots": 16, "slab_batch": a.slab,
       "collection": "rocprofv3 --pmc <one pass per counter set> --kernel-trace --output-format csv -- python3 tools/kbench.py; "
                     "FETCH_SIZE x 1024 x 2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE x 1024; means over the full-size launches",
       "note": a.note, "kernels": kernels}
open(a.out, "w").write(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))

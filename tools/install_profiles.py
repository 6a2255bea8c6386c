"""Copy what tools/collect_profiles.sh left under gpurun_out/r02p into profiles/ (tracked): the bench lines, the
rocprofv3 kernel stats of the bench command, the PMC summary bench.py reads its roofline.traffic from, and the three
counter passes condensed to one row per (dispatch, counter) of the nf:: kernels.
    python tools/install_profiles.py [--tag r02] [--src gpurun_out/r02p]"""
import argparse, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="r03")
ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "r03p"))
a = ap.parse_args()
P, O, T = a.src, os.path.join(ROOT, "profiles"), a.tag
csv.field_size_limit(1 << 30)


def condense(d, out):
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "nf::" not in r["Kernel_Name"]:
                continue
            key = (f, r["Dispatch_Id"], r["Counter_Name"])
            if key in seen:
                continue
            seen.add(key)
            rows.append([r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"], r["Workgroup_Size"],
                         r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), r["Counter_Name"], r["Counter_Value"],
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "VGPR_Count", "LDS_Block_Size", "Counter_Name", "Counter_Value", "Duration_ns"])
        w.writerows(rows)


condense(os.path.join(P, "FETCH_SIZE"), os.path.join(O, f"{T}_pmc_fetch_size.csv"))
condense(os.path.join(P, "WRITE_SIZE"), os.path.join(O, f"{T}_pmc_write_size.csv"))
condense(os.path.join(P, "sq"), os.path.join(O, f"{T}_pmc_sq.csv"))
pmc_name = "pmc_kernels.json" if os.path.exists(os.path.join(P, "pmc_kernels.json")) else "r02_pmc_kernels.json"
for src, dst in ((pmc_name, f"{T}_pmc_kernels.json"), ("trace/run_kernel_stats.csv", f"{T}_bench_kernel_stats.csv"),
                 ("bench_under_rocprof.json", f"{T}_bench_under_rocprof.json"), ("bench_batch128.json", f"{T}_bench_batch128.json"),
                 ("bench.json", f"{T}_bench.json"), ("config_bench.txt", f"{T}_config_bench.txt"), ("c3_bench.txt", f"{T}_c3_bench.txt"),
                 ("c3trace/run_kernel_stats.csv", f"{T}_c3_kernel_stats.csv"), ("train_bench.txt", f"{T}_train_bench.txt"),
                 ("traintrace/run_kernel_stats.csv", f"{T}_train_kernel_stats.csv"),
                 ("trainc3/run_kernel_stats.csv", f"{T}_train_c3_kernel_stats.csv")):
    if os.path.exists(os.path.join(P, src)):
        shutil.copy(os.path.join(P, src), os.path.join(O, dst))
for d, name in (("c3sq", "c3_pmc_sq"), ("c3FETCH_SIZE", "c3_pmc_fetch_size"), ("c3WRITE_SIZE", "c3_pmc_write_size")):
    if os.path.isdir(os.path.join(P, d)):
        condense(os.path.join(P, d), os.path.join(O, f"{T}_{name}.csv"))
line = json.loads(open(os.path.join(P, "bench.json")).read().strip().splitlines()[-1])
prof = json.load(open(os.path.join(P, pmc_name)))
print("bench", round(line["value"], 1), "configs/s; frac", round(line["roofline"]["frac"], 3), "; sources", line["kernel_src_sha"],
      "now", bench.kernel_src_sha(), "profile", prof["kernel_src_sha"])

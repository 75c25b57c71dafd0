"""gpurun_out/<run>/ (rocprofv3 csv output of tools/profile_round.sh, tools/pmc_cfg4.sh) -> profiles/<name>/:
kernel_stats.csv (the --stats summary as is) and pmc_counters.csv (per kernel and counter: launches, mean, min, max
of the per-dispatch values, summed over the counter's dimensions).   python tools/summarize_profile.py <src> <dst>"""
import collections, csv, glob, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
stats = os.path.join(src, "stats_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
agg = collections.OrderedDict()
for path in sorted(glob.glob(os.path.join(src, "*_counter_collection.csv"))):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[key] = (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"],
                      r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
    for key, v in per_dispatch.items():
        agg.setdefault((os.path.basename(path).split("_")[0], names[key], key[1]), []).append(v)
with open(os.path.join(dst, "pmc_counters.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["pass", "kernel", "grid", "workgroup", "lds_bytes", "scratch", "vgpr", "agpr", "sgpr", "counter", "launches", "mean",
                "min", "max"])
    for (pas, nm, ctr), vals in agg.items():
        if nm[0].startswith("__amd_rocclr"):
            continue
        w.writerow([pas, *nm, ctr, len(vals), f"{sum(vals)/len(vals):.6g}", f"{min(vals):.6g}", f"{max(vals):.6g}"])
print(open(os.path.join(dst, "pmc_counters.csv")).read())

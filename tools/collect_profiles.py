#!/usr/bin/env python3
"""Copy the rocprofv3 / bench outputs of one measurement session from gpurun_out/ into profiles/
(committed) and derive profiles/traffic.json.  Usage: collect_profiles.py <stats_dir> <pmc_prefix> [round]"""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):     # gpurun merges every session's files into gpurun_out/: take the latest
    return max(glob.glob(pattern), key=os.path.getmtime)


stats, pmc = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r01"
shutil.copy(newest("gpurun_out/%s/*/*_kernel_stats.csv" % stats),
            "profiles/%s_rocprof_kernel_stats_ne120x72x30.csv" % rnd)
shutil.copy("gpurun_out/bench_%s.json" % rnd, "profiles/%s_bench_ne120x72x30.json" % rnd)
for tag, name in (("generic", "generic_sweeps"), ("paired", "paired_sweeps"), ("twopass", "class_two_pass"),
                  ("classsums", "class_sum_form")):
    try:
        shutil.copy("gpurun_out/bench_%s_%s.json" % (rnd, tag), "profiles/%s_bench_ne120x72x30_%s.json" % (rnd, name))
    except FileNotFoundError:
        pass
out, tot = [], {}
for name in ("fetch", "write", "sq"):
    f = newest("gpurun_out/%s_%s/*/*_counter_collection.csv" % (pmc, name))
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        for c, vals in v.items():
            out.append((k, c, len(vals), sum(vals) / len(vals)))
            tot[(k, c)] = sum(vals) / len(vals)
with open("profiles/%s_pmc_counters_ne120x72x30.csv" % rnd, "w") as fh:
    fh.write("kernel,counter,dispatches,avg_value_per_dispatch\n")
    for r in sorted(out):
        fh.write("\"%s\",%s,%d,%.6g\n" % r)
single = any("sweep_os_kernel<double" in k or "sweep_osr_kernel<double" in k for k, _ in tot)
if single:
    e = [k for k, _ in tot if "os_contract_kernel" in k][0]
    p = [k for k, _ in tot if "sweep_os_kernel<double" in k or "sweep_osr_kernel<double" in k][0]
    # (the reference pre-pass is the same kernel on a subsample: the per-dispatch average mixes both; the sum of the two is
    #  what one step moves, so report 2 x the average)
else:
    e = [k for k, _ in tot if ("eddy" in k and "kernel<double" in k) or "flux_cls_kernel" in k][0]
    p = [k for k, _ in tot if ("project" in k and "kernel<double, 4" in k) or "sweep_op_kernel<double" in k][0]
mode = "latitude-class, single sweep" if single else "latitude-class, one pass" if "flux_cls" in e else "latitude-class" if "_cls_" in e else ("mirror-paired" if "_sym_" in e else "generic")
pm = 2 if single else 1
tr = {"workload": "ne120x72x30", "dtype": "f64", "sweeps": mode,
      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB per dispatch; FETCH_SIZE doubled per "
              "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)", "eddy_kernel": e, "project_kernel": p,
      "eddy_kernel_hbm_bytes_per_launch": int(2 * tot[(e, "FETCH_SIZE")] * 1024 + tot[(e, "WRITE_SIZE")] * 1024),
      "project_kernel_hbm_bytes_per_launch": int(pm * (2 * tot[(p, "FETCH_SIZE")] * 1024 + tot[(p, "WRITE_SIZE")] * 1024)),
      "algorithmic_bytes_per_launch": 4 * 8 * 777602 * 72 * 30}
json.dump(tr, open("profiles/traffic.json", "w"), indent=1)
print(tr)
for k in (e, p):
    busy = tot[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] / 1024
    act = tot[(k, "GRBM_GUI_ACTIVE")] / 8
    print(k, "mfma busy %.3f" % (busy / act), "cycles/XCD %.4g" % act, "MFMAs %.4g" % tot[(k, "SQ_INSTS_VALU_MFMA_MOPS_F64")])

#!/usr/bin/env python3
"""Copy the summaries of one measurement session (tools/profile_session.sh <round>, merged back under
gpurun_out/session_<round>/) into profiles/ (committed) and derive profiles/traffic.json.
   python tools/collect_profiles.py r04
Kernel tables are per (kernel, grid size): two launches of one kernel with different grids -- the reference pre-pass
and the main sweep -- are separate rows (tools/kernel_table.py)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_table  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
S = "gpurun_out/session_%s" % rnd
P = "profiles"


def cp(src, dst):
    if os.path.exists(os.path.join(S, src)):
        shutil.copy(os.path.join(S, src), os.path.join(P, dst))
        return True
    print("missing", src)
    return False


cp("bench.json", "%s_bench_ne120x72x30.json" % rnd)
for tag, name in (("classsums", "class_sum_form"), ("twopass", "class_two_pass"), ("paired", "paired_sweeps"),
                  ("exact_mirror", "exact_mirror_grid")):
    cp("bench_%s.json" % tag, "%s_bench_ne120x72x30_%s.json" % (rnd, name))
cp("tracers.log", "%s_tracers_two_per_sweep.log" % rnd)
cp("graph_probe.log", "%s_hip_graph_probe_ne30x72x1.log" % rnd)
cp("lab_d128_f32.log", "%s_lab_single_sweep_d128_f32.log" % rnd)


def write_table(kt_dir, out, title):
    rows = kernel_table.table(os.path.join(S, kt_dir))
    tot = sum(r[0] for r in rows)
    with open(os.path.join(P, out), "w") as fh:
        fh.write("# %s\n# rocprofv3 --kernel-trace, one row per (kernel, grid size); durations in microseconds\n" % title)
        fh.write("%-96s %14s %6s %10s %10s %6s\n" % ("kernel", "grid", "calls", "avg us", "min us", "%"))
        for t, n, avg, mn, name, grid in rows[:40]:
            fh.write("%-96s %14s %6d %10.1f %10.1f %6.1f\n" % (name.replace("void ", "")[:96], grid, n, avg, mn, 100 * t / tot))
    return rows


def pmc(dirname):
    """{kernel: {counter: [values per dispatch]}}"""
    f = max(glob.glob(os.path.join(S, dirname, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def long_group(vals):
    """average over the launches of the main sweep: the pre-pass launches of the same kernel are the small group"""
    groups = kernel_table.clusters(vals)
    return sum(groups[0][1]) / len(groups[0][1]), (sum(groups[1][1]) / len(groups[1][1]) if len(groups) > 1 else None)


def sweep_rows(rows, what):
    """the main sweep and the pre-pass: the two grids of the dominant kernel"""
    cand = [r for r in rows if what in r[4]]
    cand.sort(key=lambda r: -r[2])
    return cand


traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB per dispatch, averaged per (kernel, grid); "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of a streaming read at 64 B); "
                   "WRITE_SIZE as read", "shapes": {}}
jobs = [("kt_bench", "pmc_fetch", "pmc_write", "ne120x72x30", "f64", "sweep_osr_kernel<double", 777602 * 72 * 30, 8),
        ("kt_ne240x128x1_f32", "pmcf_ne240x128x1_f32", "pmcw_ne240x128x1_f32", "ne240x128x1", "f32", "sweep_os2_kernel<float", 3110402 * 128, 4),
        ("kt_ne120x72x30_f32", "pmcf_ne120x72x30_f32", "pmcw_ne120x72x30_f32", "ne120x72x30", "f32", "sweep_os2_kernel<float", 777602 * 72 * 30, 4),
        ("kt_ne30x72x91_f64", "pmcf_ne30x72x91_f64", "pmcw_ne30x72x91_f64", "ne30x72x91", "f64", "sweep_osr_kernel<double", 48602 * 72 * 91, 8)]
counters_csv = []
for kt, pf, pw, shape, dt, kern, pts, es in jobs:
    if not os.path.isdir(os.path.join(S, kt)):
        print("missing", kt)
        continue
    rows = write_table(kt, "%s_kernel_table_%s_%s.txt" % (rnd, shape, dt), "%s %s, the default path (tools/profile_session.sh)" % (shape, dt))
    sw = sweep_rows(rows, kern)
    if not sw:
        continue
    main = sw[0]
    f, w = pmc(pf), pmc(pw)
    kname = main[4].split("(")[0].replace("void ", "")
    if kname not in f:
        print("no counters for", kname)
        continue
    fmain, fpre = long_group(f[kname]["FETCH_SIZE"])
    wmain, wpre = long_group(w[kname]["WRITE_SIZE"]) if kname in w else (0.0, None)
    fetch, write = 2 * fmain * 1024, wmain * 1024
    alg = 4 * es * pts
    entry = {"dtype": dt, "kernel": kname, "grid": main[5], "avg_launch_us": main[2],
             "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch": int(fetch + write),
             "fetch_bytes": int(fetch), "write_bytes": int(write), "traffic_over_algorithmic": (fetch + write) / alg,
             "achieved_GBps_algorithmic": alg / (main[2] * 1e-6) / 1e9, "frac_of_8TBps": alg / (main[2] * 1e-6) / 8e12}
    if len(sw) > 1:
        entry["prepass"] = {"grid": sw[1][5], "avg_launch_us": sw[1][2],
                            "hbm_bytes_per_launch": int(2 * (fpre or 0.0) * 1024 + (wpre or 0.0) * 1024)}
    traffic["shapes"]["%s:%s" % (shape, dt)] = entry
    for src in (f, w):
        for k, cs in sorted(src.items()):
            for c, vals in cs.items():
                for label, g in kernel_table.clusters(vals):
                    counters_csv.append((shape + ":" + dt, k + label, len(g), c, sum(g) / len(g)))
# the headline's entries in the layout bench.py reads
h = traffic["shapes"].get("ne120x72x30:f64")
if h:
    traffic.update({"workload": "ne120x72x30", "dtype": "f64", "sweeps": "latitude-class, single sweep",
                    "project_kernel": h["kernel"], "project_kernel_hbm_bytes_per_launch": h["hbm_bytes_per_launch"],
                    "algorithmic_bytes_per_launch": h["algorithmic_bytes_per_launch"]})
if os.path.isdir(os.path.join(S, "pmc_sq")):
    sq = pmc("pmc_sq")
    for k, cs in sorted(sq.items()):
        for c, vals in cs.items():
            for label, g in kernel_table.clusters(vals):
                counters_csv.append(("ne120x72x30:f64", k + label, len(g), c, sum(g) / len(g)))
    if h and h["kernel"] in sq:
        cs = sq[h["kernel"]]
        busy, act = long_group(cs["SQ_VALU_MFMA_BUSY_CYCLES"])[0] / 1024, long_group(cs["GRBM_GUI_ACTIVE"])[0] / 8
        traffic["shapes"]["ne120x72x30:f64"]["mfma_pipe_busy"] = busy / act
        traffic["shapes"]["ne120x72x30:f64"]["mfma_f64_instructions"] = long_group(cs["SQ_INSTS_VALU_MFMA_MOPS_F64"])[0]
        traffic["shapes"]["ne120x72x30:f64"]["effective_clock_GHz"] = act / (h["avg_launch_us"] * 1e-6) / 1e9
with open(os.path.join(P, "%s_pmc_counters.csv" % rnd), "w") as fh:
    fh.write("shape,kernel,dispatches,counter,avg_value_per_dispatch\n")
    for r in counters_csv:
        fh.write("%s,\"%s\",%d,%s,%.6g\n" % r)
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic["shapes"], indent=1))

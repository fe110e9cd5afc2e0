#!/usr/bin/env python3
"""bench.py -- grid-points/s through the full TEM pipeline on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ne120x72x30] [--shard time|ncol]

A *step* is one pass of the whole hot path (theta, 7 projections, 4 native reconstructions,
eddy products, zonal apply, fused epilogue -> the ten GM16 Table-A1 outputs) over synthetic
fields already resident in HBM.  Plan build (basis + Gram + Cholesky) is timed separately.

N = 1   : the workload BASELINE.json's target is quoted on, ne120 (777602 columns) x 72 lev x 30
          snapshots, fp64, on one GPU.
N > 1   : one rank per GPU over RCCL, launched by `python -m torch.distributed.run --nproc-per-node N
          ... bench.py --gpus N`; invoked directly (`python bench.py --gpus N`) it starts that
          launcher itself as a child process, before anything touches the GPU, and returns its
          exit code.  The metric is BASELINE.json configs[3]: ONE ne120x72x30 job, its columns sharded
          over the ranks in whole latitude classes, the sums of the single sweep exchanged over RCCL/xGMI by a
          reduce-scatter over time, every rank finishing its own snapshots (all-reduces + replicated tail on plans
          that cannot run the single sweep; strong
          scaling: total work fixed).  Reported beside it under "other_workloads": configs[2],
          ne30 x 72 x 730 snapshots time-sharded (91/92 per rank at N = 8, no collective; strong
          scaling), and the weak-scaling run (every rank its own ne120x72x30 block, no collective).
          A failure of any leg is written into the record and the process exits non-zero (4; a
          stalled collective: 3).  `--shard time` makes the weak-scaling run the metric instead.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

K_HARM = 51                      # L = 50 (tem_diagnostics.py:33)
FLOPS_PER_PT = 11 * 2 * K_HARM   # 7 projections + 4 reconstructions
PEAK_F64_TFLOPS = 78.6           # MI355X FP64 matrix = vector peak (public spec)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_workload(s):
    ne, nlev, nt = s.lower().replace("ne", "").split("x")
    return int(ne), int(nlev), int(nt)


def zm_lat(dlat=1.0):
    e = np.arange(-90, 90 + dlat, dlat)
    return (e[1:] + e[:-1]) / 2


def cpu_baseline_and_parity(plan_factory, lat, lon, plev, nt_s, device, tdtype=torch.float64):
    """Time the CPU oracle (numpy port, factorised association) on a bounded sample of the same
    workload -- the same grid and levels, nt_s snapshots -- and check the GPU result on exactly
    that sample against it."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import engine, _lib
    f = engine.synth_fields(device, lat, lon, plev, nt_s, t0=0, dtype=tdtype, seed=0)     # the dtype that was timed
    host = [x.cpu().numpy() for x in f]
    t0 = time.perf_counter()
    o = orc.TEMOracle(*host, lat, plev, mode="factorised")
    ref = o.results()
    t_cpu = time.perf_counter() - t0
    plan = plan_factory()
    plan.set_tem(len(plev), nt_s, plev * 100)
    one_pass = plan.one_pass
    res, _ = plan.tem_run(*f)
    bad = plan.status()
    res = res.cpu().numpy()
    err = 0.0
    for i, n in enumerate(_lib.RESULT_NAMES):
        err = max(err, float(np.max(np.abs(res[i] - np.asarray(ref[n], np.float64))) / np.max(np.abs(ref[n]))))
    plan.close()
    pts = lat.size * len(plev) * nt_s
    try:    # threads the numpy/scipy BLAS actually used
        from threadpoolctl import threadpool_info
        cores = max([int(x.get("num_threads", 1)) for x in threadpool_info()] or [1])
    except Exception:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    return {"value": pts / t_cpu, "unit": "grid-points/s", "cores": cores, "kind": "port",
            "sample": "ne%d grid (%d cols) x %d lev x %d of the snapshots, oracle/tem_oracle.py "
                      "factorised numpy restatement, %.1f s" % (0, lat.size, len(plev), nt_s, t_cpu)}, err, bad, one_pass


def cpu_baseline_config1_literal():
    """BASELINE configs[0] (ne4 x 30 x 1), the oracle in literal-association mode: dense
    lstsq(Y0, I_N) and (Y . Y0inv) . A -- the reference's own operation order
    (sph_zonal_mean.py:389, :251), the only config where that O(N^2) form is feasible."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import synth
    lat, lon = synth.cubed_sphere_gll(4)
    plev = synth.pressure_levels(30)
    f = synth.analytic_fields(lat, lon, plev, 1, seed=0)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        orc.TEMOracle(*f, lat, plev, mode="literal").results()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": lat.size * 30 / best, "unit": "grid-points/s", "kind": "port", "seconds": best,
            "sample": "configs[0] ne4 (866 cols) x 30 lev x 1, oracle in literal mode: lstsq(Y0, I_N) and "
                      "(Y Y0inv) A as the reference associates them; best of 3"}


def self_launch(argv, ngpus):
    """`python bench.py --gpus N` without a launcher: become the parent of torch.distributed.run.
    Nothing here has touched the GPU (no HIP call, no torch.cuda query)."""
    # --standalone: the launcher's own c10d rendezvous picks a free port and keeps it (no bind/close/reuse race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(ngpus), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ne120x72x30")
    ap.add_argument("--shard", choices=["auto", "time", "ncol"], default="auto",
                    help="auto: unsharded at N = 1, ncol at N > 1 (BASELINE configs[3], strong scaling); time: every "
                         "rank its own block of snapshots (weak scaling)")
    ap.add_argument("--time-workload", default="ne30x72x730",
                    help="N > 1: the time-sharded job reported under other_workloads (BASELINE configs[2])")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-nt", type=int, default=2)
    ap.add_argument("--no-symmetry", action="store_true",
                    help="force the generic sweeps (neither latitude classes nor mirror pairing)")
    ap.add_argument("--two-pass", action="store_true",
                    help="latitude-class sweeps in their two-pass form (fields read twice)")
    ap.add_argument("--class-sums", action="store_true",
                    help="one-pass class path in its class-sum form (sweep + flux kernel) instead of the single sweep")
    ap.add_argument("--no-classes", action="store_true",
                    help="do not use the latitude-class sweeps (mirror-paired sweeps on a symmetric grid)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N > 1: skip the legs reported beside the metric (time-sharded configs[2], weak scaling)")
    ap.add_argument("--also", default="ne30x72x1,ne30x72x91,ne30x72x92,ne240x128x1:f32,ne120x72x30:f32,ne120x72x30:f64:generic,ne120x72x30:f64:lat32,ne120x72x30:f64:shard1of8,ne120x72x4",
                    help="comma list of the other BASELINE.json shapes, timed after the main one at N=1 "
                         "(shape[:f32|f64][:generic|:shardRofW]; ne30x72x91 is one rank's block of the 730-snapshot config cut evenly, "
                         "ne30x72x92 the block TimeShardedTEM cuts with multiple=aligned_snapshots (rows of whole 128-byte lines), "
                         "ne120x72x4 a time-sharded rank's block of configs[3] -- the no-collective alternative; "
                         ":generic forces the generic sweeps -- what a grid without repeated latitudes gets; :lat32 rounds the latitude "
                         "coordinate through float32 (the class sweeps stay: rounding is deterministic); :shardRofW is "
                         "rank R's step of the job ncol-sharded over W ranks, its collectives left out)")
    ap.add_argument("--exact-mirror", action="store_true",
                    help="build the southern hemisphere of the synthetic grid as the bit-for-bit mirror of the northern one "
                         "(rounds 1-3); default: latitudes as the construction leaves them, equal within a class to round-off")
    ap.add_argument("--stall-timeout", type=float, default=900.0,
                    help="N > 1: seconds the whole run may take before the watchdog reports a stalled collective")
    args = ap.parse_args()
    if args.two_pass:
        os.environ["TEMX_TWO_PASS"] = "1"
    if args.class_sums:
        os.environ["TEMX_SINGLE_SWEEP"] = "0"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))      # child ranks print the line; same exit code
    if world != args.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    # TEMX_BENCH_BACKEND=gloo rehearses the N > 1 control flow with several ranks on one GPU
    backend = os.environ.get("TEMX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from pytemdiags_amd import engine, sharding, synth

    ne, nlev, nt = parse_workload(args.workload)
    lat, lon = synth.cubed_sphere_gll(ne, mirror=args.exact_mirror)
    plev = synth.pressure_levels(nlev)
    lat_zm = zm_lat(1.0)
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    ncol = lat.size

    # ---- shard ----
    if args.shard == "auto":
        args.shard = "ncol" if world > 1 else "none"
    use_ncol = args.shard == "ncol"          # at world 1 the all-reduces are no-ops (path check)
    errors = {}

    # A collective that never completes blocks inside C code, where no Python signal handler runs: the
    # watchdog is a daemon thread; it prints what there is with the error and ends the rank with exit code 3.
    partial_rec = {"metric": "grid-points/sec through full TEM pipeline (ncol*nlev*nt)", "value": 0.0,
                   "unit": "grid-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup}
    watchdog = None
    if world > 1:
        def _stalled():
            partial_rec["errors"] = dict(errors, stall="no result after %.0f s" % args.stall_timeout)
            if rank == 0:
                print(json.dumps(partial_rec), flush=True)
            os._exit(3)
        watchdog = threading.Timer(args.stall_timeout, _stalled)
        watchdog.daemon = True
        watchdog.start()
    if use_ncol:
        # whole mirror pairs per rank: every rank's block of columns stays equatorially symmetric
        mine = sharding.symmetric_ncol_shards(lat, world)[rank]
        lat_l, lon_l, t0_l, nt_l = lat[mine], lon[mine], 0, nt
        scaling, pts_job = "strong", ncol * nlev * nt
    else:
        lat_l, lon_l, t0_l, nt_l = lat, lon, rank * nt, nt
        scaling, pts_job = "weak", ncol * nlev * nt * world

    # ---- plan (timed separately; the reference amortises it through its map cache) ----
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = engine.Plan(lat_l, lat_zm, K_HARM - 1, device=local_rank, defer_finalize=use_ncol,
                       symmetry=not args.no_symmetry, classes=not args.no_classes, fp32_fields=args.dtype == "f32")
    if use_ncol:
        runner = sharding.NcolShardedTEM(plan)
        runner.set_tem(nlev, nt_l, plev * 100)       # (+ the plan-build collectives of the time-sliced tail)
    else:
        plan.set_tem(nlev, nt_l, plev * 100)
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0

    fields = engine.synth_fields(local_rank, lat_l, lon_l, plev, nt_l, t0=t0_l, dtype=tdtype, seed=0)
    out = plan._alloc_results(False)

    if use_ncol:
        def step():
            return runner.run(*fields)
    else:
        def step():
            return plan.tem_run(*fields, out=out)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    plan.kernel_timing(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # the library launches on torch's current stream
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    nonfinite = plan.status()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    proj_ms, nproj = plan.kernel_timing_read(0)
    eddy_ms, neddy = plan.kernel_timing_read(1)
    plan.kernel_timing(False)

    one_pass_main = plan.one_pass
    ms_per_step = elapsed / args.steps * 1e3
    value = pts_job * args.steps / elapsed
    pts_rank = lat_l.size * nlev * nt_l
    esize = 8 if args.dtype == "f64" else 4

    rec = {
        "metric": "grid-points/sec through full TEM pipeline (ncol*nlev*nt)",
        "value": value, "unit": "grid-points/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "ne%d (%d cols) x %d lev x %d snapshots per %s, L=50, 1-degree zonal grid (M=180), "
                               "ten GM16 Table-A1 outputs" % (ne, ncol, nlev, nt, "GPU" if scaling == "weak" else "job"),
                   "shard": args.shard if (world > 1 or use_ncol) else "none", "ncol": int(ncol), "nlev": nlev, "nt": nt,
                   "sweeps": ("generic", "mirror-paired", "latitude-class")[plan.sweep_mode]
                             + (", single sweep" if getattr(plan, "single_sweep", False)
                                else (", one pass" if plan.one_pass else "")),
                   "mirror_paired_sweeps": bool(plan.paired)},
        # per-step HIP-event times of the same K steps (this rank): SURVEY 8(d) quotes the median of >= 20
        "ms_per_step_median": median_ms, "ms_per_step_min": step_ms[0], "ms_per_step_max": step_ms[-1],
        "value_at_median": pts_job / (median_ms * 1e-3),
        "plan_build_s": plan_s,
        # dense-operator roofline of SURVEY 8(d): max(64 B / 8 TB/s, 1122 flop / 78.6 TF) per point.  The
        # latitude-class sweeps do the MFMA work per class, not per column, so on grids with repeated
        # latitudes the pipeline can exceed the MFMA side and is bounded by the two compulsory reads.
        "pipeline_frac_of_fp64_roofline": value / world / (PEAK_F64_TFLOPS * 1e12 / FLOPS_PER_PT),
        # HBM roofline of the path as run: two compulsory reads of the four fields (8 s bytes/point), or
        # one (4 s) when the one-pass class path is in use
        "pipeline_frac_of_hbm_roofline": value / world / (PEAK_HBM_GBS * 1e9 / ((4 if plan.one_pass else 8)
                                                                              * (8 if args.dtype == "f64" else 4))),
        "hbm_roofline_bytes_per_point": (4 if plan.one_pass else 8) * (8 if args.dtype == "f64" else 4),
        "nonfinite": bool(nonfinite),
    }
    if use_ncol:
        via = "RCCL/xGMI" if backend == "nccl" else backend
        if runner.sliced:
            rec["config"]["collectives"] = (
                "per step over %s: 1 all-reduce of the reference pre-pass sums ([4][16][D] fp64 = %d bytes) + 1 reduce-scatter "
                "over time of the single sweep's projections ([4 x 101 + 3 x 51][D] fp64 = %d bytes in, 1/%d of it out per "
                "rank); K x K Gram matrices and the single sweep's two row-sum matrices once at plan build"
                % (via, 4 * 16 * nlev * nt * 8, (4 * (2 * K_HARM - 1) + 3 * K_HARM) * nlev * nt * 8, world))
            rec["config"]["tail"] = ("time sliced: every rank finishes its own snapshots (%d..%d of %d on rank 0); the ten "
                                     "results stay time-sharded" % (runner.my_snapshots()[0], runner.my_snapshots()[1] - 1, nt))
        else:
            rec["config"]["collectives"] = ("2 all-reduces per step over %s ([4][K][D] and [3][K][D] fp64 = %d bytes) + the "
                                            "K x K Gram matrix once at plan build; tail replicated on every rank"
                                            % (via, 7 * K_HARM * nlev * nt * 8))
    if neddy:
        ach = 7 * 2 * K_HARM * pts_rank / (eddy_ms * 1e-3) / 1e12
        if plan.one_pass and nproj:
            # one-pass class path: the dominant kernel is sweep 1 (the only read of the fields)
            gbs_p = 4 * esize * pts_rank / (proj_ms * 1e-3) / 1e9
            single = bool(getattr(plan, "single_sweep", False))
            rec["roofline"] = {"kernel": ("sweep_osr_kernel (the single sweep, loads of 1 row x 64 columns: theta, class sums of the four fields minus a low-degree "
                                          "reference projected to degree 2L, their three products to degree L; no class-sum stream)"
                                          if single else
                                          "sweep_opr_kernel for fp64 / sweep_op_kernel for fp32 inputs (sweep 1 of the class-sum form: theta + class sums of the fields, "
                                          "centred class co-moments of u v, u omega, v theta + 7 class projections)"),
                               "bound": "hbm", "achieved": gbs_p, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": gbs_p / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": proj_ms,
                               "launches": nproj,
                               "algorithmic": "4 fields x %d B per grid point (the one compulsory read of u, v, T, omega) x "
                                              "%d points per launch; the kernel also stores the 4 field sums of every "
                                              "latitude-class side (8 x 512 B per class-group and d-tile, see traffic)"
                                              % (esize, pts_rank)}
            rec["roofline_flux"] = {"kernel": ("os_contract_kernel (eddy-product sums from the projections: Legendre product "
                                               "linearisation on Gauss-Legendre nodes)" if single else
                                               "flux_cls_kernel (class reconstructions, n (m_u - ub)(m_v - vb) per class side, "
                                               "3 class projections)"), "avg_launch_ms": eddy_ms, "launches": neddy}
            if single:
                rec["roofline"]["algorithmic"] = ("4 fields x %d B per grid point (the one compulsory read of u, v, T, omega) x %d "
                                                  "points per launch; nothing else of that order is read or written"
                                                  % (esize, pts_rank))
        elif plan.sweep_mode == 2:
            gbs_e = 4 * esize * pts_rank / (eddy_ms * 1e-3) / 1e9
            rec["roofline"] = {"kernel": "eddy_cls_kernel (class reconstructions + eddies + products + class projections)",
                               "bound": "hbm", "achieved": gbs_e, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": gbs_e / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": eddy_ms,
                               "launches": neddy,
                               "algorithmic": "4 fields x %d B per grid point (the compulsory read of u, v, T, omega) x %d "
                                              "points per launch" % (esize, pts_rank),
                               "operator_tflops": ach,
                               "note": "operator_tflops = the dense operator's 7*2*51 flop per point over the launch time; "
                                       "the class sweep executes about 1/8 of them on this grid, which is why the kernel "
                                       "is HBM bound"}
        else:
            rec["roofline"] = {"kernel": "eddy_kernel (4 reconstructions + eddy products + 3 projections)",
                               "bound": "mfma", "achieved": ach, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_F64_TFLOPS, "traffic": None, "avg_launch_ms": eddy_ms,
                               "launches": neddy,
                               "algorithmic": "7*2*51 flop per grid point x %d points per launch "
                                              "(the operator's flops; on an equatorially symmetric grid the "
                                              "mirror-paired sweep executes 54 %% of them)" % pts_rank}
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr):
            try:
                j = json.load(open(tr))
                if (j.get("workload") == args.workload and j.get("dtype") == args.dtype
                        and j.get("sweeps", "mirror-paired") == rec["config"]["sweeps"]):
                    rec["roofline"]["traffic"] = j.get("project_kernel_hbm_bytes_per_launch" if plan.one_pass
                                                       else "eddy_kernel_hbm_bytes_per_launch")
                    rec["roofline"]["traffic_source"] = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                         "passes of this command (tools/profile_session.sh), not this run")
            except Exception:
                pass
    if nproj and not (rec["config"]["sweeps"].endswith("one pass") or rec["config"]["sweeps"].endswith("single sweep")):
        gbs = 4 * esize * pts_rank / (proj_ms * 1e-3) / 1e9
        rec["roofline_project"] = {"kernel": "project kernel (theta + 4 projections)", "bound": "hbm",
                                   "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": gbs / PEAK_HBM_GBS, "avg_launch_ms": proj_ms,
                                   "mfma_tflops": 4 * 2 * K_HARM * pts_rank / (proj_ms * 1e-3) / 1e12}

    if rank == 0 and world == 1 and args.also:
        # other BASELINE.json shapes, same pipeline, reported beside the headline (not the metric)
        rec["other_workloads"] = {}
        for wl in [w for w in args.also.split(",") if w]:
            parts = wl.split(":")                             # "ne240x128x1:f32[:generic]" -> shape, input dtype, sweeps
            name, dt_s = parts[0], (parts[1] if len(parts) > 1 else "")
            generic = "generic" in parts[2:]
            shard = [x for x in parts[2:] if x.startswith("shard")]
            dt2_t = {"": tdtype, "f64": torch.float64, "f32": torch.float32}[dt_s]
            ne2, nlev2, nt2 = parse_workload(name)
            try:
                lat2, lon2 = synth.cubed_sphere_gll(ne2, mirror=args.exact_mirror)
                if "lat32" in parts[2:]:       # a latitude coordinate that went through float32, as some grid files carry it
                    lat2 = lat2.astype(np.float32).astype(np.float64)
                plev2 = synth.pressure_levels(nlev2)
                sliced = None
                if shard:
                    # One rank's step of the ncol-sharded job, its collectives left out: the job's W plans are built
                    # over whole latitude classes and given the matrices the plan-build all-reduces leave them with
                    # (Gram matrices, and for the single sweep Gx and the subsample's Gram matrix); rank R's plan then
                    # runs exactly what NcolShardedTEM.run makes it run.
                    from pytemdiags_amd import _lib
                    r_s, w_s = (int(x) for x in shard[0][5:].split("of"))
                    blocks = sharding.symmetric_ncol_shards(lat2, w_s)
                    pw = [engine.Plan(lat2[m], lat_zm, K_HARM - 1, device=local_rank, defer_finalize=True) for m in blocks]
                    Gm = sum(q.matrix(_lib.MAT_GRAM) for q in pw).cpu().numpy()
                    for q in pw:
                        q.finalize(Gm)
                    Gm2 = sum(q.matrix(_lib.MAT_GRAM2) for q in pw).cpu().numpy()
                    for q in pw:
                        q.refine(Gm2)
                        q.configure(os_subsample=max(8, -(-32 // w_s)), single_sweep_min_groups=max(64, 640 // w_s))
                    p2 = pw[r_s - 1]
                    p2.set_tem(nlev2, nt2, plev2 * 100)
                    if p2.single_sweep and nt2 >= w_s:
                        for q in pw:
                            if q is not p2:
                                q.configure(form="single-sweep")       # (the tables depend on the grid and L only: a small
                                q.set_tem(nlev2, 1, plev2 * 100)       #  shape builds them, forced past the size threshold)
                        Gx = sum(q.matrix(_lib.MAT_GX) for q in pw).cpu().numpy()
                        Gs = sum(q.matrix(_lib.MAT_GSUB) for q in pw).cpu().numpy()
                        p2.set_os_matrices(Gx, Gs)
                        sliced = sharding.shard_bounds(nt2, w_s, r_s - 1)
                    for q in pw:
                        if q is not p2:
                            q.close()
                    lat2, lon2 = lat2[blocks[r_s - 1]], lon2[blocks[r_s - 1]]
                else:
                    forms = [x for x in parts[2:] if x in ("single-sweep", "class-sums", "two-pass")]
                    p2 = engine.Plan(lat2, lat_zm, K_HARM - 1, device=local_rank, symmetry=not generic,
                                     form=forms[0] if forms else None)
                    p2.set_tem(nlev2, nt2, plev2 * 100)
                same = name == args.workload and dt2_t == tdtype and not shard      # the headline's own fields: reuse them
                f2 = fields if same else engine.synth_fields(local_rank, lat2, lon2, plev2, nt2, dtype=dt2_t, seed=0)
                if sliced:
                    # pre-pass, [all-reduce], sweep + reduction into W time slices, [reduce-scatter], tail on this rank's slice
                    bA = torch.zeros((4, p2.KR, p2.D), dtype=torch.float64, device=dev)
                    bP = torch.zeros((w_s, p2.os_rows * nlev2 * -(-nt2 // w_s)), dtype=torch.float64, device=dev)
                    o2 = torch.empty((10, p2.M, nlev2, sliced[1] - sliced[0]), dtype=torch.float64, device=dev)

                    def step2():
                        p2.tem_os_prepass(*f2, out=bA)
                        p2.tem_os_sweep(*f2, bA, nslices=w_s, out=bP)
                        p2.tem_os_tail(bP[r_s - 1], sliced[0], sliced[1] - sliced[0], out=o2)
                else:
                    o2 = p2._alloc_results(False)

                    def step2():
                        p2.tem_run(*f2, out=o2)
                for _ in range(3):
                    step2()
                torch.cuda.synchronize()
                reps, t0 = 0, time.perf_counter()
                while reps < 5 or (time.perf_counter() - t0 < 0.3 and reps < 200):
                    step2()
                    reps += 1
                    if reps % 10 == 0:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t0) / reps
                pts2 = lat2.size * nlev2 * nt2
                rec["other_workloads"][wl] = {
                    "ms_per_step": dt2 * 1e3, "grid_points_per_s": pts2 / dt2, "reps": reps, "ncol": int(lat2.size),
                    "plan_symmetry": not generic,
                    "sweeps": ("generic", "mirror-paired", "latitude-class")[p2.sweep_mode]
                              + (", single sweep" if getattr(p2, "single_sweep", False)
                                 else (", one pass" if p2.one_pass else ""))
                              + (", time-sliced tail (snapshots %d..%d of %d), collectives left out" % (sliced[0], sliced[1] - 1, nt2)
                                 if sliced else ""),
                    "frac_of_fp64_roofline": pts2 / dt2 / (PEAK_F64_TFLOPS * 1e12 / FLOPS_PER_PT)}
                if bool(p2.status()):
                    rec["other_workloads"][wl]["nonfinite"] = True
                p2.close()
                del f2, o2
            except Exception as e:  # noqa: BLE001 - an extra shape must not cost the metric line
                rec["other_workloads"][wl] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del fields
        torch.cuda.empty_cache()
        if one_pass_main:    # the sample has few d-tiles: make it take the code path that was timed
            os.environ["TEMX_ONE_PASS"] = "1"
        cb, err, bad, op_s = cpu_baseline_and_parity(
            lambda: engine.Plan(lat, lat_zm, K_HARM - 1, device=local_rank, symmetry=not args.no_symmetry,
                                classes=not args.no_classes, fp32_fields=args.dtype == "f32"),
            lat, lon, plev, args.cpu_sample_nt, local_rank, tdtype)
        cb["sample"] = cb["sample"].replace("ne0", "ne%d" % ne)
        rec["cpu_baseline"] = cb
        try:
            rec["cpu_baseline_config1_literal"] = cpu_baseline_config1_literal()
        except Exception as e:  # noqa: BLE001 - the extra baseline must not cost the metric line
            rec["cpu_baseline_config1_literal"] = {"error": "%s: %s" % (type(e).__name__, e)}
        ptol = 1e-10 if args.dtype == "f64" else 2e-5      # fp32 fields are held to 2e-5
        rec["parity_vs_oracle_on_sample"] = {"max_field_normalised_err": err, "tolerance": ptol, "input_dtype": args.dtype,
                                            "ok": bool(err <= ptol and not bad), "one_pass": bool(op_s)}
    plan.close()

    partial_rec.update(rec)

    def timed(step_fn):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over the ranks."""
        for _ in range(max(args.warmup, 1)):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        barrier()
        e = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
        return float(e.item())

    def leg(name, fn):
        """An extra leg: its record under other_workloads; a failure is recorded and fails the run."""
        try:
            if os.environ.get("TEMX_BENCH_FAIL") == name:      # tests: a forced failure of this leg
                raise RuntimeError("forced failure of leg %r (TEMX_BENCH_FAIL)" % name)
            rec.setdefault("other_workloads", {})[name] = fn()
        except Exception as e:  # noqa: BLE001 - reported in the line, and through the exit code
            errors[name] = "%s: %s" % (type(e).__name__, e)
        torch.cuda.empty_cache()

    if world > 1 and not args.no_extras:
        del fields, out
        torch.cuda.empty_cache()

        def time_sharded():
            # BASELINE configs[2]: one job of nt_all snapshots, each rank its own contiguous block (ragged
            # blocks: 91/92 at N = 8), replicated plan, no collective on the data path; strong scaling
            ne2, nlev2, nt_all = parse_workload(args.time_workload)
            lat2, lon2 = synth.cubed_sphere_gll(ne2, mirror=args.exact_mirror)
            plev2 = synth.pressure_levels(nlev2)
            # blocks of whole cache lines per row where the count allows (730 on 8 ranks: 5 x 92 + 3 x 90, not 2 x 92 + 6 x 91)
            ta, tb = sharding.shard_bounds(nt_all, world, rank, sharding.aligned_snapshots(nlev2, 8 if args.dtype == "f64" else 4))
            p2 = engine.Plan(lat2, lat_zm, K_HARM - 1, device=local_rank, symmetry=not args.no_symmetry,
                             classes=not args.no_classes)
            p2.set_tem(nlev2, tb - ta, plev2 * 100)
            f2 = engine.synth_fields(local_rank, lat2, lon2, plev2, tb - ta, t0=ta, dtype=tdtype, seed=0)
            o2 = p2._alloc_results(False)
            e2 = timed(lambda: p2.tem_run(*f2, out=o2))
            bad2 = p2.status()
            r = {"scaling": "strong", "shard": "time", "value": lat2.size * nlev2 * nt_all * args.steps / e2,
                 "unit": "grid-points/s", "ms_per_step": e2 / args.steps * 1e3, "n_gpus": world,
                 "workload": "ne%d (%d cols) x %d lev x %d snapshots per job, about %d per rank in blocks of whole cache "
                             "lines per row, no collective" % (ne2, lat2.size, nlev2, nt_all, nt_all // world),
                 "snapshots_this_rank": tb - ta,
                 "sweeps": ("generic", "mirror-paired", "latitude-class")[p2.sweep_mode]
                           + (", one pass" if p2.one_pass else ""), "nonfinite": bool(bad2)}
            p2.close()
            if bad2:
                raise RuntimeError("non-finite values in the time-sharded run")
            return r

        def ncol_sharded():
            # BASELINE configs[3] when the metric is the weak-scaling run (--shard time)
            mine3 = sharding.symmetric_ncol_shards(lat, world)[rank]
            p3 = engine.Plan(lat[mine3], lat_zm, K_HARM - 1, device=local_rank, defer_finalize=True,
                             symmetry=not args.no_symmetry, classes=not args.no_classes)
            r3 = sharding.NcolShardedTEM(p3)
            r3.set_tem(nlev, nt, plev * 100)
            f3 = engine.synth_fields(local_rank, lat[mine3], lon[mine3], plev, nt, t0=0, dtype=tdtype, seed=0)
            e3 = timed(lambda: r3.run(*f3))
            bad3 = p3.status()
            r = {"scaling": "strong", "shard": "ncol", "value": ncol * nlev * nt * args.steps / e3,
                 "unit": "grid-points/s", "ms_per_step": e3 / args.steps * 1e3, "n_gpus": world,
                 "workload": "ne%d (%d cols) x %d lev x %d snapshots per job, columns sharded in whole latitude "
                             "classes" % (ne, ncol, nlev, nt),
                 "sweeps": ("generic", "mirror-paired", "latitude-class")[p3.sweep_mode]
                           + (", one pass" if p3.one_pass else ""), "nonfinite": bool(bad3)}
            p3.close()
            if bad3:
                raise RuntimeError("non-finite values in the ncol-sharded run")
            return r

        def weak():
            # every rank its own block of nt snapshots of the whole grid: work grows with N, no collective
            p4 = engine.Plan(lat, lat_zm, K_HARM - 1, device=local_rank, symmetry=not args.no_symmetry,
                             classes=not args.no_classes)
            p4.set_tem(nlev, nt, plev * 100)
            f4 = engine.synth_fields(local_rank, lat, lon, plev, nt, t0=rank * nt, dtype=tdtype, seed=0)
            o4 = p4._alloc_results(False)
            e4 = timed(lambda: p4.tem_run(*f4, out=o4))
            bad4 = p4.status()
            r = {"scaling": "weak", "shard": "time", "value": ncol * nlev * nt * world * args.steps / e4,
                 "unit": "grid-points/s", "ms_per_step": e4 / args.steps * 1e3, "n_gpus": world,
                 "workload": "ne%d (%d cols) x %d lev x %d snapshots per GPU, no collective" % (ne, ncol, nlev, nt),
                 "nonfinite": bool(bad4)}
            p4.close()
            if bad4:
                raise RuntimeError("non-finite values in the weak-scaling run")
            return r

        leg("%s:time_sharded" % args.time_workload, time_sharded)
        if use_ncol:
            leg("%s:weak_scaling" % args.workload, weak)
        else:
            leg("%s:ncol_sharded" % args.workload, ncol_sharded)

    if os.environ.get("TEMX_BENCH_FAIL") == "main":
        errors["main"] = "RuntimeError: forced failure of the metric leg (TEMX_BENCH_FAIL)"
    if rec.get("nonfinite"):
        errors["main"] = "non-finite values reached the zonal sums of the metric run"
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        rec["process_group"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
        # a leg may have failed on another rank only: every rank must leave with the same code
        nerr = torch.tensor([float(len(errors))], dtype=torch.float64, device=dev)
        dist.all_reduce(nerr, op=dist.ReduceOp.MAX)
        if nerr.item() > 0 and not errors:
            errors["other_rank"] = "a leg failed on another rank"
    if errors:
        rec["errors"] = errors
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if errors:
        sys.exit(4)


if __name__ == "__main__":
    main()

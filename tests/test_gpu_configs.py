"""Every BASELINE.json configuration at its full size, through the C ABI on the GPU.

configs[1] ne30 x 72 x 1 fp64 ............ vs the CPU oracle (exact shape)
configs[2] ne30 x 72 x 730 time-sharded ... one rank's block, ne30 x 72 x 91, vs the oracle
configs[3] ne120 x 72 x 30 ............... properties at full size (finite, run-to-run identical bits,
                                           one-pass == two-pass to 1e-11) + a 2-snapshot sample of
                                           the same grid vs the oracle on the one-pass path
configs[4] ne240 x 128 x 1 fp32 .......... vs the oracle (tolerance 2e-5)

Tolerances: fp64 max|d| <= 1e-10 max|ref| per output field; fp32 inputs 2e-5 (the oracle, like the
reference, rounds eddies and results to fp32).  The inputs are generated on the device
(temx_synth_fields) and copied to the host for the oracle, so both sides see identical arrays.
"""
import numpy as np
import pytest

from conftest import fieldnorm_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _zm_lat():
    e = np.arange(-90, 91, 1.0)
    return (e[1:] + e[:-1]) / 2


def _vs_oracle(ne, nlev, nt, dtype, tol, force_one_pass=None, monkeypatch=None):
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a GPU")
    if force_one_pass is not None and monkeypatch is not None:
        monkeypatch.setenv("TEMX_ONE_PASS" if force_one_pass else "TEMX_TWO_PASS", "1")
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = engine.synth_fields(0, lat, lon, plev, nt, dtype=dtype, seed=0)
    plan = engine.Plan(lat, _zm_lat(), 50)
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*f)
    assert not plan.status()
    one_pass, mode = plan.one_pass, plan.sweep_mode
    res = res.cpu().numpy()
    plan.close()
    host = [x.cpu().numpy() for x in f]
    del f
    torch.cuda.empty_cache()
    ref = orc.TEMOracle(*host, lat, plev, mode="factorised")
    worst = 0.0
    for i, n in enumerate(_lib.RESULT_NAMES):
        r = np.asarray(getattr(ref, n)(), dtype=np.float64)
        e = fieldnorm_err(res[i], r)
        worst = max(worst, e)
        assert e <= tol, (n, e)
    return worst, one_pass, mode


def test_config1_ne30x72x1_f64_vs_oracle():
    _vs_oracle(30, 72, 1, torch.float64, 1e-10)


def test_config2_one_rank_block_ne30x72x91_f64_vs_oracle():
    worst, one_pass, mode = _vs_oracle(30, 72, 91, torch.float64, 1e-10)
    import os
    if not any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS")):
        assert mode == 2 and one_pass     # the path bench.py times for this shape


def test_config3_sample_ne120x72x2_f64_vs_oracle_on_the_one_pass_path(monkeypatch):
    """The same grid, levels and code path as the headline workload, 2 of its 30 snapshots."""
    worst, one_pass, mode = _vs_oracle(120, 72, 2, torch.float64, 1e-10, force_one_pass=True,
                                       monkeypatch=monkeypatch)
    import os
    if not any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS")):
        assert one_pass


def test_config3_full_size_ne120x72x30_properties(monkeypatch):
    """Full size (53.7 GB of inputs): no non-finite value, identical bits run to run (fixed-order
    reductions), and the one-pass and two-pass forms of the class path agree to 1e-11."""
    from pytemdiags_amd import _lib, engine, synth
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a GPU")
    lat, lon = synth.cubed_sphere_gll(120)
    plev = synth.pressure_levels(72)
    f = engine.synth_fields(0, lat, lon, plev, 30, dtype=torch.float64, seed=0)
    plan = engine.Plan(lat, _zm_lat(), 50)
    plan.set_tem(72, 30, plev * 100)
    r1, _ = plan.tem_run(*f)
    r1 = r1.clone()
    r2, _ = plan.tem_run(*f)
    assert not plan.status()
    assert torch.isfinite(r1).all()
    assert torch.equal(r1, r2)
    one = plan.one_pass
    plan.close()
    if one:
        monkeypatch.setenv("TEMX_TWO_PASS", "1")
        plan2 = engine.Plan(lat, _zm_lat(), 50)
        plan2.set_tem(72, 30, plev * 100)
        assert not plan2.one_pass
        r3, _ = plan2.tem_run(*f)
        assert not plan2.status()
        for i, n in enumerate(_lib.RESULT_NAMES):
            den = float(r1[i].abs().max())
            assert float((r3[i] - r1[i]).abs().max()) <= 1e-11 * den, n
        plan2.close()
    del f
    torch.cuda.empty_cache()


def test_config3_full_size_eight_ncol_shards_equal_the_unsharded_run():
    """BASELINE configs[3] as it is partitioned at 8 GPUs: the ne120 grid cut by
    symmetric_ncol_shards(lat, 8), the staged C ABI run shard by shard on this one GPU with the three
    all-reduces written out as explicit sums (Gram matrix, [4][K][D], [3][K][D]) -- sph_zonal_mean.py:251
    is linear in the rows -- against the unsharded run of the same 53.7 GB, to 1e-11.  One shard's rows
    are gathered at a time; the eight plans stay alive because stage 2 of the one-pass class path works
    from the class sums its own stage 1 stored."""
    from pytemdiags_amd import _lib, engine, sharding, synth
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a GPU")
    W = 8
    lat, lon = synth.cubed_sphere_gll(120)
    plev = synth.pressure_levels(72)
    nt = 30
    f = engine.synth_fields(0, lat, lon, plev, nt, dtype=torch.float64, seed=0)
    full = engine.Plan(lat, _zm_lat(), 50)
    full.set_tem(72, nt, plev * 100)
    ref, _ = full.tem_run(*f)
    ref = ref.clone()
    assert not full.status()
    full.close()
    parts = sharding.symmetric_ncol_shards(lat, W)
    assert sum(p.size for p in parts) == lat.size and np.array_equal(np.sort(np.concatenate(parts)), np.arange(lat.size))
    a = np.abs(lat)
    for r in range(W - 1):                       # whole latitude classes per rank
        assert a[parts[r]].max() < a[parts[r + 1]].min()
    plans = [engine.Plan(lat[p], _zm_lat(), 50, defer_finalize=True) for p in parts]
    G = sum(pl.matrix(_lib.MAT_GRAM) for pl in plans).cpu().numpy()              # all-reduce (i)
    B4 = None
    for pl in plans:
        pl.finalize(G)
    G2 = sum(pl.matrix(_lib.MAT_GRAM2) for pl in plans).cpu().numpy()            # all-reduce (i')
    for pl, p in zip(plans, parts):
        pl.refine(G2)
        # as NcolShardedTEM.set_tem: the reference subsample spread over the ranks, the form chosen by the size of the job
        pl.configure(os_subsample=8, single_sweep_min_groups=640 // W)
        pl.set_tem(72, nt, plev * 100)
        idx = torch.as_tensor(p, device="cuda:0")
        loc = [x[idx] for x in f]
        b = pl.tem_stage1(*loc)                                                     # all-reduce (ii)
        B4 = b if B4 is None else B4 + b
        del loc
    B3 = None
    for pl, p in zip(plans, parts):
        if pl.one_pass:
            b = pl.tem_stage2_from_sums(B4)
        else:
            idx = torch.as_tensor(p, device="cuda:0")
            b = pl.tem_stage2(*[x[idx] for x in f], B4)
        B3 = b if B3 is None else B3 + b                                            # all-reduce (iii)
    import os
    if not any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS")):
        assert all(pl.sweep_mode == 2 and pl.one_pass for pl in plans)              # what each of the 8 ranks runs
    for pl in (plans[0], plans[W - 1]):          # every rank evaluates the same epilogue
        res, _ = pl.tem_stage3(B3)
        for i, n in enumerate(_lib.RESULT_NAMES):
            den = float(ref[i].abs().max())
            assert float((res[i] - ref[i]).abs().max()) <= 1e-11 * den, n
    assert not any(pl.status() for pl in plans)
    # ---- round 4: what NcolShardedTEM runs on these plans by default -- the single sweep with a time-sliced tail.
    # Plan-build all-reduces of the two row-sum matrices, per step one all-reduce (pre-pass sums) and one
    # reduce-scatter over time (projections); rank w finishes the snapshots shard_bounds(30, 8, w) = 4, 4, 4, 4, 4, 4, 3, 3.
    if all(pl.single_sweep for pl in plans):
        Gx = sum(pl.matrix(_lib.MAT_GX) for pl in plans).cpu().numpy()
        Gs = sum(pl.matrix(_lib.MAT_GSUB) for pl in plans).cpu().numpy()
        As = None
        for pl, p in zip(plans, parts):
            pl.set_os_matrices(Gx, Gs)
            idx = torch.as_tensor(p, device="cuda:0")
            a = pl.tem_os_prepass(*[x[idx] for x in f])
            As = a if As is None else As + a                                        # all-reduce
        proj = None
        for pl, p in zip(plans, parts):
            idx = torch.as_tensor(p, device="cuda:0")
            b = pl.tem_os_sweep(*[x[idx] for x in f], As, nslices=W)
            proj = b if proj is None else proj + b                                  # reduce-scatter: rank w keeps proj[w]
        outs = []
        for w, pl in enumerate(plans):
            t0, t1 = sharding.shard_bounds(nt, W, w)
            outs.append(pl.tem_os_tail(proj[w], t0, t1 - t0)[0])
        got = torch.cat(outs, dim=-1)
        assert got.shape == ref.shape
        for i, n in enumerate(_lib.RESULT_NAMES):
            den = float(ref[i].abs().max())
            assert float((got[i] - ref[i]).abs().max()) <= 1e-11 * den, n
        assert not any(pl.status() for pl in plans)
    else:
        assert any(os.environ.get(k) in ("0", "1") for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS", "TEMX_SINGLE_SWEEP", "TEMX_NO_QR"))
    for pl in plans:
        pl.close()
    del f
    torch.cuda.empty_cache()


def test_config4_ne240x128x1_f32_vs_oracle():
    _vs_oracle(240, 128, 1, torch.float32, 2e-5)

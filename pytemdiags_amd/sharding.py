"""Multi-GPU sharding of the TEM pipeline: one process per GPU, torch.distributed (RCCL on ROCm).

Two ways the path shards:

* **time sharding** -- the columns of the (N x D) operand are independent, so each rank takes a
  contiguous block of time snapshots with a replicated plan.  No collective on the data path.
* **ncol sharding** -- rows split in blocks (whole latitude classes per rank, see ``symmetric_ncol_shards``).  The
  zonal sums are linear in the rows (sph_zonal_mean.py:251) and everything after them acts along latitude and
  pressure only (tem_diagnostics.py:574-797), so the ranks exchange the sums and split the TAIL over time:
  plans that run the single sweep all-reduce the reference pre-pass sums ([4][16][D]), reduce-scatter the sweep's
  projections over time, and every rank finishes its own snapshots -- nothing of the tail is replicated, the
  results stay time-sharded (``gather_time``).  Any other plan: all-reduce of the [4][K][D] sums of (u, v, theta,
  omega) and of the [3][K][D] sums of the eddy products, every rank solving the K x K system and evaluating the
  zonal-grid epilogue redundantly.  Once at plan build: the K x K Gram matrix, the Gram matrix of the
  re-orthogonalised basis, and for the single sweep its two matrices that sum over the rows.

The driver below only sequences stages and collectives; the numerics live behind a *backend*
with the ``engine.Plan`` stage interface (``matrix``, ``finalize``, ``tem_stage1/2/3``, ``tem_os_prepass/sweep/tail``).  On
the GPU the backend is ``engine.Plan``; the world_size-2 gloo tests drive the same code with a
CPU stand-in built on the oracle.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

MAT_GRAM, MAT_GRAM2, MAT_GX, MAT_GSUB = 2, 5, 6, 7


def shard_bounds(n, world, rank, multiple=1):
    """Contiguous near-equal blocks: the first ``n % world`` ranks get one extra element.  ``multiple`` > 1: the
    blocks are whole multiples of it (a remainder goes to the last rank), see ``aligned_snapshots``."""
    n, world, multiple = int(n), int(world), int(multiple)
    if multiple > 1 and n // multiple >= world:
        units, rem = divmod(n, multiple)
        lo, hi = shard_bounds(units, world, rank)
        return lo * multiple, hi * multiple + (rem if rank == world - 1 else 0)
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def aligned_snapshots(nlev, itemsize, line=128):
    """The smallest number of snapshots m for which a row of a field -- nlev x m values, time fastest -- is a whole
    number of 128-byte lines.  The sweeps read a row in pieces of 64 columns; when rows do not start on a line every
    piece straddles one more line than it needs (PMC: 1.12 x the bytes at 72 x 91 fp64, 1.18 x at 72 x 30 fp32) and
    a step costs 7-8 % more per snapshot (ne30 x 72: 26.8 us per snapshot at 91, 24.4 at 90, 25.3 at 92).  A
    time-sharded job can choose its blocks: ``TimeShardedTEM(..., multiple=aligned_snapshots(nlev, itemsize))`` gives
    730 six-hourly snapshots on 8 ranks as 5 x 92 + 3 x 90 instead of 2 x 92 + 6 x 91."""
    import math
    return line // math.gcd(line, int(nlev) * int(itemsize))


def symmetric_ncol_shards(lat_deg, world, tol=1e-12):
    """Row index sets for ncol sharding that keep columns of equal |lat| on one rank: every
    rank's block then consists of whole latitude classes (a column, its mirror column and all the
    columns that share their latitude), so the engine can use its latitude-class / mirror-paired
    sweeps on each rank.  Columns are ordered by |lat| and cut into ``world`` near-equal runs, the
    cuts moved forward to the next class boundary.  Works for any grid (a grid without repeated
    latitudes simply gets latitude bands).  Returns ``world`` int64 index arrays, each ascending."""
    import numpy as np
    lat = np.abs(np.asarray(lat_deg, dtype=np.float64))
    n = lat.size
    order = np.argsort(lat, kind="stable")
    sl = lat[order]
    cuts = [0]
    for r in range(1, world):
        c = max(shard_bounds(n, world, r)[0], cuts[-1])
        while 0 < c < n and sl[c] - sl[c - 1] <= tol:      # do not cut inside a class
            c += 1
        cuts.append(min(c, n))
    cuts.append(n)
    return [np.sort(order[cuts[r]:cuts[r + 1]]).astype(np.int64) for r in range(world)]


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def allreduce_sum_(t, group=None):
    """In-place sum over ranks (RCCL for device tensors, gloo for CPU tensors); no-op at world 1."""
    if _world(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def reduce_scatter_sum(chunks, group=None, out=None):
    """``chunks`` is ``[world][n]`` (contiguous): rank w receives the sum over the ranks of chunk w (``[n]``).
    RCCL: one ``reduce_scatter_tensor``; back ends without it (gloo on device tensors) all-reduce and slice."""
    w = _world(group)
    if w == 1:
        return chunks.reshape(-1)
    assert chunks.shape[0] == w and chunks.is_contiguous()
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        if out is None:
            out = torch.empty(chunks.shape[1:], dtype=chunks.dtype, device=chunks.device)
        dist.reduce_scatter_tensor(out, chunks, op=dist.ReduceOp.SUM, group=group)
        return out.reshape(-1)
    dist.all_reduce(chunks, op=dist.ReduceOp.SUM, group=group)
    return chunks[rank].reshape(-1)


class NcolShardedTEM:
    """TEM pipeline over this rank's block of native columns.

    ``backend`` is a plan created over the rank's own latitudes with ``defer_finalize=True``.  Two forms of a step:

    * **time-sliced tail** (the plans run the single sweep, ``nt >= world``): everything after the zonal sums acts
      along latitude and pressure only, so the sums are exchanged by a REDUCE-SCATTER OVER TIME and each rank finishes
      the snapshots it receives -- nothing of the tail is replicated.  Per step: all-reduce of the reference
      pre-pass sums (4 x 16 x D doubles), one reduce-scatter of the projections ((4 (2L+1) + 3 (L+1)) x D doubles, an
      eighth of it arriving per rank at world 8).  ``run`` returns this rank's snapshots
      ``[..][M][nlev][t0:t1]`` (``shard_bounds(nt, world, rank)``; ``gather_time`` assembles the whole);
    * **replicated tail** (any other plan): all-reduce of the [4][K][D] sums, stage 2, all-reduce of the
      [3][K][D] sums, every rank solves and evaluates the zonal-grid epilogue redundantly; ``run`` returns the whole.

    ``tail``: "auto" (sliced when possible), "replicated", or "sliced" (raise when not possible)."""

    def __init__(self, backend, group=None, tail="auto"):
        self.backend = backend
        self.group = group
        self.world = _world(group)
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.tail = tail
        self._sliced = None                            # decided at the first step (the plan's TEM shape is set later)
        self._buf = {}
        G = backend.matrix(MAT_GRAM)
        allreduce_sum_(G, group)                       # (i) Gram matrix, K x K, once
        backend.finalize(G.detach().cpu().numpy())
        if hasattr(backend, "refine"):                 # (i') second pass of the re-orthogonalisation (temx.h)
            G2 = backend.matrix(MAT_GRAM2)
            allreduce_sum_(G2, group)
            backend.refine(G2.detach().cpu().numpy())

    def set_tem(self, nlev, nt, p_pa, p0=101325.0):
        """``backend.set_tem`` with the reference subsample of the single sweep spread over the ranks, followed by
        the plan-build collectives of the time-sliced form.  (Calling ``backend.set_tem`` directly works too: the
        collectives then run at the first step.)"""
        be = self.backend
        if hasattr(be, "configure") and self.tail != "replicated":
            # the subsample spread over the ranks; and the single sweep chosen by the size of the JOB: with a
            # time-sliced tail a rank contracts 1 / world of the columns, so the threshold scales with it
            be.configure(os_subsample=max(8, -(-32 // self.world)), single_sweep_min_groups=max(64, 640 // self.world))
        be.set_tem(nlev, nt, p_pa, p0)
        self._decide()

    def _decide(self):
        """Collective: do all ranks run the single sweep, and does the time axis cut into ``world`` slices?"""
        be = self.backend
        ok = (self.tail != "replicated" and bool(getattr(be, "single_sweep", False))
              and getattr(be, "nt", 0) is not None and int(getattr(be, "nt", 0) or 0) >= self.world)
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64,
                            device=getattr(be, "device", torch.device("cpu")))
        if self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        self._sliced = bool(flag.item() > 0.5)
        if self._sliced:
            # the two matrices of the single sweep that sum over the rows (temx_plan_set_os_matrices)
            Gx, Gs = be.matrix(MAT_GX), be.matrix(MAT_GSUB)
            allreduce_sum_(Gx, self.group)
            allreduce_sum_(Gs, self.group)
            be.set_os_matrices(Gx.detach().cpu().numpy(), Gs.detach().cpu().numpy())
        elif self.tail == "sliced":
            raise RuntimeError("NcolShardedTEM(tail='sliced'): not every rank's plan runs the single sweep, or nt < world")
        elif bool(getattr(be, "single_sweep", False)):
            # some rank cannot: all ranks take the class-sum form (its stage interface is what the replicated flow drives)
            args = be.tem_args
            be.configure(form="no-single-sweep")
            be.set_tem(*args)
        return self._sliced

    @property
    def sliced(self):
        return bool(self._sliced)

    # message buffers of the time-sliced step, kept between steps (allocated zeroed: the padding of a ragged slice is
    # never written and never read, it only travels)
    def _buffer(self, name, shape):
        be = self.backend
        key = (name, tuple(shape))
        if key not in self._buf:
            self._buf[key] = torch.zeros(tuple(shape), dtype=torch.float64, device=getattr(be, "device", torch.device("cpu")))
        return self._buf[key]

    def _slices(self, name, rows):
        be = self.backend
        ntmax = -(-int(be.nt) // self.world)
        shape = (rows, be.nlev, be.nt) if self.world == 1 else (self.world, rows * be.nlev * ntmax)
        return self._buffer(name, shape)

    def my_snapshots(self):
        """(t0, t1): the snapshots this rank's results describe (the whole run with a replicated tail)."""
        nt = int(self.backend.nt)
        return shard_bounds(nt, self.world, self.rank) if self._sliced else (0, nt)

    def run(self, ua, va, ta, wap, want_zonal=False):
        be = self.backend
        if self._sliced is None:
            self._decide()
        if self._sliced:
            As = be.tem_os_prepass(ua, va, ta, wap, out=self._buffer("As", (4, be.KR, be.D)))
            allreduce_sum_(As, self.group)             # (ii) reference pre-pass sums [4][KR][D], one message
            proj = be.tem_os_sweep(ua, va, ta, wap, As, nslices=self.world, out=self._slices("proj", be.os_rows))
            mine = reduce_scatter_sum(proj, self.group, out=self._buffer("mine", proj.shape[1:]))    # (iii) projections, one time slice per rank
            t0, t1 = shard_bounds(int(be.nt), self.world, self.rank)
            return be.tem_os_tail(mine, t0, t1 - t0, want_zonal)
        B4 = be.tem_stage1(ua, va, ta, wap)
        allreduce_sum_(B4, self.group)                 # (ii) [4][K][D] zonal sums, one message
        # one-pass class path: stage 2 works from the class sums stage 1 just stored for these fields
        if getattr(be, "one_pass", False):
            B3 = be.tem_stage2_from_sums(B4)
        else:
            B3 = be.tem_stage2(ua, va, ta, wap, B4)
        allreduce_sum_(B3, self.group)                 # (iii) [3][K][D] flux sums, one message
        return be.tem_stage3(B3, want_zonal)

    def run_tracer(self, q, va, wap, want_zonal=False):
        """Tracer TEM for one tracer; call after ``run`` on the same fields.  Time-sliced tail: all-reduce of q's
        pre-pass sums, reduce-scatter of its projections; otherwise two more all-reduces ([K][D] sums of q,
        [2][K][D] sums of q'v', q'w')."""
        be = self.backend
        if self._sliced:
            return self.run_tracers([q], va, wap, want_zonal)[0]
        if getattr(be, "tracer_one_pass", False):      # (q, v, omega) read once, see include/temx.h
            Bq = be.tracer_stage1_sums(q, va, wap)
            allreduce_sum_(Bq, self.group)
            Bq2 = be.tracer_stage2_from_sums(Bq)
        else:
            Bq = be.tracer_stage1(q)
            allreduce_sum_(Bq, self.group)
            Bq2 = be.tracer_stage2(q, va, wap, Bq)
        allreduce_sum_(Bq2, self.group)
        return be.tracer_stage3(Bq2, want_zonal)


    def _run_tracers_sliced(self, qs, va, wap, want_zonal):
        be = self.backend
        nq = len(qs)
        Asq = be.tracers_os_prepass(qs, va, wap, out=self._buffer("Asq%d" % nq, (nq, be.KR, be.D)))
        allreduce_sum_(Asq, self.group)
        projq = be.tracers_os_sweep(qs, va, wap, Asq, nslices=self.world,
                                    out=self._slices("projq%d" % nq, nq * (be.KX + 2 * be.K)))
        mine = reduce_scatter_sum(projq, self.group, out=self._buffer("mineq%d" % nq, projq.shape[1:]))
        t0, t1 = shard_bounds(int(be.nt), self.world, self.rank)
        return be.tracers_os_tail(nq, mine, t1 - t0, want_zonal)

    def run_tracers(self, qs, va, wap, want_zonal=False):
        """All tracers of the run (tem_diagnostics.py:281-301 takes a list) -> [(tres, tzon)].  Time-sliced tail: two
        tracers per sweep ((q1, q2, v, omega) read once), one all-reduce and one reduce-scatter per pair."""
        qs = list(qs)
        if not self._sliced:
            return [self.run_tracer(q, va, wap, want_zonal) for q in qs]
        out = []
        for i in range(0, len(qs), 2):
            out += self._run_tracers_sliced(qs[i:i + 2], va, wap, want_zonal)
        return out


class TimeShardedTEM:
    """Replicated plan, private time block per rank; outputs stay sharded along time."""

    def __init__(self, backend, nt_total, rank=None, world=None, group=None, multiple=1):
        self.backend = backend
        self.world = _world(group) if world is None else world
        self.rank = (dist.get_rank(group) if self.world > 1 and rank is None else (rank or 0))
        self.t0, self.t1 = shard_bounds(nt_total, self.world, self.rank, multiple)

    def run(self, ua, va, ta, wap, want_zonal=False):
        return self.backend.tem_run(ua, va, ta, wap, want_zonal)


def gather_time(res, group=None):
    """Concatenate time-sharded results [R][M][nlev][nt_local] along time on every rank
    (ragged blocks are padded to the longest one for the collective, then trimmed)."""
    w = _world(group)
    if w == 1:
        return res
    n = torch.tensor([res.shape[-1]], dtype=torch.int64, device=res.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = torch.zeros(res.shape[:-1] + (nmax,), dtype=res.dtype, device=res.device)
    pad[..., :res.shape[-1]] = res
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[..., :k] for o, k in zip(outs, sizes)], dim=-1)

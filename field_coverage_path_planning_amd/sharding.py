"""One process per GPU: shard a batch of independent fields over the ranks of a torch.distributed job.

Fields never interact, so there is no data-path collective: each rank plans a contiguous block of fields,
cut on the analytic point counts so that every GPU gets about the same number of POINTS (not fields).  The only
communication is the final gather of the per-field stats (104 B per field) and, on request, of the point arrays, over RCCL
(`torch.distributed` backend "nccl" on ROCm) -- or gloo for the CPU tests, which exercise exactly this file.

Results do not depend on the shard count: tiles are anchored at field starts and every reduction runs in a fixed order.
"""
import numpy as np

from . import _lib as L
from . import engine as E


def partition_by_points(point_counts, world_size):
    """Contiguous blocks [lo, hi) of fields, one per rank, cut where the running point total crosses k/world of the total.

    Deterministic, every field in exactly one block, blocks in rank order (some may be empty when fields are few)."""
    counts = np.asarray(point_counts, dtype=np.int64)
    n = len(counts)
    if world_size <= 0:
        raise ValueError('world_size must be positive')
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first field boundary whose running total reaches the target, never before the previous cut
        k = int(np.searchsorted(csum, target, side='left'))
        k = min(max(k, cuts[-1]), n)
        cuts.append(k)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


# Exercising the collectives where only ONE GPU exists (tests/test_gpu_rccl.py, bench.py under FCPP_BENCH_FORCE_DIST=1): with this
# set, a world of one rank does not take the "nothing to exchange" shortcuts -- the stats rows, the point arrays and the GA fitness
# really travel through the process group (ncclSend / ncclRecv addressed to the rank itself, ncclAllGather over one rank), on device
# tensors with the nccl backend, so the RCCL code path a multi-GPU job takes has run on the hardware at hand.
FORCE_COLLECTIVES = False


def _dist():
    import torch.distributed as dist
    return dist


def _send_to_self(tensors):
    """every tensor through one batch of isend / irecv addressed to this rank -> the received copies"""
    import torch
    dist = _dist()
    rank, _ = world()
    out = [torch.empty_like(t) for t in tensors]
    ops = []
    for t, o in zip(tensors, out):
        if t.numel() > 0:
            ops.append(dist.P2POp(dist.isend, t.contiguous(), rank))
            ops.append(dist.P2POp(dist.irecv, o, rank))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return out


def world():
    """(rank, world_size) of the default process group, (0, 1) when torch.distributed is not initialised."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_rows(local_rows, rows_per_rank, dst=0):
    """Gather 2-D tensors with a different number of rows per rank to `dst` (None elsewhere).

    rows_per_rank is known to every rank (it follows from the host-side partition), so no size exchange is needed: ranks
    send their block straight to the root (point-to-point over xGMI with the nccl backend: every peer uses its own link)."""
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        if FORCE_COLLECTIVES and dist.is_initialized():
            return _send_to_self([local_rows])[0]
        return local_rows
    # ONE batch of point-to-point operations (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd with the nccl backend): the root posts all
    # its receives together -- posted one by one they are served one after the other -- as gather_arrays does for the point arrays
    ops, parts = [], []
    if rank == dst:
        for r in range(ws):
            if r == dst:
                parts.append(local_rows)
                continue
            buf = torch.empty((rows_per_rank[r],) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype,
                              device=local_rows.device)
            parts.append(buf)
            if rows_per_rank[r] > 0:
                ops.append(dist.P2POp(dist.irecv, buf, r))
    elif local_rows.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, local_rows.contiguous(), dst))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return torch.cat(parts, dim=0) if rank == dst else None


def arena_point_arrays(total, like):
    """The root's full point arrays (x, y, kappa, v: float64, flagseg: int32) out of the context's OUTPUT ARENA when `like` -- this rank's
    five arrays -- live on a device whose context has one with room: the gathered arrays then lie a pitch apart like every batch's own
    (DESIGN.md section 2), instead of back to back wherever the allocator puts them.  -> list of five tensors, or None."""
    import torch
    if len(like) != 5 or not all(a.is_cuda for a in like) or [a.dtype for a in like] != [torch.float64] * 4 + [torch.int32]:
        return None
    ctx = E.get_context(like[0].device.index)
    lane, pitch = ctx.arena()
    if lane < 8 * int(total) or total <= 0:
        return None
    arr = E._ArenaArrays(ctx, int(total))
    if arr.ptrs[1] - arr.ptrs[0] != pitch:          # (the arena was full: the library's own allocation -- fine, but no better than torch's)
        return None
    return arr.tensors(int(total))


def gather_arrays(local_arrays, counts_per_rank, dst=0, alloc=None):
    """Optional gather of the point arrays: every rank holds 1-D tensors of its own block (x, y, kappa, v,
    flagseg, ...: the same list, in the same order, on every rank; counts_per_rank[r] elements each on rank r).  The root
    allocates each full array once -- alloc(total, local_arrays) -> list of tensors or None: the caller's allocator, e.g.
    arena_point_arrays -- and receives every peer's block straight into its slice -- one batch of point-to-point
    transfers (ncclGroupStart / ncclSend / ncclRecv with the nccl backend: each peer crosses its own xGMI link to the root,
    no ring, no staging copy, no concatenation).  -> list of full tensors on `dst`, None elsewhere."""
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        if FORCE_COLLECTIVES and dist.is_initialized():
            return _send_to_self([a[:int(counts_per_rank[0])] for a in local_arrays])
        return list(local_arrays)
    starts = np.concatenate([[0], np.cumsum(np.asarray(counts_per_rank, dtype=np.int64))])
    ops, out = [], None
    if rank == dst:
        out = []
        given = alloc(int(starts[-1]), local_arrays) if alloc is not None else None
        for k, a in enumerate(local_arrays):
            full = given[k] if given is not None else torch.empty(int(starts[-1]), dtype=a.dtype, device=a.device)
            full[int(starts[dst]):int(starts[dst + 1])] = a[:int(counts_per_rank[dst])]
            for r in range(ws):
                if r != dst and counts_per_rank[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, full[int(starts[r]):int(starts[r + 1])], r))
            out.append(full)
    elif counts_per_rank[rank] > 0:
        for a in local_arrays:
            ops.append(dist.P2POp(dist.isend, a[:int(counts_per_rank[rank])].contiguous(), dst))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return out


class ShardedResult:
    def __init__(self, block, local, stats_all, infos, blocks=None, points_all=None, batch=None):
        self.block = block            # (lo, hi) fields of this rank
        self.local = local            # engine.BatchResult of this rank's block (device tensors), None if the block is empty
        self.stats_all = stats_all    # rank 0: (n_fields, 13) int64 tensor of fcpp_field_stats for ALL fields; else None
        self.infos = infos            # fcpp_field_info of every field of the whole batch (host-side, same on every rank)
        self.blocks = blocks          # [(lo, hi)] of every rank
        self.points_all = points_all  # rank 0 with gather_points: (x, y, kappa, v, flagseg) of the WHOLE batch; else None
        self.batch = batch            # this rank's engine.Batch (None for an empty block / the CPU tests)

    def stats(self):
        """rank 0: dict of numpy arrays over all fields (same layout as BatchResult.stats())."""
        if self.stats_all is None:
            return None
        raw = self.stats_all.cpu().numpy()
        out = {}
        for k, (n, _) in enumerate(L.FieldStats._fields_):
            col = raw[:, k]
            out[n] = col.view(np.float64).copy() if k < L.STATS_DOUBLES else col.copy()
        return out


def _comm_tensor(t):
    """what the process group can carry: device tensors with nccl (RCCL), host tensors with gloo (the CPU tests, and the
    one-GPU rehearsal of a multi-rank job)"""
    return t.cpu() if _dist().is_initialized() and _dist().get_backend() == 'gloo' else t


def plan_sharded(specs, vehicle, options=None, device=None, compute=None, mode=1, gather_points=False, batch=None, buffers=None, infos=None,
                 counts=None):
    """Plan `specs` (a list of engine.FieldSpec or an engine.FieldTable) across all ranks of the current process group; per-field stats are gathered to rank 0, and so are the point
    arrays (x, y, kappa, v, flagseg) when gather_points is set.

    compute(specs_block, vehicle, options) -> (n_block, 13) int64 tensor [, list of 1-D point arrays] replaces the GPU batch in
    the CPU (gloo) tests.  batch / buffers: reuse this rank's engine.Batch (and output buffers) of an earlier call with the same
    specs -- the setup (fcpp_batch_create) is then skipped, as a caller that plans the same fields repeatedly would; infos: the
    fcpp_plan_count result of an earlier call, or counts: the points per field (engine.plan_points) -- the sizing is then skipped too.

    The sizing -- what every rank needs of ALL fields to cut the blocks -- is the points per field: on a GPU rank it comes from the
    device (engine.plan_points: the plan function one thread per field, 8 bytes per field back; identical on every rank, so no collective
    is needed to agree on the partition), in the CPU tests from fcpp_plan_count on the host."""
    import torch
    options = options or E.make_options()
    rank, ws = world()
    if counts is None and infos is None and compute is None and ws == 1:
        blocks = [(0, len(specs))]        # one rank: nothing to cut, nothing to size (res.counts stays None: batch.info.counts() has them)
    else:
        if counts is None:
            if infos is None and compute is None:
                counts = E.plan_points(specs, vehicle, options, device=device)
            else:
                if infos is None:
                    infos = E.plan_count(specs, vehicle, options)
                counts = infos.counts() if hasattr(infos, 'counts') else np.asarray([i.n_main + i.n_head for i in infos], dtype=np.int64)
        counts = np.asarray(counts, dtype=np.int64)
        blocks = partition_by_points(counts, ws)
    lo, hi = blocks[rank]
    local, arrays = None, None
    if compute is not None:
        got = compute(specs[lo:hi], vehicle, options)
        stats_local, arrays = got if isinstance(got, tuple) else (got, None)
    elif hi > lo:
        if batch is None:
            batch = E.Batch(specs[lo:hi], vehicle, options, device=device)
        local = batch.run(buffers, mode=mode)
        stats_local = local.stats_raw
        arrays = [local.x, local.y, local.kappa, local.v, local.flagseg]
    else:
        dev = torch.device('cuda', device if device is not None else torch.cuda.current_device())
        stats_local = torch.zeros((0, L.STATS_WORDS), dtype=torch.int64, device=dev)
        arrays = [torch.empty(0, dtype=torch.float64, device=dev) for _ in range(4)] + [torch.empty(0, dtype=torch.int32, device=dev)]
    stats_all = gather_rows(_comm_tensor(stats_local), [b[1] - b[0] for b in blocks], dst=0)
    points_all = None
    if gather_points:
        per_rank = [int(np.sum(counts[a:b])) for a, b in blocks] if counts is not None else [int(batch.total_points) if batch is not None else 0]
        # (the root's full arrays out of the output arena when the transfers carry device tensors: RCCL; the gloo rehearsal gathers on the host)
        points_all = gather_arrays([_comm_tensor(a) for a in arrays], per_rank, dst=0, alloc=arena_point_arrays)
    res = ShardedResult((lo, hi), local, stats_all, infos, blocks, points_all, batch)
    res.counts = counts           # points per field of the whole batch (None: one rank planned everything without sizing)
    return res


def partition_even(n_items, world_size):
    """Contiguous blocks [lo, hi) of n_items equal-cost items (GA chromosomes), one per rank, in rank order; the first
    n_items % world_size ranks get one item more."""
    if world_size <= 0:
        raise ValueError('world_size must be positive')
    q, r = divmod(int(n_items), world_size)
    cuts = [0]
    for k in range(world_size):
        cuts.append(cuts[-1] + q + (1 if k < r else 0))
    return [(cuts[k], cuts[k + 1]) for k in range(world_size)]


def ga_fitness_sharded(routes, D, order_mode=0, device=None, compute=None, with_distance=False):
    """Evaluate a GA population across the ranks ("shard population across GPUs, all-gather 8 B fitness per
    chromosome"; GA:168-181).  Every rank holds the same `routes` (pop, n) and D (n, n) -- the population of a generation is
    produced from the same counter-based draws on every rank --, evaluates its contiguous block of chromosomes with
    fcpp_ga_fitness, and the blocks are all-gathered, so every rank returns the fitness of the WHOLE population (what selection
    needs next), identical to one process evaluating it: a chromosome's tour is summed by one wavefront in a fixed order wherever
    it is evaluated.  with_distance: also all-gather the tour lengths (another 8 B per chromosome).
    -> fitness (pop,) float64 tensor [, distance (pop,) float64 tensor]

    compute(routes_block, D, order_mode) -> (distance, fitness) 1-D float64 tensors replaces the GPU call in the CPU (gloo) tests."""
    import torch
    dist = _dist()
    rank, ws = world()
    r = routes if isinstance(routes, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(routes, dtype=np.int32))
    n = int(D.shape[0])
    r = r.reshape(-1, n)
    pop = int(r.shape[0])
    blocks = partition_even(pop, ws)
    lo, hi = blocks[rank]
    if compute is not None:
        d_loc, f_loc = compute(r[lo:hi], D, order_mode)
    elif hi > lo:
        d_loc, f_loc = E.ga_fitness(r[lo:hi], D, order_mode, device=device)
    else:
        dev = torch.device('cuda', device if device is not None else torch.cuda.current_device())
        d_loc = f_loc = torch.empty(0, dtype=torch.float64, device=dev)
    if ws == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
        return (f_loc, d_loc) if with_distance else f_loc
    # one all-gather of equal-sized blocks (the last ranks' blocks padded by at most one element): a single collective whatever the
    # population size, [fitness | distance] side by side when both are asked for
    width = blocks[0][1] - blocks[0][0]
    cols = 2 if with_distance else 1
    send = torch.zeros((cols, width), dtype=torch.float64, device=f_loc.device)
    send[0, :hi - lo] = f_loc
    if with_distance:
        send[1, :hi - lo] = d_loc
    send = _comm_tensor(send)
    recv = [torch.empty_like(send) for _ in range(ws)]
    dist.all_gather(recv, send)
    out = [torch.cat([recv[k][c, :blocks[k][1] - blocks[k][0]] for k in range(ws)]).to(f_loc.device) for c in range(cols)]
    return (out[0], out[1]) if with_distance else out[0]

#!/usr/bin/env python3
"""bench.py -- validated path points/sec of the HIP hot path on MI355X.

One "step" = one pass of the whole hot path (sample every path point, curvature, curvature clamp, forward/backward speed
sweeps, a_lat / geofence / obstacle validation, per-field metrics) over one batch of synthetic fields whose descriptors are
already resident in HBM.

Headline (the workload BASELINE.json's metric names): a batch of 4096 fields of 500 x 200 m (BASELINE.json configs[0], the
reference's own case) per GPU, planned in the reference's own model -- circular arcs at the reference's sampling (2 points per
swath line, 20 per U-turn, 15 per corner), the mode that is pinned to the reference's outputs (tests/golden) -- 1691 points per
field.  With --gpus N every rank plans its own such batch (weak scaling) and the per-field stats go to rank 0 over RCCL.

The same JSON line carries a `configs` array: the same batch with clothoid turns, and BASELINE.json's other configurations --
cfg2 (1024 random rectangles; reference sampling, 0.5 m, 0.1 m), cfg3 (5000 x 2000 m, 32 obstacles, 0.05 m), cfg4 (GA, 128 nodes,
population 4096, 500 generations) and cfg5 (65 536 parallelograms) -- each with its points, ms, points/s, per-kernel HIP-event
times, roofline fraction of its dominant kernel and its own CPU baseline.  With --gpus N > 1 cfg5 is the SHARDED job
(sharding.plan_sharded: contiguous blocks cut on the analytic point counts, strong scaling) with per-GPU and aggregate rates and
the time of the optional point-array gather; cfg2 at 0.1 m stays as a weak-scaled second figure.

`python bench.py --gpus N` starts its own workers (torch.distributed.run, one per GPU) before this process touches the GPU;
under an external launcher (RANK / WORLD_SIZE set) it is a worker itself.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BYTES_PER_POINT = 36          # x, y, kappa, v as float64 + one uint32 flag/segment word
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = 'validated path points/sec (Clothoid+speed+geofence) on 500x200m field batch'      # BASELINE.json, verbatim


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--fields', type=int, default=4096, help='fields of 500 x 200 m per GPU in the headline batch')
    ap.add_argument('--mode', type=int, default=1, help='1 = fused pipeline (default), 0 = staged pipeline')
    ap.add_argument('--configs', default='all',
                    help="'all', 'none' or a comma list of: cfg1_clothoid,cfg1_clothoid_dense,cfg2_ref,cfg2_0.5,cfg2_0.1,cfg3,cfg4,cfg5")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=8.0, help='seconds of CPU baseline for the headline (the other configs get 3 s each)')
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: become the launcher.  Nothing in this process has touched the GPU (no torch import),
    the workers are children, and this process exits with their code."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


# ---- CPU baseline: the C oracle on a bounded sample of the same workload, on the box's own host cores ------------------------------
def cpu_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return min(cores, 16)           # a one-GPU box's CPU share is 16 cores, whatever the host exposes


def cpu_baseline_fields(make_ofield, n_fields, oopt, budget_s, what):
    """oracle.plan_field over the first fields of the workload: one thread first, then every core (fields are independent; ctypes
    releases the GIL inside the C call), about budget_s / 2 seconds each."""
    import oracle as orc
    from concurrent.futures import ThreadPoolExecutor
    veh = orc.Vehicle.make()

    def plan(k):
        rc, p = orc.plan_field(make_ofield(k % n_fields), veh, oopt)
        assert rc == 0
        return p.n

    pts1, t1, k = 0, 0.0, 0
    while t1 < budget_s / 2 and k < 4 * n_fields:
        t0 = time.perf_counter()
        pts1 += plan(k)
        t1 += time.perf_counter() - t0
        k += 1
    cores = cpu_cores()
    m = max(cores, int(cores * (budget_s / 2) / (t1 / k)))
    m = min(m, 64 * n_fields)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        ptsn = sum(ex.map(plan, range(m), chunksize=max(1, m // (8 * cores))))
    tn = time.perf_counter() - t0
    return {'value': ptsn / tn, 'unit': 'points/s', 'cores': cores, 'kind': 'port', 'single_core_value': pts1 / t1,
            'sample': f'{m} plans of {what} ({ptsn} points, {tn:.1f} s wall on {cores} threads; one thread: {k} plans, {pts1} points, '
                      f'{t1:.1f} s) through oracle/fcpp_oracle.c (sequential C restatement of the reference loops, gcc -O2)'}


# ---- one planner configuration on this rank --------------------------------------------------------------------------------------
def run_planner(E, torch, specs, opt, steps, warmup, mode=1, placement=1, fence=None, after_step=None, stats_of=None):
    """-> dict(points, ms_per_step, kernels {name: ms}, dominant kernel + its points, batch, bufs, res): K timed steps bracketed by
    fence() (barrier + synchronize), per-kernel HIP events recorded inside the timed region.  stats_of(i, batch): the stats tensor step i
    writes (default: the one of alloc()); after_step(i, res): called after step i has been enqueued."""
    fence = fence or torch.cuda.synchronize
    batch = E.Batch(specs, E.make_vehicle(), opt)
    plain_ms = None
    if placement > 1:
        # the plain allocation first (its figure is reported beside the calibrated one), then the fastest of it and `placement` more
        # candidate sets under the batch's own step (Batch.alloc(best_of): setup only, engine.py)
        bufs0 = batch.alloc()
        for _ in range(2):
            batch.run(bufs0, mode=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            batch.run(bufs0, mode=mode)
        torch.cuda.synchronize()
        plain_ms = (time.perf_counter() - t0) / steps * 1e3
        bufs = batch.alloc(best_of=placement, include=[bufs0])
        del bufs0
    else:
        bufs = batch.alloc()
    res = None
    k = 0

    def step():
        nonlocal res, k
        res = batch.run(bufs[:5] + (stats_of(k, batch),) if stats_of else bufs, mode=mode)
        if after_step:
            after_step(k, res)
        k += 1

    for _ in range(warmup):
        step()
    fence()
    # per-kernel HIP events inside the timed region, on a sample of its steps (a profiled step dispatches every kernel with its own
    # start / stop events and costs ~15 us more than a plain one: on every step that would be 14 % of the headline's 110 us)
    batch.set_profiling(True, every=max(8, steps // 8))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kernels, prof_runs = batch.stage_times()
    batch.set_profiling(False)
    q_pts, g_pts = batch.point_split()
    n_points = batch.total_points
    dom = max(kernels, key=kernels.get)
    stage_points = batch.stage_points()
    dom_points = stage_points[dom]
    return {'stage_points': stage_points, 'prof_runs': prof_runs,'points': n_points, 'dt': dt, 'ms_per_step': dt / steps * 1e3, 'kernels': kernels, 'dominant': dom, 'dominant_points': dom_points,
            'quiet_points': q_pts, 'general_points': g_pts, 'batch': batch, 'bufs': bufs, 'res': res, 'placement': getattr(batch, 'placement', None),
            'plain_ms': plain_ms}


def roofline_of(r, traffic_key=None):
    dom, dom_ms, dom_points = r['dominant'], r['kernels'][r['dominant']], r['dominant_points']
    achieved = BYTES_PER_POINT * dom_points / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    pipe_ms = sum(r['kernels'].values())
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'traffic.json')
    if traffic_key and os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(f'{dom}|{traffic_key}')
    note = None
    if dom == 'k_plan_sparse':
        # (the secondary ceiling.  Counters of the kept profiles, not measured in this run.)
        note = ('k_plan_sparse is bound by fp64 vector issue, not by HBM: 640-680 vector instructions per 64-lane wavefront of ~52 output points, '
                'vector ALU busy ~100 % of the kernel (profiles/r02_counter_table.txt); its HBM fraction is low by construction, the '
                'step-level figure is step_frac')
    return {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'note': note,
            'traffic': traffic, 'kernel_ms': dom_ms, 'algorithmic_bytes_per_launch': BYTES_PER_POINT * dom_points,
            'kernel_points_per_launch': dom_points, 'all_kernels_ms': r['kernels'], 'all_kernels_points': r['stage_points'],
            'profiled_steps': r['prof_runs'], 'pipeline_ms': pipe_ms,
            'pipeline_frac': (BYTES_PER_POINT * r['points'] / (pipe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pipe_ms > 0 else 0.0,
            'step_frac': BYTES_PER_POINT * r['points'] / (r['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS}


def config_entry(name, workload, r, cpu=None, extra=None, traffic_key=None):
    e = {'name': name, 'workload': workload, 'points': r['points'], 'ms_per_step': r['ms_per_step'],
         'value': r['points'] / (r['ms_per_step'] * 1e-3), 'unit': 'points/s', 'dtype': 'f64',
         'quiet_points': r['quiet_points'], 'general_points': r['general_points'], 'roofline': roofline_of(r, traffic_key or name), 'cpu_baseline': cpu}
    if r.get('placement'):
        e['placement'] = r['placement']
    if r.get('plain_ms'):
        e['placement1'] = {'ms_per_step': r['plain_ms'], 'value': r['points'] / (r['plain_ms'] * 1e-3),
                           'step_frac': BYTES_PER_POINT * r['points'] / (r['plain_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           'note': 'output arrays as the allocator returns them (device time of the same steps, no per-kernel events)'}
    if extra:
        e.update(extra)
    return e


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from field_coverage_path_planning_amd import engine as E
    from field_coverage_path_planning_amd import sharding as S
    from field_coverage_path_planning_amd import workloads as WL

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE {world}')
    # FCPP_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives carry host tensors); the measured configuration is nccl = RCCL, one rank per GPU
    backend = os.environ.get('FCPP_BENCH_BACKEND', 'nccl')
    local = local % torch.cuda.device_count() if backend == 'gloo' else local
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cdev = dev if backend == 'nccl' else torch.device('cpu')          # where the collectives' tensors live
    if world > 1:
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
    # everything runs on a stream of its own: on the legacy default stream the same launches take 3-8 % longer at the 0.05-0.1 ms
    # steps of the reference's sampling (0.0855 vs 0.0827 ms on cfg1 x 4096, 0.050 vs 0.047 ms on cfg2; larger steps do not care)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    # device warm-up, before any step of the bench: a fresh box starts at idle clocks, and the first timed region (the headline: K steps
    # of 0.09 ms) would otherwise carry their ramp (observed once: 0.29 instead of 0.09 ms per step)
    wa = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            wa = (wa @ wa).clamp_(-1.0, 1.0)
        torch.cuda.synchronize()
    del wa

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    want = args.configs.split(',') if args.configs not in ('all', 'none') else (
        ['cfg1_clothoid', 'cfg1_clothoid_dense', 'cfg2_ref', 'cfg2_0.5', 'cfg2_0.1', 'cfg3', 'cfg4', 'cfg5'] if args.configs == 'all' else [])
    cpu_on = world == 1 and not args.no_cpu_baseline

    # ---- headline: 4096 x (500 x 200 m) per GPU, arcs at the reference's sampling (the pinned mode) ------------------------------
    # the only collective of the path: the gather of every step's per-field stats on rank 0.  Every step writes its stats into the next
    # slot of a ring on the device; after GATHER_EVERY steps their slots go to the root in ONE asynchronous gather on RCCL's own
    # stream (a collective per 0.1 ms step would cost more host time than the step's kernels take), double-buffered: while one half of
    # the ring travels, the steps fill the other.
    LH1 = WL.cfg1_batch(args.fields)
    GATHER_EVERY = 8
    pending = []
    ring = gather_bufs = None
    steps_done = [0]

    def stats_slot(i, batch):
        nonlocal ring
        if ring is None:
            ring = torch.zeros((2 * GATHER_EVERY, batch.n_fields, E.L.STATS_WORDS), dtype=torch.int64, device=dev)
        return ring[i % (2 * GATHER_EVERY)]

    def send_group(g):
        """gather the ring half that holds the stats of steps [8 g, 8 g + 8)"""
        nonlocal gather_bufs
        if gather_bufs is None:
            gather_bufs = [[torch.empty((GATHER_EVERY,) + tuple(ring.shape[1:]), dtype=ring.dtype, device=cdev) for _ in range(world)]
                           if rank == 0 else None for _ in range(2)]
        if pending:
            pending[-1].wait()          # the other half has left before the coming steps refill it (for RCCL: a stream-side wait)
        half = g & 1
        part = ring[half * GATHER_EVERY:(half + 1) * GATHER_EVERY]
        pending.append(dist.gather(part if backend == 'nccl' else part.cpu(), gather_bufs[half], dst=0, async_op=True))

    def count_and_gather(i, res):
        steps_done[0] = i + 1
        if world > 1 and (i + 1) % GATHER_EVERY == 0:
            send_group(i // GATHER_EVERY)

    def fence_headline():
        if world > 1:
            if steps_done[0] % GATHER_EVERY:          # an unfinished group at a fence travels as it is (and again when it is complete)
                send_group(steps_done[0] // GATHER_EVERY)
            if pending:
                pending[-1].wait()
        fence()

    r = run_planner(E, torch, WL.specs_from_lh(E, LH1), E.make_options(), args.steps, args.warmup, mode=args.mode, fence=fence_headline,
                    after_step=count_and_gather, stats_of=stats_slot if world > 1 else None)
    dt = allmax(r['dt'])
    total_points = allsum(r['points'])
    out = None
    if rank == 0:
        st = r['res'].stats()
        assert int(st['n_viol'].sum()) == 0 and np.isfinite(st['main_len_m']).all()
        assert (r['batch'].info[0].n_main, r['batch'].info[0].n_head) == (1256, 435)          # README_en.md:206-207
        out = {
            'metric': METRIC, 'value': total_points * args.steps / dt, 'unit': 'points/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': f'cfg1 x {args.fields}: batch of {args.fields} fields of 500 x 200 m per GPU (BASELINE.json configs[0], README_en.md:199-215), '
                            f'default VehicleParams, the reference\'s own model: circular-arc turns at the reference\'s sampling '
                            f'(2 / 20 / 15 / 20 points per line / U-turn / corner / headland side), 1691 points per field -- the mode pinned to '
                            f'the reference\'s outputs; the clothoid variants of the same batch are configs[cfg1_clothoid*] below',
                'points_per_gpu_step': r['points'], 'fields_per_gpu': args.fields,
                'pipeline': 'staged (7 kernels)' if args.mode == 0 else 'fused: k_plan_quiet (closed-form runs and spans) + k_plan_sparse (wave tiles, one point per lane) + k_plan_fused (all other tiles) + k_reduce_stats',
                'quiet_points': r['quiet_points'], 'general_points': r['general_points'],
            },
            'roofline': roofline_of(r, 'cfg1'),
            'cpu_baseline': None,
        }
    if rank == 0 and cpu_on:
        import oracle as orc
        out['cpu_baseline'] = cpu_baseline_fields(lambda k: orc.make_field(L=float(LH1[k, 0]), H=float(LH1[k, 1])), len(LH1), orc.Options.make(),
                                                  args.cpu_budget, '500 x 200 m fields, arcs, reference sampling')
    r['batch'].close()
    del r
    torch.cuda.empty_cache()

    # ---- the other configurations --------------------------------------------------------------------------------------------------
    configs = []

    def planner_config(name, workload, specs, opt, steps, warmup, make_ofield=None, n_ofields=0, oopt=None, placement=1, what='', extra_fn=None,
                       budget=3.0):
        rr = run_planner(E, torch, specs, opt, steps, warmup, placement=placement, fence=fence)
        cpu = None
        if rank == 0 and cpu_on and make_ofield is not None:
            cpu = cpu_baseline_fields(make_ofield, n_ofields, oopt, budget, what)
        extra = extra_fn(rr) if extra_fn else None
        if rank == 0:
            configs.append(config_entry(name, workload, rr, cpu, extra))
        rr['batch'].close()
        del rr
        torch.cuda.empty_cache()

    if world == 1:
        import oracle as orc
        if 'cfg1_clothoid' in want:
            planner_config('cfg1_clothoid', f'cfg1 x {args.fields}, clothoid turns (line-clothoid-arc-clothoid-line) at the reference\'s sample counts',
                           WL.specs_from_lh(E, LH1), E.make_options(1, 0.0), args.steps, args.warmup,
                           lambda k: orc.make_field(L=500.0, H=200.0), len(LH1), orc.Options.make(1, 1, 0.0, 0.5), what='500 x 200 m fields, clothoid, reference sampling')
        if 'cfg1_clothoid_dense' in want:
            planner_config('cfg1_clothoid_dense', f'cfg1 x {args.fields}, clothoid turns, uniform 0.1 m sample spacing; output arrays calibrated (placement1: the plain allocation)',
                           WL.specs_from_lh(E, LH1), E.make_options(1, 0.1), max(3, args.steps // 20), 2,
                           lambda k: orc.make_field(L=500.0, H=200.0), len(LH1), orc.Options.make(1, 1, 0.1, 0.5), what='500 x 200 m fields, clothoid, 0.1 m',
                           placement=2)
        LH2 = WL.cfg2_rectangles()
        for key, tm, sp, st_, wu in (('cfg2_ref', 0, 0.0, args.steps, args.warmup), ('cfg2_0.5', 1, 0.5, max(5, args.steps // 10), 2),
                                      ('cfg2_0.1', 1, 0.1, max(3, args.steps // 20), 2)):
            if key not in want:
                continue
            wl = (f'cfg2: 1024 random rectangular fields (edges U[100,1000) m, seed 1024), '
                  f'{"arc turns at the reference sampling" if sp == 0 else f"clothoid turns, {sp} m sample spacing"}')
            if key == 'cfg2_0.1':
                # the figure with the plain allocation first, then with the output arrays chosen by Batch.alloc(best_of=3): three candidate
                # sets, the batch's own step timed on each, the fastest kept (setup only)
                planner_config('cfg2_0.1_placement1', wl + ', output arrays as the allocator returns them', WL.specs_from_lh(E, LH2),
                               E.make_options(tm, sp), st_, wu)
                planner_config('cfg2_0.1', wl + ', output arrays = the fastest of 4 candidate sets (the plain allocation and 3 more) under the batch\'s own step (setup only)', WL.specs_from_lh(E, LH2),
                               E.make_options(tm, sp), st_, wu, lambda k: orc.make_field(L=float(LH2[k, 0]), H=float(LH2[k, 1])), len(LH2),
                               orc.Options.make(tm, 1, sp, 0.5), placement=3, what=f'cfg2 fields, clothoid, {sp} m')
            else:
                planner_config(key, wl + (', output arrays calibrated (placement1: the plain allocation)' if sp > 0 else ''), WL.specs_from_lh(E, LH2),
                               E.make_options(tm, sp), st_, wu,
                               lambda k: orc.make_field(L=float(LH2[k, 0]), H=float(LH2[k, 1])), len(LH2), orc.Options.make(tm, 1, sp, 0.5),
                               what=f'cfg2 fields, {"arcs, reference sampling" if sp == 0 else f"clothoid, {sp} m"}', placement=3 if sp > 0 else 1)
        if 'cfg3' in want:
            (L3, H3), obst = WL.cfg3_field()

            def cfg3_extra(rr):
                st3 = rr['res'].stats()
                return {'n_in_obstacle': int(st3['n_in_obstacle'][0]), 'n_outside': int(st3['n_outside'][0])}
            planner_config('cfg3', 'cfg3: one 5000 x 2000 m field, 32 convex eight-gon obstacles, clothoid turns, 0.05 m sample spacing; output arrays calibrated (placement1: the plain allocation)',
                           [E.FieldSpec(field_length=L3, field_width=H3, obstacles=obst)], E.make_options(1, 0.05), max(5, args.steps // 10), 2,
                           lambda k: orc.make_field(L=1000.0, H=400.0, obstacles=[[(x / 5, y / 5) for x, y in o] for o in obst]), 1,
                           orc.Options.make(1, 1, 0.05, 0.5), what='a 1000 x 400 m field with the 32 obstacles scaled by 1/5, clothoid, 0.05 m (1/25 of cfg3)',
                           extra_fn=cfg3_extra, budget=6.0, placement=3)
        if 'cfg4' in want:
            configs.append(run_cfg4(E, torch, WL, cpu_on))
    if world > 1 and 'cfg4' in want:
        entry = run_cfg4_sharded(E, S, WL, torch, dev, world, fence, allmax)
        if rank == 0:
            configs.append(entry)
    if 'cfg5' in want or world > 1:
        entry = run_cfg5(E, S, WL, torch, dist, np, rank, world, dev, cdev, fence, allmax, max(5, args.steps // 10), cpu_on)
        if rank == 0:
            configs.append(entry)
    if world > 1 and 'cfg2_0.1' in want:
        LH2 = WL.cfg2_rectangles(seed=1024 + rank)
        rr = run_planner(E, torch, WL.specs_from_lh(E, LH2), E.make_options(1, 0.1), max(3, args.steps // 20), 2, fence=fence)
        dt2, pts2 = allmax(rr['dt']), allsum(rr['points'])
        if rank == 0:
            e = config_entry('cfg2_0.1_weak', f'cfg2 weak-scaled: 1024 random rectangles PER GPU (seed 1024 + rank), clothoid, 0.1 m; rank 0\'s kernels', rr)
            e.update({'value': pts2 * max(3, args.steps // 20) / dt2, 'points': pts2, 'n_gpus': world, 'scaling': 'weak'})
            configs.append(e)
        rr['batch'].close()

    if rank == 0:
        out['configs'] = configs
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run_cfg4_sharded(E, S, WL, torch, dev, world, fence, allmax):
    """cfg4 over the ranks: the population cut into contiguous blocks, fcpp_ga_fitness per block, one all-gather of the
    fitness (8 B per chromosome) to every rank -- sharding.ga_fitness_sharded; 501 evaluations of the resident population, as
    fitness_only of the one-GPU entry."""
    D, routes = WL.cfg4_ga()
    Dd, rd = torch.as_tensor(D, device=dev), torch.as_tensor(routes, device=dev)
    fit = None
    for _ in range(5):
        fit = S.ga_fitness_sharded(rd, Dd, device=dev.index)
    same = bool(torch.equal(fit.to(dev), E.ga_fitness(rd, Dd, device=dev.index)[1]))     # every rank holds the whole population's fitness
    fence()
    t0 = time.perf_counter()
    for _ in range(501):
        S.ga_fitness_sharded(rd, Dd, device=dev.index)
    fence()
    dt = allmax(time.perf_counter() - t0)
    evals = 501 * routes.shape[0]
    return {'name': 'cfg4_fitness_sharded', 'workload': f'cfg4 population (4096 tours of 128 nodes) evaluated in {world} contiguous blocks, one per rank, '
            'fitness all-gathered to every rank after each of 501 evaluations (sharding.ga_fitness_sharded)', 'n_gpus': world, 'scaling': 'strong',
            'ms_501_evaluations': dt * 1e3, 'value': evals / dt, 'unit': 'chromosome evaluations/s', 'dtype': 'f64',
            'identical_to_one_process': same,
            'note': 'a collective per 11 us kernel: the figure is the all-gather latency, reported so that the sharded GA path has a measured line'}


def run_cfg4(E, torch, WL, cpu_on):
    """cfg4: the GA on 128 nodes, population 4096, 500 generations = 501 population evaluations (fcpp_ga_evolve)."""
    D, routes = WL.cfg4_ga()

    class Cfg:
        population_size, max_generations, crossover_rate, mutation_rate = 4096, 500, 0.85, 0.02
        elite_size, tournament_size, convergence_threshold = 20, 5, 10 ** 9          # never converges early: all 500 generations run

    Dd = torch.as_tensor(D, device='cuda')
    E.ga_evolve(Dd, routes, Cfg, seed=4096)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        _, _, hb, _, res = E.ga_evolve(Dd, routes, Cfg, seed=4096)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    evals = 501 * 4096
    # fitness alone (the SURVEY 8a-15 operator): 501 launches over the resident population
    rd = torch.as_tensor(routes, device='cuda')
    E.ga_fitness(rd, Dd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(501):
        E.ga_fitness(rd, Dd)
    torch.cuda.synchronize()
    dtf = time.perf_counter() - t0
    cpu = None
    if cpu_on:
        import oracle as orc
        t0 = time.perf_counter()
        gens = 12
        orc.ga_evolve(D, routes, max_generations=gens, convergence_threshold=10 ** 9, seed=4096)
        tc = time.perf_counter() - t0
        cpu = {'value': (gens + 1) * 4096 / tc, 'unit': 'chromosome evaluations/s', 'cores': 1, 'kind': 'port',
               'sample': f'{gens} generations of the same run (pop 4096, n 128) through oracle/fcpp_oracle.c: orc_ga_evolve, {tc:.1f} s on one thread'}
    bytes_per_chrom = 4 * 128 + 8
    return {'name': 'cfg4', 'workload': 'cfg4: GA over 128 nodes, population 4096, 500 generations (501 population evaluations), whole loop on the device',
            'generations': int(res.generations), 'ms_total': dt * 1e3, 'us_per_generation': dt / 500 * 1e6,
            'value': evals / dt, 'unit': 'chromosome evaluations/s', 'dtype': 'f64', 'gathers_per_s': evals * 128 / dt,
            'best_distance_m': float(res.best_distance),
            'fitness_only': {'ms_501_launches': dtf * 1e3, 'chromosomes_per_s': evals / dtf, 'gathers_per_s': evals * 128 / dtf},
            'roofline': {'bound': 'hbm', 'kernel': 'k_ga_generation', 'achieved': evals * bytes_per_chrom / dt / 1e9, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': evals * bytes_per_chrom / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                         'note': 'algorithmic HBM bytes per chromosome = 4 n + 8 = 520 B (SURVEY.md 8d): the loop is bound by the latency of its 501 '
                                 'dependent launches, each as long as one workgroup\'s bookkeeping chain (D and the population live in L2 / LDS), not by HBM; '
                                 'the fraction is tiny by construction'},
            'cpu_baseline': cpu}


def run_cfg5(E, S, WL, torch, dist, np, rank, world, dev, cdev, fence, allmax, steps, cpu_on):
    """cfg5: 65 536 parallelograms through sharding.plan_sharded -- one block of fields per rank, cut on the analytic point counts."""
    V = WL.cfg5_parallelograms()
    specs = WL.specs_from_vertices(E, V)
    veh, opt = E.make_vehicle(), E.make_options()
    t0 = time.perf_counter()
    res = S.plan_sharded(specs, veh, opt, device=dev.index)          # sets up this rank's batch
    t_setup = time.perf_counter() - t0
    batch, infos = res.batch, res.infos
    # output arrays as the allocator returns them first (the plain figure), then the fastest of three candidate sets under the batch's
    # own step (Batch.alloc(best_of=3, include=[the plain one]): setup only, see engine.py) for everything below
    bufs = batch.alloc()
    for _ in range(2):
        batch.run(bufs)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(bufs)
    torch.cuda.synchronize()
    dt_plain = allmax(time.perf_counter() - t0)
    bufs = batch.alloc(best_of=3, include=[bufs])
    for _ in range(2):
        res = S.plan_sharded(specs, veh, opt, device=dev.index, batch=batch, buffers=bufs, infos=infos)
    fence()
    # timed: the device work of every rank + the stats gather (sizing and batch setup were done once, above: setup_s)
    batch.set_profiling(True, every=max(4, steps // 4))
    t0 = time.perf_counter()
    for _ in range(steps):
        res = S.plan_sharded(specs, veh, opt, device=dev.index, batch=batch, buffers=bufs, infos=infos)
    fence()
    dt_job = allmax(time.perf_counter() - t0)
    kernels, _ = batch.stage_times()
    batch.set_profiling(False)
    # device-only view of the same job: the kernels of this rank's block, no host-side partition in the loop
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        r2 = batch.run(bufs)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    fence()
    my_points = batch.total_points
    per_rank = torch.zeros(world, 2, dtype=torch.float64, device=cdev)
    per_rank[rank, 0], per_rank[rank, 1] = float(my_points), dt_local / steps
    if world > 1:
        dist.all_reduce(per_rank)
    dt_dev = allmax(dt_local)
    # the optional point-array gather, once
    fence()
    t0 = time.perf_counter()
    resg = S.plan_sharded(specs, veh, opt, device=dev.index, batch=batch, buffers=bufs, infos=infos, gather_points=True)
    fence()
    t_with_gather = allmax(time.perf_counter() - t0)
    entry = None
    if rank == 0:
        total = int(sum(i.n_main + i.n_head for i in res.infos))
        st = res.stats()
        assert res.stats_all.shape[0] == len(specs) and int(st['n_viol'].sum()) == 0
        if resg.points_all is not None:
            assert all(int(a.numel()) == total for a in resg.points_all)
        q_pts, g_pts = batch.point_split()
        dom = max(kernels, key=kernels.get)
        stage_points = batch.stage_points()
        dom_points = stage_points[dom]
        achieved = BYTES_PER_POINT * dom_points / (kernels[dom] * 1e-3) / 1e9
        tpath = os.path.join(REPO, 'profiles', 'traffic.json')
        traffic5 = json.load(open(tpath)).get(f'{dom}|cfg5') if os.path.exists(tpath) else None      # (counters of the whole job on one GPU)
        pr = per_rank.cpu().numpy()
        entry = {'name': 'cfg5', 'workload': 'cfg5: 65 536 parallelograms (base / height U[100,1000) m, angle U[60,120) deg, rotation U[-pi/4,pi/4), seed 65536), '
                                             'arc turns at the reference sampling, sharded over the ranks by sharding.plan_sharded '
                                             '(contiguous blocks cut on the analytic point counts; the only collective is the stats gather); '
                                             'output arrays = the fastest of 4 candidate sets (the plain allocation and 3 more) under the batch\'s own step (setup only), the plain allocation alone in placement1',
                 'n_gpus': world, 'scaling': 'strong', 'points': total, 'setup_s': t_setup,
                 'ms_per_step': dt_dev / steps * 1e3, 'value': total * steps / dt_dev, 'unit': 'points/s', 'dtype': 'f64',
                 'ms_per_job_with_stats_gather': dt_job / steps * 1e3,
                 'placement': getattr(batch, 'placement', None),
                 'placement1': {'ms_per_step': dt_plain / steps * 1e3, 'value': total * steps / dt_plain,
                                'step_frac': BYTES_PER_POINT * total / (dt_plain / steps) / 1e9 / HBM_PEAK_GBS / world,
                                'note': 'output arrays as the allocator returns them'},
                 'per_gpu': [{'rank': k, 'points': int(pr[k, 0]), 'ms_per_step': float(pr[k, 1] * 1e3), 'points_per_s': float(pr[k, 0] / pr[k, 1])}
                             for k in range(world)],
                 'point_array_gather': {'ms_job_with_gather': t_with_gather * 1e3, 'bytes_to_root': int(BYTES_PER_POINT * (total - pr[0, 0])),
                                        'note': 'one plan_sharded(gather_points=True): every peer sends its x, y, kappa, v, flagseg block straight into '
                                                'the root\'s arrays (dist.batch_isend_irecv)'},
                 'quiet_points': q_pts, 'general_points': g_pts,
                 'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                              'traffic': traffic5 if world == 1 else None, 'algorithmic_bytes_per_launch': BYTES_PER_POINT * dom_points,
                              'kernel_ms': kernels[dom], 'kernel_points_per_launch': dom_points, 'all_kernels_ms': kernels, 'all_kernels_points': stage_points,
                              'rank': 0, 'step_frac': BYTES_PER_POINT * total / (dt_dev / steps) / 1e9 / HBM_PEAK_GBS / world},
                 'cpu_baseline': None}
        if cpu_on:
            import oracle as orc
            entry['cpu_baseline'] = cpu_baseline_fields(lambda k: orc.make_field(verts=[(float(a), float(b)) for a, b in V[k]]), len(V), orc.Options.make(),
                                                        3.0, 'cfg5 parallelograms, arcs, reference sampling')
    batch.close()
    return entry


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""bench.py -- validated path points/sec of the HIP hot path on MI355X.

One "step" = one PLAN CALL of the reference for a whole batch: plan_complete_coverage (MLP:387-465) sets a NEW field up and generates its
path in one call, and so does a step here (round 5) -- `engine.Batch.plan(table)` = fcpp_batch_plan: the batch's setup on the device, its
output arrays, the whole hot path (sample every path point, curvature, curvature clamp, forward/backward speed sweeps, a_lat / geofence /
obstacle validation, per-field metrics) -- and the stream drained; the field records (128 B per field) are RESIDENT IN DEVICE MEMORY when the
region starts (engine.FieldTable.to_device(): inputs in HBM, as the tier's measurement rule has them; the same call on records in pinned host
memory, read by the device across PCIe where they lie, rides along as `ms_pinned_records` / `value_pinned_records`).  `value` / `ms_per_step`:
the median of REPS_SHORT regions of K such calls, each region bracketed by barrier + synchronize.

`value_step` / `ms_step` is what rounds 1-4 reported as `value`: the hot path re-run on a batch that is already set up (the kernels alone,
inputs resident in HBM) -- the figure the kernels' roofline fractions refer to.  `end_to_end` (diagnostic) splits a plan call made through
the three separate entries (create / alloc / run) into its parts, with `setup_ms` = pack / host_plan / templates / tiler / image / h2d.

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
# the secondary ceiling: float64 vector issue.  A vector instruction of a 64-wide wavefront occupies its SIMD for 4 cycles;
# MI355X: 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md).  valu_frac = vector instructions of a launch x 4 cycles / (1024 SIMDs x clock
# x kernel time); the instruction counts are the SQ_INSTS_VALU counters of the committed profiles (profiles/valu.json), the time is this run's.
SIMDS, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 4
VALU_BOUND_KERNELS = ('k_plan_sparse', 'k_plan_sparse_fields', 'k_plan_fused')      # bound by float64 vector issue (profiles/*_counter_table.txt); the quiet kernels by HBM


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--fields', type=int, default=4096, help='fields of 500 x 200 m per GPU in the headline batch')
    ap.add_argument('--mode', type=int, default=1, help='1 = fused pipeline (default), 0 = staged pipeline')
    ap.add_argument('--configs', default='all',
                    help="'all', 'none' or a comma list of: cfg1_clothoid,cfg1_x16384,cfg1_clothoid_dense,cfg2_ref,cfg2_0.5,cfg2_0.1,cfg3,cfg3_avoid,cfg4,cfg5,single_field")
    ap.add_argument('--calibrate', type=int, default=0, help='opt-in: also report the dense configs and cfg5 with Batch.alloc(best_of=N) output arrays (never the primary figure)')
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


def cpu_baseline_python(orc, budget_s):
    """The per-point stages of the hot path (curvature, clamp, sweeps, metrics, verifier: SURVEY.md 8a rows 10-14) the way the reference runs
    them -- one Python iteration per path point with numpy scalars (oracle/py_loops.py, pinned bit for bit to the reference's own outputs in
    tests/golden) -- and as whole-array numpy, on ONE core, on the reference's own 500 x 200 m path (1691 points): SURVEY.md 8d's CPU baseline,
    measured on this box in this run."""
    import numpy as np
    from oracle import py_loops as PL
    rc, p = orc.plan_field(orc.make_field(L=500.0, H=200.0))
    assert rc == 0
    kind = p.flagseg & 7
    nominal = np.select([kind == 0, kind == 1, kind == 2, kind == 3, kind == 4, kind == 5], [9.0, 4.0, 15.0, 15.0, 4.0, 2.5]).astype(np.float64)

    class V:
        max_lateral_accel, safety_factor, max_longitudinal_accel = 2.0, 0.85, 1.5
    loops, n1, t1 = PL.time_stages(p.xy, nominal, V, 'loops', budget_s / 2)
    vec, n2, t2 = PL.time_stages(p.xy, nominal, V, 'numpy', budget_s / 2)
    v_loops, _ = PL.speed_plan_loops(p.xy, nominal, V)
    assert float(np.abs(v_loops - p.v).max()) < 1e-9
    return {'python_loops_value': loops, 'numpy_value': vec, 'python_cores': 1,
            'python_sample': f'curvature + clamp + sweeps + metrics + verifier of the 1691-point path of a 500 x 200 m field: {n1} passes with per-point Python loops '
                             f'({t1:.1f} s), {n2} as whole-array numpy ({t2:.1f} s), one core (oracle/py_loops.py)'}


# ---- one planner configuration on this rank --------------------------------------------------------------------------------------
REPS = 5          # repetitions of the K-step timed region; the median is reported
# A K = 20 region of the sparse configurations is ~1 ms long -- shorter than the device's clock / power transient after a load sets in
# (tools/short_region.py, profiles/r04_short_region.log: regions 3-6 of a series read 5-10 % slower than the first two and than everything
# from the tenth on).  Steps that write < 1 GB are therefore timed over REPS_SHORT regions (25 x K steps, ~30 ms): every region is K steps
# bracketed as the contract says, all of them are listed in bench_detail.json (timed_region), the median is `value`.
REPS_SHORT = 25
SHORT_STEP_BYTES = 1 << 30


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def fresh_regions(E, torch, table, veh, opt, steps, reps, fence, after_call=None):
    """`reps` regions of `steps` FRESH plan calls (engine.Batch.plan: fcpp_batch_plan + the stream drained; the previous call's batch and
    arrays are released inside the region, as a caller in a loop releases them) -> (seconds per region, points per call)"""
    dts, batch, res, n_points, k = [], None, None, 0, 0
    drain = torch.cuda.current_stream().synchronize
    for _ in range(max(1, reps)):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            if batch is not None:
                batch.close()
            res = None
            batch, res = E.Batch.plan(table, veh, opt)
            if after_call:
                after_call(k, res)
            k += 1
            drain()                     # (the call's stream drained: 2.5 us less than torch.cuda.synchronize(), which waits for every stream of the device)
        fence()
        dts.append(time.perf_counter() - t0)
        n_points = batch.total_points
    if batch is not None:
        batch.close()
    return dts, n_points


def sustained_regions(E, torch, table, veh, opt, steps, reps, fence):
    """`reps` regions of `steps` fresh plan calls with TWO calls in flight: the calls alternate between two streams, batch k + 1 is set up (a chain
    of latency-bound kernels) on one while batch k's step (bound by vector issue / HBM) still runs on the other; a batch is released when the
    call two later needs its stream, its results complete by then.  What a caller that plans batch after batch and consumes the results
    asynchronously sustains -- NOT `value`, which drains the stream after every call.  -> seconds per region"""
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    held = [None, None]
    dts = []
    for _ in range(max(1, reps) + 1):          # (the first region warms the second stream's scratch and allocation up: dropped)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i & 1
            with torch.cuda.stream(streams[k]):
                if held[k] is not None:
                    streams[k].synchronize()          # (the results of the call two back: consumed)
                    held[k][0].close()
                    held[k] = None
                held[k] = E.Batch.plan(table, veh, opt)
        for st in streams:
            st.synchronize()
        fence()
        dts.append(time.perf_counter() - t0)
    for k in range(2):
        if held[k] is not None:
            held[k][0].close()
            held[k] = None
    return dts[1:]


def run_planner(E, torch, table, opt, steps, warmup, mode=1, calibrate=0, fence=None, after_step=None, stats_of=None, reps=REPS, e2e_reps=5,
                fresh_steps=0, fresh_reps=0, after_fresh=None, records='device', pinned_too=False):
    """-> dict(points, ms_per_step, kernels {name: ms}, dominant kernel + its points, end_to_end, batch, bufs, res).

    1. the plan call end to end, e2e_reps times: a fresh batch from `table` (engine.FieldTable), its output arrays, one step, the stream
       drained; the batch of the last repetition is kept for
    2. the timed region: `reps` x K steps bracketed by fence() (barrier + synchronize), median reported, per-kernel HIP events recorded
       inside it on a sample of the steps.
    The output arrays come from Batch.alloc() with its default layout rule ('auto': large batches get the five arrays >= 24 GiB apart
    inside one allocation; reported per entry as output_arrays).  calibrate > 1 (opt-in, reported beside the primary figure, never as it):
    Batch.alloc(best_of=calibrate) picks the fastest of `calibrate` further candidate sets under the batch's own step.
    stats_of(i, batch): the stats tensor step i writes (default: the one of alloc()); after_step(i, res): called after step i has
    been enqueued."""
    fence = fence or torch.cuda.synchronize
    veh = E.make_vehicle()
    e2e, batch, bufs = [], None, None
    # the field records: resident in DEVICE memory when the timed regions start (records='device', round 5b: FieldTable.to_device() -- inputs in
    # HBM), or in pinned host memory, read by the device across PCIe where they lie ('pinned': rounds 4-5a).  The host paths of the library
    # (cfg3's single field, AVOID mode) copy device records back first -- inside the call, on the clock.
    # A batch the HOST sets up (cfg3's single field, AVOID mode) keeps its records in pinned host memory: the host plans them, a table on the
    # device would be copied back first (30-40 us of every call, measured: cfg3 0.442 -> 0.480 ms).
    if records == 'device' and hasattr(table, 'to_device'):
        probe = E.Batch(table, veh, opt)
        on_device = probe.setup_path() == 'device'
        probe.close()
        if on_device:
            table.to_device()
        else:
            records = 'pinned'
    if records != 'device' and hasattr(table, 'pin'):
        table.pin()
    for rep in range(max(1, e2e_reps)):
        if batch is not None:
            batch.close()
            batch = bufs = None
        fence()
        t0 = time.perf_counter()
        batch = E.Batch(table, veh, opt)
        t1 = time.perf_counter()
        bufs = batch.alloc()
        batch.run(bufs, mode=mode)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st = batch.setup_times()
        e2e.append({'ms': (t2 - t0) * 1e3, 'create_ms': (t1 - t0) * 1e3, 'alloc_run_sync_ms': (t2 - t1) * 1e3, 'setup_ms': st})
    pageable_ms = None
    if e2e_reps >= 100 and hasattr(table, 'pin'):
        # the same call on a copy of the table in PAGEABLE memory (the library copies the records to the device first), a quarter of the repetitions
        import numpy as np
        plain = E.FieldTable(np.array(table.rec), table.poly_offsets, table.poly_x, table.poly_y)
        ms = []
        for rep in range(e2e_reps // 4):
            fence()
            t0 = time.perf_counter()
            b2 = E.Batch(plain, veh, opt)
            b2.run(b2.alloc(), mode=mode)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            b2.close()
        pageable_ms = sorted(ms[1:])[(len(ms) - 2) // 2]
    n_points = batch.total_points
    if reps == REPS and BYTES_PER_POINT * n_points < SHORT_STEP_BYTES:
        reps = REPS_SHORT
    # the plan call as ONE library call, K of them per region (the headline's `value`; every configuration's `value_fresh`)
    fresh = None
    if fresh_steps > 0:
        fw, _ = fresh_regions(E, torch, table, veh, opt, min(fresh_steps, 5), 1, fence)          # (warm-up: the arena, the spare allocations)
        fdts, fpts = fresh_regions(E, torch, table, veh, opt, fresh_steps, fresh_reps or reps, fence, after_fresh)
        assert fpts == n_points
        fresh = {'dts': fdts, 'steps': fresh_steps, 'ms_per_call': median(fdts) / fresh_steps * 1e3, 'records': records}
        if pinned_too and records == 'device':
            # the same regions with the records in pinned host memory (rounds 4-5a's `value`)
            import numpy as np
            tp = E.FieldTable(np.array(table.rec), table.poly_offsets, table.poly_x, table.poly_y).pin()
            fresh_regions(E, torch, tp, veh, opt, min(fresh_steps, 5), 1, fence)
            pdts, _ = fresh_regions(E, torch, tp, veh, opt, fresh_steps, fresh_reps or reps, fence)
            fresh['pinned_ms_per_call'] = median(pdts) / fresh_steps * 1e3
    warm = e2e[1:] or e2e                 # (the first repetition of a process pays the context's pinned staging memory and the allocator)
    mid = sorted(warm, key=lambda r: r['ms'])[(len(warm) - 1) // 2]
    end_to_end = {'ms': mid['ms'], 'points_per_s': n_points / (mid['ms'] * 1e-3), 'create_ms': mid['create_ms'],
                  'alloc_run_sync_ms': mid['alloc_run_sync_ms'], 'setup_ms': {k: (round(v, 4) if isinstance(v, float) else v) for k, v in mid['setup_ms'].items()},
                  'first_ms': e2e[0]['ms'], 'all_ms': [round(r['ms'], 3) for r in e2e[:16]], 'reps': len(e2e),
                  'what': 'fresh batch in a warm context: engine.Batch(table) [fcpp_batch_create: the setup, on the device where the device planner '
                          'takes the batch, else host plan + tiler + image + one H2D copy] + output arrays + one step + stream drained; the field '
                          'records ' + ('resident in device memory (FieldTable.to_device())' if records == 'device' else 'in pinned host memory (FieldTable.pin(): read by the device where they lie)') +
                          '; pageable_ms: the same call on pageable records; median of the repetitions after the first; the reference times this call (plan_complete_coverage, MLP:387-465)'}
    if pageable_ms is not None:
        end_to_end['pageable_ms'] = pageable_ms
    res = None
    k = 0

    def step():
        nonlocal res, k
        res = batch.run(bufs[:5] + (stats_of(k, batch),) if stats_of else bufs, mode=mode)
        if after_step:
            after_step(k, res)
        k += 1

    # Which kernels does a step launch?  The last warm-up step carries per-kernel events.
    one_kernel = None
    for w in range(warmup):
        if w == warmup - 1:
            batch.set_profiling(True, every=1)
        step()
    if warmup > 0:
        torch.cuda.synchronize()
        kw, _ = batch.stage_times()
        batch.set_profiling(False)
        live = [name for name, ms in kw.items() if ms > 0.0]
        one_kernel = live[0] if len(live) == 1 else None
    fence()
    dts = []
    if one_kernel:
        # ONE launch per step (the headline since round 4): the kernel's average launch duration = HIP events around the K launches of
        # the timed region, on the stream they are launched on (torch's current stream IS the context's), / K -- nothing between the
        # launches.  (A dispatch that carries its own start / stop events does not overlap its neighbours' ramp and tail and costs the
        # stream ~20 us: three of them in a region of 20 steps of 0.055 ms were 10 % of it, tools/short_region.py.)
        ev_ms = []
        for _ in range(max(1, reps)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
            fence()
            dts.append(time.perf_counter() - t0)
            ev_ms.append(e0.elapsed_time(e1))
        kernels = {name: 0.0 for name in kw}
        kernels[one_kernel] = median(ev_ms) / steps
        prof_runs = len(ev_ms) * steps
        kernel_timing = f'HIP events around the {steps} launches of each timed region / {steps} (one kernel per step)'
    else:
        # per-kernel HIP events inside the timed region, on a sample of its steps (a profiled step dispatches every kernel with its own
        # start / stop events and costs ~20 us more than a plain one)
        batch.set_profiling(True, every=max(8, steps // 8))
        for _ in range(max(1, reps)):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            dts.append(time.perf_counter() - t0)
        kernels, prof_runs = batch.stage_times()
        batch.set_profiling(False)
        kernel_timing = f'start / stop HIP events on every kernel of every {max(8, steps // 8)}-th step of the timed regions'
    dt = median(dts)
    calibrated = None
    if calibrate > 1:
        bufs2 = batch.alloc(best_of=calibrate, include=[bufs])
        for _ in range(2):
            batch.run(bufs2, mode=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            batch.run(bufs2, mode=mode)
        torch.cuda.synchronize()
        cal_ms = (time.perf_counter() - t0) / steps * 1e3
        calibrated = {'ms_per_step': cal_ms, 'value': n_points / (cal_ms * 1e-3), 'placement': getattr(batch, 'placement', None),
                      'step_frac': BYTES_PER_POINT * n_points / (cal_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      'note': f'opt-in Batch.alloc(best_of={calibrate}): the fastest of the plain allocation and {calibrate} more candidate sets under the batch\'s own step (setup only); NOT the primary figure'}
        del bufs2
    q_pts, g_pts = batch.point_split()
    dom = max(kernels, key=kernels.get)
    stage_points = batch.stage_points()
    dom_points = stage_points[dom]
    return {'stage_points': stage_points, 'prof_runs': prof_runs, 'points': n_points, 'dt': dt, 'dts': dts, 'ms_per_step': dt / steps * 1e3, 'kernels': kernels,
            'dominant': dom, 'dominant_points': dom_points, 'quiet_points': q_pts, 'general_points': g_pts, 'batch': batch, 'bufs': bufs, 'res': res,
            'calibrated': calibrated, 'end_to_end': end_to_end, 'steps': steps, 'kernel_timing': kernel_timing, 'layout': getattr(batch, 'layout', None),
            'fresh': fresh}


def roofline_of(r, traffic_key=None):
    """The roofline record of one configuration.  `frac` / `kernel` name the kernel with the longest launch (two kernels of the headline are
    0.2 us apart: which one that is flips with noise); `step_frac` -- 36 B x all points of the step / ms_per_step / 8 TB/s -- is the figure
    that does not, and what the compact line reports as its `roofline.frac` beside every kernel's own fraction.  `traffic`: HBM bytes per
    launch from the PMC counters of the committed profiles (profiles/traffic.json: a constant of the builder's profiling run on this
    workload, NOT measured in this run; `traffic_source` says so)."""
    dom, dom_ms, dom_points = r['dominant'], r['kernels'][r['dominant']], r['dominant_points']
    achieved = BYTES_PER_POINT * dom_points / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    pipe_ms = sum(r['kernels'].values())
    traffic = step_traffic = None
    tpath = os.path.join(REPO, 'profiles', 'traffic.json')
    point_kernels = [k for k, ms in r['kernels'].items() if ms > 0 and k != 'k_reduce_stats']
    if traffic_key and os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic = tj.get(f'{dom}|{traffic_key}')
        parts = [tj.get(f'{k}|{traffic_key}') for k in point_kernels]
        step_traffic = sum(parts) if parts and all(p is not None for p in parts) else None
    note = None
    if dom in ('k_plan_sparse', 'k_plan_sparse_fields'):
        # (the secondary ceiling.  Counters of the kept profiles, not measured in this run.)
        note = ('the wave-tile kernels are bound by fp64 vector issue and dependent loads, not by HBM (profiles/*_counter_table.txt); '
                'the step-level figure is step_frac')
    step_achieved = BYTES_PER_POINT * r['points'] / (r['ms_per_step'] * 1e-3) / 1e9
    valu, valu_src = {}, None
    vpath = os.path.join(REPO, 'profiles', 'valu.json')
    if traffic_key and os.path.exists(vpath):
        vj = json.load(open(vpath))
        for k, ms in r['kernels'].items():
            insts = vj.get(f'{k}|{traffic_key}')
            if insts and ms > 0:
                valu[k] = insts * VALU_CYCLES / (SIMDS * CLOCK_HZ * ms * 1e-3)
        valu_src = 'profiles/valu.json (SQ_INSTS_VALU per launch of the committed rocprofv3 --pmc pass) x 4 cycles / (1024 SIMDs x 2.4 GHz x this run\'s kernel time)'
    bound = 'valu_f64' if dom in VALU_BOUND_KERNELS else 'hbm'
    return {'bound': bound, 'valu_frac': valu.get(dom), 'valu_fracs': valu or None, 'valu_source': valu_src if valu else None, 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'note': note,
            'traffic': traffic, 'traffic_source': 'profiles/traffic.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of the builder\'s profiling run; a constant, not measured in this run)' if traffic is not None or step_traffic is not None else None,
            'kernel_ms': dom_ms, 'algorithmic_bytes_per_launch': BYTES_PER_POINT * dom_points,
            'kernel_points_per_launch': dom_points, 'all_kernels_ms': r['kernels'], 'all_kernels_points': r['stage_points'],
            'profiled_steps': r['prof_runs'], 'kernel_timing': r.get('kernel_timing'), 'pipeline_ms': pipe_ms,
            'pipeline_frac': (BYTES_PER_POINT * r['points'] / (pipe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pipe_ms > 0 else 0.0,
            'step_kernels': 'step: ' + ' + '.join(point_kernels), 'step_achieved': step_achieved, 'step_traffic': step_traffic,
            'step_frac': step_achieved / HBM_PEAK_GBS}


def timed_region_of(r):
    return {'reps': len(r['dts']), 'steps_per_rep': r['steps'], 'ms_per_step_each_rep': [round(d / r['steps'] * 1e3, 5) for d in r['dts']], 'reported': 'median'}


def config_entry(name, workload, r, cpu=None, extra=None, traffic_key=None):
    e = {'name': name, 'workload': workload, 'points': r['points'], 'ms_per_step': r['ms_per_step'],
         'value': r['points'] / (r['ms_per_step'] * 1e-3), 'unit': 'points/s', 'dtype': 'f64', 'timed_region': timed_region_of(r),
         'value_end_to_end': r['end_to_end']['points_per_s'], 'end_to_end': r['end_to_end'], 'setup_ms': r['end_to_end']['setup_ms'],
         'output_arrays': r['layout'],
         'quiet_points': r['quiet_points'], 'general_points': r['general_points'], 'roofline': roofline_of(r, traffic_key or name), 'cpu_baseline': cpu}
    if r.get('fresh'):
        e['ms_fresh'] = r['fresh']['ms_per_call']
        e['records'] = r['fresh'].get('records')          # (where the field records lay: 'device' = resident in HBM, 'pinned' = host memory the device reads / the host plans)
        e['value_fresh'] = r['points'] / (r['fresh']['ms_per_call'] * 1e-3)
        e['fresh_region'] = {'reps': len(r['fresh']['dts']), 'calls_per_rep': r['fresh']['steps'],
                             'ms_per_call_each_rep': [round(d / r['fresh']['steps'] * 1e3, 5) for d in r['fresh']['dts']], 'reported': 'median',
                             'what': 'a plan call = engine.Batch.plan(table) [fcpp_batch_plan] + the stream drained'}
    else:
        e['ms_fresh'], e['value_fresh'] = r['end_to_end']['ms'], r['end_to_end']['points_per_s']
    if r.get('calibrated'):
        e['calibrated'] = r['calibrated']
    if extra:
        e.update(extra)
    return e


# the reference's own cost as PUBLISHED / measured once in the build container: constants with
# provenance, kept in the detail file only.  What the reference's loops cost ON THIS BOX is measured in every run: cpu_baseline.python_loops_value
# (oracle/py_loops.py: the per-point Python loops of MLP:490-504, 558-587, 1383-1408 restated, pinned bit for bit to the reference's outputs).
REFERENCE_MEASURED = {
    'plan_points_per_s': 6.1e4, 'plan_ms_500x200': 27.6, 'published_plan_ms_500x200': 46.0, 'published_points_per_s': 3.7e4,
    'ga_chromosomes_per_s': 3.9e4, 'ga_other_operators_s_per_generation': 0.38,
    'provenance': 'reference code (multi_layer_planner_v3.plan_complete_coverage, genetic_algorithm_solver) timed in the build container, one core, '
                  'Shapely stubbed (BASELINE.md section 2); published: README_en.md:206-208 (0.046 s for 1691 points, hardware unstated). '
                  'Constants, not measured in this run -- the reference cannot travel to the GPU box.',
}



# ---- the ONE line the driver parses: compact (< 4 KB); everything else goes to bench_detail.json and stderr ----------------------------
COMPACT_LIMIT = 4000


def _r(v, nd=6):
    if isinstance(v, float):
        return float(f'{v:.{nd}g}')
    return v


def compact_line(out):
    """The last stdout line: BASELINE.json's metric on the headline workload -- `value` = the fresh plan call, `value_step` = the hot path on a
    batch already set up -- with `roofline` (the dominant kernel: its own launch duration and fraction, what bounds it, the call's fraction
    as `end_to_end_frac`; every kernel of the step as [ms per launch, points per launch, HBM fraction]) and `cpu_baseline`, plus one short
    row per configuration.  The full record (per-configuration entries with their setup splits, timed regions, notes) is bench_detail.json."""
    rf, cfg = out['roofline'], out['config']
    tr = out.get('timed_region') or {}
    kern = {}
    for name, ms in rf['all_kernels_ms'].items():
        pts = rf['all_kernels_points'].get(name, 0)
        if ms > 0:
            kern[name] = [_r(ms, 5), int(pts), _r(BYTES_PER_POINT * pts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if name != 'k_reduce_stats' else None]
    c = {
        'metric': out['metric'], 'value': _r(out['value'], 7), 'unit': out['unit'], 'n_gpus': out['n_gpus'], 'steps': out['steps'], 'warmup': out['warmup'],
        'ms_per_step': _r(out['ms_per_step'], 6), 'higher_is_better': True, 'scaling': out['scaling'], 'vs_baseline': out.get('vs_baseline'),
        'vs_baseline_of': out.get('vs_baseline_of'), 'dtype': 'f64', 'data': 'synthetic',
        'step': 'one plan call (fcpp_batch_plan: setup on the device + output arrays + the hot path) + stream drained; field records resident in HBM',
        'ms_pinned_records': _r(out.get('ms_pinned_records'), 5), 'value_pinned_records': _r(out.get('value_pinned_records'), 5),
        'value_step': _r(out.get('value_step'), 6), 'ms_step': _r(out.get('ms_step'), 5),
        'value_sustained': _r(out.get('value_sustained'), 5), 'ms_sustained': _r(out.get('ms_sustained'), 5),
        'config': {'workload': cfg['workload'][:200], 'turn_model': cfg['turn_model'], 'points_per_gpu_step': cfg['points_per_gpu_step'],
                   'fields_per_gpu': cfg['fields_per_gpu'], 'setup': cfg.get('setup')},
        'roofline': {'bound': rf['bound'], 'kernel': rf['kernel'], 'frac': _r(rf['frac'], 4), 'achieved': _r(rf['achieved'], 6), 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'kernel_ms': _r(rf['kernel_ms'], 5), 'algorithmic_bytes_per_launch': rf['algorithmic_bytes_per_launch'],
                     'valu_frac': _r(rf.get('valu_frac'), 3), 'traffic': rf.get('traffic'), 'traffic_source': (rf.get('traffic_source') or '')[:60] or None,
                     'step_frac': _r(rf['step_frac'], 4), 'end_to_end_frac': _r(out.get('end_to_end_frac'), 4),
                     'kernel_timing': (rf.get('kernel_timing') or '')[:100], 'kernels': kern},
        'cpu_baseline': None, 'end_to_end_ms': _r(out['end_to_end']['ms'], 5),
        'value_clothoid': _r(out.get('value_clothoid'), 6), 'value_step_clothoid': _r(out.get('value_step_clothoid'), 6), 'frac_clothoid': _r(out.get('frac_clothoid'), 3),
        'rccl_ranks': out.get('rccl_ranks'), 'per_rank_points_per_s': out.get('per_rank_points_per_s'),
        'host_threads': out.get('host_threads'), 'detail': 'bench_detail.json',
        # (every region is K plan calls between two fences; their number and the spread ride along, each one's value is in the detail file)
        'timed_regions': {'n': tr.get('reps'), 'reported': 'median', 'first_ms': _r(tr['ms_per_step_each_rep'][0], 5), 'min_ms': _r(min(tr['ms_per_step_each_rep']), 5),
                          'max_ms': _r(max(tr['ms_per_step_each_rep']), 5)} if tr.get('ms_per_step_each_rep') else None,
    }
    if out.get('forced_dist'):
        c['forced_dist'] = True
    cb = out.get('cpu_baseline')
    if cb:
        c['cpu_baseline'] = {'value': _r(cb['value'], 5), 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb['sample'][:120],
                             'single_core_value': _r(cb.get('single_core_value'), 5), 'python_loops_value': _r(cb.get('python_loops_value'), 4),
                             'numpy_value': _r(cb.get('numpy_value'), 4), 'python_cores': cb.get('python_cores')}
    rows = {}
    for e in out.get('configs', []):
        rr = e.get('roofline') or {}
        cpu = e.get('cpu_baseline') or {}
        dev = (e.get('setup_ms') or {}).get('device_setup')
        rows[e['name']] = [_r(e.get('ms_fresh'), 4), _r(e.get('value_fresh'), 4), _r(e.get('ms_per_step', e.get('ms_total')), 5),
                           _r(rr.get('step_frac', rr.get('frac')), 3), _r(cpu.get('value'), 4), None if dev is None else ('device' if dev else 'host')]
    c['configs'] = {'columns': ['ms_plan_call', 'value_plan_call', 'ms_step', 'step_frac', 'cpu_baseline_value', 'setup'], **rows}
    line = json.dumps(c, separators=(',', ':'))
    for drop in ('per_rank_points_per_s', 'configs'):          # (never needed so far: a guard, not a plan)
        if len(line) <= COMPACT_LIMIT:
            break
        c.pop(drop, None)
        line = json.dumps(c, separators=(',', ':'))
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    # ONE JSON line on stdout, nothing else: RCCL prints a version banner and gloo its connection messages to stdout when the process
    # group comes up -- from here on file descriptor 1 is stderr, and the line is written to the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
    # host threads of the batch setup (fcpp_parallel.h reads FCPP_THREADS when its pool starts): N ranks of one node share its cores
    try:
        n_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        n_cores = os.cpu_count() or 1
    host_threads = int(os.environ.get('FCPP_THREADS') or max(1, min(16, n_cores // max(world, 1))))
    os.environ['FCPP_THREADS'] = str(host_threads)
    # FCPP_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives carry host tensors); the measured configuration is nccl = RCCL, one rank per GPU
    backend = os.environ.get('FCPP_BENCH_BACKEND', 'nccl')
    local = local % torch.cuda.device_count() if backend == 'gloo' else local
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cdev = dev if backend == 'nccl' else torch.device('cpu')          # where the collectives' tensors live
    # FCPP_BENCH_FORCE_DIST=1 with --gpus 1: a process group of ONE rank over RCCL, and the collective legs of the path (the headline's
    # stats-ring gather, the sharded job's gathers) really issue, addressed to the rank itself -- the RCCL code path on the hardware at hand
    force_dist = world == 1 and os.environ.get('FCPP_BENCH_FORCE_DIST') == '1'
    use_dist = world > 1 or force_dist
    if force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        S.FORCE_COLLECTIVES = True
    if use_dist:
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}), **({'rank': 0, 'world_size': 1} if force_dist else {}))
    # everything runs on a stream of its own: on the legacy default stream the same launches take 3-8 % longer at the 0.05-0.1 ms
    # steps of the reference's sampling (0.0855 vs 0.0827 ms on cfg1 x 4096, 0.050 vs 0.047 ms on cfg2; larger steps do not care)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    # the context's output arena, once, at start-up (context creation in the sense of SURVEY.md 8d: an allocation of this size takes the
    # driver seconds): every configuration's output arrays come out of it, five lanes 24 GiB apart
    t0 = time.perf_counter()
    arena = None
    try:
        E.get_context(local).reserve_outputs(lane_gib=24.0, pitch_gib=24.0)
        arena = {'lane_GiB': 24.0, 'pitch_GiB': 24.0, 'reserve_ms': round((time.perf_counter() - t0) * 1e3, 1)}
    except Exception as ex:          # (a device without the room: the plain layout, reported as such per entry)
        arena = {'error': str(ex)[:200]}
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
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    want = args.configs.split(',') if args.configs not in ('all', 'none') else (
        ['cfg1_clothoid', 'cfg1_x16384', 'cfg1_clothoid_dense', 'cfg2_ref', 'cfg2_0.5', 'cfg2_0.1', 'cfg3', 'cfg3_avoid', 'cfg4', 'cfg5', 'single_field'] if args.configs == 'all' else [])
    cpu_on = world == 1 and not args.no_cpu_baseline

    # ---- headline: 4096 x (500 x 200 m) per GPU, arcs at the reference's sampling (the pinned mode) ------------------------------
    # the only collective of the path: the gather of every step's per-field stats on rank 0.  Every step writes its stats into the next
    # slot of a ring on the device; after GATHER_EVERY steps their slots go to the root in ONE asynchronous gather on RCCL's own
    # stream (a collective per 0.1 ms step would cost more host time than the step's kernels take), double-buffered: while one half of
    # the ring travels, the steps fill the other.
    LH1 = WL.cfg1_batch(args.fields)
    GATHER_EVERY = 20
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
        if use_dist and (i + 1) % GATHER_EVERY == 0:
            send_group(i // GATHER_EVERY)

    def fence_headline():
        if use_dist:
            if steps_done[0] % GATHER_EVERY:          # an unfinished group at a fence travels as it is (and again when it is complete)
                send_group(steps_done[0] // GATHER_EVERY)
            if pending:
                pending[-1].wait()
        fence()

    # `value`: regions of K FRESH plan calls (Batch.plan + the stream drained), REPS_SHORT of them; `value_step`: the same regions of K steps on
    # the batch that is already set up, with the kernel's own launch duration by HIP events (what the roofline fraction refers to).  With
    # several ranks the per-field stats of every call go to rank 0 as before (a device copy into the ring, a gather per GATHER_EVERY calls).
    def after_fresh(k, res):
        if use_dist:
            stats_slot(k, res.batch).copy_(res.stats_raw)
        count_and_gather(k, res)

    r = run_planner(E, torch, E.FieldTable.from_rectangles(LH1), E.make_options(), args.steps, args.warmup, mode=args.mode, fence=fence_headline,
                    after_step=count_and_gather, stats_of=stats_slot if use_dist else None, e2e_reps=50, fresh_steps=args.steps, fresh_reps=REPS_SHORT,
                    after_fresh=after_fresh, pinned_too=(world == 1))
    sustained = None
    if world == 1:
        t_h = E.FieldTable.from_rectangles(LH1)
        t_h.to_device()
        sustained = median(sustained_regions(E, torch, t_h, E.make_vehicle(), E.make_options(), args.steps, REPS_SHORT, fence_headline))
    dt = allmax(r['dt'])
    fresh_dt = allmax(median(r['fresh']['dts']))
    total_points = allsum(r['points'])
    e2e_ms = allmax(r['end_to_end']['ms'])
    # every rank's own rate (its points x K / its own time of the K-call region), so that the scaling run checks itself
    rates = torch.zeros(world, dtype=torch.float64, device=cdev)
    rates[rank] = r['points'] * args.steps / median(r['fresh']['dts'])
    if use_dist:
        dist.all_reduce(rates)
    per_rank_rates = [float(f'{v:.5g}') for v in rates.cpu().tolist()]
    out = None
    if rank == 0:
        st = r['res'].stats()
        assert int(st['n_viol'].sum()) == 0 and np.isfinite(st['main_len_m']).all()
        assert (r['batch'].info[0].n_main, r['batch'].info[0].n_head) == (1256, 435)          # README_en.md:206-207
        fresh_ms = fresh_dt / args.steps * 1e3
        out = {
            'metric': METRIC, 'value': total_points * args.steps / fresh_dt, 'unit': 'points/s',
            'n_gpus': world, **({'forced_dist': 'one-rank RCCL process group, collectives addressed to the rank itself (FCPP_BENCH_FORCE_DIST=1)'} if force_dist else {}), 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': fresh_ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'step': 'one plan call of the reference for the whole batch (plan_complete_coverage, MLP:387-465: a NEW field set up and its path generated): '
                    'engine.Batch.plan(table) = fcpp_batch_plan (the setup on the device, the output arrays from the context\'s arena, the hot path) + the stream drained; '
                    'field records resident in device memory when the region starts (FieldTable.to_device(); ms_pinned_records: the same call on records in pinned host memory)',
            'timed_region': {'reps': len(r['fresh']['dts']), 'steps_per_rep': args.steps, 'ms_per_step_each_rep': [round(d / args.steps * 1e3, 5) for d in r['fresh']['dts']],
                             'reported': 'median'},
            # the hot path on the batch that is set up (rounds 1-4's `value`): K steps per region, the kernels alone
            'value_step': total_points * args.steps / dt, 'ms_step': dt / args.steps * 1e3, 'timed_region_step': timed_region_of(r),
            'end_to_end_frac': BYTES_PER_POINT * r['points'] / (fresh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            **({'ms_pinned_records': r['fresh']['pinned_ms_per_call'], 'value_pinned_records': r['points'] / (r['fresh']['pinned_ms_per_call'] * 1e-3)}
               if r['fresh'].get('pinned_ms_per_call') else {}),
            # two plan calls in flight on two streams: the sustained fresh-batch rate, reported as such
            **({'value_sustained': total_points * args.steps / sustained, 'ms_sustained': sustained / args.steps * 1e3,
                'sustained': 'fresh plan calls alternating between two streams, batch k + 1 set up while batch k\'s step runs; not `value`'} if sustained else {}),
            # the same call through the three separate entries (create / alloc / run), split into its parts
            'value_end_to_end': total_points / (e2e_ms * 1e-3), 'end_to_end': r['end_to_end'], 'setup_ms': r['end_to_end']['setup_ms'],
            'config': {
                'workload': f'cfg1 x {args.fields}: {args.fields} fields of 500 x 200 m per GPU, ARC turns (the reference\'s own model, pinned to its outputs) at the '
                            f'reference\'s sampling, 1691 points per field; a step = one fresh plan call of the batch '
                            f'(BASELINE.json configs[0], README_en.md:199-215; default VehicleParams; 2 / 20 / 15 / 20 points per line / U-turn / corner / headland side); '
                            f'the clothoid turn model on the same batch: value_clothoid, configs[cfg1_clothoid*]',
                'turn_model': 'arc (reference, pinned)', 'points_per_gpu_step': r['points'], 'fields_per_gpu': args.fields,
                'pipeline': 'staged (7 kernels)' if args.mode == 0 else 'fused: k_plan_quiet (closed-form runs and spans) + k_plan_sparse / k_plan_sparse_fields (wave tiles, two points per lane; the latter also reduces its fields) + k_plan_fused (all other tiles) + k_reduce_stats',
                'quiet_points': r['quiet_points'], 'general_points': r['general_points'], 'output_arrays': r['layout'],
            },
            'roofline': roofline_of(r, 'cfg1'),
            'cpu_baseline': None,
            'reference_published': REFERENCE_MEASURED,
            'rccl_ranks': (dist.get_world_size() if use_dist else 1) if backend == 'nccl' else 0,
            'per_rank_points_per_s': per_rank_rates,
            'host_threads': host_threads,
            'output_arena': arena,
        }
        out['config']['setup'] = r['batch'].setup_path() if hasattr(r['batch'], 'setup_path') else 'host'
    if rank == 0 and cpu_on:
        import oracle as orc
        out['cpu_baseline'] = cpu_baseline_fields(lambda k: orc.make_field(L=float(LH1[k, 0]), H=float(LH1[k, 1])), len(LH1), orc.Options.make(),
                                                  args.cpu_budget, '500 x 200 m fields, arcs, reference sampling')
        out['cpu_baseline'].update(cpu_baseline_python(orc, args.cpu_budget / 2))
        # BASELINE.md publishes no points/s figure for this metric: vs_baseline stays null; the ratio against this run's own CPU baseline (the C
        # port of the reference's loops on the box's cores) rides along under its own name
        out['vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
    r['batch'].close()
    del r
    torch.cuda.empty_cache()

    # ---- the other configurations --------------------------------------------------------------------------------------------------
    configs = []

    def planner_config(name, workload, table, opt, steps, warmup, make_ofield=None, n_ofields=0, oopt=None, calibrate=0, what='', extra_fn=None,
                       budget=3.0, e2e_reps=5, cpu_fn=None, traffic_key=None):
        # (the plan call as one library call: `steps` calls per region for the configurations at the reference's sampling -- 0.1-0.2 ms a call --,
        # a few calls for the dense ones -- milliseconds each)
        sparse = e2e_reps >= 100
        rr = run_planner(E, torch, table, opt, steps, warmup, calibrate=calibrate, fence=fence, e2e_reps=min(e2e_reps, 20),
                         fresh_steps=steps if sparse else 3, fresh_reps=REPS_SHORT if sparse else 5)
        cpu = None
        if rank == 0 and cpu_on and cpu_fn is not None:
            cpu = cpu_fn()
        elif rank == 0 and cpu_on and make_ofield is not None:
            cpu = cpu_baseline_fields(make_ofield, n_ofields, oopt, budget, what)
        extra = extra_fn(rr) if extra_fn else None
        entry = config_entry(name, workload, rr, cpu, extra, traffic_key) if rank == 0 else None
        if entry is not None:
            configs.append(entry)
        rr['batch'].close()
        del rr
        torch.cuda.empty_cache()
        return entry

    if world == 1:
        import oracle as orc
        T1 = E.FieldTable.from_rectangles(LH1)
        if 'cfg1_clothoid' in want:
            e = planner_config('cfg1_clothoid', f'cfg1 x {args.fields}, clothoid turns (line-clothoid-arc-clothoid-line) at the reference\'s sample counts',
                               T1, E.make_options(1, 0.0), args.steps, args.warmup,
                               lambda k: orc.make_field(L=500.0, H=200.0), len(LH1), orc.Options.make(1, 1, 0.0, 0.5), what='500 x 200 m fields, clothoid, reference sampling',
                               e2e_reps=200)
            # (BASELINE.json's metric names the clothoid sampler: the same batch with clothoid turns, at top level beside the pinned arc model)
            out['value_clothoid'], out['ms_per_step_clothoid'] = e['value_fresh'], e['ms_fresh']
            out['value_step_clothoid'], out['ms_step_clothoid'], out['frac_clothoid'] = e['value'], e['ms_per_step'], e['roofline']['frac']
        if 'cfg1_x16384' in want:
            # the headline's fields, four times as many: 1 GB of output per step -- beyond the 256 MiB Infinity Cache that the headline's 249 MB fit into
            T16 = E.FieldTable.from_rectangles(WL.cfg1_batch(16384))
            planner_config('cfg1_x16384', 'cfg1 x 16384: the headline\'s fields, 16 384 of them (1.0 GB of output per step: beyond the 256 MiB Infinity Cache)',
                           T16, E.make_options(), args.steps, args.warmup, e2e_reps=100, traffic_key='cfg1')
        if 'cfg1_clothoid_dense' in want:
            planner_config('cfg1_clothoid_dense', f'cfg1 x {args.fields}, clothoid turns, uniform 0.1 m sample spacing',
                           T1, E.make_options(1, 0.1), max(3, args.steps // 20), 2,
                           lambda k: orc.make_field(L=500.0, H=200.0), len(LH1), orc.Options.make(1, 1, 0.1, 0.5), what='500 x 200 m fields, clothoid, 0.1 m',
                           calibrate=args.calibrate, e2e_reps=3)
        LH2 = WL.cfg2_rectangles()
        T2 = E.FieldTable.from_rectangles(LH2)
        for key, tm, sp, st_, wu in (('cfg2_ref', 0, 0.0, args.steps, args.warmup), ('cfg2_0.5', 1, 0.5, max(5, args.steps // 10), 2),
                                      ('cfg2_0.1', 1, 0.1, max(3, args.steps // 20), 2)):
            if key not in want:
                continue
            wl = (f'cfg2: 1024 random rectangular fields (edges U[100,1000) m, seed 1024), '
                  f'{"arc turns at the reference sampling" if sp == 0 else f"clothoid turns, {sp} m sample spacing"}')
            planner_config(key, wl, T2, E.make_options(tm, sp), st_, wu,
                           lambda k: orc.make_field(L=float(LH2[k, 0]), H=float(LH2[k, 1])), len(LH2), orc.Options.make(tm, 1, sp, 0.5),
                           what=f'cfg2 fields, {"arcs, reference sampling" if sp == 0 else f"clothoid, {sp} m"}', calibrate=args.calibrate if sp > 0 else 0,
                           e2e_reps=200 if sp == 0 else 3)
        if 'cfg3' in want:
            (L3, H3), obst = WL.cfg3_field()

            def cfg3_extra(rr):
                st3 = rr['res'].stats()
                return {'n_in_obstacle': int(st3['n_in_obstacle'][0]), 'n_outside': int(st3['n_outside'][0])}

            def cfg3_cpu():
                # the configuration itself, not a scaled stand-in: the whole 5000 x 2000 m field with its 32 obstacles at 0.05 m through
                # the oracle, once, on one thread (one path cannot be split over threads by the sequential loops)
                t0 = time.perf_counter()
                rc, p = orc.plan_field(orc.make_field(L=L3, H=H3, obstacles=obst), orc.Vehicle.make(), orc.Options.make(1, 1, 0.05, 0.5))
                tc = time.perf_counter() - t0
                assert rc == 0
                return {'value': p.n / tc, 'unit': 'points/s', 'cores': 1, 'kind': 'port', 'single_core_value': p.n / tc,
                        'sample': f'the whole cfg3 field (5000 x 2000 m, 32 obstacles, clothoid, 0.05 m: {p.n} points) once through oracle/fcpp_oracle.c on one thread, {tc:.1f} s'}
            planner_config('cfg3', 'cfg3: one 5000 x 2000 m field, 32 convex eight-gon obstacles, clothoid turns, 0.05 m sample spacing',
                           E.FieldTable.from_specs([E.FieldSpec(field_length=L3, field_width=H3, obstacles=obst)]), E.make_options(1, 0.05), max(5, args.steps // 10), 2,
                           extra_fn=cfg3_extra, calibrate=args.calibrate, e2e_reps=3, cpu_fn=cfg3_cpu)
        if 'cfg3_avoid' in want:
            (L3, H3), obst = WL.cfg3_field()

            def cfg3a_extra(rr):
                st3 = rr['res'].stats()
                fs = rr['res'].flagseg
                return {'n_in_obstacle': int(st3['n_in_obstacle'][0]), 'detour_points': int(((fs & E.L.KIND_MASK) == E.L.KIND_DETOUR).sum())}

            def cfg3a_cpu():
                t0 = time.perf_counter()
                rc, p = orc.plan_field(orc.make_field(L=L3, H=H3, obstacles=obst), orc.Vehicle.make(), orc.Options.make(1, 1, 0.05, 0.5, obstacle_mode=1))
                tc = time.perf_counter() - t0
                assert rc == 0
                return {'value': p.n / tc, 'unit': 'points/s', 'cores': 1, 'kind': 'port', 'single_core_value': p.n / tc,
                        'sample': f'the whole cfg3 field with obstacle-aware swaths ({p.n} points) once through oracle/fcpp_oracle.c on one thread, {tc:.1f} s'}
            planner_config('cfg3_avoid', 'cfg3 with obstacle-aware swaths (fcpp_options.obstacle_mode = AVOID, SURVEY.md 8f-4): every swath that meets one of '
                           'the 32 obstacles is clipped and driven around it; clothoid turns, 0.05 m',
                           E.FieldTable.from_specs([E.FieldSpec(field_length=L3, field_width=H3, obstacles=obst)]), E.make_options(1, 0.05, avoid_obstacles=True),
                           max(5, args.steps // 10), 2, extra_fn=cfg3a_extra, e2e_reps=3, cpu_fn=cfg3a_cpu)
        if 'cfg4' in want:
            configs.append(run_cfg4(E, torch, WL, cpu_on))
        if 'single_field' in want:
            out['single_field_ms'] = single_field_latency(torch)
    if world > 1 and 'cfg4' in want:
        entry = run_cfg4_sharded(E, S, WL, torch, dev, world, fence, allmax)
        if rank == 0:
            configs.append(entry)
    if 'cfg5' in want or world > 1:
        entry = run_cfg5(E, S, WL, torch, dist, np, rank, world, dev, cdev, fence, allmax, max(5, args.steps // 10), cpu_on, args.calibrate)
        if rank == 0:
            configs.append(entry)
    if world > 1 and 'cfg2_0.1' in want:
        LH2 = WL.cfg2_rectangles(seed=1024 + rank)
        rr = run_planner(E, torch, E.FieldTable.from_rectangles(LH2), E.make_options(1, 0.1), max(3, args.steps // 20), 2, fence=fence, e2e_reps=2)
        dt2, pts2 = allmax(rr['dt']), allsum(rr['points'])
        if rank == 0:
            e = config_entry('cfg2_0.1_weak', f'cfg2 weak-scaled: 1024 random rectangles PER GPU (seed 1024 + rank), clothoid, 0.1 m; rank 0\'s kernels', rr)
            e.update({'value': pts2 * max(3, args.steps // 20) / dt2, 'points': pts2, 'n_gpus': world, 'scaling': 'weak'})
            configs.append(e)
        rr['batch'].close()

    if rank == 0:
        out['configs'] = configs
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        detail = json.dumps(out)
        for path in (os.path.join(REPO, 'bench_detail.json'), os.path.join(REPO, 'gpurun_out', 'bench_detail.json')):
            try:
                if os.path.isdir(os.path.dirname(path)):
                    with open(path, 'w') as fh:
                        fh.write(detail + '\n')
            except OSError:
                pass
        sys.stderr.write(detail + '\n')
        sys.stderr.flush()
        os.write(real_stdout, (compact_line(out) + '\n').encode())


def single_field_latency(torch):
    """The drop-in planner on ONE field -- the reference's own case, 500 x 200 m (README_en.md:199-215: 0.046 s published; 27.6 ms measured
    for the reference code in the build container): constructor, plan_complete_coverage (warm: the planner keeps
    its batch), verify_all_corners_coverage, verify_curvature_constraints; median of 7 calls after a first one."""
    from field_coverage_path_planning_amd.multi_layer_planner_v3 import TwoLayerPathPlannerV37, VehicleParams
    t0 = time.perf_counter()
    pl = TwoLayerPathPlannerV37(VehicleParams(), field_length=500, field_width=200)
    r = pl.plan_complete_coverage()
    torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    rows = {'ctor': [], 'plan': [], 'plan_again': [], 'corners': [], 'verify': []}
    for _ in range(7):
        t0 = time.perf_counter()
        pl = TwoLayerPathPlannerV37(VehicleParams(), field_length=500, field_width=200)
        t1 = time.perf_counter()
        r = pl.plan_complete_coverage()
        t2 = time.perf_counter()
        r = pl.plan_complete_coverage()
        t3 = time.perf_counter()
        pl.verify_all_corners_coverage(r['headland'])
        t4 = time.perf_counter()
        pl.verify_curvature_constraints(r['main_work']['path'], r['main_work']['speeds'])
        t5 = time.perf_counter()
        for k, v in zip(rows, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            rows[k].append(v * 1e3)
    assert len(r['main_work']['path']) == 1256 and len(r['headland']['path']) == 435
    return {'ctor': median(rows['ctor']), 'plan': median(rows['plan']), 'plan_same_planner_again': median(rows['plan_again']),
            'corners': median(rows['corners']), 'verify': median(rows['verify']),
            'first_ctor_plus_plan_of_the_process': first, 'reference_published_plan_ms': 46.0, 'reference_measured_plan_ms': 27.6,
            'note': 'host wall-clock ms per call, results copied to the host as numpy arrays (what the reference returns); plan includes coverage_rate'}


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
        gens = 500                      # (the whole run: ~2.5 s on one thread)
        orc.ga_evolve(D, routes, max_generations=gens, convergence_threshold=10 ** 9, seed=4096)
        tc = time.perf_counter() - t0
        cpu = {'value': (gens + 1) * 4096 / tc, 'unit': 'chromosome evaluations/s', 'cores': 1, 'kind': 'port',
               'sample': f'{gens} generations of the same run (pop 4096, n 128) through oracle/fcpp_oracle.c: orc_ga_evolve, {tc:.2f} s on one thread'}
    bytes_per_chrom = 4 * 128 + 8
    return {'name': 'cfg4', 'workload': 'cfg4: GA over 128 nodes, population 4096, 500 generations (501 population evaluations), whole loop on the device',
            'generations': int(res.generations), 'ms_total': dt * 1e3, 'us_per_generation': dt / 500 * 1e6,
            'value': evals / dt, 'unit': 'chromosome evaluations/s', 'dtype': 'f64', 'gathers_per_s': evals * 128 / dt,
            'best_distance_m': float(res.best_distance),
            'fitness_only': {'ms_501_launches': dtf * 1e3, 'chromosomes_per_s': evals / dtf, 'gathers_per_s': evals * 128 / dtf},
            'roofline': {'bound': 'latency', 'kernel': 'k_ga_generation', 'achieved': evals * bytes_per_chrom / dt / 1e9, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': evals * bytes_per_chrom / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                         'note': 'algorithmic HBM bytes per chromosome = 4 n + 8 = 520 B (SURVEY.md 8d): the loop is bound by the latency of its 501 '
                                 'dependent launches, each as long as one workgroup\'s bookkeeping chain (D and the population live in L2 / LDS), not by HBM; '
                                 'the fraction is tiny by construction'},
            'cpu_baseline': cpu}


def run_cfg5(E, S, WL, torch, dist, np, rank, world, dev, cdev, fence, allmax, steps, cpu_on, calibrate=0):
    """cfg5: 65 536 parallelograms through sharding.plan_sharded -- one block of fields per rank, cut on the analytic point counts."""
    V = WL.cfg5_parallelograms()
    t0 = time.perf_counter()
    table = E.FieldTable.from_vertices(V).to_device()        # (records resident in device memory; the ranks' blocks are contiguous slices of them)
    t_table = (time.perf_counter() - t0) * 1e3
    veh, opt = E.make_vehicle(), E.make_options()
    # ---- the job end to end, E2E_REPS times: sizing of all fields (every rank, threaded, no collective), this rank's fresh batch, its
    # output arrays, one step, the stats gathered on rank 0
    e2e, batch, bufs, res = [], None, None, None
    for rep in range(9):             # (the first is dropped: allocations; median of the other eight)
        if batch is not None:
            batch.close()
            batch = bufs = res = None
        fence()
        t0 = time.perf_counter()
        # sizing of ALL fields, on this rank's GPU (fcpp_plan_points): what the ranks cut their blocks on -- one rank has nothing to cut
        counts = E.plan_points(table, veh, opt, device=dev.index) if world > 1 else None
        t1 = time.perf_counter()
        res = S.plan_sharded(table, veh, opt, device=dev.index, counts=counts)          # this rank's batch + buffers + one step + stats gather
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        batch = res.batch
        e2e.append({'ms': allmax((t2 - t0) * 1e3), 'count_ms': (t1 - t0) * 1e3, 'plan_sharded_ms': (t2 - t1) * 1e3, 'setup_ms': batch.setup_times()})
    bufs = (res.local.x, res.local.y, res.local.kappa, res.local.v, res.local.flagseg, res.local.stats_raw)
    warm = sorted(e2e[1:], key=lambda r: r['ms'])
    mid = warm[(len(warm) - 1) // 2]
    for _ in range(2):
        batch.run(bufs)
    fence()
    # ---- the step: the kernels of this rank's block, REPS repetitions of `steps` steps, median
    batch.set_profiling(True, every=max(4, steps // 4))
    dts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        for _ in range(steps):
            batch.run(bufs)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    kernels, _ = batch.stage_times()
    batch.set_profiling(False)
    dt_local = median(dts)
    fence()
    # ---- the job per step with the stats gather (sizing and batch setup done once, above)
    for _ in range(2):
        res = S.plan_sharded(table, veh, opt, device=dev.index, batch=batch, buffers=bufs, counts=counts)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = S.plan_sharded(table, veh, opt, device=dev.index, batch=batch, buffers=bufs, counts=counts)
    fence()
    dt_job = allmax(time.perf_counter() - t0)
    my_points = batch.total_points
    per_rank = torch.zeros(world, 2, dtype=torch.float64, device=cdev)
    per_rank[rank, 0], per_rank[rank, 1] = float(my_points), dt_local / steps
    if dist.is_initialized():
        dist.all_reduce(per_rank)
    dt_dev = allmax(dt_local)
    calibrated = None
    if calibrate > 1:
        bufs2 = batch.alloc(best_of=calibrate, include=[bufs])
        for _ in range(2):
            batch.run(bufs2)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            batch.run(bufs2)
        torch.cuda.synchronize()
        dt_cal = allmax(time.perf_counter() - t0)
        calibrated = {'ms_per_step': dt_cal / steps * 1e3, 'placement': getattr(batch, 'placement', None),
                      'note': f'opt-in Batch.alloc(best_of={calibrate}); NOT the primary figure'}
        del bufs2
    # the optional point-array gather, once
    fence()
    t0 = time.perf_counter()
    resg = S.plan_sharded(table, veh, opt, device=dev.index, batch=batch, buffers=bufs, counts=counts, gather_points=True)
    fence()
    t_with_gather = allmax(time.perf_counter() - t0)
    entry = None
    if rank == 0:
        total = int(counts.sum()) if counts is not None else int(batch.total_points)
        st = res.stats()
        assert res.stats_all.shape[0] == len(table) and int(st['n_viol'].sum()) == 0
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
                                             '(contiguous blocks cut on the analytic point counts; the only collective is the stats gather)',
                 'n_gpus': world, 'scaling': 'strong', 'points': total, 'output_arrays': getattr(batch, 'layout', None),
                 'ms_per_step': dt_dev / steps * 1e3, 'value': total * steps / dt_dev, 'unit': 'points/s', 'dtype': 'f64',
                 'timed_region': {'reps': len(dts), 'steps_per_rep': steps, 'ms_per_step_each_rep': [round(d / steps * 1e3, 5) for d in dts], 'reported': 'median (rank 0 shown)'},
                 'value_end_to_end': total / (mid['ms'] * 1e-3), 'ms_fresh': mid['ms'], 'value_fresh': total / (mid['ms'] * 1e-3),
                 'end_to_end': {'ms': mid['ms'], 'points_per_s': total / (mid['ms'] * 1e-3), 'count_ms': mid['count_ms'], 'plan_sharded_ms': mid['plan_sharded_ms'],
                                'setup_ms': {k: (round(v, 4) if isinstance(v, float) else v) for k, v in mid['setup_ms'].items()},
                                'first_ms': e2e[0]['ms'], 'all_ms': [round(r['ms'], 3) for r in e2e], 'table_from_vertices_ms': t_table,
                                'what': 'the sharded job from the field table to the gathered stats, fresh batch in a warm context (max over ranks): '
                                        'with more than one rank fcpp_plan_points over all fields on every rank\'s own GPU (no collective; one rank has nothing to cut) + this rank\'s engine.Batch (set up on the device) + '
                                        'output arrays + one step + stats gather; median of the repetitions after the first'},
                 'setup_ms': {k: (round(v, 4) if isinstance(v, float) else v) for k, v in mid['setup_ms'].items()},
                 'ms_per_job_with_stats_gather': dt_job / steps * 1e3,
                 'per_gpu': [{'rank': k, 'points': int(pr[k, 0]), 'ms_per_step': float(pr[k, 1] * 1e3), 'points_per_s': float(pr[k, 0] / pr[k, 1])}
                             for k in range(world)],
                 'point_array_gather': {'ms_job_with_gather': t_with_gather * 1e3, 'bytes_to_root': int(BYTES_PER_POINT * (total - pr[0, 0])),
                                        'note': 'one plan_sharded(gather_points=True): every peer sends its x, y, kappa, v, flagseg block straight into '
                                                'the root\'s arrays (dist.batch_isend_irecv)'},
                 'quiet_points': q_pts, 'general_points': g_pts,
                 'roofline': {'bound': 'valu_f64' if dom in VALU_BOUND_KERNELS else 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                              'traffic': traffic5 if world == 1 else None, 'algorithmic_bytes_per_launch': BYTES_PER_POINT * dom_points,
                              'kernel_ms': kernels[dom], 'kernel_points_per_launch': dom_points, 'all_kernels_ms': kernels, 'all_kernels_points': stage_points,
                              'rank': 0, 'step_frac': BYTES_PER_POINT * total / (dt_dev / steps) / 1e9 / HBM_PEAK_GBS / world},
                 'cpu_baseline': None}
        if calibrated:
            entry['calibrated'] = calibrated
        if cpu_on:
            import oracle as orc
            entry['cpu_baseline'] = cpu_baseline_fields(lambda k: orc.make_field(verts=[(float(a), float(b)) for a, b in V[k]]), len(V), orc.Options.make(),
                                                        3.0, 'cfg5 parallelograms, arcs, reference sampling')
    batch.close()
    return entry


if __name__ == '__main__':
    main()

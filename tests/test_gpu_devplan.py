"""The setup of a batch ON THE DEVICE (csrc/fcpp_devplan.hip) against the same setup on the host (csrc/fcpp_host.cpp + fcpp_tiler.cpp): every table of
the batch image byte for byte, fcpp_field_info byte for byte, and the results of a step bit for bit.  The host path is the checker: it is
what every oracle / golden parity test of rounds 1-3 ran (and still runs under FCPP_SETUP_HOST)."""
import ctypes as C
import os

import numpy as np
import pytest

from field_coverage_path_planning_amd import _lib as L
from field_coverage_path_planning_amd import engine as E
from field_coverage_path_planning_amd import workloads as WL

pytestmark = pytest.mark.gpu

TABLES = ['fields', 'prims', 'tiles', 'wave_tiles', 'general_ids', 'chunks', 'span_chunks', 'stat_ids', 'stat_first', 'stat_run', 'red_paths', 'field_work',
          'open_wave_ids', 'seg', 'seg_mask', 'partial', 'field_junc', 'work_totals', 'obs_off', 'obs_x', 'obs_y', 'obs_bbox', 'field_packs']


def _both(table, veh, opt):
    """-> (batch set up on the device, the same batch set up on the host without primitive sharing)"""
    ctx = E.get_context()
    ctx.set_setup('device')
    try:
        bd = E.Batch(table, veh, opt)
    finally:
        ctx.set_setup('host')
    os.environ['FCPP_NO_SHARE'] = '1'          # the device planner gives every field its own primitives
    try:
        bh = E.Batch(table, veh, opt)
    finally:
        del os.environ['FCPP_NO_SHARE']
        ctx.set_setup('auto')
    assert bd.setup_path() == 'device' and bh.setup_path() == 'host'
    return bd, bh


def _compare(bd, bh, what=''):
    import torch
    assert bd.total_points == bh.total_points, what
    assert bytes(bd.info.array.tobytes()) == bytes(bh.info.array.tobytes()), what
    for k, name in enumerate(TABLES):
        a, b = bd.debug_table(k), bh.debug_table(k)
        assert a.size == b.size, (what, name, a.size, b.size)
        if name == 'red_paths':          # (room for every field; only the fields k_reduce_stats reduces are listed)
            used = 4 * sum(bd.reduce_classes())
            a, b = a[:used], b[:used]
        if not np.array_equal(a, b):
            bad = np.flatnonzero(a != b)
            raise AssertionError(f'{what}: table {name} differs at byte {bad[0]} of {a.size} ({bad.size} bytes differ)')
    assert bd.reduce_classes() == bh.reduce_classes() and bd.point_split() == bh.point_split() and bd.stage_points() == bh.stage_points()
    rd, rh = bd.run(), bh.run()
    torch.cuda.synchronize()
    for name in ('x', 'y', 'kappa', 'v', 'flagseg', 'stats_raw'):
        assert torch.equal(getattr(rd, name), getattr(rh, name)), (what, name)
    ad, dd = bd.connectors()
    ah, dh = bh.connectors()
    assert torch.equal(torch.nan_to_num(ad), torch.nan_to_num(ah)) and torch.equal(torch.nan_to_num(dd), torch.nan_to_num(dh))
    bd.close()
    bh.close()


def test_shared_math_device_equals_host_bit_for_bit():
    import torch
    lib = L.load()
    ctx = E.get_context()
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.uniform(-np.pi, np.pi, 100000), rng.uniform(-1, 1, 100000), rng.uniform(-1e4, 1e4, 50000), [0.0, 1.0, -1.0],
                        0.5 + rng.uniform(-1.5e-8, 1.5e-8, 2000), np.cos(np.deg2rad(89.0)) + rng.uniform(-1.5e-8, 1.5e-8, 2000),   # (fc_acos's refined windows)
                        np.cos(np.deg2rad(91.0)) + rng.uniform(-1.5e-8, 1.5e-8, 2000), 0.5 + np.arange(-20, 21) * 2.0 ** -53])
    b = rng.uniform(-1e3, 1e3, a.size)
    for fn in range(6):          # (4, 5: the rotation's correctly rounded sine / cosine and angle, round 5)
        arg = np.clip(a, -1, 1) if fn == 2 else a
        h0, h1 = np.empty_like(arg), np.empty_like(arg)
        L.check(lib.fcpp_debug_math(fn, arg.size, C.c_void_p(arg.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(h0.ctypes.data), C.c_void_p(h1.ctypes.data)))
        da, db = torch.as_tensor(arg, device='cuda'), torch.as_tensor(b, device='cuda')
        d0, d1 = torch.empty_like(da), torch.zeros_like(da)
        L.check(lib.fcpp_debug_math_dev(ctx.handle, fn, arg.size, C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(d0.data_ptr()),
                                        C.c_void_p(d1.data_ptr())))
        assert np.array_equal(d0.cpu().numpy().view(np.uint64), h0.view(np.uint64)), fn
        if fn in (0, 4):
            assert np.array_equal(d1.cpu().numpy().view(np.uint64), h1.view(np.uint64))


def test_headline_and_cfg2_tables_equal_the_hosts():
    _compare(*_both(E.FieldTable.from_rectangles(WL.cfg1_batch(256)), E.make_vehicle(), E.make_options()), 'cfg1 x 256')
    _compare(*_both(E.FieldTable.from_rectangles(WL.cfg2_rectangles()), E.make_vehicle(), E.make_options()), 'cfg2')
    _compare(*_both(E.FieldTable.from_rectangles(WL.cfg2_rectangles()), E.make_vehicle(), E.make_options(1, 0.0)), 'cfg2 clothoid')


def test_parallelograms_tables_equal_the_hosts():
    V = WL.cfg5_parallelograms(4096)
    _compare(*_both(E.FieldTable.from_vertices(V), E.make_vehicle(), E.make_options()), 'cfg5 x 4096')
    _compare(*_both(E.FieldTable.from_vertices(V[:777]), E.make_vehicle(), E.make_options(ring_order=1)), 'cfg5 ring 1')


def test_speculative_and_exact_layouts_build_the_same_tables():
    """Small batches are laid out by per-field capacities and filled before the host has the totals (fcpp_api.cpp: try_device_setup);
    FCPP_SETUP_EXACT lays them out from the totals as large batches are.  Both against the host's tables -- the headline's fields (within
    the capacities), cfg2's (spans of up to fourteen chunks), and a vehicle with ten headland loops (beyond them: the fill pass runs twice)."""
    cases = [(E.FieldTable.from_rectangles(WL.cfg1_batch(300)), E.make_vehicle()), (E.FieldTable.from_rectangles(WL.cfg2_rectangles()), E.make_vehicle()),
             (E.FieldTable.from_rectangles(WL.cfg2_rectangles()[:200]), E.make_vehicle(working_width=0.8))]
    for table, veh in cases:
        for exact in (False, True):
            if exact:
                os.environ['FCPP_SETUP_EXACT'] = '1'
            try:
                _compare(*_both(table, veh, E.make_options()), f'exact={exact}')
            finally:
                os.environ.pop('FCPP_SETUP_EXACT', None)


def test_a_batch_beyond_the_one_scan_limit_equals_the_hosts():
    """More than 8192 fields: the counting pass, then the three-kernel scans -- the scan of the points derives what depends on the spans'
    alignment; at most 8192: ONE scan of one launch after the pass.  Both against the host's tables -- and the experiment FCPP_COUNT_CHUNKS
    (the counting pass in chunks on a second stream beside the planner's chunks; read once per process, so in a process of its own)."""
    V = WL.cfg5_parallelograms(9001)
    _compare(*_both(E.FieldTable.from_vertices(V), E.make_vehicle(), E.make_options()), 'cfg5 x 9001')
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from tests.test_gpu_devplan import _compare, _both; "
            "from field_coverage_path_planning_amd import engine as E, workloads as WL; "
            "_compare(*_both(E.FieldTable.from_vertices(WL.cfg5_parallelograms(9001)), E.make_vehicle(), E.make_options()), 'chunks'); print('chunks equal')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, FCPP_COUNT_CHUNKS='3'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'chunks equal' in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    _compare(*_both(E.FieldTable.from_vertices(V[:8192]), E.make_vehicle(), E.make_options()), 'cfg5 x 8192')


def _random_quads(rng, n):
    """convex quadrilaterals of every shape class, some too small to plan (they raise: status < 0), some degenerate"""
    out = np.empty((n, 4, 2))
    for k in range(n):
        w, h = rng.uniform(20, 900, 2)
        q = np.array([[0, 0], [w, 0], [w, h], [0, h]], dtype=np.float64)
        kind = rng.integers(0, 4)
        if kind == 1:
            q[2:, 0] += rng.uniform(-0.4, 0.4) * h
        elif kind == 2:
            q += rng.uniform(-0.12, 0.12, (4, 2)) * min(w, h)
        rot = rng.uniform(-np.pi, np.pi) if kind else 0.0
        q = q @ np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]]) + rng.uniform(-50, 50, 2) * (kind > 0)
        out[k] = q
    return out


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_random_fields_vehicles_and_points_tables_equal_the_hosts(seed):
    rng = np.random.default_rng(seed)
    n = 1500
    V = _random_quads(rng, n)
    V[::97] *= 0.01                                    # fields too small to plan: they raise (status < 0) on both paths alike
    starts = np.where(rng.random((n, 1)) < 0.5, rng.uniform(0, 600, (n, 2)), np.nan)
    ends = np.where(rng.random((n, 1)) < 0.5, rng.uniform(0, 600, (n, 2)), np.nan)
    table = E.FieldTable.from_vertices(V, start_points=starts, end_points=ends)
    vehicles = [E.make_vehicle(), E.make_vehicle(working_width=2.0, min_turn_radius=5.0), E.make_vehicle(working_width=6.0, min_turn_radius=6.0, max_work_speed_kmh=12.0),
                E.make_vehicle(max_longitudinal_accel=0.3, headland_turn_speed_kmh=6.0), E.make_vehicle(working_width=1.0, min_turn_radius=4.0)]
    veh = vehicles[seed % len(vehicles)]
    for tm, ring, tol in ((0, 0, 1e-6), (1, 1, 0.0), (0, 0, -0.5)):
        bd, bh = _both(table, veh, E.make_options(tm, 0.0, ring_order=ring, geofence_tol=tol))
        assert (bd.info.array['status'] != 0).any() and (bd.info.array['status'] == 0).any()
        _compare(bd, bh, f'seed {seed} turn model {tm} ring {ring} tol {tol}')


def test_fields_with_obstacle_polygons_flag_mode():
    rng = np.random.default_rng(5)
    specs = []
    for k in range(300):
        Lx, Hy = rng.uniform(150, 700, 2)
        obs = [[(float(cx + r * np.cos(t)), float(cy + r * np.sin(t))) for t in np.arange(6) * np.pi / 3]
               for cx, cy, r in zip(rng.uniform(30, Lx - 30, 3), rng.uniform(30, Hy - 30, 3), rng.uniform(3, 25, 3))] if k % 3 else None
        specs.append(E.FieldSpec(field_length=float(Lx), field_width=float(Hy), obstacles=obs))
    _compare(*_both(E.FieldTable.from_specs(specs), E.make_vehicle(), E.make_options()), 'obstacles, flag mode')


def test_what_the_device_planner_does_not_take_goes_to_the_host():
    """obstacle-aware swaths, and dense sampling of fields WITH obstacles (a run per line and turn); dense fields without obstacles are the
    device's since round 5 (test_dense_sampling_tables_equal_the_hosts)"""
    ctx = E.get_context()
    t = E.FieldTable.from_rectangles(WL.cfg2_rectangles(64))
    sq = [[(40.0, 40.0), (50.0, 40.0), (50.0, 50.0), (40.0, 50.0)]]
    t_obs = E.FieldTable.from_specs([E.FieldSpec(field_length=300.0 + 7 * k, field_width=200.0 + 3 * k, obstacles=sq) for k in range(24)])
    # (ten headland loops = 83 primitives per field: more than the dense block's lane each)
    for tab, opt, veh in ((t_obs, E.make_options(1, 0.5), E.make_vehicle()), (t, E.make_options(avoid_obstacles=True), E.make_vehicle()),
                          (t, E.make_options(0, 0.3), E.make_vehicle(working_width=0.8))):
        b = E.Batch(tab, veh, opt)
        assert b.setup_path() == 'host'
        b.close()
        ctx.set_setup('device')
        try:
            with pytest.raises(L.FcppError):
                E.Batch(tab, veh, opt)
        finally:
            ctx.set_setup('auto')
    b = E.Batch(t, E.make_vehicle(), E.make_options())
    assert b.setup_path() == 'device'
    b.close()


def test_turns_that_are_not_closed_form_make_the_whole_path_general():
    """a slow-accelerating vehicle: no span, the device tiler's window slides over thousands of points per field"""
    LH = np.array([[900.0, 700.0], [300.0, 1500.0], [120.0, 90.0], [2500.0, 2200.0]])
    bd, bh = _both(E.FieldTable.from_rectangles(LH), E.make_vehicle(max_longitudinal_accel=0.01), E.make_options())
    assert bd.point_split()[0] == 0 and bd.total_points > 20000
    _compare(bd, bh, 'no closed-form turns')
    bd, bh = _both(E.FieldTable.from_rectangles(LH), E.make_vehicle(working_width=6.0, min_turn_radius=6.0, max_work_speed_kmh=12.0), E.make_options())
    _compare(bd, bh, 'fast work speed')


def test_malformed_obstacle_ranges_are_refused_by_the_device_path():
    t = E.FieldTable.from_specs([E.FieldSpec(field_length=300.0, field_width=200.0, obstacles=[[(50.0, 50.0), (60.0, 50.0), (55.0, 60.0)]])])
    t.rec['n_obstacles'][0] = 5
    E.get_context().set_setup('device')
    try:
        with pytest.raises(L.FcppError):
            E.Batch(t, E.make_vehicle(), E.make_options())
    finally:
        E.get_context().set_setup('auto')


@pytest.mark.gpu
@pytest.mark.parametrize('opt', [dict(sample_spacing=0.5), dict(turn_model=1, sample_spacing=0.2), dict(turn_model=1, sample_spacing=0.1, clothoid_frac=0.3),
                                 dict(), dict(sample_spacing=0.7, avoid_obstacles=True)])
def test_chunk_lists_expanded_on_the_device_equal_the_hosts(opt):
    """Host-built images (dense sampling, obstacle-aware swaths, fields beyond the device planner) carry chunk GROUPS; the chunk lists of
    k_plan_quiet are expanded from them on the device (k_expand_chunks).  FCPP_HOST_CHUNKS=1 keeps the host's own lists: every table of the
    two batches -- the two chunk lists among them -- byte for byte, and the results of a step bit for bit."""
    import torch
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_parity import _random_fields, DEFAULT_VP, _veh
    specs, _ = _random_fields(21, 10, with_points=True)
    specs2, _ = _random_fields(22, 6, para=True, with_obstacles=True)
    o = E.make_options(**opt)
    ctx = E.get_context()
    ctx.set_setup('host')
    try:
        bd = E.Batch(specs + specs2, _veh(DEFAULT_VP), o)
        os.environ['FCPP_HOST_CHUNKS'] = '1'
        try:
            bh = E.Batch(specs + specs2, _veh(DEFAULT_VP), o)
        finally:
            del os.environ['FCPP_HOST_CHUNKS']
    finally:
        ctx.set_setup('auto')
    assert bd.total_points == bh.total_points and bd.stage_points() == bh.stage_points()
    n_chunk_bytes = 0
    for k, name in enumerate(TABLES):
        a, b = bd.debug_table(k), bh.debug_table(k)
        if name == 'red_paths':          # (room for every field; only the fields k_reduce_stats reduces are listed)
            used = 4 * sum(bd.reduce_classes())
            a, b = a[:used], b[:used]
        assert a.size == b.size and np.array_equal(a, b), (name, a.size, b.size)
        if name in ('chunks', 'span_chunks'):
            n_chunk_bytes += a.size
    if opt.get('sample_spacing', 0.0) > 0:
        assert n_chunk_bytes > 0
        assert bd.setup_times()['image_bytes'] < bh.setup_times()['image_bytes']          # the lists no longer travel
    rd, rh = bd.run(), bh.run()
    torch.cuda.synchronize()
    for name in ('x', 'y', 'kappa', 'v', 'flagseg', 'stats_raw'):
        assert torch.equal(getattr(rd, name), getattr(rh, name)), name
    bd.close(); bh.close()


def test_one_call_plan_equals_create_alloc_run():
    """Batch.plan (fcpp_batch_plan: creation, output arrays and one step in ONE library call) = Batch() + alloc() + run(), bit for bit; with
    the arrays from the context's output arena and from an allocation of their own; the batch serves further steps on the same arrays."""
    import torch
    rng = np.random.default_rng(21)
    LH = rng.uniform(100.0, 1000.0, size=(1500, 2))
    LH[7] = (15.0, 200.0)                                   # a field that raises (MLP:597-598): zero points, zero statistics
    table = E.FieldTable.from_rectangles(LH)
    veh = E.make_vehicle()

    def bits(t):
        return t.view(torch.int64) if t.dtype == torch.float64 else t

    for opt in (E.make_options(), E.make_options(1, 0.0), E.make_options(1, 0.5)):
        ref_b = E.Batch(table, veh, opt)
        ref = ref_b.run()
        torch.cuda.synchronize()
        want = [t.clone() for t in (ref.x, ref.y, ref.kappa, ref.v, ref.flagseg, ref.stats_raw)]
        ref_b.close()
        saved = E._contexts.get(0)
        for arena in (False, True):
            if arena:                                       # a context of its own with a small arena (the shared one stays as it is)
                E._contexts[0] = E.Context(0)
                E._contexts[0].reserve_outputs(lane_gib=0.25, pitch_gib=0.25)
            try:
                b, r = E.Batch.plan(table, veh, opt)
                fits = 8 * want[0].numel() <= (1 << 28)           # (arrays beyond a lane get an allocation of their own)
                assert b.layout['layout'] == ('arena' if arena and fits else 'plain') and b.total_points == want[0].numel()
                torch.cuda.synchronize()
                for got, w in zip((r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw), want):
                    assert torch.equal(bits(got), bits(w))
                r.x.zero_()
                r2 = b.run((r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw))
                torch.cuda.synchronize()
                assert torch.equal(bits(r2.x), bits(want[0])) and torch.equal(r2.stats_raw, want[5])
                assert b.info[7].status == E.L.EINVAL and b.info[8].n_main > 0
                b.close()
                del r, r2
            finally:
                if arena:
                    del E._contexts[0]
                    if saved is not None:
                        E._contexts[0] = saved


def test_dense_sampling_tables_equal_the_hosts():
    """Round 5: batches at dense sampling whose fields have no obstacles are set up on the device too -- the span of all complete passes, the
    quiet zones of the last swath line and of the headland straights as runs with their chunks, and between them general tiles (fine
    samplings) or wave tiles cut by the window cut, a stretch after the other (0.18 m and coarser) -- and must equal the host's tables
    byte for byte.  Fields with obstacles keep a run per line and turn: the host's (setup_path says so)."""
    R = WL.cfg2_rectangles()[:64]
    V = WL.cfg5_parallelograms(48)
    for tm, sp in ((1, 0.1), (0, 0.12), (1, 0.05), (1, 0.5), (0, 0.25), (1, 1.7)):
        _compare(*_both(E.FieldTable.from_rectangles(R), E.make_vehicle(), E.make_options(tm, sp)), f'rectangles, turn model {tm}, {sp} m')
    _compare(*_both(E.FieldTable.from_vertices(V), E.make_vehicle(), E.make_options(1, 0.1)), 'parallelograms, clothoid, 0.1 m')
    _compare(*_both(E.FieldTable.from_vertices(V), E.make_vehicle(), E.make_options(1, 0.5)), 'parallelograms, clothoid, 0.5 m')
    _compare(*_both(E.FieldTable.from_vertices(V), E.make_vehicle(working_width=1.2), E.make_options(0, 0.15)), 'parallelograms, three headland loops, 0.15 m')
    _compare(*_both(E.FieldTable.from_vertices(V), E.make_vehicle(working_width=1.2), E.make_options(0, 0.4)), 'parallelograms, three headland loops, 0.4 m')
    (L_, H_), obst = WL.cfg3_field()
    b = E.Batch([E.FieldSpec(field_length=L_ / 10, field_width=H_ / 10 + k, obstacles=[[(20.0, 20.0), (30.0, 20.0), (30.0, 30.0), (20.0, 30.0)]]) for k in range(20)],
                E.make_vehicle(), E.make_options(1, 0.1))
    assert b.setup_path() == 'host'
    b.close()


@pytest.mark.parametrize('seed', [21, 22, 23])
def test_random_fields_at_dense_sampling_tables_equal_the_hosts(seed):
    """Random quadrilaterals of every shape class (some too small to plan: they raise on both paths alike), start / end points, three
    vehicles (one, two and four headland loops), both turn models and ring orders, samplings on both sides of the wave-tile limit: the
    device's dense setup against the host's tables, byte for byte, and a step's results bit for bit."""
    rng = np.random.default_rng(seed)
    n = 96
    V = np.empty((n, 4, 2))
    for k in range(n):
        w, h = rng.uniform(90, 380, 2)
        q = np.array([[0, 0], [w, 0], [w, h], [0, h]], dtype=np.float64)
        kind = rng.integers(0, 3)
        if kind == 1:
            q[2:, 0] += rng.uniform(-0.35, 0.35) * h
        elif kind == 2:
            q += rng.uniform(-0.08, 0.08, (4, 2)) * min(w, h)
        rot = rng.uniform(-np.pi, np.pi) if kind else 0.0
        V[k] = q @ np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]]) + rng.uniform(-50, 50, 2) * (kind > 0)
    V[::31] *= 0.01
    starts = np.where(rng.random((n, 1)) < 0.5, rng.uniform(0, 300, (n, 2)), np.nan)
    ends = np.where(rng.random((n, 1)) < 0.5, rng.uniform(0, 300, (n, 2)), np.nan)
    table = E.FieldTable.from_vertices(V, start_points=starts, end_points=ends)
    vehicles = [E.make_vehicle(), E.make_vehicle(working_width=2.0, min_turn_radius=5.0), E.make_vehicle(working_width=1.0, min_turn_radius=4.0)]
    veh = vehicles[seed % len(vehicles)]
    for tm, ring, sp in ((0, 0, 0.12), (1, 1, 0.5), (1, 0, 0.2), (0, 1, 0.8)):
        bd, bh = _both(table, veh, E.make_options(tm, sp, ring_order=ring))
        assert (bd.info.array['status'] != 0).any() and (bd.info.array['status'] == 0).any()
        _compare(bd, bh, f'seed {seed} turn model {tm} ring {ring} spacing {sp}')


def test_field_records_in_device_memory_equal_host_records():
    """Round 5: a FieldTable kept in DEVICE memory (FieldTable.to_device(): fcpp_batch_plan / fcpp_batch_create / fcpp_plan_points read the
    records where they lie, nothing crosses PCIe before the first kernel) plans bit for bit what the same table in pageable and in pinned host
    memory plans -- on the device path (a batch), on the host paths that copy the records back (a handful of fields, FCPP_SETUP=host, AVOID
    mode), for a contiguous slice (a shard), and for the sizing calls."""
    import torch
    rng = np.random.default_rng(77)
    LH = rng.uniform(100.0, 900.0, size=(700, 2))
    LH[5] = (15.0, 200.0)                                   # a field that raises
    veh = E.make_vehicle()

    def bits(t):
        return t.view(torch.int64) if t.dtype == torch.float64 else t

    def planned(table, opt):
        b, r = E.Batch.plan(table, veh, opt)
        torch.cuda.synchronize()
        out = [bits(t).clone() for t in (r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw)], b.setup_path(), bytes(b.info.array.tobytes())
        b.close()
        return out

    ctx = E.get_context()
    for opt in (E.make_options(), E.make_options(1, 0.5)):
        host = E.FieldTable.from_rectangles(LH)
        pinned = E.FieldTable.from_rectangles(LH).pin()
        dev = E.FieldTable.from_rectangles(LH).to_device()
        assert dev._dev is not None and dev._dev.is_cuda and dev.c_args()[0] is not None
        want, path, info = planned(host, opt)
        assert path == 'device'
        for t in (pinned, dev):
            got, p, i = planned(t, opt)
            assert p == 'device' and i == info
            assert all(torch.equal(a, b) for a, b in zip(got, want))
        # a shard of the table on the device is on the device; a handful of fields goes to the host, which reads a copy
        for sl in (slice(100, 431), slice(3, 9)):
            w, pw, iw = planned(host[sl], opt)
            g, pg, ig = planned(dev[sl], opt)
            assert dev[sl]._dev is not None and pg == pw and ig == iw and all(torch.equal(a, b) for a, b in zip(g, w))
        assert planned(dev[3:9], opt)[1] == 'host'
        ctx.set_setup('host')
        try:
            g, pg, ig = planned(dev, opt)
            w, pw, iw = planned(host, opt)
        finally:
            ctx.set_setup('auto')
        assert pg == pw == 'host' and ig == iw and all(torch.equal(a, b) for a, b in zip(g, w))
        assert np.array_equal(E.plan_points(dev, veh, opt), E.plan_points(host, veh, opt))
        assert bytes(E.plan_count(dev, veh, opt).array.tobytes()) == bytes(E.plan_count(host, veh, opt).array.tobytes())
    # AVOID mode is planned on the host: records (and obstacle ranges) from the device copy
    specs = [E.FieldSpec(field_length=400.0 + 10 * k, field_width=220.0, obstacles=[[(150.0, 100.0), (170.0, 100.0), (170.0, 112.0), (150.0, 112.0)]])
             for k in range(20)]
    opt = E.make_options(avoid_obstacles=True)
    w, pw, iw = planned(E.FieldTable.from_specs(specs), opt)
    g, pg, ig = planned(E.FieldTable.from_specs(specs).to_device(), opt)
    assert pg == pw == 'host' and ig == iw and all(torch.equal(a, b) for a, b in zip(g, w))

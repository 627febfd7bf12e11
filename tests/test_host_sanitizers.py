"""The host-side planner under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (sanitizers on the host
library; the GPU pool runs none).  tests/native/host_sanitize_driver.cpp drives fcpp::build_host_plan -- what fcpp_plan_count and
fcpp_batch_create decide on the host -- with random and hostile inputs; any sanitizer report aborts the driver."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def driver():
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    out = os.path.join(REPO, 'build', 'host_sanitize_driver')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(REPO, 'tests', 'native', 'host_sanitize_driver.cpp'),
            os.path.join(REPO, 'field_coverage_path_planning_amd', 'csrc', 'fcpp_host.cpp')]
    cmd = [gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer',
           '-ffp-contract=off', '-pthread', '-o', out] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_host_planner_clean_under_asan_ubsan(driver, seed):
    r = subprocess.run([driver, str(seed), '400'], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr and 'LeakSanitizer' not in r.stderr, r.stderr[-4000:]
    words = r.stdout.split()
    assert words[0] == 'planned' and int(words[1]) > 500 and int(words[3]) > 100      # both outcomes exercised


@pytest.fixture(scope='module')
def tiler_driver():
    """tests/native/tiler_check_driver.cpp: host plan + tiler + image of random batches, every table checked (bounds, every path point
    planned by exactly one piece of kernel work), under ASan + UBSan"""
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    out = os.path.join(REPO, 'build', 'tiler_check_driver')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(REPO, 'field_coverage_path_planning_amd', 'csrc')
    srcs = [os.path.join(REPO, 'tests', 'native', 'tiler_check_driver.cpp'), os.path.join(csrc, 'fcpp_host.cpp'), os.path.join(csrc, 'fcpp_tiler.cpp')]
    cmd = [gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer',
           '-ffp-contract=off', '-pthread', '-Wno-unknown-pragmas', '-o', out] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def _run_tiler(driver, seed, rounds, **env):
    r = subprocess.run([driver, str(seed), str(rounds)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1', **env))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr and 'LeakSanitizer' not in r.stderr, r.stderr[-4000:]
    w = r.stdout.split()
    return dict(zip(w[0::2], w[1::2]))


@pytest.mark.parametrize('seed', [1, 2])
def test_batch_image_is_complete_and_independent_of_threads_and_sharing(tiler_driver, seed):
    """The image fcpp_batch_create uploads: (1) every path point of every field is planned by exactly one general tile, wave tile
    or chunk, all indices in bounds (checked inside the driver); (2) byte-identical for 1, 3 and 8 host threads; (3) the same per-field
    content whether equal fields share one plan (the headline's 4096 equal fields) or are planned one by one."""
    one = _run_tiler(tiler_driver, seed, 60, FCPP_THREADS='1')
    assert int(one['fields']) > 500 and int(one['shared']) > 100 and int(one['wave']) > 1000
    for t in ('3', '8'):
        assert _run_tiler(tiler_driver, seed, 60, FCPP_THREADS=t) == one
    alone = _run_tiler(tiler_driver, seed, 60, FCPP_THREADS='4', FCPP_NO_SHARE='1')
    assert int(alone['shared']) == 0 and alone['semantic'] == one['semantic'] and alone['tiles'] == one['tiles']


def test_malformed_obstacle_tables_are_refused_before_any_read():
    """A polygon table with NULL / decreasing / negative offsets or NULL coordinates must come back as FCPP_ESIZE / FCPP_EINVAL from
    fcpp_plan_count (both obstacle modes), not as a host crash: the table is validated before the planner touches it."""
    import ctypes as C
    import numpy as np
    from field_coverage_path_planning_amd import engine as E, _lib as L
    lib = L.load()
    table = E.FieldTable.from_rectangles([[500.0, 200.0]])
    table.rec['n_obstacles'], table.rec['obstacle_first'] = 1, 0
    arr, _polys, _keep = table.c_args()
    info = E.InfoTable(1)
    good_off = np.array([0, 4], dtype=np.int64)
    xs, ys = np.array([240.0, 260.0, 260.0, 240.0]), np.array([90.0, 90.0, 110.0, 110.0])
    ptr = lambda a, t: a.ctypes.data_as(t) if a is not None else C.cast(None, t)
    cases = [(1, None, xs, ys), (1, np.array([1, 4], dtype=np.int64), xs, ys), (1, np.array([0, -3], dtype=np.int64), xs, ys),
             (2, np.array([0, 4, 2], dtype=np.int64), xs, ys), (1, good_off, None, ys), (1, good_off, xs, None), (-1, good_off, xs, ys)]
    for avoid in (False, True):
        opt = E.make_options(avoid_obstacles=avoid)
        for n_polys, off, x, y in cases:
            polys = L.Polys(n_polys, ptr(off, L.c_i64_p), ptr(x, L.c_double_p), ptr(y, L.c_double_p))
            rc = lib.fcpp_plan_count(C.byref(E.make_vehicle()), C.byref(opt), 1, arr, C.byref(polys), info._c)
            assert rc in (L.ESIZE, L.EINVAL), (avoid, n_polys, rc)
        # a field whose obstacle range leaves the table
        table.rec['obstacle_first'] = 1
        polys = L.Polys(1, ptr(good_off, L.c_i64_p), ptr(xs, L.c_double_p), ptr(ys, L.c_double_p))
        assert lib.fcpp_plan_count(C.byref(E.make_vehicle()), C.byref(opt), 1, arr, C.byref(polys), info._c) == L.ESIZE
        table.rec['obstacle_first'] = 0
        assert lib.fcpp_plan_count(C.byref(E.make_vehicle()), C.byref(opt), 1, arr, C.byref(polys), info._c) == 0 and info[0].status == 0


def test_non_finite_and_degenerate_parameters_are_refused():
    """found by the run above: an infinite or denormal working width passed `W > 0` and reached an integer conversion"""
    from field_coverage_path_planning_amd import engine as E, _lib as L
    spec = [E.FieldSpec(field_length=500.0, field_width=200.0)]
    for kw in (dict(working_width=float('inf')), dict(working_width=1e-300), dict(min_turn_radius=float('inf')),
               dict(max_work_speed_kmh=float('nan')), dict(safety_factor=float('inf'))):
        with pytest.raises(L.FcppError) as ei:
            E.plan_count(spec, E.make_vehicle(**kw), E.make_options())
        assert ei.value.code == L.EINVAL
    for sp in (float('inf'), 1e-300):
        with pytest.raises(L.FcppError):
            E.plan_count(spec, E.make_vehicle(), E.make_options(1, sp))
    # a field of 3e8 swaths: refused (the swath index has 24 bits), the others of the batch are planned
    big = E.FieldSpec(field_vertices=[(0.0, 0.0), (1e9, 0.0), (1e9, 1e9), (0.0, 1e9)])
    infos = E.plan_count(spec + [big], E.make_vehicle(), E.make_options())
    assert infos[0].status == 0 and infos[1].status == L.ESIZE and infos[1].n_main == 0

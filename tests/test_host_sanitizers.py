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
           '-ffp-contract=off', '-o', out] + srcs
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

"""bench.py's last stdout line is what the driver parses: it must stay compact (the driver keeps an 8 KB tail; round 3's 27 KB line came back
`parsed: null`) and carry `roofline` and `cpu_baseline`.  No GPU needed: the line is built from a synthetic result record."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _fake_run(points=6926336):
    # (the headline since round 4: ONE launch per step)
    kernels = {'k_plan_quiet_spans': 0.0, 'k_plan_quiet': 0.0, 'k_plan_sparse': 0.0, 'k_plan_fused': 0.0, 'k_reduce_stats': 0.0, 'k_plan_sparse_fields': 0.0626}
    stage_points = {'k_plan_quiet_spans': 0, 'k_plan_quiet': 0, 'k_plan_sparse': 0, 'k_plan_fused': 0, 'k_reduce_stats': points, 'k_plan_sparse_fields': points}
    return {'dominant': 'k_plan_sparse_fields', 'dominant_points': points, 'kernel_timing': 'HIP events around the 20 launches of each timed region / 20 (one kernel per step)', 'kernels': kernels, 'stage_points': stage_points, 'prof_runs': 16, 'points': points,
            'ms_per_step': 0.0641}


def _fake_out(n_configs=12):
    r = _fake_run()
    e2e = {'ms': 0.19, 'points_per_s': 3.6e10, 'create_ms': 0.13, 'setup_ms': {'host_plan': 0.1}, 'what': 'x' * 400}
    out = {'metric': bench.METRIC, 'value': 4.2e10, 'unit': 'points/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 0.165, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'step': 'one plan call ' + 'y' * 300,
           'value_step': 1.08e11, 'ms_step': 0.0641, 'value_sustained': 6.0e10, 'ms_sustained': 0.1155, 'ms_pinned_records': 0.1741, 'value_pinned_records': 3.98e10, 'end_to_end_frac': 36 * 6926336 / 0.165e-3 / 8e12,
           'value_end_to_end': 3.6e10, 'end_to_end': e2e,
           'config': {'workload': 'w' * 700, 'turn_model': 'arc (reference, pinned)', 'points_per_gpu_step': r['points'], 'fields_per_gpu': 4096, 'setup': 'device'},
           'timed_region': {'reps': 25, 'steps_per_rep': 20, 'ms_per_step_each_rep': [0.0563, 0.0565, 0.0594] + [0.0545] * 22, 'reported': 'median'},
           'roofline': bench.roofline_of(r, 'cfg1'), 'value_clothoid': 4.0e10, 'value_step_clothoid': 1.0e11, 'frac_clothoid': 0.54, 'rccl_ranks': 1,
           'per_rank_points_per_s': [4.2e10] * 8, 'host_threads': 16,
           'cpu_baseline': {'value': 3.3e7, 'unit': 'points/s', 'cores': 16, 'kind': 'port', 'single_core_value': 5e6, 'sample': 's' * 600,
                            'python_loops_value': 3.2e4, 'numpy_value': 3.9e6, 'python_cores': 1, 'python_sample': 'p' * 300}}
    out['configs'] = [{'name': f'cfg_with_a_long_name_{k}', 'workload': 'v' * 500, 'ms_per_step': 1.9255567982327193, 'value': 141933305343.59515,
                       'ms_fresh': 2.7701, 'value_fresh': 98660000000.0, 'setup_ms': {'device_setup': k % 2},
                       'roofline': bench.roofline_of(r, None), 'end_to_end': dict(e2e), 'cpu_baseline': {'value': 33199297.29, 'sample': 'q' * 300}} for k in range(n_configs)]
    return out


def test_compact_line_is_short_and_complete():
    line = bench.compact_line(_fake_out())
    assert '\n' not in line and len(line) < 4096
    d = json.loads(line)
    assert d['metric'] == bench.METRIC and d['unit'] == 'points/s' and d['higher_is_better'] is True
    # `value` is the fresh plan call, the step on a batch already set up rides along
    assert d['value'] == 4.2e10 and d['ms_per_step'] == 0.165 and d['value_step'] == 1.08e11 and d['ms_step'] == 0.0641 and d['step'].startswith('one plan call')
    # ... and the sustained rate of two plan calls in flight, under its own name
    assert d['value_sustained'] == 6.0e10 and d['ms_sustained'] == 0.1155
    # `value` has the field records resident in HBM; the same call on records in pinned host memory rides along (round 5b)
    assert d['ms_pinned_records'] == 0.1741 and d['value_pinned_records'] == 3.98e10 and 'HBM' in d['step']
    rf = d['roofline']
    # the dominant kernel: its own launch duration and fraction, and what bounds it (the wave-tile kernels: float64 vector issue)
    assert rf['kernel'] == 'k_plan_sparse_fields' and rf['bound'] == 'valu_f64' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0
    assert abs(rf['frac'] - 36 * 6926336 / 0.0626e-3 / 8e12) < 1e-3 and abs(rf['kernel_ms'] - 0.0626) < 1e-6
    assert rf['algorithmic_bytes_per_launch'] == 36 * 6926336 and abs(rf['achieved'] - 36 * 6926336 / 0.0626e-3 / 1e9) < 1.0
    assert abs(rf['end_to_end_frac'] - 36 * 6926336 / 0.165e-3 / 8e12) < 1e-3 and abs(rf['step_frac'] - 36 * 6926336 / 0.0641e-3 / 8e12) < 1e-3
    if os.path.exists(os.path.join(os.path.dirname(bench.__file__), 'profiles', 'valu.json')):
        assert 0.1 < rf['valu_frac'] < 1.0
    assert set(rf['kernels']) == {'k_plan_sparse_fields'} and rf['kernel_timing'].startswith('HIP events')
    ms, pts, frac = rf['kernels']['k_plan_sparse_fields']
    assert abs(frac - 36 * pts / (ms * 1e-3) / 8e12) < 1e-3
    assert rf['traffic_source'].startswith('profiles/traffic.json') and rf['traffic'] > 0
    cb = d['cpu_baseline']
    assert cb['cores'] == 16 and cb['kind'] == 'port' and len(cb['sample']) <= 160 and cb['python_loops_value'] == 3.2e4 and cb['numpy_value'] == 3.9e6 and cb['python_cores'] == 1
    assert d['vs_baseline'] is None and len(d['config']['workload']) <= 200
    assert d['value_clothoid'] == 4.0e10 and d['frac_clothoid'] == 0.54
    assert len(d['configs']) == 13 and d['configs']['columns'][:3] == ['ms_plan_call', 'value_plan_call', 'ms_step']
    # ... and where every configuration's batch was set up (round 5: dense sampling on the device too)
    assert d['configs']['columns'][-1] == 'setup' and d['configs']['cfg_with_a_long_name_1'][-1] == 'device' and d['configs']['cfg_with_a_long_name_2'][-1] == 'host'
    # every region is K steps between two fences; their number, the first one's own value and the spread ride along
    assert d['timed_regions'] == {'n': 25, 'reported': 'median', 'first_ms': 0.0563, 'min_ms': 0.0545, 'max_ms': 0.0594}
    assert 'turn' in d['config']['workload'].lower() or d['config']['turn_model']


def test_compact_line_drops_optional_parts_before_it_grows_past_the_limit():
    line = bench.compact_line(_fake_out(n_configs=60))
    assert len(line) <= bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d['roofline'] and d['cpu_baseline']

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
    e2e = {'ms': 1.04, 'points_per_s': 6.7e9, 'create_ms': 0.93, 'setup_ms': {'host_plan': 0.47}, 'what': 'x' * 400}
    out = {'metric': bench.METRIC, 'value': 1.08e11, 'unit': 'points/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 0.0641, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': 3000.0, 'vs_baseline_of': 'cpu_baseline.value of this run', 'dtype': 'f64', 'data': 'synthetic',
           'value_end_to_end': 6.7e9, 'end_to_end': e2e,
           'config': {'workload': 'w' * 700, 'turn_model': 'arc (reference, pinned)', 'points_per_gpu_step': r['points'], 'fields_per_gpu': 4096, 'setup': 'device'},
           'timed_region': {'reps': 25, 'steps_per_rep': 20, 'ms_per_step_each_rep': [0.0563, 0.0565, 0.0594] + [0.0545] * 22, 'reported': 'median'},
           'roofline': bench.roofline_of(r, 'cfg1'), 'value_clothoid': 1.0e11, 'rccl_ranks': 1, 'per_rank_points_per_s': [1.08e11] * 8, 'host_threads': 16,
           'cpu_baseline': {'value': 3.3e7, 'unit': 'points/s', 'cores': 16, 'kind': 'port', 'single_core_value': 5e6, 'sample': 's' * 600}}
    out['configs'] = [{'name': f'cfg_with_a_long_name_{k}', 'workload': 'v' * 500, 'ms_per_step': 1.9255567982327193, 'value': 141933305343.59515,
                       'roofline': bench.roofline_of(r, None), 'end_to_end': dict(e2e), 'cpu_baseline': {'value': 33199297.29, 'sample': 'q' * 300}} for k in range(n_configs)]
    return out


def test_compact_line_is_short_and_complete():
    line = bench.compact_line(_fake_out())
    assert '\n' not in line and len(line) < 4096
    d = json.loads(line)
    assert d['metric'] == bench.METRIC and d['unit'] == 'points/s' and d['higher_is_better'] is True
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0
    # the headline fraction is the step's (it does not flip between two kernels 0.2 us apart), every kernel's own fraction rides beside it
    assert abs(rf['frac'] - 36 * 6926336 / 0.0641e-3 / 8e12) < 1e-3 and rf['frac'] == rf['step_frac']
    assert set(rf['kernels']) == {'k_plan_sparse_fields'} and rf['kernel_timing'].startswith('HIP events')
    ms, pts, frac = rf['kernels']['k_plan_sparse_fields']
    assert abs(frac - 36 * pts / (ms * 1e-3) / 8e12) < 1e-3
    assert rf['traffic_source'].startswith('profiles/traffic.json') and rf['traffic'] > 0
    assert d['cpu_baseline']['cores'] == 16 and d['cpu_baseline']['kind'] == 'port' and len(d['cpu_baseline']['sample']) <= 160
    assert d['vs_baseline'] == 3000.0 and len(d['config']['workload']) <= 200
    assert len(d['configs']) == 13 and d['configs']['columns'][0] == 'ms_per_step'
    # every region is K steps between two fences; their number, the first one's own value and the spread ride along
    assert d['timed_regions'] == {'n': 25, 'reported': 'median', 'first_ms': 0.0563, 'min_ms': 0.0545, 'max_ms': 0.0594}


def test_compact_line_drops_optional_parts_before_it_grows_past_the_limit():
    line = bench.compact_line(_fake_out(n_configs=60))
    assert len(line) <= bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d['roofline'] and d['cpu_baseline']

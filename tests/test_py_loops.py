"""The Python restatements of the per-point stages (oracle/py_loops.py: per-point loops as the reference runs them, and whole-array numpy)
against the reference's own outputs in tests/golden/golden_kernels.npz -- they are the `python_loops` / `numpy` CPU baselines of bench.py."""
import numpy as np

from oracle import py_loops as PL


class _Veh:
    def __init__(self, a):
        (self.working_width, self.min_turn_radius, self.max_work_speed_kmh, self.max_headland_speed_kmh, self.headland_turn_speed_kmh,
         self.max_lateral_accel, self.max_longitudinal_accel, self.safety_factor) = (float(x) for x in a)


def _paths(g):
    offs = g['sp_offsets']
    for k in range(len(offs) - 1):
        yield k, g['sp_path'][offs[k]:offs[k + 1]], slice(offs[k], offs[k + 1])


def test_python_loops_are_the_reference_bit_for_bit(golden_kernels):
    g = golden_kernels
    veh, veh2 = _Veh(g['vp_default']), _Veh(g['vp2'])
    for k, xy, sl in _paths(g):
        v_in = g['sp_v_in'][sl]
        out, _ = PL.speed_plan_loops(xy, v_in, veh)
        assert np.array_equal(out, g['sp_v_out'][sl])
        assert np.array_equal(PL.speed_plan_loops(xy, v_in, veh2)[0], g['sp2_v_out'][sl])
        assert np.array_equal(PL.smooth_loops(xy, v_in, veh.max_longitudinal_accel), g['sp_v_smooth_only'][sl])
        st = PL.verify_loops(xy, g['sp_v_out'][sl], veh)
        assert np.array_equal(st, g['ver_stats'][k])
        ln, tm = PL.metrics(xy, g['sp_v_out'][sl])
        assert ln == g['len_m'][k] and tm == g['time_s'][k]
        if k >= 1:
            assert np.array_equal(PL.verify_loops(xy, np.full(len(xy), 15.0), veh), g['ver15_stats'][k - 1])
    for tri, kap in zip(g['curv_tri'], g['curv_kappa']):
        assert PL.curvature_loops(tri[0], tri[1], tri[2]) == kap


def test_numpy_variant_within_tolerance(golden_kernels):
    g = golden_kernels
    veh, veh2 = _Veh(g['vp_default']), _Veh(g['vp2'])
    for k, xy, sl in _paths(g):
        v_in = g['sp_v_in'][sl]
        np.testing.assert_allclose(PL.speed_plan_numpy(xy, v_in, veh)[0], g['sp_v_out'][sl], rtol=0, atol=1e-9)
        np.testing.assert_allclose(PL.speed_plan_numpy(xy, v_in, veh2)[0], g['sp2_v_out'][sl], rtol=0, atol=1e-9)
        np.testing.assert_allclose(PL.smooth_numpy(xy, v_in, veh.max_longitudinal_accel), g['sp_v_smooth_only'][sl], rtol=0, atol=1e-9)
        st, ref = PL.verify_numpy(xy, g['sp_v_out'][sl], veh), g['ver_stats'][k]
        np.testing.assert_allclose(st[[0, 1, 3, 4]], ref[[0, 1, 3, 4]], rtol=1e-12, atol=1e-13)
        assert st[2] == ref[2] and st[5] == ref[5]
        if len(xy) >= 3:
            assert PL.speed_plan_numpy(xy, v_in, veh)[1] == PL.speed_plan_loops(xy, v_in, veh)[1]


def test_both_variants_on_a_whole_plan(golden_plans):
    """the reference's 500 x 200 m plan: main || headland through both variants = the reference's speeds"""
    g = golden_plans
    veh = _Veh(g['cfg1_500x200/vp'])
    xy = np.vstack([g['cfg1_500x200/main_path'], g['cfg1_500x200/head_path']])
    want = np.concatenate([g['cfg1_500x200/main_v'], g['cfg1_500x200/head_v']])
    nominal = np.empty(len(xy))
    # nominal speeds are not stored; the planned ones are a fixed point of the speed plan (clamp and sweeps are idempotent)
    nominal[:] = want
    assert np.array_equal(PL.speed_plan_loops(xy, nominal, veh)[0], want)
    np.testing.assert_allclose(PL.speed_plan_numpy(xy, nominal, veh)[0], want, rtol=0, atol=1e-9)
    st = PL.verify_loops(xy, want, veh)
    np.testing.assert_allclose(st[[0, 1, 3, 4]], g['cfg1_500x200/ver'][[0, 1, 3, 4]], rtol=1e-12, atol=1e-13)

"""Full-size runs (BASELINE.json configs) checked through size-independent properties on the device.

cfg2 = 1024 random rectangles, clothoid turns, 0.1 m spacing: 1.0e9 points, 36.5 GB of output per run.  The oracle cannot
follow at this size in test time, so the run is checked by what must hold for ANY valid plan:
  * the defining inequality of the sweeps on every segment, the clamp inequality and the a_lat bound on every point,
  * idempotence: planning the speeds of the produced path again changes nothing (the output is a fixed point),
  * checksums: the device-reduced stats equal independent torch reductions of the arrays,
  * both pipelines (fused / staged) agree to 1e-9, and a re-run is bit-identical,
  * a sample of whole fields equals the oracle.
"""
import numpy as np
import pytest
import torch

import oracle as orc
from field_coverage_path_planning_amd import _lib as L
from field_coverage_path_planning_amd import engine as E

pytestmark = pytest.mark.gpu


def _cfg2(n_fields=1024, seed=1024):
    rng = np.random.default_rng(seed)
    LH = rng.uniform(100.0, 1000.0, size=(n_fields, 2))
    return [E.FieldSpec(field_length=float(a), field_width=float(b)) for a, b in LH], LH


def _field_ids(batch, device):
    """int32 tensors: start offset of every field, and per-point 'first point of a field' / 'seam' masks."""
    offs = torch.tensor([i.point_offset for i in batch.info] + [batch.total_points], dtype=torch.int64, device=device)
    seam = torch.tensor([i.point_offset + i.n_main for i in batch.info], dtype=torch.int64, device=device)
    return offs, seam


def test_cfg2_full_size_properties():
    specs, LH = _cfg2()
    veh = E.make_vehicle()
    opt = E.make_options(L.TURN_CLOTHOID, 0.1)
    batch = E.Batch(specs, veh, opt)
    n = batch.total_points
    assert n > 1.0e9
    bufs = batch.alloc()                 # one set of output arrays for every run of this test (a set of this size is laid out over ~130 GiB)
    res = batch.run(bufs, mode=1)
    dev = res.x.device
    st = res.stats()
    offs, seam = _field_ids(batch, dev)

    # -- segment kinds and nominal speeds
    fs = res.flagseg.view(torch.int32)
    kind = fs & 7
    nominal = torch.tensor([9.0, 4.0, 15.0, 15.0, 4.0, 2.5, 0.0, 0.0], dtype=torch.float64, device=dev)[kind.long()]
    assert bool((res.v <= nominal).all()) and bool((res.v > 0).all())
    assert int((kind == L.KIND_SWATH).sum()) == sum(int(i.n_swaths) * _n_line(i, batch) for i in batch.info)
    assert int(((fs & L.FLAG_ALAT) != 0).sum()) == 0 and int(st['n_viol'].sum()) == 0
    assert int(((fs & L.FLAG_OUTSIDE) != 0).sum()) == int(st['n_outside'].sum())

    # -- clamp and lateral acceleration on every point
    ms = res.v / 3.6
    alat = ms * ms * res.kappa
    assert bool((res.kappa >= 0).all())
    assert float(alat.max()) <= veh.max_lateral_accel * veh.safety_factor ** 2 * (1 + 1e-9)
    np.testing.assert_allclose(float(res.kappa.max()), st['max_kappa'].max(), rtol=0, atol=0)
    np.testing.assert_allclose(float(alat.max()), st['max_alat'].max(), rtol=1e-12)
    del alat

    # -- the sweeps' defining inequality on every segment inside a field: |u_i - u_(i-1)| <= 2 a d  (u = (v/3.6)^2)
    u = ms * ms
    del ms
    dx = res.x[1:] - res.x[:-1]
    dy = res.y[1:] - res.y[:-1]
    d = torch.sqrt(dx * dx + dy * dy)
    del dx, dy
    inside = torch.ones(n - 1, dtype=torch.bool, device=dev)
    inside[offs[1:-1] - 1] = False                      # pairs straddling two fields
    du = (u[1:] - u[:-1]).abs()
    slack = du - 2 * veh.max_longitudinal_accel * d
    ok = (slack <= 1e-9) | (d < 1e-6) | ~inside
    assert bool(ok.all()), int((~ok).sum())
    del du, slack, ok, u

    # -- checksums: path length of each layer = sum of the segment lengths that belong to it
    layer_pairs = inside.clone()
    layer_pairs[seam - 1] = False                       # the seam main|headland belongs to neither layer
    total_len = float((d * layer_pairs).sum())
    np.testing.assert_allclose(st['main_len_m'].sum() + st['head_len_m'].sum(), total_len, rtol=1e-10)
    del d, inside, layer_pairs

    # -- a re-run is bit-identical; the staged pipeline agrees to 1e-9
    keep = {k: getattr(res, k).clone() for k in ('x', 'y', 'kappa', 'v')}
    keep_fs, keep_stats = res.flagseg.clone(), res.stats_raw.clone()
    res2 = batch.run(bufs, mode=1)
    for k in keep:
        assert torch.equal(getattr(res2, k), keep[k]), k
    assert torch.equal(res2.flagseg, keep_fs) and torch.equal(res2.stats_raw, keep_stats)
    res0 = batch.run(bufs, mode=0)
    for k, tol in (('x', 1e-9), ('y', 1e-9), ('kappa', 1e-7), ('v', 1e-7)):
        assert float((getattr(res0, k) - keep[k]).abs().max()) <= tol, k
    assert torch.equal(res0.flagseg, keep_fs)
    st0 = res0.stats()
    np.testing.assert_allclose(st0['main_len_m'], st['main_len_m'], rtol=1e-12)
    np.testing.assert_allclose(st0['main_time_s'], st['main_time_s'], rtol=1e-10)
    del res0, res2

    # -- idempotence of the speed plan on the produced paths (first 64 fields, 6e7 points)
    hi = int(offs[64])
    out, nadj = E.speed_plan(keep['x'][:hi], keep['y'][:hi], keep['v'][:hi], veh, clamp=True, offsets=offs[:65])
    assert float((out - keep['v'][:hi]).abs().max()) <= 1e-9 and int(nadj.sum()) == 0

    # -- three whole fields against the oracle
    oopt = orc.Options.make(1, 1, 0.1)
    for i in (0, 511, 1023):
        rc, p = orc.plan_field(orc.make_field(L=float(LH[i, 0]), H=float(LH[i, 1])), orc.Vehicle.make(), oopt)
        sl = res.field_slice(i)
        assert rc == 0 and p.n == sl.stop - sl.start
        np.testing.assert_allclose(keep['x'][sl].cpu().numpy(), p.xy[:, 0], rtol=0, atol=1e-9)
        np.testing.assert_allclose(keep['y'][sl].cpu().numpy(), p.xy[:, 1], rtol=0, atol=1e-9)
        np.testing.assert_allclose(keep['v'][sl].cpu().numpy(), p.v, rtol=0, atol=1e-6)
        assert np.array_equal(keep_fs[sl].cpu().numpy().view(np.uint32), p.flagseg)
        np.testing.assert_allclose(st['main_len_m'][i], p.main_len_m, rtol=1e-10)
        np.testing.assert_allclose(st['main_time_s'][i], p.main_time_s, rtol=1e-10)
    batch.close()


def test_cfg5_full_size_properties():
    """BASELINE.json configs[4] at full size: 65 536 rotated parallelograms at the reference's sampling (2.7e8 points, 9.8 GB of output) --
    the span kernel on tilted fields, 6e5 wave tiles, the 8-lane reduction class; checked by what must hold for any valid plan, by both
    pipelines agreeing, by the partition the sharded job cuts, and by whole fields against the oracle."""
    from field_coverage_path_planning_amd import sharding as S, workloads as WL
    V = WL.cfg5_parallelograms()
    specs = WL.specs_from_vertices(E, V)
    veh, opt = E.make_vehicle(), E.make_options()
    batch = E.Batch(specs, veh, opt)
    n = batch.total_points
    assert len(batch.info) == 65536 and all(i.status == 0 for i in batch.info) and 2.6e8 < n < 2.9e8
    # every path reduced by the 8-lane class of k_reduce_stats or, fields of three or four wave tiles, by its own workgroup of k_plan_sparse_fields
    assert sum(batch.reduce_classes()[1:]) == 0 and batch.reduce_classes()[0] > 30000
    bufs = batch.alloc()                 # one set of output arrays for every run of this test (a set of this size is laid out over ~130 GiB)
    res = batch.run(bufs, mode=1)
    dev = res.x.device
    st = res.stats()
    offs, seam = _field_ids(batch, dev)

    fs = res.flagseg.view(torch.int32)
    kind = fs & 7
    nominal = torch.tensor([9.0, 4.0, 15.0, 15.0, 4.0, 2.5, 0.0, 0.0], dtype=torch.float64, device=dev)[kind.long()]
    assert bool((res.v <= nominal).all()) and bool((res.v > 0).all()) and bool((res.kappa >= 0).all())
    for flag, key in ((L.FLAG_ALAT, 'n_viol'), (L.FLAG_OUTSIDE, 'n_outside'), (L.FLAG_OBSTACLE, 'n_in_obstacle')):
        assert int(((fs & flag) != 0).sum()) == int(st[key].sum()), key
    assert int(st['n_outside'].sum()) > 0 and int(st['n_in_obstacle'].sum()) == 0      # U-turns of skewed fields leave the polygon
    # per-field flag counts: a segmented sum over the fields
    fid = torch.repeat_interleave(torch.arange(len(batch.info), device=dev), offs[1:] - offs[:-1])
    per_field = torch.zeros(len(batch.info), dtype=torch.int64, device=dev).index_add_(0, fid, ((fs & L.FLAG_OUTSIDE) != 0).long())
    assert np.array_equal(per_field.cpu().numpy(), st['n_outside'])
    del fid, per_field
    ms = res.v / 3.6
    alat = ms * ms * res.kappa
    assert float(alat.max()) <= veh.max_lateral_accel * veh.safety_factor ** 2 * (1 + 1e-9)
    np.testing.assert_allclose(float(res.kappa.max()), st['max_kappa'].max(), rtol=0, atol=0)
    np.testing.assert_allclose(float(alat.max()), st['max_alat'].max(), rtol=1e-12)
    del alat
    # the sweeps' inequality on every segment inside a field, and the length checksum
    u = ms * ms
    dx, dy = res.x[1:] - res.x[:-1], res.y[1:] - res.y[:-1]
    d = torch.sqrt(dx * dx + dy * dy)
    del dx, dy, ms
    inside = torch.ones(n - 1, dtype=torch.bool, device=dev)
    inside[offs[1:-1] - 1] = False
    ok = ((u[1:] - u[:-1]).abs() - 2 * veh.max_longitudinal_accel * d <= 1e-9) | (d < 1e-6) | ~inside
    assert bool(ok.all()), int((~ok).sum())
    layer_pairs = inside.clone()
    layer_pairs[seam - 1] = False
    np.testing.assert_allclose(st['main_len_m'].sum() + st['head_len_m'].sum(), float((d * layer_pairs).sum()), rtol=1e-10)
    del d, inside, layer_pairs, ok, u

    # a re-run is bit-identical; the staged pipeline agrees
    keep = {k: getattr(res, k).clone() for k in ('x', 'y', 'kappa', 'v')}
    keep_fs, keep_stats = res.flagseg.clone(), res.stats_raw.clone()
    res2 = batch.run(bufs, mode=1)
    assert all(torch.equal(getattr(res2, k), keep[k]) for k in keep) and torch.equal(res2.flagseg, keep_fs) and torch.equal(res2.stats_raw, keep_stats)
    res0 = batch.run(bufs, mode=0)
    for k, tol in (('x', 1e-9), ('y', 1e-9), ('kappa', 1e-9), ('v', 1e-9)):
        assert float((getattr(res0, k) - keep[k]).abs().max()) <= tol, k
    assert torch.equal(res0.flagseg, keep_fs)
    st0 = res0.stats()
    for key in ('main_len_m', 'head_len_m', 'main_time_s', 'head_time_s'):
        np.testing.assert_allclose(st0[key], st[key], rtol=1e-10, err_msg=key)
    for key in ('n_viol', 'n_outside', 'n_adjusted'):
        assert np.array_equal(st0[key], st[key]), key
    del res0, res2

    # the blocks of the sharded job: contiguous, every GPU within one field of its share of the points
    counts = [i.n_main + i.n_head for i in batch.info]
    for world in (2, 4, 8):
        blocks = S.partition_by_points(counts, world)
        loads = [sum(counts[lo:hi]) for lo, hi in blocks]
        assert blocks[0][0] == 0 and blocks[-1][1] == 65536 and max(loads) <= n / world + max(counts)

    # whole fields against the oracle (first, last, and a spread in between)
    for i in (0, 1, 4097, 21845, 43690, 65534, 65535):
        verts = [(float(a), float(b)) for a, b in V[i]]
        rc, p = orc.plan_field(orc.make_field(verts=verts), orc.Vehicle.make(), orc.Options.make())
        sl = res.field_slice(i)
        assert rc == 0 and p.n == sl.stop - sl.start, i
        np.testing.assert_allclose(keep['x'][sl].cpu().numpy(), p.xy[:, 0], rtol=0, atol=1e-9)
        np.testing.assert_allclose(keep['y'][sl].cpu().numpy(), p.xy[:, 1], rtol=0, atol=1e-9)
        np.testing.assert_allclose(keep['kappa'][sl].cpu().numpy(), p.kappa, rtol=0, atol=1e-9)
        np.testing.assert_allclose(keep['v'][sl].cpu().numpy(), p.v, rtol=0, atol=1e-9)
        assert np.array_equal(keep_fs[sl].cpu().numpy().view(np.uint32), p.flagseg), i
        for key, ref in (('main_len_m', p.main_len_m), ('head_len_m', p.head_len_m), ('main_time_s', p.main_time_s), ('head_time_s', p.head_time_s)):
            np.testing.assert_allclose(st[key][i], ref, rtol=1e-10, err_msg=key)
        assert (st['n_outside'][i], st['n_adjusted'][i], st['n_viol'][i]) == (p.n_outside, p.n_adjusted, p.n_viol), i
    batch.close()


def _n_line(info, batch):
    """samples per swath line of a field: (n_main + n_turn) / P - n_turn with n_turn known from the batch options"""
    # n_main = P * n_line + (P - 1) * n_turn
    P = info.n_swaths
    n_turn = batch._n_turn if hasattr(batch, '_n_turn') else None
    if n_turn is None:
        # derive n_turn from two fields with different P (same for the whole batch)
        a, b = None, None
        for i in batch.info:
            if a is None:
                a = i
            elif i.n_swaths != a.n_swaths and (i.n_main - a.n_main) % 1 == 0:
                b = i
                break
        # with n_line unknown per field this cannot be solved from counts alone; use the host formula instead
        import math
        veh_R, ds = 8.0, batch.options.sample_spacing
        from oracle import lib as olib
        Re = olib().orc_cac_fit_radius(math.pi, veh_R, batch.options.clothoid_frac, batch.options.clothoid_fit)
        T = olib().orc_cac_length(math.pi, Re, batch.options.clothoid_frac)
        n_turn = max(2, int(math.ceil(T / ds)) + 1)
        batch._n_turn = n_turn
    return (info.n_main + n_turn) // P - n_turn

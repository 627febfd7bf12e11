"""A batch that ran on one torch stream and is closed while another is bound hands its device tables to the next batch
(fcpp_batch_destroy keeps the allocation as the context's spare): the destroy must drain the stream the batch really ran on."""
import numpy as np
import pytest

from field_coverage_path_planning_amd import engine as E

pytestmark = pytest.mark.gpu


def test_batch_closed_under_another_stream_does_not_disturb_its_own_runs():
    import torch
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(7)
    LH_a = rng.uniform(100.0, 1000.0, size=(1024, 2))
    LH_b = rng.uniform(100.0, 1000.0, size=(1024, 2))
    veh, opt = E.make_vehicle(), E.make_options()
    # reference results, everything on the default stream
    ref = {}
    for key, LH in (('a', LH_a), ('b', LH_b)):
        bt = E.Batch(E.FieldTable.from_rectangles(LH), veh, opt)
        r = bt.run()
        torch.cuda.synchronize()
        ref[key] = [t.clone() for t in (r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw)]
        bt.close()
    side = torch.cuda.Stream(device=dev)
    for _ in range(5):
        with torch.cuda.stream(side):
            a = E.Batch(E.FieldTable.from_rectangles(LH_a), veh, opt)
            bufs = a.alloc()
            for _ in range(20):             # a queue of steps on the side stream
                ra = a.run(bufs)
        # default stream bound: close a (its tables become the spare), create b at once (its H2D copy and setup kernels reuse them)
        a.close()
        b = E.Batch(E.FieldTable.from_rectangles(LH_b), veh, opt)
        rb = b.run()
        torch.cuda.synchronize()
        for got, want in zip((ra.x, ra.y, ra.kappa, ra.v, ra.flagseg, ra.stats_raw), ref['a']):
            assert torch.equal(got, want)
        for got, want in zip((rb.x, rb.y, rb.kappa, rb.v, rb.flagseg, rb.stats_raw), ref['b']):
            assert torch.equal(got, want)
        b.close()

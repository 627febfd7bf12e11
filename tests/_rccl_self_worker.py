"""tests/test_gpu_rccl.py: the collective legs of a sharded job over RCCL (torch.distributed backend "nccl") on the ONE GPU of the test
box -- a process group of one rank, sharding.FORCE_COLLECTIVES set, so that the stats rows, the point arrays and the GA fitness really
travel through ncclSend / ncclRecv / ncclAllGather on DEVICE tensors (no gloo, no host staging), plus the headline's asynchronous
gather of a stats ring.  usage: _rccl_self_worker.py PORT OUT_NPZ"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
port, out = sys.argv[1], sys.argv[2]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

from field_coverage_path_planning_amd import engine as E, sharding as S, workloads as WL      # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == 'nccl'
S.FORCE_COLLECTIVES = True

# 1. plan_sharded with the stats gather and the point-array gather: device tensors through the process group
table = E.FieldTable.from_vertices(WL.cfg5_parallelograms(192, seed=65536))
res = S.plan_sharded(table, E.make_vehicle(), E.make_options(), device=0, gather_points=True)
assert res.stats_all.is_cuda and all(a.is_cuda for a in res.points_all)
assert res.stats_all.data_ptr() != res.local.stats_raw.data_ptr()          # received copies, not the local tensors
torch.cuda.synchronize()
got = {'stats': res.stats_all.cpu().numpy(), 'local_stats': res.local.stats_raw.cpu().numpy()}
for k, (a, b) in enumerate(zip(res.points_all, (res.local.x, res.local.y, res.local.kappa, res.local.v, res.local.flagseg))):
    got[f'a{k}'], got[f'l{k}'] = a.cpu().numpy(), b.cpu().numpy()

# 2. the GA population: fitness of every block all-gathered (one rank: ncclAllGather of the whole block)
D, routes = WL.cfg4_ga(128, 1024)
fit, dst = S.ga_fitness_sharded(torch.as_tensor(routes, device=dev), torch.as_tensor(D, device=dev), device=0, with_distance=True)
d1, f1 = E.ga_fitness(routes, D)
got['fit'], got['fit_local'], got['dist'], got['dist_local'] = fit.cpu().numpy(), f1.cpu().numpy(), dst.cpu().numpy(), d1.cpu().numpy()

# 3. the headline's stats ring: asynchronous dist.gather of device tensors on RCCL's stream, double-buffered as in bench.py
ring = torch.arange(2 * 8 * 64 * 13, dtype=torch.int64, device=dev).reshape(16, 64, 13)
bufs = [[torch.empty((8, 64, 13), dtype=torch.int64, device=dev)] for _ in range(2)]
pending = []
for g in range(4):
    if pending:
        pending[-1].wait()
    half = g & 1
    pending.append(dist.gather(ring[half * 8:(half + 1) * 8], bufs[half], dst=0, async_op=True))
pending[-1].wait()
torch.cuda.synchronize()
got['ring0'], got['ring1'], got['ring'] = bufs[0][0].cpu().numpy(), bufs[1][0].cpu().numpy(), ring.cpu().numpy()
np.savez(out, **got)
dist.barrier()
dist.destroy_process_group()
print('rccl self worker OK')

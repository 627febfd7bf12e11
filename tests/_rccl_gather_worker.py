"""Child of tests/test_gpu_rccl.py: fcpp_gather (the C ABI's final gather) over a ONE-rank RCCL communicator created here through
ctypes -- ncclGetUniqueId / ncclCommInitRank on the RCCL the process has loaded (torch's) -- with flags = 1, so that the rank's own
block travels through ncclSend / ncclRecv instead of a device-to-device copy.  Usage: _rccl_gather_worker.py out.npz"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from field_coverage_path_planning_amd import _lib as L, engine as E, workloads as WL

out = sys.argv[1]
torch.cuda.set_device(0)
ctx = E.get_context(0)
# the RCCL library of this process: torch's bundled copy (same SONAME as the system's)
rccl = None
for line in open('/proc/self/maps'):
    if 'librccl' in line:
        rccl = C.CDLL(line.split()[-1])
        break
if rccl is None:
    torch.cuda.nccl.version()          # (loads it)
    for line in open('/proc/self/maps'):
        if 'librccl' in line:
            rccl = C.CDLL(line.split()[-1])
            break
assert rccl is not None, 'RCCL is not loaded'


class UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

b = E.Batch(E.FieldTable.from_vertices(WL.cfg5_parallelograms(64, seed=3)), E.make_vehicle(), E.make_options())
r = b.run()
torch.cuda.synchronize()
n, nf = b.total_points, b.n_fields
send = [r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw]
recv = [torch.zeros_like(t) for t in send]
elem = (C.c_int32 * 6)(8, 8, 8, 8, 4, 8 * L.STATS_WORDS)
sp = (C.c_void_p * 6)(*[t.data_ptr() for t in send])
rp = (C.c_void_p * 6)(*[t.data_ptr() for t in recv])
ctx.bind_stream()
lib = ctx.lib
# the point arrays and the statistics are different counts: two calls (elements of the points, then fields)
cnt = (C.c_int64 * 1)(n)
L.check(lib.fcpp_gather(ctx.handle, comm, 0, 1, 0, 5, sp, elem, cnt, rp, 1))
cnt_f = (C.c_int64 * 1)(nf)
sp6, rp6, el6 = (C.c_void_p * 1)(send[5].data_ptr()), (C.c_void_p * 1)(recv[5].data_ptr()), (C.c_int32 * 1)(8 * L.STATS_WORDS)
L.check(lib.fcpp_gather(ctx.handle, comm, 0, 1, 0, 1, sp6, el6, cnt_f, rp6, 1))
# ... and once without the flag: the root's own block as a device-to-device copy
recv2 = [torch.zeros_like(t) for t in send[:5]]
rp2 = (C.c_void_p * 5)(*[t.data_ptr() for t in recv2])
L.check(lib.fcpp_gather(ctx.handle, None, 0, 1, 0, 5, sp, elem, cnt, rp2, 0))
torch.cuda.synchronize()
assert all(torch.equal(a, c) for a, c in zip(send, recv)), 'gathered arrays differ'
assert all(torch.equal(a, c) for a, c in zip(send[:5], recv2))
rccl.ncclCommDestroy(comm)
np.savez(out, n=n, nf=nf, ok=1)
print('rccl gather worker OK', n, nf)

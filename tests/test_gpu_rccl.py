"""The RCCL legs of the multi-GPU path on the hardware that exists (one GPU): torch.distributed backend "nccl" with a one-rank
process group and sharding.FORCE_COLLECTIVES -- the stats gather, the direct point-array gather (batch_isend_irecv = ncclSend /
ncclRecv), the GA all-gather and the headline's asynchronous stats-ring gather all issue on DEVICE tensors (tests/_rccl_self_worker.py).
What comes back must be byte-identical to the local tensors.  The worker runs as a child with a hard timeout: a collective that hangs
is killed, not waited for."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_job_collectives_over_rccl_on_one_gpu(tmp_path):
    out = str(tmp_path / 'rccl.npz')
    port = str(36500 + (os.getpid() % 2000))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    p = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_rccl_self_worker.py'), port, out], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0 and 'rccl self worker OK' in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])
    g = np.load(out)
    assert g['stats'].shape == (192, 13) and np.array_equal(g['stats'], g['local_stats'])
    for k in range(5):
        assert g[f'a{k}'].size > 0 and np.array_equal(g[f'a{k}'], g[f'l{k}']), k
    assert np.array_equal(g['fit'], g['fit_local']) and np.array_equal(g['dist'], g['dist_local'])
    assert np.array_equal(g['ring0'], g['ring'][:8]) and np.array_equal(g['ring1'], g['ring'][8:])


def test_c_abi_gather_over_a_one_rank_rccl_communicator(tmp_path):
    """fcpp_gather (include/fcpp.h): ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the context's stream over the caller's communicator --
    here a one-rank communicator made through ctypes, the rank's own block forced through it (flags = 1)."""
    out = str(tmp_path / 'gather.npz')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    p = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_rccl_gather_worker.py'), out], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'rccl gather worker OK' in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])
    g = np.load(out)
    assert int(g['ok']) == 1 and int(g['n']) > 0

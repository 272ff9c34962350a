"""The RCCL branch of the multi-GPU driver on ONE GPU: a world-size-1 ``nccl`` process group (``backend="nccl"`` is
RCCL on ROCm) with ``force_collectives=True`` runs exactly the calls a rank of the 8-GPU job issues --
``broadcast`` of X, ``broadcast`` + ``scatter`` + ``all_gather_into_tensor`` of the per-cell state
(``multi.broadcast_state``), ``all_gather`` of the result table -- on device buffers, and the sharded grouped driver
on top of them must return the bits of the plain single-process evaluation.  One child process, so that a stuck
rendezvous cannot hang the test session."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, socket, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from gaussian_processes_amd import multi, synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine, fit_eval_group

with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

N, d, UNITS = 300, 64, 5                       # N not a multiple of 128; five units in groups of two
grid = syn.grid_for(d); lower, upper = syn.limits()
X0 = torch.from_numpy(syn.stimuli(N, d))
X = multi.broadcast_stimuli(X0, (N, d), dev, force_collectives=True)
assert X.is_cuda and torch.equal(X.cpu(), X0)

# per-cell state: broadcast of (r, m), scatter + all_gather_into_tensor of V, in both element types
r_np, m_np = syn.cell_inputs(N)
g = torch.Generator().manual_seed(11)
A = torch.randn(N, N, dtype=torch.float64, generator=g)
V0 = A @ A.T / N + torch.eye(N, dtype=torch.float64)
for dtype in (torch.float64, torch.float32):
    r, m, V = multi.broadcast_state(torch.from_numpy(r_np), torch.from_numpy(m_np), V0, N, dev, dtype=dtype,
                                    force_collectives=True)
    assert r.is_cuda and V.shape == (N, N) and V.dtype == dtype
    assert torch.equal(V.cpu(), V0.to(dtype)) and torch.equal(r.cpu(), torch.from_numpy(r_np).to(dtype))
    assert torch.equal(m.cpu(), torch.from_numpy(m_np).to(dtype))
r, m, V = multi.broadcast_state(torch.from_numpy(r_np), torch.from_numpy(m_np), V0, N, dev, force_collectives=True)

engines = [GPFitEngine(N, d) for _ in range(2)]
thetas = [syn.theta_eval(u) for u in range(UNITS)]
la, l0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]

def row(o):
    return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]

def group_fn(us):
    outs = fit_eval_group(engines[:len(us)], [thetas[u] for u in us], lower, upper, grid, X, r, m, V, la, l0)
    return [row(o) for o in outs]

table = multi.run_sharded(UNITS, None, dev, group_fn=group_fn, group=2, force_collectives=True)   # all_gather of the table
plain = torch.tensor([row(engines[0].fit_eval(thetas[u], lower, upper, grid, X, r, m, V, la, l0)) for u in range(UNITS)],
                     dtype=torch.float64)
assert table.is_cuda and table.shape == (UNITS, multi.RESULT_WIDTH)
assert torch.equal(table.cpu(), plain), (table.cpu() - plain).abs().max()
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", float(table[0, 0]))
"""


def test_rccl_collectives_of_the_multi_gpu_driver_on_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, "-c", _CHILD % ROOT], env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert any(line.startswith("RCCL_OK") for line in p.stdout.splitlines()), p.stdout[-1500:]

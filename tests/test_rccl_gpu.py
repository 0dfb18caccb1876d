"""The RCCL code path itself: `parallel.DataParallel(backend="nccl")` - torch.distributed's "nccl" backend IS
RCCL on ROCm - initialised on the GPU box as a one-rank group: init_process_group, broadcast_flat and allreduce_flat of an arena
of the generators' size (85.7 M fp32 = 343 MB, three 128 MB chunks), asynchronous handles, barrier, max_over_ranks.  A sum over
one rank must leave every element unchanged.  (Two GPUs exchanging bytes need a multi-GPU node: the driver's SCALE run.)"""
import os
import socket
import subprocess
import sys

import pytest

from conftest import PKG_NAME, ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import importlib, os, sys, time
import torch
sys.path.insert(0, %(root)r)
par = importlib.import_module(%(pkg)r + ".parallel")
import torch.distributed as dist
dp = par.DataParallel(backend="nccl")
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", dp.device_index)
n = 85_700_000
arena = torch.randn(n, device=dev)
ref = arena.clone()
par.broadcast_flat(arena)
torch.cuda.synchronize()
assert torch.equal(arena, ref), "broadcast changed the source rank's buffer"
t0 = time.perf_counter()
works = par.allreduce_flat(arena, wait=False)
assert len(works) == 3, "343 MB arena = three 128 MB all-reduce calls, got %%d" %% len(works)
dp.wait(works)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert torch.equal(arena, ref), "sum all-reduce over one rank changed the buffer"
small = torch.arange(18600, device=dev, dtype=torch.float32)
par.allreduce_flat(small)
torch.cuda.synchronize()
assert torch.equal(small, torch.arange(18600, device=dev, dtype=torch.float32))
dp.barrier()
assert par.max_over_ranks(1.25) == 1.25
print("rccl ok: backend=%%s world=%%d; 343 MB all-reduce in 3 chunks %%.1f ms; 74 kB all-reduce; barrier; max_over_ranks" %% (dist.get_backend(), dist.get_world_size(), dt * 1e3))
dist.destroy_process_group()
"""


def test_rccl_backend_runs_on_one_rank(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SSCG_DP_BACKEND", None)
    env.pop("SSCG_DP_SHARED_GPU", None)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "pkg": PKG_NAME}], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rccl ok: backend=nccl world=1" in r.stdout

"""The data-parallel exchange on CPU: world_size 2 over gloo (one process per rank), exercising the same
parallel.py code the MI355X run uses with RCCL: flat-arena all-reduce in chunks, broadcast from rank 0,
max-over-ranks timing, and the shard/average identity that makes the sharded step equal the full-batch step."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG_NAME


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = importlib.import_module(PKG_NAME + ".parallel")
    from oracle import fixtures as FX
    from oracle import nets
    dp = par.DataParallel(backend="gloo")
    assert dp.rank == rank and dp.world_size == world
    # 1. chunked all-reduce of a flat arena == sum over ranks
    n = 100003
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    par.allreduce_flat(flat, chunk=4096)
    ok1 = torch.equal(flat, torch.arange(n, dtype=torch.float32) * 3)
    # 2. broadcast: every rank ends with rank 0's weights
    w = torch.full((5000,), float(rank + 7))
    par.broadcast_flat(w, 0, chunk=1024)
    ok2 = bool((w == 7).all())
    # 3. max over ranks
    ok3 = par.max_over_ranks(1.0 + rank) == float(world)
    # 4. per-sample net (PixelDiscriminator, InstanceNorm): mean-loss gradients of the global batch equal the
    #    all-reduced shard gradients times 1/world  (the scale optim.FusedAdam folds into its kernel)
    C = 3
    sd = {k: v.double().requires_grad_(True) for k, v in FX.net_weights("pixel_3", "pixel", (C,), torch.float64).items()}
    x = FX.net_input("pixel_3", (4, C, 16, 16), torch.float64)

    def grads(xb):
        for v in sd.values():
            v.grad = None
        ((nets.pixel_discriminator(sd, xb) - 1.0) ** 2).mean().backward()
        return torch.cat([v.grad.flatten() for v in sd.values()])
    g_full = grads(x)
    g_shard = grads(x[rank * 2:(rank + 1) * 2]).float()
    par.allreduce_flat(g_shard)
    ok4 = float((g_shard.double() / world - g_full).abs().max() / g_full.abs().max()) < 1e-6
    # 5. the census behind bench.py's `rccl` object: every rank reports its device and its own step time; the process group's world
    #    size, the per-rank list in rank order and the number of DISTINCT devices come back identical on every rank.  (CPU ranks all
    #    report "cpu" = one device for two ranks: exactly the condition under which bench.py refuses to call a run an N-GPU run.)
    census = par.rank_census(10.0 + rank)
    ok5 = (census["world_size"] == world and census["per_rank_ms"] == [10.0, 11.0] and census["devices"] == ["cpu", "cpu"]
           and census["distinct_devices"] == 1 and census["backend"] == "gloo" and census["version"])
    # 6. bucketed exchange (SSCG_DP_BUCKETS; parallel.GradBuckets) on a CPU arena: parameters report "last gradient kernel queued" in
    #    reverse order, two of them never report (outside the counted paths) - their bucket goes out from finish().  Every element is
    #    exchanged exactly once: the result is the plain sum over ranks, whatever the bucketing.
    import types
    sizes = [640, 64, 1280, 256, 64, 2048, 128, 704]
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    params = [torch.nn.Parameter(torch.zeros(1)) for _ in sizes]
    opt = types.SimpleNamespace(slices={p_: (o, n_) for p_, o, n_ in zip(params, offs, sizes)}, grad=torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1))
    gb = par.GradBuckets(opt, 3)
    ok6 = gb.n == 3 and gb.bounds[0][0] == 0 and gb.bounds[-1][1] == sum(sizes) and all(a[1] == b[0] for a, b in zip(gb.bounds, gb.bounds[1:]))
    gb.begin()
    for p_ in reversed(params[2:]):          # (params 0 and 1 never report)
        gb.ready(p_)
    early = list(gb.order)
    for w in gb.finish():
        w.wait()
    ok6 = ok6 and early == [2, 1] and gb.order == [2, 1, 0] and torch.equal(opt.grad, torch.arange(sum(sizes), dtype=torch.float32) * 3)
    dp.barrier()
    q.put((rank, ok1, ok2, ok3, ok4, bool(ok5), bool(ok6)))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r

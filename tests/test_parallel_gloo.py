"""The data-parallel exchange on CPU: world_size 2 over gloo (one process per rank), exercising the same
parallel.py code the MI355X run uses with RCCL: flat-arena all-reduce in chunks, broadcast from rank 0,
max-over-ranks timing, and the shard/average identity that makes the sharded step equal the full-batch step."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG_NAME, load_sub


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
    census = par.rank_census(10.0 + rank, host_issue_ms=5.0 + rank, affinity={"numa_node": rank, "cpus": 4, "pinned": True})
    ok5 = (census["world_size"] == world and census["per_rank_ms"] == [10.0, 11.0] and census["devices"] == ["cpu", "cpu"]
           and census["distinct_devices"] == 1 and census["backend"] == "gloo" and census["version"]
           and census["per_rank_host_issue_ms"] == [5.0, 6.0] and [a["numa_node"] for a in census["per_rank_affinity"]] == [0, 1])
    # 5b. the preflight bench.py runs before any step (round 6): per-rank batch < 2 is refused on every rank, two ranks on ONE device
    #     ("cpu", "cpu") are refused unless the one-GPU test rig is declared, and then the gathered rows come back in rank order;
    #     a process group switches the reverse-order buckets on (SSCG_DP_BUCKETS unset -> DEFAULT_BUCKETS, "0" -> one piece).
    def refused(fn):
        try:
            fn()
        except SystemExit as e:
            return "preflight" in str(e)
        return False
    rows = par.preflight(dp, 8, shared_gpu_ok=True)
    os.environ.pop("SSCG_DP_BUCKETS", None)
    nb_default = par.dp_buckets(world)
    os.environ["SSCG_DP_BUCKETS"] = "0"
    nb_off = par.dp_buckets(world)
    os.environ.pop("SSCG_DP_BUCKETS")
    ok5 = (ok5 and refused(lambda: par.preflight(dp, 1, shared_gpu_ok=True)) and refused(lambda: par.preflight(dp, 8))
           and [r_["rank"] for r_ in rows] == [0, 1] and nb_default == par.DEFAULT_BUCKETS == 4 and nb_off == 0)
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


def test_rank_pinning_follows_the_gpus_numa_node(tmp_path, monkeypatch):
    """parallel.pin_to_device_node (round 6): the rank's affinity becomes the CPUs sysfs lists as local to the GPU's PCI
    function, intersected with the affinity it had; no answer from sysfs, SSCG_DP_PIN=0 or an empty intersection leave it alone."""
    par = load_sub("parallel")
    assert par.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and par.parse_cpulist("") == []
    have = sorted(os.sched_getaffinity(0))
    monkeypatch.setattr(par, "device_pci_address", lambda index: "0000:%02x:00.0" % (0x10 + index))
    d = tmp_path / "bus" / "pci" / "devices" / "0000:10:00.0"
    d.mkdir(parents=True)
    (d / "numa_node").write_text("1\n")
    half = have[:max(1, len(have) // 2)]
    (d / "local_cpulist").write_text(",".join(str(c) for c in half) + ",9999\n")
    try:
        assert par.device_node_cpus(0, str(tmp_path)) == (1, half + [9999])
        assert par.device_node_cpus(1, str(tmp_path)) == (None, [])                   # no such device in the tree
        monkeypatch.setenv("SSCG_DP_PIN", "0")
        assert par.pin_to_device_node(0, str(tmp_path))["pinned"] is False and sorted(os.sched_getaffinity(0)) == have
        monkeypatch.delenv("SSCG_DP_PIN")
        info = par.pin_to_device_node(0, str(tmp_path))
        if len(half) < len(have):
            assert info == {"numa_node": 1, "cpus": len(half), "pinned": True} and sorted(os.sched_getaffinity(0)) == half
        else:                                                                          # a one-CPU container: nothing to narrow
            assert info["pinned"] is False
        assert par.pin_to_device_node(1, str(tmp_path))["pinned"] is False
    finally:
        os.sched_setaffinity(0, have)


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

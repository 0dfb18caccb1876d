"""Data parallelism for the step: one process per MI355X, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-device (arch/ops.py:31-34 uses gpu_ids[0] only); this layer is new.
The path shards over the batch dimension with exactly one exchange per optimiser step: a sum all-reduce of
the optimiser's flat gradient arena (G: ~85.7 M fp32 = 343 MB after gen_loss.backward(), D: ~18.6 k floats
after discriminator_loss.backward()); the 1/world scale is folded into the fused Adam kernel.
BatchNorm statistics, image pools and data streams stay per rank (the reference's per-replica semantics);
initial weights are broadcast from rank 0.  The arena is reduced in a few large chunks so that an xGMI
ring is bandwidth- not latency-bound (7 links x ~153 GB/s per GPU)."""
import os

import torch
import torch.distributed as dist

CHUNK_ELEMS = 32 * 1024 * 1024   # 128 MB fp32 per all-reduce call


class DataParallel:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("SSCG_DP_SHARED_GPU"):      # test rig: all ranks on GPU 0 (with SSCG_DP_BACKEND=gloo) - exercises the
            self.local_rank = 0                       # multi-rank control flow of the step / bench on a 1-GPU box
        if not dist.is_initialized():
            backend = backend or os.environ.get("SSCG_DP_BACKEND")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
        if torch.cuda.is_available():
            # side lanes made before the group existed are low-priority streams: beside RCCL's stream those cost +30 % (functional.side_priority)
            from . import functional as F
            F.reset_side_streams()

    def attach(self, g_opt, d_opt, nets):
        """Make every rank start from rank 0's weights and tell the optimisers the world size."""
        for opt in (g_opt, d_opt):
            opt.world_size = self.world_size
            broadcast_flat(opt.arena)
            _refresh_operand_copies(opt)
        for net in nets:
            for t in list(net.parameters()) + list(net.buffers()):
                if getattr(t, "_sscg_grad", None) is None:   # arena-resident parameters were broadcast above
                    dist.broadcast(t.data, 0)

    def attach_one(self, opt, nets):
        """Single-optimiser drivers (supervised_model)."""
        opt.world_size = self.world_size
        broadcast_flat(opt.arena)
        _refresh_operand_copies(opt)
        for net in nets:
            for t in list(net.parameters()) + list(net.buffers()):
                if getattr(t, "_sscg_grad", None) is None:
                    dist.broadcast(t.data, 0)

    def sync_grads(self, opt):
        allreduce_flat(opt.grad)

    def begin_backward(self, opt):
        """SSCG_DP_BUCKETS=n (n > 1): exchange `opt`'s gradient arena in n buckets, each as soon as the backward pass has queued the
        last gradient kernel of its parameters (reverse-order overlap with the backward).  Call before the step's
        forwards; `sync_grads_async` then returns the handles of the buckets already in flight plus the rest."""
        n = int(os.environ.get("SSCG_DP_BUCKETS", "0"))
        if n < 2:
            return
        b = getattr(opt, "_sscg_buckets", None)
        if b is None or b.n_asked != n:
            b = opt._sscg_buckets = GradBuckets(opt, n)
        b.begin()

    def sync_grads_async(self, opt):
        """Start the all-reduce of an optimiser's gradient arena and return the work handles: the generator
        exchange (343 MB) then runs over xGMI while the discriminator step computes (nothing in the D step reads
        generator weights, so applying the G update after it is the same arithmetic as model.py:474)."""
        b = getattr(opt, "_sscg_buckets", None)
        if b is not None and b.active:
            return b.finish()
        return allreduce_flat(opt.grad, wait=False)

    @staticmethod
    def wait(works):
        for w in works or ():
            w.wait()

    def barrier(self):
        dist.barrier()


class GradBuckets:
    """Reverse-order bucketed exchange of one optimiser's gradient arena (off unless SSCG_DP_BUCKETS=n).

    The arena is in parameter order; a bucket is a contiguous range of whole parameters.  The backward pass reaches the parameters
    roughly last-to-first, several times over (every DeepLab is used by two or three passes of the step): functional counts a
    parameter's uses in the forward and reports it (`functional.GRAD_READY`) when the backward has queued its LAST gradient kernel;
    a bucket whose parameters have all reported is exchanged at once - on a stream of its own, behind one event per lane that may
    carry gradient kernels - while the rest of the backward runs.  Buckets that never complete (parameters outside the counted
    paths, unused parameters) go out from `finish()`, after the backward.  The launch order is a function of the step's graph
    alone, hence identical on every rank (collectives must be issued in one order everywhere); the sums are the same
    all-reduces over the same elements as the one-piece exchange, so the result does not depend on the bucketing."""

    def __init__(self, opt, n):
        from . import functional as F
        self.F, self.opt, self.n_asked = F, opt, n
        items = sorted(((off, cnt, p) for p, (off, cnt) in opt.slices.items()), key=lambda t: t[0])
        total = opt.grad.numel()
        self.bounds, self.bucket_of, self.size = [], {}, []
        start, b = 0, 0
        for i, (off, cnt, p) in enumerate(items):
            self.bucket_of[id(p)] = b
            if len(self.size) <= b:
                self.size.append(0)
            self.size[b] += 1
            end = items[i + 1][0] if i + 1 < len(items) else total
            if (end >= (b + 1) * total / n and b < n - 1) or i + 1 == len(items):
                self.bounds.append((start, end))
                start, b = end, b + 1
        self.n = len(self.bounds)
        self.active = False
        self.comm = torch.cuda.Stream(device=opt.grad.device) if opt.grad.is_cuda else None

    def begin(self):
        for p in self.opt.slices:
            p._sscg_uses = 0
        self.left = list(self.size)
        self.launched = [False] * self.n
        self.works = []
        self.order = []
        self.active = True
        self.F.GRAD_READY[0] = self.ready

    def ready(self, p):
        b = self.bucket_of.get(id(p))
        if b is None or not self.active:
            return
        self.left[b] -= 1
        if self.left[b] == 0 and not self.launched[b]:
            self._launch(b)

    def _launch(self, b):
        lo, hi = self.bounds[b]
        self.launched[b] = True
        self.order.append(b)
        flat = self.opt.grad[lo:hi]
        if self.comm is None:
            self.works += allreduce_flat(flat, wait=False)
            return
        dev = flat.device
        evs = self.F.lane_events(dev)
        for ev in evs:
            self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            self.works += allreduce_flat(flat, wait=False)

    def finish(self):
        """After the backward pass: what has not gone out yet, then every handle (in launch order)."""
        self.F.GRAD_READY[0] = None
        for b in range(self.n - 1, -1, -1):
            if not self.launched[b]:
                self._launch(b)
        self.active = False
        if self.comm is not None:       # `wait()` on a handle orders the CALLING stream behind the collective; the comm stream's own
            torch.cuda.current_stream(self.opt.grad.device).wait_stream(self.comm)   # prologue (event waits) is ordered here
        return self.works


def _refresh_operand_copies(opt):
    """The optimiser's operand copies of its parameters (bf16 shadow / split planes) follow a broadcast of the arena."""
    if opt.arena16 is not None:
        opt.refresh_shadow()
    if getattr(opt, "arena_x3", None) is not None:
        opt.refresh_split()


def allreduce_flat(flat, chunk=CHUNK_ELEMS, wait=True):
    """In-place sum all-reduce of a 1-D buffer in large chunks (async; `wait=False` returns the work handles)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and dist.get_backend() != "nccl"):
        return flat if wait else []
    works = []
    n = flat.numel()
    for off in range(0, n, chunk):
        works.append(dist.all_reduce(flat[off:min(n, off + chunk)], op=dist.ReduceOp.SUM, async_op=True))
    if not wait:
        return works
    for w in works:
        w.wait()
    return flat


def broadcast_flat(flat, src=0, chunk=CHUNK_ELEMS):
    # (a one-rank RCCL group still runs the collective: `SSCG_FORCE_DP=1` / tests/test_rccl_gpu.py exercise the real code path)
    if not dist.is_initialized() or (dist.get_world_size() == 1 and dist.get_backend() != "nccl"):
        return flat
    n = flat.numel()
    for off in range(0, n, chunk):
        dist.broadcast(flat[off:min(n, off + chunk)], src)
    return flat


def device_identity(index=None):
    """A string that names the physical GPU this rank computes on: the device UUID where torch exposes it, else PCI domain:bus:device
    (two ranks on one GPU report the same string; a CPU-only rank reports "cpu")."""
    if not torch.cuda.is_available():
        return "cpu"
    index = torch.cuda.current_device() if index is None else index
    p = torch.cuda.get_device_properties(index)
    uuid = getattr(p, "uuid", None)
    if uuid is not None:
        return str(uuid)
    return "pci %04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", index), getattr(p, "pci_device_id", 0))


def collective_version():
    """(library, version string) of the collective backend in use: RCCL reports through torch.cuda.nccl.version() on ROCm."""
    if not dist.is_initialized():
        return None, None
    backend = dist.get_backend()
    if backend == "nccl":
        try:
            v = torch.cuda.nccl.version()
            return "rccl", ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:          # the figure is a report, never a reason to fail a run
            return "rccl", "unknown (%s)" % type(e).__name__
    return backend, torch.__version__


def rank_census(ms_per_step):
    """What the job actually ran on, gathered from every rank (bench.py's `rccl` object): world size as the process group sees it,
    the physical device of each rank, how many DISTINCT devices that is, the collective library's version and every rank's own
    ms per step.  The bench line's `n_gpus` is WORLD_SIZE from the environment; this is the evidence beside it."""
    me = {"rank": int(os.environ.get("RANK", "0")), "device": device_identity(), "ms_per_step": round(float(ms_per_step), 3)}
    if dist.is_initialized() and dist.get_world_size() > 1:
        rows = [None] * dist.get_world_size()
        dist.all_gather_object(rows, me)
    else:
        rows = [me]
    rows.sort(key=lambda r: r["rank"])
    lib, ver = collective_version()
    return {"backend": lib, "version": ver, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "distinct_devices": len(set(r["device"] for r in rows)), "devices": [r["device"] for r in rows],
            "per_rank_ms": [r["ms_per_step"] for r in rows]}


def max_over_ranks(value):
    """Max of a host float over ranks (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

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
        # the device this rank computes on: LOCAL_RANK among the visible devices - or device 0 where a launcher has narrowed every
        # rank's visibility to its own GPU (HIP_VISIBLE_DEVICES per process: one visible device, LOCAL_RANK still counts up)
        self.device_index = self.local_rank
        if torch.cuda.is_available() and self.local_rank >= torch.cuda.device_count():
            self.device_index = 0
        if not dist.is_initialized():
            backend = backend or os.environ.get("SSCG_DP_BACKEND")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.device_index)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            try:
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
            except Exception as e:          # a port somebody else holds, a launcher that died: say where the rendezvous was, not only torch's trace
                raise RuntimeError("rank %d/%d: no %s process group at %s:%s (%s: %s) - a MASTER_PORT clash shows up here; pick a free "
                                   "port (bench.py --gpus N without a launcher finds one itself)" % (
                                       self.rank, self.world_size, backend, os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"],
                                       type(e).__name__, e)) from e
        # the rank's host thread issues ~5 000 launches per step: keep it (and everything it spawns) on the cores of the GPU's own
        # NUMA node - eight ranks on two sockets otherwise share one scheduler domain and cross the socket link for every doorbell
        self.affinity = pin_to_device_node(self.device_index) if torch.cuda.is_available() else None
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
        n = dp_buckets(self.world_size)
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


DEFAULT_BUCKETS = 4


def dp_buckets(world_size):
    """Number of reverse-order gradient buckets: SSCG_DP_BUCKETS when set (0 / 1 = the one-piece exchange after the backward), else
    DEFAULT_BUCKETS whenever a process group exists (round 6: at world size 8 the ring moves 2 * 7/8 * 343 MB per rank over xGMI;
    in four pieces three of them are on the wire while the backward still computes)."""
    v = os.environ.get("SSCG_DP_BUCKETS")
    if v is not None and v != "":
        return int(v)
    return DEFAULT_BUCKETS if dist.is_initialized() else 0


def device_pci_address(index):
    """dddd:bb:dd.f of a visible device (torch's device properties carry domain / bus / device; GPUs are function 0)."""
    p = torch.cuda.get_device_properties(index)
    return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)


def parse_cpulist(text):
    """'0-31,64-95' -> sorted list of CPU numbers (the sysfs cpulist format)."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def device_node_cpus(index, sysfs="/sys"):
    """(numa_node, cpus) of the GPU's PCI function as the kernel reports them (`numa_node`, `local_cpulist`); (None, []) when sysfs
    has no answer (containers without the PCI tree, numa_node == -1 with an empty list)."""
    try:
        base = os.path.join(sysfs, "bus/pci/devices", device_pci_address(index))
        node = int(open(os.path.join(base, "numa_node")).read())
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
        if node < 0:
            node = None
        return node, cpus
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None, []


def pin_to_device_node(index, sysfs="/sys"):
    """Restrict this process (the issue thread and whatever it spawns later) to the CPUs local to GPU `index`, intersected with
    the affinity it already has (a launcher's or a container's cpuset wins).  SSCG_DP_PIN=0 leaves the affinity alone.
    Returns what was done: {"numa_node", "cpus": count, "pinned": bool}."""
    node, cpus = device_node_cpus(index, sysfs)
    info = {"numa_node": node, "cpus": len(os.sched_getaffinity(0)), "pinned": False}
    if os.environ.get("SSCG_DP_PIN", "1") == "0" or not cpus:
        return info
    want = set(cpus) & os.sched_getaffinity(0)
    if not want or want == os.sched_getaffinity(0):
        return info
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return info
    info.update(cpus=len(want), pinned=True)
    return info


def preflight(dp, batch_per_rank, shared_gpu_ok=False):
    """Fail before any step runs, with one clear message, on what would make an N-GPU figure meaningless or crash it late:
    per-rank batch below 2 (model.py:435's squeeze_(0) drops the batch dimension at B = 1), ranks that do not sit
    on N distinct devices, a rank whose device is not the one its LOCAL_RANK names.  Collective: every rank calls it."""
    if batch_per_rank < 2:
        raise SystemExit("preflight: per-rank batch %d < 2 - the step needs B >= 2 on every rank (reference model.py:435 squeezes the "
                         "batch dimension away at B = 1; BASELINE's DDP configurations are 64 / 8 = 8 and 32 / 8 = 4)" % batch_per_rank)
    if dp is None:
        return None
    me = {"rank": dp.rank, "device": device_identity(), "local_rank": getattr(dp, "device_index", dp.local_rank),
          "current": torch.cuda.current_device() if torch.cuda.is_available() else -1, "affinity": getattr(dp, "affinity", None)}
    rows = [None] * dist.get_world_size()
    if dist.get_world_size() > 1:
        dist.all_gather_object(rows, me)
    else:
        rows = [me]
    rows.sort(key=lambda r: r["rank"])
    distinct = len(set(r["device"] for r in rows))
    if distinct != len(rows) and not shared_gpu_ok:
        raise SystemExit("preflight: %d ranks on %d distinct devices %s - not a %d-GPU job (SSCG_DP_SHARED_GPU=1 marks the one-GPU test "
                         "rig)" % (len(rows), distinct, [r["device"] for r in rows], len(rows)))
    for r in rows:
        if r["current"] >= 0 and r["current"] != r["local_rank"]:
            raise SystemExit("preflight: rank %d computes on device %d but its LOCAL_RANK / visibility says %d" % (r["rank"], r["current"], r["local_rank"]))
    return rows


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
    """A string that names the physical GPU this rank computes on: the device UUID where torch exposes it, and PCI domain:bus:device
    (two ranks on one GPU report the same string; a CPU-only rank reports "cpu")."""
    if not torch.cuda.is_available():
        return "cpu"
    index = torch.cuda.current_device() if index is None else index
    p = torch.cuda.get_device_properties(index)
    pci = "pci %04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", index), getattr(p, "pci_device_id", 0))
    uuid = getattr(p, "uuid", None)
    # (both: a runtime that reports one placeholder UUID for every device must not make eight GPUs look like one to the preflight)
    return pci if uuid is None else "%s %s" % (uuid, pci)


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


def rank_census(ms_per_step, host_issue_ms=None, affinity=None):
    """What the job actually ran on, gathered from every rank (bench.py's `rccl` object): world size as the process group sees it,
    the physical device of each rank, how many DISTINCT devices that is, the collective library's version and every rank's own
    ms per step.  The bench line's `n_gpus` is WORLD_SIZE from the environment; this is the evidence beside it."""
    me = {"rank": int(os.environ.get("RANK", "0")), "device": device_identity(), "ms_per_step": round(float(ms_per_step), 3),
          "host_issue_ms": None if host_issue_ms is None else round(float(host_issue_ms), 2), "affinity": affinity}
    if dist.is_initialized() and dist.get_world_size() > 1:
        rows = [None] * dist.get_world_size()
        dist.all_gather_object(rows, me)
    else:
        rows = [me]
    rows.sort(key=lambda r: r["rank"])
    lib, ver = collective_version()
    return {"backend": lib, "version": ver, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "distinct_devices": len(set(r["device"] for r in rows)), "devices": [r["device"] for r in rows],
            "per_rank_ms": [r["ms_per_step"] for r in rows],
            "per_rank_host_issue_ms": [r.get("host_issue_ms") for r in rows],
            "per_rank_affinity": [r.get("affinity") for r in rows]}


def max_over_ranks(value):
    """Max of a host float over ranks (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

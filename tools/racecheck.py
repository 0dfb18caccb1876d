"""Launch-log race detector for the multi-stream step (debug tool; SSCG_RACECHECK=1 or `racecheck.install()`).

The step (model.semisuper_cycleGAN.step) issues ~5500 kernels per iteration on four or five HIP streams.  Every ordering between
two streams is something THIS package asked for (wait_stream / wait_event / record_stream / the allocator) - a missing one is a data
race that only shows when the streams' timing shifts.  This module checks the orderings on the host, without relying on timing:

  * a vector clock per stream: every libsscg launch ticks its stream's component; `a.wait_stream(b)`, `event.record(s)` /
    `s.wait_event(event)`, host synchronisation and the autograd engine's producer -> consumer stream hand-over merge clocks;
  * shadow memory: per address range the last write (stream, clock, clock vector, kernel) and the last read per stream.  Which
    pointer arguments of an entry point are read and which are written comes from include/sscg.h (`const T*` = read), their extents
    from the tensor whose `data_ptr()` produced them;
  * a launch that touches a range must be ordered (happens-before, by the clocks) behind every conflicting earlier access of another
    stream: read-after-write, write-after-write, write-after-read -> report;
  * allocator reuse: when an address range comes back as a NEW storage (torch's caching allocator hands a freed block to the next
    allocation of the same stream at once), every earlier access of ANOTHER stream to it must either be ordered before the new
    owner's stream or have been announced with `record_stream` (then the allocator itself waited) -> else report "reuse".

What it cannot see: kernels torch launches itself (`torch.cat`, `copy_`, autograd's own accumulation) - a handful per step, all on
the stream of their neighbours - and RCCL's internal stream, which is modelled at `torch.distributed.all_reduce`.

The core (`RaceCore`) is plain Python and is unit-tested on the CPU (tests/test_racecheck.py); the torch layer patches
`torch.cuda.Stream/Event`, `Tensor.data_ptr/record_stream` and wraps the library handle.
"""
import os
import re
import sys

from sortedcontainers import SortedDict

BIG = 8 << 20          # ranges of at least this many bytes live in a short list (arenas, full-resolution maps), the rest in a sorted map


class Rec(object):
    __slots__ = ("start", "end", "w", "reads", "alloc")

    def __init__(self, start, end, alloc):
        self.start, self.end, self.alloc = start, end, alloc
        self.w = None          # (stream, clock, clock vector at the write, what)
        self.reads = {}        # stream -> (clock, what)


class Alloc(object):
    __slots__ = ("base", "nbytes", "pool", "recorded", "weak", "gen")

    def __init__(self, base, nbytes, pool, weak, gen):
        self.base, self.nbytes, self.pool, self.weak, self.gen = base, nbytes, pool, weak, gen
        self.recorded = set()   # streams announced with record_stream: the allocator waits for them before it reuses the block


class RaceCore(object):
    """Vector clocks + shadow memory.  Streams are any hashable keys."""

    def __init__(self):
        self.vc = {}
        self.events = {}
        self.host = {}                  # what the host has waited for: precedes everything issued afterwards
        self.small = SortedDict()       # start -> {end: Rec}
        self.big = []
        self.allocs = SortedDict()      # base -> Alloc
        self.reports = {}
        self.order = []
        self.launches = 0
        self.gen = 0
        self.names = {}
        self.maxlen = 1

    # ---- clocks
    def clock(self, s):
        v = self.vc.get(s)
        if v is None:
            v = self.vc[s] = {s: 0}
        if self.host:
            self._join(v, self.host)
        return v

    @staticmethod
    def _join(dst, src):
        for k, c in src.items():
            if dst.get(k, 0) < c:
                dst[k] = c

    def tick(self, s):
        v = self.clock(s)
        v[s] = v.get(s, 0) + 1
        self.launches += 1
        return v

    def wait_stream(self, waiter, other):
        if waiter != other:
            self._join(self.clock(waiter), self.clock(other))

    def record_event(self, ev, s):
        """`ev`: any object with a writable attribute (torch.cuda.Event), or a hashable key."""
        c = dict(self.clock(s))
        try:
            ev._rc_clock = c
        except AttributeError:
            self.events[ev] = c

    def _event_clock(self, ev):
        c = getattr(ev, "_rc_clock", None)
        if c is None:
            try:
                c = self.events.get(ev)
            except TypeError:
                c = None
        return c

    def wait_event(self, s, ev):
        c = self._event_clock(ev)
        if c is not None:
            self._join(self.clock(s), c)

    def host_sync(self, s=None):
        """The host waited for stream s (None: for the device)."""
        for k in ([s] if s is not None else list(self.vc)):
            self._join(self.host, self.clock(k))

    def host_sync_event(self, ev):
        c = self._event_clock(ev)
        if c is not None:
            self._join(self.host, c)

    def merge_from_writer(self, s, ptr):
        """The autograd engine orders a consumer node's stream behind the producer of each incoming gradient."""
        rec = self._exact(ptr)
        if rec is not None and rec.w is not None and rec.w[0] != s:
            self._join(self.clock(s), rec.w[2])
            # ... and announces the gradient to the allocator on the consumer's stream (InputBuffer::add calls record_stream;
            # confirmed on this torch by tests/aids/engine_handover_probe.py)
            self.record_stream(ptr, s)

    # ---- shadow memory
    def _exact(self, ptr):
        d = self.small.get(ptr)
        if d:
            for rec in d.values():
                if rec.w is not None:
                    return rec
        for rec in self.big:
            if rec.start == ptr and rec.w is not None:
                return rec
        return None

    def _overlapping(self, start, end):
        out = []
        for k in self.small.irange(start - self.maxlen, end, inclusive=(True, False)):
            for rec in self.small[k].values():
                if rec.end > start:
                    out.append(rec)
        for rec in self.big:
            if rec.start < end and rec.end > start:
                out.append(rec)
        return out

    def _rec(self, start, end, alloc):
        if end - start >= BIG:
            for rec in self.big:
                if rec.start == start and rec.end == end:
                    return rec
            rec = Rec(start, end, alloc)
            self.big.append(rec)
            return rec
        d = self.small.get(start)
        if d is None:
            d = self.small[start] = {}
        rec = d.get(end)
        if rec is None:
            rec = d[end] = Rec(start, end, alloc)
            if end - start > self.maxlen:
                self.maxlen = end - start
        return rec

    def _alloc_of(self, ptr):
        i = self.allocs.bisect_right(ptr) - 1
        if i >= 0:
            a = self.allocs.peekitem(i)[1]
            if ptr < a.base + a.nbytes:
                return a
        return None

    def _report(self, kind, new, old, rng):
        key = (kind, new[0], new[2], old[0], old[2])
        ent = self.reports.get(key)
        if ent is None:
            ent = self.reports[key] = {"kind": kind, "count": 0, "new": new, "old": old, "range": rng, "launch": self.launches}
            self.order.append(key)
        ent["count"] += 1

    def access(self, v, s, start, nbytes, write, what):
        """One pointer argument of a launch on stream s whose (already ticked) clock vector is v."""
        if nbytes <= 0:
            return
        end = start + nbytes
        c = v[s]
        for rec in self._overlapping(start, end):
            w = rec.w
            if w is not None and w[0] != s and v.get(w[0], 0) < w[1]:
                self._report("write-after-write" if write else "read-after-write", (s, c, what), (w[0], w[1], w[3]), (start, end))
            if write:
                for rs, (rc, rwhat) in rec.reads.items():
                    if rs != s and v.get(rs, 0) < rc:
                        self._report("write-after-read", (s, c, what), (rs, rc, rwhat), (start, end))
        rec = self._rec(start, end, self._alloc_of(start))
        if write:
            rec.w = (s, c, dict(v), what)
            rec.reads = {}
        else:
            rec.reads[s] = (c, what)

    # ---- allocator
    def new_storage(self, base, nbytes, s, weak=None):
        """[base, base + nbytes) is a storage the detector has not seen before: whatever used that range earlier must be ordered before
        the new owner's stream s (or have been announced to the allocator)."""
        end = base + nbytes
        v = self.clock(s)
        for rec in self._overlapping(base, end):
            oa = rec.alloc
            acc = ([(rec.w[0], rec.w[1], rec.w[3])] if rec.w is not None else []) + [(rs, rc, rw) for rs, (rc, rw) in rec.reads.items()]
            for (t, c, what) in acc:
                if t == s or v.get(t, 0) >= c:
                    continue
                if oa is not None and t in oa.recorded:
                    continue
                self._report("reuse", (s, v.get(s, 0), "new storage of %d bytes" % nbytes), (t, c, what), (rec.start, rec.end))
        # forget the range
        for k in list(self.small.irange(base - self.maxlen, end, inclusive=(True, False))):
            d = self.small[k]
            for e in [e for e, rec in d.items() if rec.end > base and rec.start < end]:
                del d[e]
            if not d:
                del self.small[k]
        self.big = [rec for rec in self.big if not (rec.start < end and rec.end > base)]
        i = max(self.allocs.bisect_right(base) - 1, 0)
        dead = []
        for k in self.allocs.islice(i):
            if k >= end:
                break
            if k + self.allocs[k].nbytes > base:
                dead.append(k)
        for k in dead:
            del self.allocs[k]
        self.gen += 1
        a = self.allocs[base] = Alloc(base, nbytes, s, weak, self.gen)
        return a

    def record_stream(self, ptr, s):
        a = self._alloc_of(ptr)
        if a is not None:
            a.recorded.add(s)

    # ---- output
    def name(self, s):
        return self.names.get(s, hex(s) if isinstance(s, int) else str(s))

    def summary(self):
        lines = ["racecheck: %d launches, %d streams, %d distinct reports" % (self.launches, len(self.vc), len(self.reports))]
        for key in self.order:
            r = self.reports[key]
            lines.append("  [%s] x%d  %s @%s  vs earlier  %s @%s  (first at launch %d, range %#x+%d)" % (
                r["kind"], r["count"], r["new"][2], self.name(r["new"][0]), r["old"][2], self.name(r["old"][0]), r["launch"],
                r["range"][0], r["range"][1] - r["range"][0]))
        return "\n".join(lines)


# ----------------------------------------------------------------------------------------------- include/sscg.h -> read / write table
def parse_header(path):
    """{entry point: [(arg name, kind)]} with kind in "r" (const pointer), "w" (mutable pointer), "stream", "desc", "-" (scalar)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    table = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(sscg_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2)
        row = []
        for a in [x.strip() for x in args.split(",")] if args.strip() not in ("", "void") else []:
            arg = re.split(r"[\s\*]+", a)[-1]
            if "*" not in a:
                kind = "-"
            elif arg == "stream":
                kind = "stream"
            elif "sscg_conv_desc" in a:
                kind = "desc"
            elif a.startswith("const"):
                kind = "r"
            else:
                kind = "w"
            row.append((arg, kind))
        table[name] = row
    return table


# ----------------------------------------------------------------------------------------------- torch layer
CORE = None
_STATE = {"installed": False, "extent": {}, "table": None, "fn_depth": 0}


def _where(skip=2, depth=5):
    f = sys._getframe(skip)
    out = []
    while f is not None and len(out) < depth:
        co = f.f_code
        fn = os.path.basename(co.co_filename)
        if fn not in ("racecheck.py", "_lib.py", "function.py", "grad_mode.py", "module.py"):
            out.append("%s:%d" % (co.co_name, f.f_lineno))
        f = f.f_back
    return "<".join(out)


def _cur_stream():
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _span_bytes(t):
    n = 1
    for sz, st in zip(t.shape, t.stride()):
        if sz == 0:
            return 0
        n += (sz - 1) * abs(st)
    return n * t.element_size()


class _Checked(object):
    """Library handle that logs every launch into CORE before it runs."""

    def __init__(self, lib):
        self._lib = lib
        self._cache = {}
        self._batches = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            fn = self._cache[name] = self._wrap(name, getattr(self._lib, name))
        return fn

    def note_batch(self, table_ptr, pairs):
        """functional._transpose_batch: the (source, destination) pointers behind a job table in device memory."""
        self._batches[table_ptr] = pairs

    def _wrap(self, name, fn):
        row = _STATE["table"].get(name)
        if row is None or not any(k == "stream" for _, k in row):
            return fn           # size queries and the like: no launch

        def call(*a):
            core = CORE
            if core is not None:
                s = a[-1] or 0
                v = core.tick(s)
                where = None
                ext = _STATE["extent"]
                if name == "sscg_weight_krsc_to_crsk_batch":
                    where = "%s(%s)" % (name, _where())
                    for src, dst in self._batches.get(a[0], ()):
                        core.access(v, s, src, ext.get(src, 4), False, where + ".w")
                        core.access(v, s, dst, ext.get(dst, 4), True, where + ".wt")
                for (arg, kind), val in zip(row, a):
                    if kind not in ("r", "w") or val is None:
                        continue
                    if name == "sscg_weighted_sum":
                        if arg == "terms":
                            for p in val:
                                if where is None:
                                    where = "%s(%s)" % (name, _where())
                                core.access(v, s, int(p), 4, False, where + ".terms")
                        if arg in ("terms", "w"):
                            continue
                    if not isinstance(val, int):
                        continue
                    if where is None:
                        where = "%s(%s)" % (name, _where())
                    n = ext.get(val)
                    if n is None:
                        al = core._alloc_of(val)
                        n = (al.base + al.nbytes - val) if al is not None else 4
                    core.access(v, s, val, n, kind == "w", where + "." + arg)
            return fn(*a)
        return call


# ----------------------------------------------------------------------------------------------- schedule fuzzer
FUZZ = {"on": False, "rng": None, "prob": 0.02, "max_cycles": 6000000, "busy": None, "busy_prob": 0.005, "sleeps": 0}


def fuzz(seed=None, prob=0.02, max_cycles=6000000, busy=False):
    """Switch the schedule fuzzer on (seed) or off (None).  While on, a spin kernel of random length (torch.cuda._sleep, up to
    `max_cycles` shader clocks: 6e6 = 2.5 ms) is queued in front of a random `prob` of the launches, on the launch's own stream: the streams drift against
    each other by milliseconds, in a different pattern per seed - any ordering the schedule only gets from timing breaks, and
    the step's results stop being bitwise equal to the serial schedule's.  `busy`: also keep a further stream occupied (what
    RCCL's stream or a user's copy stream does to the hardware queues)."""
    import random
    import torch
    FUZZ["on"] = seed is not None
    FUZZ["rng"] = random.Random(seed)
    FUZZ["prob"], FUZZ["max_cycles"] = prob, max_cycles
    if busy and FUZZ["busy"] is None:
        FUZZ["busy"] = torch.cuda.Stream()
    if not busy:
        FUZZ["busy"] = None


class _Fuzzed(object):
    """Library handle that perturbs the streams' relative timing (see fuzz())."""

    def __init__(self, lib):
        self._lib = lib
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real = getattr(self._lib, name)
            if name.endswith(("_workspace", "_bytes", "_applies", "_version")):
                fn = real
            else:
                def fn(*a, _real=real):
                    if FUZZ["on"]:
                        import torch
                        r = FUZZ["rng"]
                        if r.random() < FUZZ["prob"]:
                            torch.cuda._sleep(int(r.random() * FUZZ["max_cycles"]) + 1)     # on the current stream = the launch's
                            FUZZ["sleeps"] += 1
                        if FUZZ["busy"] is not None and r.random() < FUZZ["busy_prob"]:
                            with torch.cuda.stream(FUZZ["busy"]):
                                torch.cuda._sleep(int(r.random() * FUZZ["max_cycles"]) + 1)
                    return _real(*a)
            self._cache[name] = fn
        return fn


def install(header=None):
    """Patch torch's stream / event / tensor entry points and return the core.  Idempotent."""
    global CORE
    if _STATE["installed"]:
        return CORE
    import torch
    from torch.multiprocessing.reductions import StorageWeakRef
    here = os.path.dirname(os.path.abspath(__file__))
    _STATE["table"] = parse_header(header or os.path.join(os.path.dirname(here), "include", "sscg.h"))
    core = CORE = RaceCore()
    base_data_ptr = torch._C.TensorBase.data_ptr
    ext = _STATE["extent"]

    def data_ptr(self):
        p = base_data_ptr(self)
        if p and self.is_cuda:
            try:
                st = self.untyped_storage()
                sb, sn, cd = st.data_ptr(), st.nbytes(), st._cdata
            except Exception:
                return p
            a = core.allocs.get(sb)
            if a is None or a.weak is None or a.weak.cdata != cd:
                core.new_storage(sb, sn, _cur_stream(), StorageWeakRef(st))
            ext[p] = _span_bytes(self)
        return p
    torch.Tensor.data_ptr = data_ptr

    base_record_stream = torch._C.TensorBase.record_stream

    def record_stream(self, stream):
        if self.is_cuda:
            data_ptr(self)
            core.record_stream(base_data_ptr(self), stream.cuda_stream)
        return base_record_stream(self, stream)
    torch.Tensor.record_stream = record_stream

    S, E = torch.cuda.Stream, torch.cuda.Event
    o_wait_stream, o_wait_event, o_record_event, o_ssync = S.wait_stream, S.wait_event, S.record_event, S.synchronize
    o_erecord, o_ewait, o_esync = E.record, E.wait, E.synchronize

    def wait_stream(self, other):
        core.wait_stream(self.cuda_stream, other.cuda_stream)
        return o_wait_stream(self, other)

    def wait_event(self, event):
        core.wait_event(self.cuda_stream, event)
        return o_wait_event(self, event)

    def record_event(self, event=None):
        ev = o_record_event(self, event)
        core.record_event(ev, self.cuda_stream)
        return ev

    def ssync(self):
        r = o_ssync(self)
        core.host_sync(self.cuda_stream)
        return r

    def erecord(self, stream=None):
        core.record_event(self, stream.cuda_stream if stream is not None else _cur_stream())
        return o_erecord(self, stream) if stream is not None else o_erecord(self)

    def ewait(self, stream=None):
        core.wait_event(stream.cuda_stream if stream is not None else _cur_stream(), self)
        return o_ewait(self, stream) if stream is not None else o_ewait(self)

    def esync(self):
        r = o_esync(self)
        core.host_sync_event(self)
        return r
    S.wait_stream, S.wait_event, S.record_event, S.synchronize = wait_stream, wait_event, record_event, ssync
    E.record, E.wait, E.synchronize = erecord, ewait, esync

    o_sync = torch.cuda.synchronize

    def synchronize(device=None):
        r = o_sync(device)
        core.host_sync(None)
        return r
    torch.cuda.synchronize = synchronize

    # RCCL / gloo: the collective runs on the backend's own stream, ordered behind the caller's current stream; Work.wait() orders
    # the caller's stream behind it
    try:
        import torch.distributed as dist
        o_all_reduce = dist.all_reduce

        class _Work(object):
            def __init__(self, w):
                self._w = w

            def wait(self, *a, **k):
                r = self._w.wait(*a, **k) if self._w is not None else True
                core.wait_stream(_cur_stream(), "rccl")
                return r

            def __getattr__(self, n):
                return getattr(self._w, n)

        def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
            if tensor.is_cuda:
                core.wait_stream("rccl", _cur_stream())
                v = core.tick("rccl")
                p = data_ptr(tensor)
                core.access(v, "rccl", p, _span_bytes(tensor), True, "all_reduce(%s)" % _where())
            w = o_all_reduce(tensor, op=op, group=group, async_op=async_op)
            if not async_op:
                if tensor.is_cuda:
                    core.wait_stream(_cur_stream(), "rccl")
                return w
            return _Work(w)
        dist.all_reduce = all_reduce
        core.names["rccl"] = "rccl"
    except Exception:
        pass
    _STATE["installed"] = True
    return core


def wrap_functions(module):
    """Model the autograd engine's stream hand-over: before a node's backward runs on its stream, that stream is ordered behind the
    producer of each incoming gradient."""
    import torch

    def wrap(orig):
        def backward(ctx, *grads):
            core = CORE
            if core is not None:
                s = _cur_stream()
                for g in grads:
                    if isinstance(g, torch.Tensor) and g.is_cuda:
                        core.merge_from_writer(s, torch._C.TensorBase.data_ptr(g))
            return orig(ctx, *grads)
        return backward
    for obj in list(vars(module).values()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function \
                and "backward" in vars(obj) and not getattr(obj, "_sscg_racewrapped", False):
            obj.backward = staticmethod(wrap(obj.backward))
            obj._sscg_racewrapped = True


def name_streams(F, device):
    """Readable stream names in the report."""
    import torch
    core = CORE
    if core is None:
        return
    core.names[torch.cuda.default_stream(device).cuda_stream] = "main"
    for (dev, lane), s in F.SideStream._streams.items():
        core.names[s.cuda_stream] = "side%d" % lane
    for (dev, lane), s in F.ForkStream._streams.items():
        core.names[s.cuda_stream] = "fork%d" % lane


def report(file=None):
    if CORE is not None:
        print(CORE.summary(), file=file or sys.stderr)
    return CORE

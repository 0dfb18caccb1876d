"""Host-side operators: torch tensors in, libsscg.so kernels underneath.

Two layers:
  * raw ops (`conv2d_fwd`, `norm_stats`, ...) - thin wrappers over the C ABI (include/sscg.h) that
    allocate outputs through torch's caching allocator and launch on torch's current HIP stream;
  * `torch.autograd.Function`s (`Conv2dFn`, `NormActFn`, ...) that give those kernels the semantics
    of the stock torch.nn modules the reference composes (arch/ops.py:40-57, model.py:268-273).

Tensors keep the reference's logical NCHW shapes; physically they are channels-last (NHWC), which is
what every kernel assumes.  There is no fallback: a non-HIP tensor raises.
"""
import ctypes as C

import os
import weakref

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZEROS, PAD_REFLECT, ConvDesc, check, lib

CL = torch.channels_last


# ----------------------------------------------------------------------------- plumbing
_raw_stream = torch._C._cuda_getCurrentRawStream     # the handle only: torch.cuda.current_stream() builds a Stream object (~6 us)
_cur_dev = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)     # the binding itself: torch.cuda.current_device() adds a lazy-init check per call


def _stream():
    return _raw_stream(_cur_dev())


def _ptr(t):
    return None if t is None else t.data_ptr()


F32, BF16, BF16X3 = _lib.F32, _lib.BF16, _lib.BF16X3
_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _need_hip(t, f32_only=False):
    if not t.is_cuda:
        raise _lib.SscgError("sscg kernels run on the MI355X only: got a %s tensor (no CPU fallback)" % t.device)
    if t.dtype != torch.float32 and (f32_only or t.dtype != torch.bfloat16):
        raise _lib.SscgError("%s tensor expected, got %s" % ("fp32" if f32_only else "fp32 or bf16", t.dtype))


def _dt(t):
    return _DT[t.dtype]


def _same_dtype(*ts):
    d = ts[0].dtype
    for t in ts[1:]:
        if t is not None and t.dtype != d:
            raise _lib.SscgError("operands of one element type expected, got %s and %s" % (d, t.dtype))
    return _DT[d]


class _Workspace:
    """One growable scratch buffer per (device, stream): reuse is ordered by the stream it is used on."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 16)
        key = (device, _raw_stream(device.index if device.index is not None else _cur_dev()))
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


_WS = _Workspace()

_HIPRT = []


def _hip_runtime():
    """The HIP runtime torch itself loaded (a second copy's stream handles would be foreign to torch): its path is taken from this
    process's own mappings; None when it cannot be found or lacks the symbol."""
    import ctypes
    if not _HIPRT:
        h = None
        try:
            path = None
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64.so" in line:
                        path = line.split(None, 5)[-1].strip()
                        break
            if path is not None:
                h = ctypes.CDLL(path)       # already mapped: dlopen returns the loaded library
                h.hipStreamCreateWithPriority.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.c_int]
                h.hipStreamCreateWithPriority.restype = ctypes.c_int
        except (OSError, AttributeError):
            h = None
        _HIPRT.append(h)
    return _HIPRT[0]


def _make_stream(device, priority=0):
    """A HIP stream of the given priority (HIP's scale: -1 high, 0 normal, 1 low).  torch only creates normal / high priority
    streams; a LOW priority one comes from hipStreamCreateWithPriority and is wrapped (it lives as long as the process).  Where
    that is not possible (runtime not found, call refused) the lane is a normal-priority torch stream: slower by ~1 %, never wrong."""
    import ctypes
    if priority <= 0:
        return torch.cuda.Stream(device=device, priority=priority)
    rt = _hip_runtime()
    if rt is None:
        return torch.cuda.Stream(device=device, priority=0)
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = rt.hipStreamCreateWithPriority(ctypes.byref(h), 1, priority)   # 1 = hipStreamNonBlocking
    if rc != 0 or not h.value:
        return torch.cuda.Stream(device=device, priority=0)
    return torch.cuda.ExternalStream(h.value, device=device)


FORK_PRIORITY = int(os.environ.get("SSCG_FORK_PRIORITY", "0"))


def side_priority():
    """Priority of the side lanes (weight gradients, frozen generators, D step): LOW on a single GPU - the dispatcher then prefers
    the critical path's workgroups whenever both are ready (-1.3 ms per config-2 step, same-box A/B) - but NORMAL once a process
    group is up: with RCCL's stream in flight the low-priority lanes cost +30 % (176-190 vs 145 ms, `SSCG_FORCE_DP=1`, measured)."""
    env = os.environ.get("SSCG_SIDE_PRIORITY")
    if env is not None:
        return int(env)
    import torch.distributed as dist
    return 0 if (dist.is_available() and dist.is_initialized()) else 1


def side_priority_for(pixels_per_batch):
    """... and only where the lanes compete for CUs: below ~128 K pixels per batch the kernels do not fill the chip, a low-priority
    lane is merely late at the next join (64x64, batch 2: 48-51 -> 55-61 ms per step)."""
    p = side_priority()
    return p if (p <= 0 or os.environ.get("SSCG_SIDE_PRIORITY") is not None or pixels_per_batch >= 131072) else 0


def set_side_priority(prio):
    """Choose the side lanes' priority - effective only while the process has no lanes yet.  ONE set of lanes per process: a second
    set (a small model at normal priority, then a large one at low priority) oversubscribes the hardware queues - the step ran 172
    instead of 142 ms and produced non-finite losses on this runtime (`tools/nan_probe.sh`, profiles/r03_experiments.txt item 12).
    The first model that steps decides; returns the priority in force."""
    if not SideStream._streams:
        SideStream.priority = prio
    return SideStream.priority


def reset_side_streams():
    """Drop the cached side lanes (they are re-created, with the priority that holds now, on their next use)."""
    if SideStream._streams:
        flush_side_work()
        torch.cuda.synchronize()
        SideStream._streams.clear()
        _STREAM_BY_HANDLE.clear()
    SideStream.priority = None


class SideStream:
    """A second HIP stream per device for work that is independent of the main stream's critical path:
    weight gradients (needed only by the optimiser step) and the frozen generators' forward (needed only by
    the discriminator step).  Two kernels in flight let the dispatcher fill the CUs that one launch with an
    awkward workgroup count (8712-row DeepLab maps: 548 workgroups on 256 CUs) leaves idle."""

    enabled = True
    _streams = {}

    # A/B aid only: with a third lane (or a second lane set) streams share hardware queues, and the overlapped D step then produces
    # non-finite losses (open issue, profiles/r03_experiments.txt items 12-13)
    lanes = int(os.environ.get("SSCG_SIDE_LANES", "2"))   # measured: 4 streams in flight (main, fork lane, 2 side lanes) is the sweet spot; a 5th costs 20 %

    priority = None     # fixed when the first lane is made: set_side_priority() of the first model that steps, else side_priority()

    @classmethod
    def get(cls, device, lane=0):
        s = cls._streams.get((device, lane))
        if s is None:
            if cls.priority is None:
                cls.priority = side_priority()
            s = _make_stream(device, cls.priority)
            cls._streams[(device, lane)] = s
        return s

    @classmethod
    def join(cls, device=None):
        """Make the current stream wait for everything queued on the side stream(s) (deferred work is launched first)."""
        flush_side_work(device)
        for (dev, _), s in cls._streams.items():
            if device is None or dev == device:
                torch.cuda.current_stream(dev).wait_stream(s)


class ForkStream:
    """A third HIP stream: the second lane of the forked generator forwards (model.semisuper_cycleGAN.step).  The
    step's five trainable DeepLab passes form two independent chains, Gis(onehot) -> Gsi(fake_img) and
    Gsi(unl) -> Gis(fake_gt), plus Gsi(l_img); autograd replays each op's backward on the stream of its forward, so
    forking the forward also runs the two backward chains side by side."""

    _streams = {}

    @classmethod
    def get(cls, device, lane=0):
        s = cls._streams.get((device, lane))
        if s is None:
            s = _make_stream(device, FORK_PRIORITY)
            cls._streams[(device, lane)] = s
        return s

    @classmethod
    def join(cls, device):
        for (dev, _), s in cls._streams.items():
            if dev == device:
                torch.cuda.current_stream(device).wait_stream(s)


def lane_events(device):
    """An event on every lane that may carry gradient kernels of the current backward pass - the calling stream, the fork lane, the
    side lanes (deferred side-lane work is launched first): a stream that waits for all of them sees every gradient kernel queued
    so far (parallel.GradBuckets orders a bucket's all-reduce behind them)."""
    flush_side_work(device)
    streams = [torch.cuda.current_stream(device)]
    for reg in (ForkStream._streams, SideStream._streams):
        for (dev, _), st in reg.items():
            if dev == device:
                streams.append(st)
    evs = []
    for st in streams:
        ev = torch.cuda.Event()
        ev.record(st)
        evs.append(ev)
    return evs


# Data parallelism with bucketed gradient exchange (parallel.GradBuckets): GRAD_READY[0] is called with a parameter once the LAST
# gradient kernel of this backward pass that accumulates into its arena slice has been queued.  "Last" is known by counting: every
# forward of a conv whose weight will receive a gradient notes one use (_note_use), every backward of one takes it back (_note_done).
GRAD_READY = [None]
# grad mode of the CALLER of the autograd.Function being applied (inside forward() it is always off): ctx.needs_input_grad is True
# for a trainable weight under torch.no_grad() as well, but such a forward never sees a backward - it must neither count a use nor keep
# pre-norm activations alive for one (the unused Gis(lab_gt) pass of a step, evaluation, validation)
_CALLER_GRAD = [True]


def _will_backward(ctx, i=None):
    need = any(ctx.needs_input_grad) if i is None else ctx.needs_input_grad[i]
    return bool(need) and _CALLER_GRAD[0]


def _note_use(*params):
    if GRAD_READY[0] is not None:
        for p in params:
            if p is not None and getattr(p, "_sscg_grad", None) is not None:
                p._sscg_uses = getattr(p, "_sscg_uses", 0) + 1


def _note_done(*params):
    hook = GRAD_READY[0]
    if hook is not None:
        for p in params:
            if p is not None and getattr(p, "_sscg_grad", None) is not None:
                p._sscg_uses = getattr(p, "_sscg_uses", 0) - 1
                if p._sscg_uses == 0:
                    hook(p)


def d_stream(device):
    """Stream of the overlapped discriminator step: the LAST side lane - idle between the end of a backward pass and
    the next one, which is when the D step runs.  Not a stream of its own: a fifth stream in flight costs ~20 % (the
    hardware runs four queues side by side; more are time-sliced)."""
    return SideStream.get(device, SideStream.lanes - 1)


def debug_wait_for_d(device, names):
    """Bisection aid (SSCG_DBG_TOPWAIT=main,fork,side0,...): order the named streams behind the D stream."""
    d = d_stream(device)
    for n in names:
        if n == "main":
            s = torch.cuda.current_stream(device)
        elif n == "fork":
            s = ForkStream.get(device)
        else:
            s = SideStream.get(device, int(n[4:]))
        if s.cuda_stream != d.cuda_stream:
            s.wait_stream(d)


def run_on_side_stream(device, tensors, fn, lane=0, defer=False):
    """Launch `fn`'s kernels on the side stream after everything queued so far on the current stream.
    `tensors` are the buffers fn reads: the allocator must not recycle them before the side stream is done.

    defer=True (weight / bias gradient kernels: nothing reads their result before SideStream.join): the call is queued per lane
    and launched in batches of SIDE_BATCH - one stream switch (current_stream + wait_stream + stream guard + record_stream:
    ~20 us of host time) per batch instead of per convolution; 440 switches per step were the largest single item of the
    backward's issue cost.  A parameter's gradient kernels keep their order (FIFO on their own lane)."""
    if not SideStream.enabled:
        return fn()
    lane %= SideStream.lanes
    if defer and SIDE_BATCH[0] > 1:
        q = _SIDE_PENDING.setdefault((device, lane), [])
        q.append((fn, tensors, _stream()))
        if len(q) >= SIDE_BATCH[0]:
            _flush_lane(device, lane)
        return None
    main = torch.cuda.current_stream(device)
    side = SideStream.get(device, lane)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r = fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    return r


# (8 until the end of round 6; re-swept on the final kernels, profiles/r06_experiments.txt item 27: 4 / 8 / 16 alike, 32 -0.45 and
# -0.8 ms per config-2 step on two boxes, 64 +0.4; config 3 indifferent)
SIDE_BATCH = [int(os.environ.get("SSCG_SIDE_BATCH", "32"))]
_SIDE_PENDING = {}      # (device, lane) -> [(fn, tensors, raw handle of the stream that produced the tensors)]
_STREAM_BY_HANDLE = {}  # raw stream handle -> torch.cuda.Stream (built once per stream: the object costs ~6 us to make)


def _stream_object(device, handle):
    so = _STREAM_BY_HANDLE.get((device, handle))
    if so is None:
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream == handle:
            so = cur
        else:       # a producer stream that is not current any more: every stream this package creates is in the two registries
            for reg in (SideStream._streams, ForkStream._streams):
                for key, st in reg.items():
                    if key[0] == device and st.cuda_stream == handle:
                        so = st
            if so is None:
                so = torch.cuda.ExternalStream(handle, device=device)
        _STREAM_BY_HANDLE[(device, handle)] = so
    return so


def _flush_lane(device, lane):
    q = _SIDE_PENDING.get((device, lane))
    if not q:
        return
    _SIDE_PENDING[(device, lane)] = []
    side = SideStream.get(device, lane)
    for h in {h for _, _, h in q}:
        if h != side.cuda_stream:
            side.wait_stream(_stream_object(device, h))      # behind everything its producers have queued so far
    with torch.cuda.stream(side):
        for fn, _, _ in q:
            fn()
    for _, tensors, _ in q:
        for t in tensors:
            if t is not None:
                t.record_stream(side)


def flush_side_work(device=None):
    """Launch every deferred side-lane call (SideStream.join does this first)."""
    for (dev, lane) in list(_SIDE_PENDING.keys()):
        if device is None or dev == device:
            _flush_lane(dev, lane)


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=CL)


def to_nhwc(x):
    """Logical NCHW tensor -> channels-last memory (HIP transpose kernel when a copy is needed)."""
    _need_hip(x)
    if x.dim() != 4:
        raise _lib.SscgError("4-D tensor expected")
    if x.is_contiguous(memory_format=CL):
        return x
    _need_hip(x, f32_only=True)      # bf16 tensors only exist inside the networks, where every kernel writes channels-last
    if not x.is_contiguous():
        x = x.contiguous()
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h, w, x.device)
    check(lib.sscg_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), n, c, h, w, _stream()), "sscg_nchw_to_nhwc")
    return y


def to_nchw(x):
    """channels-last tensor -> standard contiguous NCHW memory."""
    if x.dtype == torch.bfloat16:
        x = cast(x, torch.float32)
    _need_hip(x)
    if x.is_contiguous():
        return x
    x = to_nhwc(x)
    n, c, h, w = x.shape
    y = torch.empty((n, c, h, w), dtype=x.dtype, device=x.device)
    check(lib.sscg_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), n, c, h, w, _stream()), "sscg_nhwc_to_nchw")
    return y


# What `--dtype f32` computes its contractions with: False = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), True = the fp32-accurate split
# contraction on the bf16 matrix cores (conv_split.hip) wherever the library offers it.
F32_SPLIT = [os.environ.get("SSCG_F32_SPLIT", "1") == "1"]
_MODE = ["f32s" if F32_SPLIT[0] else "f32"]
FUSE_STATS = [os.environ.get("SSCG_FUSE_STATS", "1") != "0"]    # norm statistics from the producing conv's epilogue (K3/K4)
# The reduction pass of a norm layer's backward from the epilogue of the data gradient that produces its upstream gradient
# (sscg_conv2d_dgrad_bsums): ~260 col_reduce launches fewer per step.  fp32 tensors: ON - the split kernel takes
# the sums in its coalesced store phase (config 2 134.4 -> 134.0 ms, host issue time -2 ms per step; the first version, sums in the
# MFMA layout, LOST 0.7 ms).  bf16 tensors: ON since round 6 - rounds 4-5 took them in conv16_kernel's MFMA layout (per-element
# 2-byte gathers of the layer's input: config 3 +1.9 ... +3 ms) and a first store-phase version lost too (64 -> 111 VGPRs); the round-6
# store phase (16-byte row segments requested with the addend's, fp32 over a thread's rows, shuffles across a wave's row lanes,
# fp64 across the waves, one record per tile; its own instances, 128 VGPRs without the fan-in code) wins 1.05 ms at config 3
# (profiles/r06_experiments.txt item 17).  SSCG_FUSE_BSUMS=1 / 0 forces both on / off, "f32" = fp32 tensors only.
_FB = os.environ.get("SSCG_FUSE_BSUMS", "1")
FUSE_BSUMS = [_FB != "0"]            # master switch (tests flip it)
FUSE_BSUMS_BF16 = [_FB == "1"]


def set_conv_precision(mode):
    """Arithmetic of the networks (host-side policy; the library itself keeps no mode - every call carries its dtypes):
      "f32"   (default, the reference's dtype, BASELINE config 2): fp32 tensors.  Which contraction that means is the build-wide
              policy F32_SPLIT (env SSCG_F32_SPLIT, default on): "f32s" below, or "f32x";
      "f32s"  fp32 tensors; forward / data-gradient contractions of every conv with >= 32 source channels (a multiple of 32) run as
              the fp32-accurate split contraction on the BF16 matrix cores (conv_split.hip: each operand as three bfloat16 pieces,
              six exact piece products accumulated in fp32; the weights' pieces are kept by the Adam kernel / the operand-copy
              cache, the activations are split between LDS and the matrix cores); weight gradients, few-channel stems and heads
              run the exact fp32 MFMA kernels;
      "f32x"  fp32 tensors, exact fp32 MFMA contractions everywhere (v_mfma_f32_32x32x2_f32);
      "bf16"  (BASELINE configs 3/5): activations and conv-weight operand copies are bfloat16 in HBM, bf16 LDS tiles,
              v_mfma_f32_32x32x16_bf16 with fp32 accumulation; network inputs/outputs, norm statistics, losses, weight
              gradients, master weights and Adam moments stay fp32;
      "bf16c" (round-1 mode): fp32 tensors, operands rounded to bf16 between LDS and the matrix cores."""
    m = {"f32": "f32", "fp32": "f32", "float32": "f32", "bf16": "bf16", "bfloat16": "bf16", "bf16c": "bf16c", "f32s": "f32s",
         "f32x": "f32x"}.get(str(mode).lower())
    if m is None:
        raise _lib.SscgError("conv precision must be 'f32', 'f32x', 'f32s', 'bf16' or 'bf16c', got %r" % (mode,))
    if m == "f32":          # fp32 tensors: which contraction "f32" means is a build-wide policy (F32_SPLIT); "f32x" / "f32s" name one
        m = "f32s" if F32_SPLIT[0] else "f32"
    elif m == "f32x":
        m = "f32"
    if m != _MODE[0]:
        # which operand copies (transposed fp32 / bf16 / split planes) a weight needs depends on the mode: forget the old mode's
        # registrations, or every later step keeps rebuilding copies nothing reads
        for _, kinds in globals().get("_WT_USERS", {}).values():
            kinds.clear()
    _MODE[0] = m


def get_conv_precision():
    return _MODE[0]


if os.environ.get("SSCG_CONV_PRECISION"):          # tools/ and ad-hoc runs; main.py / bench.py take --dtype
    set_conv_precision(os.environ["SSCG_CONV_PRECISION"])


def cast(x, dtype):
    """dtype conversion on the device (fp32 <-> bf16, RNE); layout (strides) preserved."""
    _need_hip(x)
    if x.dtype == dtype:
        return x
    if not (x.is_contiguous() or x.is_contiguous(memory_format=CL)):
        x = x.contiguous()
    y = torch.empty_like(x, dtype=dtype, memory_format=torch.preserve_format)
    check(lib.sscg_cast(x.data_ptr(), _dt(x), y.data_ptr(), _DT[dtype], x.numel(), _stream()), "sscg_cast")
    return y


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


_DESC_CACHE = {}


TUNING = [int(os.environ.get("SSCG_TUNING", "0"), 0)]        # sscg_conv_desc.tuning of every descriptor built from here on (tools/ and tile-class tests; 0 = the library's plan)
WGRAD_TUNING = [int(os.environ.get("SSCG_WGRAD_TUNING", "0"), 0)]  # sscg_conv_desc.wgrad_tuning, likewise (env: A/B aid)


def tuning(tile_class=None, split=0, wgrad_class=None, wgrad_splits=0, wgrad_flags=0):
    """Set the per-call tuning fields of the descriptors built from here on (None / 0 = the library's own plans).  Returns the
    previous pair: `old = F.tuning(tile_class=1); ...; F.TUNING[0], F.WGRAD_TUNING[0] = old`."""
    old = (TUNING[0], WGRAD_TUNING[0])
    TUNING[0] = (0 if tile_class is None else tile_class + 1) | (split << 8)
    WGRAD_TUNING[0] = (0 if wgrad_class is None else wgrad_class + 1) | (wgrad_splits << 8) | (wgrad_flags << 24)
    return old


def make_desc(xshape, wshape, stride, pad, dil, pad_mode=PAD_ZEROS, act=ACT_NONE, slope=0.0, xdt=F32, wdt=F32, ydt=F32, prec=0, wplane=0):
    """ConvDesc of a call, memoised (a step re-issues the same few dozen geometries thousands of times, and the host side
    of a launch is what bounds the 4-stream schedule)."""
    key = (tuple(xshape), tuple(wshape), stride, pad, dil, pad_mode, act, slope, xdt, wdt, ydt, prec, TUNING[0], WGRAD_TUNING[0], wplane)
    d = _DESC_CACHE.get(key)
    if d is None:
        d = _DESC_CACHE[key] = _build_desc(xshape, wshape, stride, pad, dil, pad_mode, act, slope)
        d.x_dtype, d.w_dtype, d.y_dtype, d.precision = xdt, wdt, ydt, prec
        d.tuning, d.w_plane, d.wgrad_tuning = TUNING[0], wplane, WGRAD_TUNING[0]
    return d


_SPLIT_OK = {}


def split_applies(xshape, wshape, stride, pad, dil, pad_mode, kind):
    """Does the library's split contraction (fp32 accuracy on the bf16 matrix cores) serve this product?  kind: 0 forward,
    1 data gradient, 2 weight gradient.  The library decides (sscg_conv2d_split_applies); the answer is cached per geometry."""
    key = (tuple(xshape), tuple(wshape), stride, pad, dil, pad_mode, kind)
    v = _SPLIT_OK.get(key)
    if v is None:
        d = _build_desc(xshape, wshape, stride, pad, dil, pad_mode, ACT_NONE, 0.0)
        v = _SPLIT_OK[key] = bool(lib.sscg_conv2d_split_applies(C.byref(d), kind))
    return v


# Products the split mode covers (bisection aid).  The weight gradient has no pre-split operand: BOTH of its operands are split
# between LDS and the matrix cores, which makes the kernel VALU-bound and no faster than the exact fp32 one when it runs alone -
# but it then needs 2.7x fewer matrix-core cycles, and in the step's four-stream schedule those go to the forward / data-gradient
# kernels running beside it: 156.6 -> 148.7 ms per step (SSCG_WGRAD_SPLIT=0 restores the exact fp32 weight gradient).
SPLIT_KINDS = {"fwd", "dgrad"} | ({"wgrad"} if os.environ.get("SSCG_WGRAD_SPLIT", "1") == "1" else set())


def _prec(kind="fwd"):
    """`precision` field for fp32-tensor contractions: 1 = bf16 rounding between LDS and the matrix cores (both bf16 modes); the
    split mode selects its kernels through the weight operand's dtype (SSCG_BF16X3), not through this field."""
    if _MODE[0] == "f32s" and kind == "wgrad" and "wgrad" in SPLIT_KINDS:
        return 2
    return 0 if _MODE[0] in ("f32", "f32s") else 1


_PLAN_SIZES = {}


def _ws_bytes(d, which):
    """Workspace size of a conv entry point.  The split plans depend on the geometry and on the descriptor's tuning fields: cached
    per (descriptor object - one per geometry / dtype / tuning, make_desc - , entry point)."""
    key = (id(d), which)
    v = _PLAN_SIZES.get(key)
    if v is None:
        v = _PLAN_SIZES[key] = getattr(lib, "sscg_conv2d_%s_workspace" % which)(C.byref(d))
    return v


def _stats_bytes(d, g, l):
    key = (id(d), "stats", g, l)
    v = _PLAN_SIZES.get(key)
    if v is None:
        v = _PLAN_SIZES[key] = lib.sscg_conv2d_fwd_stats_bytes(C.byref(d), g, l)
    return v


def _build_desc(xshape, wshape, stride, pad, dil, pad_mode, act, slope):
    n, c, h, w = xshape
    k, c2, r, s = wshape
    if c != c2:
        raise _lib.SscgError("conv channel mismatch: input %d vs weight %d" % (c, c2))
    d = ConvDesc()
    d.N, d.H, d.W, d.C = n, h, w, c
    d.K, d.R, d.S = k, r, s
    d.P, d.Q = conv_out_size(h, r, stride, pad, dil), conv_out_size(w, s, stride, pad, dil)
    d.stride, d.pad, d.dil = stride, pad, dil
    d.pad_mode, d.act, d.slope = pad_mode, act, slope
    return d


# ----------------------------------------------------------------------------- per-kernel timing hook (bench.py)
class ConvProfile:
    """Collects (kind, algorithmic flops, HIP-event pair) for every conv launch while active.  Events are
    recorded on torch's current stream - the stream the kernels are launched on."""

    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        ConvProfile.active = self
        return self

    def __exit__(self, *a):
        ConvProfile.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, flops, key, e0, e1, nbytes in self.records:      # kind = (fwd|dgrad|wgrad, kernel family f32|bf16)
            s = out.setdefault(kind, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0, "shapes": {}})
            ms = e0.elapsed_time(e1)
            s["launches"] += 1
            s["flops"] += flops
            s["bytes"] += nbytes
            s["ms"] += ms
            sh = s["shapes"].setdefault(key, [0, 0.0, 0.0])
            sh[0] += 1
            sh[1] += flops
            sh[2] += ms
        return out


def _unprofiled(fn):
    prof, ConvProfile.active = ConvProfile.active, None
    try:
        return fn()
    finally:
        ConvProfile.active = prof


def _timed(kind, d, fn, extra_bytes=0):
    """extra_bytes: algorithmic reads of a launch's fused store phase beyond operands and result (the fan-in addend, the normalisation
    layer's input and mask source of a data gradient that takes the backward sums)."""
    prof = ConvProfile.active
    if prof is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    flops = 2.0 * d.N * d.P * d.Q * d.K * d.R * d.S * d.C
    key = "%dx%dx%d c%d k%d r%d s%d p%d d%d" % (d.N, d.H, d.W, d.C, d.K, d.R, d.stride, d.pad, d.dil)
    # which kernel family the library dispatches to (include/sscg.h, "Supported dtype combinations")
    if kind == "fwd":
        b16 = d.x_dtype == BF16 and d.w_dtype == BF16
    elif kind == "dgrad":
        b16 = d.y_dtype == BF16 and d.w_dtype == BF16
    else:
        b16 = d.x_dtype == BF16 and d.y_dtype == BF16 and d.K >= 32 and d.R * d.S * d.C >= 32
    # algorithmic HBM bytes of the launch: every operand read once, the result written once
    esz = {F32: 4, BF16: 2, BF16X3: 6}
    nbytes = (d.N * d.H * d.W * d.C * esz[d.x_dtype] + d.N * d.P * d.Q * d.K * esz[d.y_dtype]
              + d.K * d.R * d.S * d.C * (4 if kind == "wgrad" else esz[d.w_dtype])) + extra_bytes
    prof.records.append(((kind, "split" if d.w_dtype == BF16X3 else ("bf16" if b16 else "f32")), flops, key, e0, e1, float(nbytes)))
    return r


# ----------------------------------------------------------------------------- raw ops
def _fwd_operands(x, w, geom=None):
    """(weight operand, its dtype code, plane stride) for a forward whose input is x: bf16 tiles when x is a bf16 activation
    with a multiple of 64 channels; the three-plane split copy in the split mode where the library's split kernels serve the
    geometry `geom` = (stride, pad, dil, pad_mode); else the fp32 kernel on the fp32 master weight (stems, few-channel inputs)."""
    if x.dtype == torch.bfloat16:
        if x.shape[1] % 64:
            raise _lib.SscgError("bf16 activations with %d channels: the bf16 conv kernels need a multiple of 64" % x.shape[1])
        return weight_bf16(w), BF16, 0
    if _MODE[0] == "f32s" and "fwd" in SPLIT_KINDS and geom is not None and split_applies(x.shape, w.shape, *geom, 0):
        ws, plane = weight_split(w)
        return ws, BF16X3, plane
    return w, F32, 0


def _out_dtype(out_f32):
    return torch.bfloat16 if (_MODE[0] == "bf16" and not out_f32) else torch.float32


PAD_STEMS = [os.environ.get("SSCG_PAD_STEMS", "1") != "0"]


def resize_channels(x, c_new):
    """[N, C, H, W] (channels-last) or a [K, C, R, S] weight -> the same with c_new channels: zero-padded or cut (fp32)."""
    x = x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c_new, h, w, x.device)
    check(lib.sscg_resize_channels(x.data_ptr(), y.data_ptr(), n * h * w, c, c_new, _stream()), "sscg_resize_channels")
    return y


def _padded_stem(xshape, wshape, stride, pad, dil, pad_mode, kind):
    """Split mode: a convolution over 17-31 source channels (the 21 / 20-channel stems on one-hot / softmax maps: arch/generators.py:73,
    373) runs as a 32-channel convolution on the split contraction - the extra channels are zero in the activation AND in the weight
    copy - instead of the exact-fp32 kernel's ragged-channel path (57 TFLOP/s): returns 32 when the padded geometry is served."""
    c = xshape[1]
    if not (PAD_STEMS[0] and _MODE[0] == "f32s" and 16 < c < 32 and wshape[0] >= 16):
        return 0
    kname = "fwd" if kind == 0 else "dgrad"
    if kname not in SPLIT_KINDS:
        return 0
    xs = (xshape[0], 32, xshape[2], xshape[3])
    ws = (wshape[0], 32, wshape[2], wshape[3])
    return 32 if split_applies(xs, ws, stride, pad, dil, pad_mode if kind == 0 else PAD_ZEROS, kind) else 0


def _padded_head(xshape, wshape, stride, pad, dil):
    """Split mode: the data gradient of a convolution with 17-31 OUTPUT channels (the reduction runs over taps x output channels: 21
    is no multiple of the 32-wide k-tile) runs with dy and the filters padded to 32 output channels: returns 32 when served."""
    k = wshape[0]
    if not (PAD_STEMS[0] and _MODE[0] == "f32s" and 16 < k < 32 and "dgrad" in SPLIT_KINDS):
        return 0
    return 32 if split_applies(xshape, (32,) + tuple(wshape[1:]), stride, pad, dil, PAD_ZEROS, 1) else 0


def _pad_filters(w, k_new):
    """[K, C, R, S] -> [k_new, C, R, S] with zero filters appended (a device copy of the K * R * S * C leading elements)."""
    w = w if w.is_contiguous(memory_format=CL) else w.contiguous(memory_format=CL)
    wp = empty_nhwc(k_new, w.shape[1], w.shape[2], w.shape[3], w.device)
    fill_(wp, 0.0)
    wp[:w.shape[0]].copy_(w)
    return wp


def _padded_weight(w, c_new):
    """The weight with zero-padded source channels, cached on the weight (rebuilt when the optimiser rewrites it); its own operand
    copies (split planes, transposed split planes) are cached on the padded tensor and die with it."""
    return _cached_copy(w, "_sscg_wpad", lambda: resize_channels(w.detach(), c_new))


def conv2d_fwd(x, w, bias, stride=1, pad=0, dil=1, pad_mode=PAD_ZEROS, act=ACT_NONE, slope=0.0, out_f32=True, stats=None):
    """y = act(conv(x, w) + bias).  out_f32: keep the output fp32 in bf16 mode (network heads).
    stats = (G, L): also return the epilogue's column statistics buffer (None when the fusion does not apply)."""
    if act == ACT_TANH and w.shape[0] > 32 and x.dtype == torch.bfloat16:
        # (no such layer in the reference's nets: the bf16 kernels carry tanh in their 32-column heads' class only)
        return act_fwd(conv2d_fwd(x, w, bias, stride, pad, dil, pad_mode, ACT_NONE, 0.0, out_f32), ACT_TANH, 0.0)
    if x.dtype == torch.float32:
        cp = _padded_stem(x.shape, w.shape, stride, pad, dil, pad_mode, 0)
        if cp:
            # (the profile books this launch - padding pass included - under the ORIGINAL geometry: algorithmic FLOP of the 21-channel conv)
            d0 = make_desc(x.shape, w.shape, stride, pad, dil, pad_mode, act, slope, F32, BF16X3, F32, _prec(), 0)
            return _timed("fwd", d0, lambda: _unprofiled(lambda: conv2d_fwd(resize_channels(x, cp), _padded_weight(w, cp), bias, stride, pad, dil,
                                                                             pad_mode, act, slope, out_f32, stats)))
    # (a fused tanh behind more than 32 output channels - no such layer in the reference's nets - stays on the exact kernel: the split
    # family's epilogue carries tanh in its heads' class only)
    wop, wdt, wplane = _fwd_operands(x, w, None if (act == ACT_TANH and w.shape[0] > 32) else (stride, pad, dil, pad_mode))
    ydt = _out_dtype(out_f32)
    d = make_desc(x.shape, w.shape, stride, pad, dil, pad_mode, act, slope, _dt(x), wdt, _DT[ydt], _prec(), wplane)
    y = empty_nhwc(d.N, d.K, d.P, d.Q, x.device, ydt)
    ws = _WS.get(_ws_bytes(d, "fwd"), x.device)
    if stats is not None:
        nb = _stats_bytes(d, int(stats[0]), int(stats[1]))
        if nb:
            sbuf = torch.empty(nb, dtype=torch.uint8, device=x.device)
            _timed("fwd", d, lambda: check(lib.sscg_conv2d_fwd_stats(C.byref(d), x.data_ptr(), wop.data_ptr(), _ptr(bias), y.data_ptr(),
                                                                     int(stats[0]), int(stats[1]), sbuf.data_ptr(), nb, ws.data_ptr(),
                                                                     ws.numel(), _stream()), "sscg_conv2d_fwd_stats"))
            return y, (d, sbuf)
    _timed("fwd", d, lambda: check(lib.sscg_conv2d_fwd(C.byref(d), x.data_ptr(), wop.data_ptr(), _ptr(bias), y.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), _stream()), "sscg_conv2d_fwd"))
    return (y, None) if stats is not None else y


def norm_stats_from_conv(cs, per_sample_glc, eps, running_mean=None, running_var=None, momentum=0.1):
    """mean / rstd (and the running-statistics update) from the statistics a conv epilogue left (conv2d_fwd(..., stats=))."""
    d, sbuf = cs
    g, l, c = per_sample_glc
    mean = torch.empty((g, c), dtype=torch.float32, device=sbuf.device)
    rstd = torch.empty((g, c), dtype=torch.float32, device=sbuf.device)
    check(lib.sscg_norm_stats_from_conv(C.byref(d), sbuf.data_ptr(), g, l, eps, mean.data_ptr(), rstd.data_ptr(), _ptr(running_mean),
                                        _ptr(running_var), momentum, _stream()), "sscg_norm_stats_from_conv")
    return mean, rstd


def conv2d_fwd_norm(x, w, bias, stride, pad, dil, pad_mode, out_f32, glc, eps, running_mean=None, running_var=None, momentum=0.1):
    """(y, mean, rstd): y = conv(x, w) + bias with the batch statistics of the normalisation layer that follows (view [G][L][C] of y;
    running statistics updated): the conv's epilogue takes the statistics, one small launch finalises them.  mean is None when no conv
    epilogue takes statistics for this geometry (the caller runs a statistics pass over y)."""
    g, l, c = glc
    y, cs = conv2d_fwd(x, w, bias, stride, pad, dil, pad_mode, ACT_NONE, 0.0, out_f32, stats=(g, l))
    if cs is None:
        return y, None, None
    mean, rstd = norm_stats_from_conv(cs, glc, eps, running_mean, running_var, momentum)
    return y, mean, rstd


FUSE_FRONT = [os.environ.get("SSCG_FUSE_FRONT", "1") != "0"]      # A/B aid: PixelDiscriminator's two leading convs as separate launches
FRONT_CALLS = [0]                                                  # fused front launches issued (tests assert the path taken)


def conv2d_front_applies(x, w1, w, stride, pad, dil, pad_mode):
    """Does ONE launch serve Conv2d(cin, 64, 1x1) -> LeakyReLU -> Conv2d(64, K, 1x1) (PixelDiscriminator's front half,
    arch/discriminators.py:70-73) on this input?  fp32 tensors in the split mode, cin in {3, 4, 20, 21}; the library decides the rest."""
    if not (FUSE_FRONT[0] and _MODE[0] == "f32s" and "fwd" in SPLIT_KINDS and x.dtype == torch.float32 and x.is_cuda):
        return False
    if not (tuple(w1.shape[2:]) == (1, 1) and tuple(w.shape[2:]) == (1, 1) and w1.shape[0] == w.shape[1] == 64
            and stride == 1 and pad == 0 and dil == 1 and pad_mode == PAD_ZEROS):
        return False
    n, cin, h, wd = x.shape
    key = ("front", n, cin, h, wd, w.shape[0], TUNING[0])
    v = _SPLIT_OK.get(key)
    if v is None:
        ws, plane = None, w.numel()
        d = make_desc((n, 64, h, wd), w.shape, 1, 0, 1, PAD_ZEROS, ACT_NONE, 0.0, F32, BF16X3, F32, _prec(), plane)
        v = _SPLIT_OK[key] = bool(lib.sscg_conv2d_front_applies(C.byref(d), cin))
    return v


def conv2d_front_fwd(x, w1, b1, slope1, w, bias, glc=None, want_h1=False):
    """(y, h1 or None, cs or None): y = conv(lrelu(conv(x, w1) + b1), w) + bias in one launch; h1 (the 64-channel map between the two
    convs) is written only when asked for; glc = (G, L, C): the launch also leaves the statistics records of the norm layer behind
    it (`cs` for norm_stats_from_conv)."""
    x = to_nhwc(x)
    n, cin, h, wd = x.shape
    wop, plane = weight_split(w)
    d = make_desc((n, 64, h, wd), w.shape, 1, 0, 1, PAD_ZEROS, ACT_NONE, 0.0, F32, BF16X3, F32, _prec(), plane)
    y = empty_nhwc(n, w.shape[0], h, wd, x.device, torch.float32)
    h1 = empty_nhwc(n, 64, h, wd, x.device, torch.float32) if want_h1 else None
    g = l = nb = 0
    sbuf = None
    if glc is not None:
        g, l = int(glc[0]), int(glc[1])
        nb = _stats_bytes(d, g, l)
        if not nb:
            raise _lib.SscgError("conv2d_front_fwd: no fused statistics for this geometry")
        sbuf = torch.empty(nb, dtype=torch.uint8, device=x.device)
    FRONT_CALLS[0] += 1
    w1f = w1.detach().reshape(64, cin)
    if not w1f.is_contiguous():
        w1f = w1f.contiguous()
    # (booked under the second conv's geometry + the front conv's algorithmic FLOP; the bytes are the launch's own: x in, y [+ h1] out)
    _timed("fwd", d, lambda: check(lib.sscg_conv2d_front_fwd(C.byref(d), x.data_ptr(), cin, w1f.data_ptr(), _ptr(b1), float(slope1), _ptr(h1),
                                                             wop.data_ptr(), _ptr(bias), y.data_ptr(), g, l, _ptr(sbuf), nb, _stream()),
                                   "sscg_conv2d_front_fwd"))
    return y, h1, ((d, sbuf) if sbuf is not None else None)


def weight_transposed(w, dtype=torch.float32):
    """[K][R][S][C] -> [C][R][S][K] (the operand layout of sscg_conv2d_dgrad), as fp32 or bf16; dtype "x3" = the three bf16
    planes of the split contraction (a flat bf16 tensor of 3 * numel elements)."""
    k, c, r, s = w.shape
    if dtype == "x3":
        wt = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=w.device)
        check(lib.sscg_weight_krsc_to_crsk(w.data_ptr(), F32, wt.data_ptr(), BF16X3, k, r * s, c, _stream()), "sscg_weight_krsc_to_crsk")
        return wt
    src = w
    if dtype == torch.bfloat16:
        src = weight_bf16(w)            # transposing the bf16 shadow reads half the bytes
    wt = torch.empty((c, k, r, s), dtype=dtype, device=w.device, memory_format=CL)
    check(lib.sscg_weight_krsc_to_crsk(src.data_ptr(), _dt(src), wt.data_ptr(), _DT[dtype], k, r * s, c, _stream()),
          "sscg_weight_krsc_to_crsk")
    return wt


def dgrad_operand(w, xshape, stride, pad, dil, dy_dtype=torch.float32):
    """The transposed operand copy [C][R][S][K] of weight `w` in the form the data gradient of this geometry reads in the current
    mode (uncached: tools and kernel tests; the networks use the per-parameter cache of conv2d_dgrad_param)."""
    if dy_dtype == torch.bfloat16:
        return weight_transposed(w, torch.bfloat16)
    if _MODE[0] == "f32s" and "dgrad" in SPLIT_KINDS and split_applies(xshape, w.shape, stride, pad, dil, PAD_ZEROS, 1):
        return weight_transposed(w, "x3")
    return weight_transposed(w, torch.float32)


def _bsums_bytes(d, g, l):
    key = (id(d), "bsums", g, l)
    v = _PLAN_SIZES.get(key)
    if v is None:
        v = _PLAN_SIZES[key] = lib.sscg_conv2d_dgrad_bsums_bytes(C.byref(d), g, l)
    return v


def _dgrad_add_applies(d):
    key = (id(d), "dgrad_add")
    v = _PLAN_SIZES.get(key)
    if v is None:
        v = _PLAN_SIZES[key] = bool(lib.sscg_conv2d_dgrad_add_applies(C.byref(d)))
    return v


def conv2d_dgrad(dy, wt, xshape, wshape, stride, pad, dil, bias=None, act=ACT_NONE, slope=0.0, out_dtype=torch.float32, bsums=None,
                 addend=None, z=None):
    """dx = act(dgrad(dy, wt) + bias); `wt` is the transposed operand copy [C][R][S][K] (weight_transposed), fp32 for an
    fp32 dy, bf16 for a bf16 dy.
    bsums = (nx, mean, rstd, gamma, beta, (G, L, C), act, slope[, residual]): dx is the gradient at the output of
    act(norm(nx) [+ residual]); the launch also takes that layer's backward sums in its epilogue (`residual` true: the mask is read
    off the unit's output `z`, this conv's own input).
    addend: a tensor like dx - the gradient another consumer of the same tensor left - joined in the store phase (dx = dgrad + addend;
    the sums then see the total).
    With bsums or addend the result is (dx, records, joined): records None where the library does not fuse the geometry (the caller
    runs the ordinary reduction pass), joined False where the addend was NOT added (the caller adds)."""
    if act == ACT_TANH and xshape[1] > 32 and dy.dtype == torch.bfloat16 and bsums is None and addend is None:
        return act_fwd(conv2d_dgrad(dy, wt, xshape, wshape, stride, pad, dil, bias, ACT_NONE, 0.0, out_dtype), ACT_TANH, 0.0)
    wdt = BF16X3 if (wt.dim() == 1 and wt.dtype == torch.bfloat16) else _dt(wt)      # the split copy is a flat tensor of three planes
    d = make_desc(xshape, wshape, stride, pad, dil, xdt=_DT[out_dtype], wdt=wdt, ydt=_dt(dy), prec=_prec("dgrad"))
    dx = empty_nhwc(d.N, d.C, d.H, d.W, dy.device, out_dtype)
    ws = _WS.get(_ws_bytes(d, "dgrad"), dy.device)
    want3 = bsums is not None or addend is not None
    plain = bias is None and act == ACT_NONE
    f32 = out_dtype == torch.float32 and dy.dtype == torch.float32
    if addend is not None and not (plain and addend.dtype == out_dtype == dy.dtype and tuple(addend.shape) == tuple(dx.shape)
                                   and addend.stride() == dx.stride() and _dgrad_add_applies(d)):
        addend = None
    joinable = addend is not None
    if joinable and not f32:
        bsums = None            # (bf16 tensors: the addend joins, the sums of a joined gradient are not taken - conv16_kernel)
    if bsums is not None:
        nx, mean, rstd, gamma, beta, (g, l, c), nact, nslope = bsums[:8]
        res = len(bsums) > 8 and bsums[8]
        nb = _bsums_bytes(d, g, l) if (plain and c == d.C and nx.dtype == out_dtype and dy.dtype == out_dtype) else 0
        if res and not (f32 and z is not None and z.dtype == torch.float32 and z.stride() == dx.stride()):
            nb = 0
        if nb:
            sums = torch.empty(nb, dtype=torch.uint8, device=dy.device)
            fused_reads = dx.numel() * dx.element_size() * (1 + (1 if res else 0) + (1 if addend is not None else 0))      # nx [, z] [, addend]
            _timed("dgrad", d, lambda: check(lib.sscg_conv2d_dgrad_bsums(
                C.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), nx.data_ptr(), z.data_ptr() if res else None, _ptr(addend),
                mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), _ptr(beta), g, l, nact, nslope, sums.data_ptr(), nb, ws.data_ptr(),
                ws.numel(), _stream()), "sscg_conv2d_dgrad_bsums"), extra_bytes=fused_reads)
            return dx, (d, sums, res), joinable
    if joinable:
        _timed("dgrad", d, lambda: check(lib.sscg_conv2d_dgrad_add(C.byref(d), dy.data_ptr(), wt.data_ptr(), addend.data_ptr(), dx.data_ptr(),
                                                                   ws.data_ptr(), ws.numel(), _stream()), "sscg_conv2d_dgrad_add"),
               extra_bytes=dx.numel() * dx.element_size())
        return dx, None, True
    _timed("dgrad", d, lambda: check(lib.sscg_conv2d_dgrad(C.byref(d), dy.data_ptr(), wt.data_ptr(), _ptr(bias), dx.data_ptr(),
                                                           act, slope, ws.data_ptr(), ws.numel(), _stream()), "sscg_conv2d_dgrad"))
    return (dx, None, False) if want3 else dx


def conv2d_dgrad_param(dy, w, xshape, wshape, stride, pad, dil, bias=None, act=ACT_NONE, slope=0.0, out_dtype=torch.float32, bsums=None,
                       addend=None, z=None):
    """conv2d_dgrad with the transposed operand copy of parameter `w` taken from the per-parameter cache, in the element
    type the kernel for dy reads (bf16 tiles for a bf16 dy, fp32 otherwise)."""
    if dy.dtype == torch.float32 and out_dtype == torch.float32 and bias is None and act == ACT_NONE:
        cp = _padded_stem(xshape, wshape, stride, pad, dil, PAD_ZEROS, 1) if (bsums is None and addend is None) else 0
        if cp:      # the padded weight's data gradient has 32 channels: the extra ones (zero weights: zero gradient) are cut again
            wp = _padded_weight(w, cp)
            xs = (xshape[0], cp, xshape[2], xshape[3])
            d0 = make_desc(xshape, wshape, stride, pad, dil, xdt=F32, wdt=BF16X3, ydt=F32, prec=_prec("dgrad"))
            return _timed("dgrad", d0, lambda: _unprofiled(lambda: resize_channels(
                conv2d_dgrad(dy, _cached_wt(wp, "x3"), xs, tuple(wp.shape), stride, pad, dil), xshape[1])))
        kp = _padded_head(xshape, wshape, stride, pad, dil)
        if kp:      # a 21 / 20-channel head (the DeepLab classifiers): dy and the filters padded to 32 OUTPUT channels - zero gradients
            wp = _cached_copy(w, "_sscg_wpadk", lambda: _pad_filters(w.detach(), kp))       # against zero filters add nothing to dx; the
            d0 = make_desc(xshape, wshape, stride, pad, dil, xdt=F32, wdt=BF16X3, ydt=F32, prec=_prec("dgrad"))      # result has dx's own shape:
            return _timed("dgrad", d0, lambda: _unprofiled(lambda: conv2d_dgrad(      # the fused store phases (fan-in, backward sums) apply
                resize_channels(dy, kp), _cached_wt(wp, "x3"), xshape, tuple(wp.shape), stride, pad, dil, bsums=bsums, addend=addend, z=z)))
    if dy.dtype == torch.bfloat16:
        if wshape[0] % 64:
            raise _lib.SscgError("bf16 output gradients with %d channels: the bf16 conv kernels need a multiple of 64" % wshape[0])
        wt = _cached_wt(w, torch.bfloat16)
    elif (_MODE[0] == "f32s" and "dgrad" in SPLIT_KINDS and out_dtype == torch.float32
          and split_applies(xshape, wshape, stride, pad, dil, PAD_ZEROS, 1)):
        wt = _cached_wt(w, "x3")
    else:
        wt = _cached_wt(w, torch.float32)
    return conv2d_dgrad(dy, wt, xshape, wshape, stride, pad, dil, bias, act, slope, out_dtype, bsums, addend, z)


def conv2d_wgrad(x, dy, wshape, stride, pad, dil, pad_mode=PAD_ZEROS, out=None, accumulate=False):
    d = make_desc(x.shape, wshape, stride, pad, dil, pad_mode, xdt=_dt(x), ydt=_dt(dy), prec=_prec("wgrad"))
    if out is None:
        k, c, r, s = wshape
        out = torch.empty((k, c, r, s), dtype=torch.float32, device=x.device, memory_format=CL)
        accumulate = False
    ws = _WS.get(_ws_bytes(d, "wgrad"), x.device)
    _timed("wgrad", d, lambda: check(lib.sscg_conv2d_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), out.data_ptr(),
                                                           1.0 if accumulate else 0.0, ws.data_ptr(), ws.numel(), _stream()),
                                     "sscg_conv2d_wgrad"))
    return out


def colsum(x2d_rows, cols, x, out=None, accumulate=False):
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
        accumulate = False
    nb = _cached_size(lib.sscg_colsum_workspace, x2d_rows, cols)
    ws = _WS.get(nb, x.device)
    check(lib.sscg_colsum(x.data_ptr(), _dt(x), out.data_ptr(), x2d_rows, cols, 1.0 if accumulate else 0.0, ws.data_ptr(),
                          ws.numel(), _stream()), "sscg_colsum")
    return out


_SIZE_CACHE = {}


def _cached_size(fn, *args):
    """Workspace size queries that depend on the geometry alone (no tuning hook changes them)."""
    key = (id(fn),) + args
    v = _SIZE_CACHE.get(key)
    if v is None:
        v = _SIZE_CACHE[key] = fn(*args)
    return v


def _glc(x, per_sample):
    return _glc_shape(x.shape, per_sample)


def _glc_shape(shape, per_sample):
    """[G][L][C] view of an NHWC tensor.  per_sample: True = InstanceNorm (G = N), False = BatchNorm (G = 1), an int
    k > 1 = BatchNorm over k batches stacked along N (G = k, each group N/k samples)."""
    n, c, h, w = shape
    if per_sample is True:
        return n, h * w, c
    k = 1 if per_sample is False else int(per_sample)
    if k < 1 or n % k:
        raise _lib.SscgError("batch of %d does not split into %d groups" % (n, k))
    return k, (n // k) * h * w, c


def norm_stats(x, per_sample, eps=1e-5, running_mean=None, running_var=None, momentum=0.1):
    g, l, c = _glc(x, per_sample)
    mean = torch.empty((g, c), dtype=torch.float32, device=x.device)
    rstd = torch.empty((g, c), dtype=torch.float32, device=x.device)
    nb = _cached_size(lib.sscg_norm_stats_workspace, g, l, c)
    ws = _WS.get(nb, x.device)
    check(lib.sscg_norm_stats(x.data_ptr(), _dt(x), g, l, c, eps, mean.data_ptr(), rstd.data_ptr(), _ptr(running_mean),
                              _ptr(running_var), momentum, ws.data_ptr(), ws.numel(), _stream()), "sscg_norm_stats")
    return mean, rstd


def norm_apply(x, mean, rstd, gamma, beta, residual, per_sample, act=ACT_NONE, slope=0.0):
    g, l, c = _glc(x, per_sample)
    y = torch.empty_like(x, memory_format=CL)
    check(lib.sscg_norm_apply(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), _ptr(beta), _ptr(residual),
                              y.data_ptr(), _same_dtype(x, residual), g, l, c, act, slope, _stream()), "sscg_norm_apply")
    return y


def norm_bwd(dy, x, y, mean, rstd, gamma, per_sample, act, slope, stats_grad=True, want_dres=False, dgamma=None,
             dbeta=None, beta=None, overwrite=False):
    """y None (ReLU / LeakyReLU, no residual in the forward): the activation mask is recomputed from x, gamma, beta."""
    g, l, c = _glc(x, per_sample)
    dx = torch.empty_like(x, memory_format=CL)
    dres = torch.empty_like(x, memory_format=CL) if want_dres else None
    nb = _cached_size(lib.sscg_norm_bwd_workspace, g, l, c)
    ws = _WS.get(nb, x.device)
    check(lib.sscg_norm_bwd(dy.data_ptr(), x.data_ptr(), _ptr(y), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), _ptr(beta),
                            dx.data_ptr(), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _same_dtype(x, dy, y), g, l, c, act, slope,
                            (1 if stats_grad else 0) | (2 if overwrite else 0), ws.data_ptr(), ws.numel(), _stream()), "sscg_norm_bwd")
    return dx, dres


def norm_bwd_from_sums(rec, dy, x, mean, rstd, gamma, beta, per_sample, act, slope, dgamma=None, dbeta=None, y=None, want_dres=False):
    """norm_bwd (batch statistics, gradients of gamma / beta written) for a dy whose backward sums the data gradient that produced it
    already took (conv2d_dgrad(..., bsums=)): finalize + apply.  The mask is recomputed from x, or read off the unit's forward output
    y when a residual joined it (want_dres: its gradient, the masked dy).  Returns (dx, dres)."""
    d, sums = rec[0], rec[1]
    g, l, c = _glc(x, per_sample)
    dx = torch.empty_like(x, memory_format=CL)
    dres = torch.empty_like(x, memory_format=CL) if want_dres else None
    ws = _WS.get(g * c * 8, x.device)
    check(lib.sscg_norm_bwd_from_sums(C.byref(d), sums.data_ptr(), dy.data_ptr(), x.data_ptr(), _ptr(y), mean.data_ptr(), rstd.data_ptr(),
                                      _ptr(gamma), _ptr(beta), dx.data_ptr(), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _same_dtype(x, dy), g, l,
                                      c, act, slope, 2, ws.data_ptr(), ws.numel(), _stream()), "sscg_norm_bwd_from_sums")
    return dx, dres


def norm_head_applies(c):
    """Does the one-pass norm -> activation -> 1x1 single-channel conv (PixelDiscriminator's tail) serve a C-channel map?"""
    return bool(lib.sscg_norm_head_applies(int(c)))


def norm_head_fwd(x, mean, rstd, gamma, beta, w, bias, per_sample, act, slope):
    """out[N,1,H,W] (fp32) = bias + sum_c w[c] * act(norm(x)) - one pass over x, the normalised map is never written."""
    g, l, c = _glc(x, per_sample)
    n, _, h, wd = x.shape
    out = torch.empty((n, 1, h, wd), dtype=torch.float32, device=x.device)
    check(lib.sscg_norm_head_fwd(x.data_ptr(), _dt(x), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), _ptr(beta), w.data_ptr(),
                                 _ptr(bias), out.data_ptr(), g, l, c, act, slope, _stream()), "sscg_norm_head_fwd")
    return out


def norm_head_bwd(dout, w, x, mean, rstd, gamma, beta, per_sample, act, slope, stats_grad, dwb, dgamma=None, dbeta=None):
    """Backward of norm_head_fwd: the gradient at x; dwb[:C] / dwb[C] (written) = the head's weight / bias gradient."""
    g, l, c = _glc(x, per_sample)
    dx = torch.empty_like(x, memory_format=CL)
    nb = _cached_size(lib.sscg_norm_head_bwd_workspace, g, l, c)
    ws = _WS.get(nb, x.device)
    check(lib.sscg_norm_head_bwd(dout.data_ptr(), w.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), _ptr(beta),
                                 dx.data_ptr(), dwb.data_ptr(), dwb.data_ptr() + 4 * c, _ptr(dgamma), _ptr(dbeta), _dt(x), g, l, c, act,
                                 slope, (1 if stats_grad else 0) | 2 | 4, ws.data_ptr(), ws.numel(), _stream()), "sscg_norm_head_bwd")
    return dx


def rstd_from_var(var, eps):
    out = torch.empty_like(var)
    check(lib.sscg_rstd_from_var(var.data_ptr(), out.data_ptr(), var.numel(), eps, _stream()), "sscg_rstd_from_var")
    return out


def act_fwd(x, act, slope=0.0):
    y = torch.empty_like(x, memory_format=torch.preserve_format)
    check(lib.sscg_act_fwd(x.data_ptr(), y.data_ptr(), _dt(x), x.numel(), act, slope, _stream()), "sscg_act_fwd")
    return y


def act_bwd(dy, y, act, slope=0.0):
    dx = torch.empty_like(y, memory_format=torch.preserve_format)
    check(lib.sscg_act_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), _same_dtype(y, dy), y.numel(), act, slope, _stream()), "sscg_act_bwd")
    return dx


def add(a, b):
    y = torch.empty_like(a, memory_format=torch.preserve_format)
    check(lib.sscg_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), _same_dtype(a, b), a.numel(), _stream()), "sscg_add")
    return y


def fill_(x, v):
    if x.dtype != torch.float32:
        if v != 0.0 or (x.numel() * x.element_size()) % 4:
            raise _lib.SscgError("fill_ of a %s tensor supports zero only" % x.dtype)
        check(lib.sscg_fill(x.data_ptr(), x.numel() * x.element_size() // 4, 0.0, _stream()), "sscg_fill")
        return x
    check(lib.sscg_fill(x.data_ptr(), x.numel(), float(v), _stream()), "sscg_fill")
    return x


def dropout(x, p, seed):
    y = torch.empty_like(x, memory_format=torch.preserve_format)
    check(lib.sscg_dropout(x.data_ptr(), y.data_ptr(), _dt(x), x.numel(), p, seed, _stream()), "sscg_dropout")
    return y


def gauss_noise(x, sigma, seed):
    """utils.GaussianNoise.forward on the device: x + sigma * x * N(0, 1) (utils.py:133-139)."""
    _need_hip(x, f32_only=True)
    y = torch.empty_like(x, memory_format=torch.preserve_format)
    check(lib.sscg_gauss_noise(x.data_ptr(), y.data_ptr(), x.numel(), float(sigma), int(seed), _stream()), "sscg_gauss_noise")
    return y


def pool_out_size(h):
    """MaxPool2d(3, 2, 1, ceil_mode=True) output size (torch rule: last window must start inside input+left pad)."""
    o = -((-(h + 2 - 3)) // 2) + 1
    if (o - 1) * 2 >= h + 1:
        o -= 1
    return o


def maxpool_fwd(x):
    n, c, h, w = x.shape
    p, q = pool_out_size(h), pool_out_size(w)
    y = empty_nhwc(n, c, p, q, x.device, x.dtype)
    idx = torch.empty((n, c, p, q), dtype=torch.uint8, device=x.device, memory_format=CL)
    check(lib.sscg_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), _dt(x), n, h, w, c, p, q, _stream()),
          "sscg_maxpool3x3s2_fwd")
    return y, idx


def maxpool_bwd(dy, idx, xshape):
    n, c, h, w = xshape
    p, q = dy.shape[2], dy.shape[3]
    dx = empty_nhwc(n, c, h, w, dy.device, dy.dtype)
    check(lib.sscg_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), _dt(dy), n, h, w, c, p, q, _stream()),
          "sscg_maxpool3x3s2_bwd")
    return dx


def upsample_fwd(x, oh, ow):
    _need_hip(x, f32_only=True)          # the resize sits on the fp32 side of a network head (model.py:390-392)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, oh, ow, x.device)
    check(lib.sscg_upsample_bilinear_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, oh, ow, _stream()), "sscg_upsample_bilinear_fwd")
    return y


def upsample_bwd(dy, h, w):
    n, c, oh, ow = dy.shape
    dx = empty_nhwc(n, c, h, w, dy.device)
    check(lib.sscg_upsample_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), n, h, w, c, oh, ow, _stream()), "sscg_upsample_bilinear_bwd")
    return dx


def reflect_pad(x, pad):
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h + 2 * pad, w + 2 * pad, x.device, x.dtype)
    check(lib.sscg_reflect_pad(x.data_ptr(), y.data_ptr(), _dt(x), n, h, w, c, pad, _stream()), "sscg_reflect_pad")
    return y


def reflect_pad_bwd(dy, pad):
    n, c, oh, ow = dy.shape
    dx = empty_nhwc(n, c, oh - 2 * pad, ow - 2 * pad, dy.device, dy.dtype)
    check(lib.sscg_reflect_pad_bwd(dy.data_ptr(), dx.data_ptr(), _dt(dy), n, oh - 2 * pad, ow - 2 * pad, c, pad, _stream()),
          "sscg_reflect_pad_bwd")
    return dx


def softmax_fwd(x):
    _need_hip(x, f32_only=True)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=CL)
    check(lib.sscg_softmax_fwd(x.data_ptr(), y.data_ptr(), n * h * w, c, _stream()), "sscg_softmax_fwd")
    return y


def softmax_bwd(dy, y):
    n, c, h, w = y.shape
    dx = torch.empty_like(y, memory_format=CL)
    check(lib.sscg_softmax_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), n * h * w, c, _stream()), "sscg_softmax_bwd")
    return dx


def argmax_onehot(x, want_index=False):
    """`x.max(1)[1]` + make_one_hot in one pass (model.py:435-437).  Returns (onehot NHWC, index [N,H,W] or None)."""
    _need_hip(x, f32_only=True)
    x = to_nhwc(x)
    n, c, h, w = x.shape
    oh = torch.empty_like(x, memory_format=CL)
    idx = torch.empty((n, h, w), dtype=torch.int64, device=x.device) if want_index else None
    check(lib.sscg_argmax_onehot(x.data_ptr(), oh.data_ptr(), _ptr(idx), n * h * w, c, _stream()), "sscg_argmax_onehot")
    return oh, idx


def argmax_index(x):
    _need_hip(x, f32_only=True)
    x = to_nhwc(x)
    n, c, h, w = x.shape
    idx = torch.empty((n, h, w), dtype=torch.int64, device=x.device)
    check(lib.sscg_argmax_onehot(x.data_ptr(), None, idx.data_ptr(), n * h * w, c, _stream()), "sscg_argmax_onehot")
    return idx


def confusion_hist(label_true, label_pred, num_classes, hist=None):
    """hist[C*t + p] += 1 over the pixels with 0 <= t < C (utils.py:363-369); `hist` int64 [C, C] on the device."""
    if not (label_true.is_cuda and label_pred.is_cuda):
        raise _lib.SscgError("sscg kernels run on the MI355X only: got a %s tensor (no CPU fallback)" % label_true.device)
    lt = label_true.to(torch.int64).contiguous()
    lp = label_pred.to(torch.int64).contiguous()
    if lt.numel() != lp.numel():
        raise _lib.SscgError("confusion_hist: label/prediction sizes differ")
    if hist is None:
        hist = torch.empty((num_classes, num_classes), dtype=torch.int64, device=lt.device)
        check(lib.sscg_fill(hist.data_ptr(), 2 * hist.numel(), 0.0, _stream()), "sscg_fill")   # 2 fp32 zeros per int64 zero
    check(lib.sscg_confusion_hist(lt.data_ptr(), lp.data_ptr(), lt.numel(), num_classes, hist.data_ptr(), _stream()),
          "sscg_confusion_hist")
    return hist


def image_u8_to_f32(img_u8, mean, std):
    """uint8 [B,H,W,C] (HWC pixels as PIL decoded them) -> fp32 logical [B,C,H,W], channels-last, ((u/255) - mean) / std:
    ToTensor + Normalize of data_utils/__init__.py:126-150 on the device."""
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or img_u8.dim() != 4:
        raise _lib.SscgError("image_u8_to_f32: uint8 [B,H,W,C] tensor on the MI355X expected")
    b, h, w, c = img_u8.shape
    src = img_u8.contiguous()
    dst = torch.empty((b, h, w, c), dtype=torch.float32, device=src.device)
    check(lib.sscg_image_u8_to_f32(src.data_ptr(), dst.data_ptr(), b * h * w, c, mean.data_ptr(), std.data_ptr(), _stream()),
          "sscg_image_u8_to_f32")
    return dst.permute(0, 3, 1, 2)


def label_lut(gt_u8, lut):
    """uint8 label ids [B,H,W] -> int64 [B,1,H,W] through a 256-entry table (ToLabel + Relabel / encode_segmap)."""
    if not gt_u8.is_cuda or gt_u8.dtype != torch.uint8 or lut.dtype != torch.int64 or lut.numel() != 256:
        raise _lib.SscgError("label_lut: uint8 labels on the MI355X and an int64 table of 256 entries expected")
    src = gt_u8.contiguous()
    dst = torch.empty(src.shape, dtype=torch.int64, device=src.device)
    check(lib.sscg_label_lut(src.data_ptr(), dst.data_ptr(), src.numel(), lut.data_ptr(), _stream()), "sscg_label_lut")
    return dst.unsqueeze(1)


def label_onehot(labels, num_classes):
    """utils.make_one_hot (utils.py:314-350): labels int64 [N,1,H,W] -> fp32 one-hot [N,C,H,W] (NHWC memory)."""
    if not labels.is_cuda or labels.dtype != torch.int64:
        raise _lib.SscgError("int64 HIP label tensor expected")
    labels = labels.contiguous()
    n, _, h, w = labels.shape
    oh = empty_nhwc(n, num_classes, h, w, labels.device)
    check(lib.sscg_label_onehot(labels.data_ptr(), oh.data_ptr(), n * h * w, num_classes, _stream()), "sscg_label_onehot")
    return oh


def _scalar(device):
    return torch.empty((), dtype=torch.float32, device=device)


def _loss_ws(device):
    return _WS.get(lib.sscg_loss_workspace(0), device)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, shadow_bf16=None, shadow_split=None):
    """shadow_bf16: bf16 arena rewritten with the parameters; shadow_split: the 3-plane split arena (3 * numel bf16) instead."""
    sh, sdt = (shadow_split, BF16X3) if shadow_split is not None else (shadow_bf16, BF16)
    check(lib.sscg_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(sh), sdt, p.numel(), lr, beta1, beta2,
                             eps, step, grad_scale, _stream()), "sscg_adam_step")


# ----------------------------------------------------------------------------- autograd functions
_WEIGHT_EPOCH = [0]     # weights owned by no arena optimiser (stock torch optimisers rewrite `_version` instead)


def bump_weight_epoch(epoch_cell=None):
    """Called by the optimiser after it rewrote parameters: invalidates the cached transposed copies of ITS weights
    (`epoch_cell` is the one-element list its parameters carry as `_sscg_epoch`; the discriminators' update must not
    make the generators rebuild 450 transposes)."""
    (epoch_cell if epoch_cell is not None else _WEIGHT_EPOCH)[0] += 1


_WT_USERS = {}      # id -> (weakref of the weight, set of operand-copy kinds a pass has asked for: "t32", "t16", "w16")


def _note_user(w, kind):
    ent = _WT_USERS.get(id(w))
    if ent is None:
        key = id(w)
        ent = _WT_USERS[key] = (weakref.ref(w, lambda _r, key=key: _WT_USERS.pop(key, None)), set())
    ent[1].add(kind)


def refresh_transposed_weights(weights=(), all_users=True):
    """Bring the cached operand copies (transposed fp32 / transposed bf16 / plain bf16) of `weights` - and, with
    `all_users`, of every weight a pass has asked for - up to date on the CURRENT stream.  The step calls this before it
    forks its forward passes over two streams: their backward passes then only read the cache (a lazily rebuilt copy would
    be written on one stream and read on the other).

    `all_users` is only safe on a stream that is ordered behind EVERY optimiser's last update and every reader of the
    stale copies (a rebuild frees the old copy: the allocator may hand its block to the next allocation at once)."""
    _IN_REFRESH[0] = True
    try:
        _refresh_all(weights, all_users)
    finally:
        _IN_REFRESH[0] = False


def _refresh_all(weights, all_users):
    bf16 = _MODE[0] == "bf16"
    jobs = []
    if _MODE[0] != "f32s":       # (split mode: which copy a weight needs depends on its geometry - the registry below knows)
        for w in weights:
            kind = "t16" if (bf16 and w.shape[0] % 64 == 0) else "t32"
            _note_user(w, kind)
            if not all_users:
                _refresh_kinds(w, (kind,), jobs)
    if not all_users:
        if _MODE[0] == "f32s":
            for w in weights:
                ent = _WT_USERS.get(id(w))
                if ent is not None:
                    _refresh_kinds(w, ent[1], jobs)
        _transpose_batch(jobs)
        return
    for r, kinds in list(_WT_USERS.values()):
        w = r()
        if w is None:
            continue
        _refresh_kinds(w, kinds, jobs)
    _transpose_batch(jobs)


_WT_ATTR = {"tx3": "_sscg_wtx3", "t16": "_sscg_wt16", "t32": "_sscg_wt"}
BATCH_TRANSPOSES = [os.environ.get("SSCG_BATCH_TRANSPOSES", "1") != "0"]


def _refresh_kinds(w, kinds, jobs=None):
    """Bring the operand copies `kinds` of weight w up to date.  With `jobs` (a list), a stale TRANSPOSED copy is not rebuilt here:
    the weight is appended and `_transpose_batch` rebuilds all of them in one launch."""
    for kind, dtype in (("t32", torch.float32), ("t16", torch.bfloat16), ("tx3", "x3")):
        if kind in kinds:
            if jobs is not None and BATCH_TRANSPOSES[0]:
                ent = getattr(w, _WT_ATTR[kind], None)
                tag = _wtag(w)
                if ent is None or ent.tag != tag:
                    jobs.append((w, kind, ent, tag))
            else:
                _cached_wt(w, dtype)
    if "w16" in kinds:
        weight_bf16(w)
    if "wx3" in kinds:
        weight_split(w)


class _WtJob(C.Structure):          # include/sscg.h: sscg_wt_job
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("w_dtype", C.c_int32), ("wt_dtype", C.c_int32),
                ("K", C.c_int32), ("RS", C.c_int32), ("C", C.c_int32), ("block0", C.c_int32)]


_WT_TABLES = {}      # (source, destination) pointers of a batch -> (device table, blocks): the same weights come back every step


def _transpose_batch(jobs):
    """The stale transposed operand copies of a whole step in ONE launch (sscg_weight_krsc_to_crsk_batch; 234 launches of ~6 us at the
    top of a config-2 step otherwise).  A copy that exists is rewritten IN PLACE - `refresh_transposed_weights` runs ordered behind
    every reader of the stale copies, which is also what freeing them (the per-weight path) needs - so the pointers, and with them
    the job table in device memory, are the same from the second step on."""
    assert _IN_REFRESH[0]
    if len(jobs) < 2:
        for w, kind, _, _ in jobs:
            _cached_wt(w, {"t32": torch.float32, "t16": torch.bfloat16, "tx3": "x3"}[kind])
        return
    rows = []
    blocks = 0
    keep = []
    fast = not hasattr(lib, "note_batch")       # (racecheck.py wants every buffer's extent: it sees them in data_ptr())
    cur = _stream()
    for w, kind, ent, tag in jobs:
        src = weight_bf16(w) if kind == "t16" else w
        row = getattr(ent, "row", None) if (fast and ent is not None) else None
        if row is not None and ent.ev is None and row[0] == (tag[2] if src is w else src.data_ptr()):
            rows.append(row + (blocks,))        # same source, same destination as the last time: rewritten in place
            blocks += ent.nblk
            ent.tag, ent.stream = tag, cur
            continue
        k, c, r, s = w.shape
        temp = not src.is_contiguous(memory_format=CL)
        if temp:
            src = src.contiguous(memory_format=CL)
            keep.append(src)
        t = None
        if ent is not None:
            if ent.ev is None and ent.t.device == w.device:
                t = ent.t               # built by a refresh: its readers are the ones our caller is ordered behind
            else:
                ent.retire()            # built lazily (a model's first step): the allocator keeps the block for its readers
        if t is None:
            t = (torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=w.device) if kind == "tx3" else
                 torch.empty((c, k, r, s), dtype=torch.bfloat16 if kind == "t16" else torch.float32, device=w.device, memory_format=CL))
        wt_dtype = {"t32": F32, "t16": BF16, "tx3": BF16X3}[kind]
        row = (src.data_ptr(), t.data_ptr(), _dt(src), wt_dtype, k, r * s, c)
        rows.append(row + (blocks,))
        copy = _Copy.__new__(_Copy)
        copy.tag, copy.t, copy.stream, copy.ev, copy.readers = tag, t, cur, None, None
        copy.nblk = -(-c // 32) * -(-k // 32) * r * s
        copy.row = None if temp else row
        blocks += copy.nblk
        keep.append((w, kind, copy))
    key = tuple(rows)
    tab = _WT_TABLES.get(key)
    if tab is None:
        if len(_WT_TABLES) >= 16:           # (a table may still be in use on another stream)
            torch.cuda.synchronize()
            _WT_TABLES.clear()
        host = (_WtJob * len(rows))(*[_WtJob(*row) for row in rows])
        dev_t = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(jobs[0][0].device)
        torch.cuda.current_stream().synchronize()       # once per table: later steps launch it from whichever lane refreshes
        tab = _WT_TABLES[key] = (dev_t, rows)
    _note_batch(tab)
    check(lib.sscg_weight_krsc_to_crsk_batch(tab[0].data_ptr(), len(rows), blocks, _stream()), "sscg_weight_krsc_to_crsk_batch")
    for item in keep:
        if isinstance(item, tuple):
            w, kind, copy = item
            try:
                setattr(w, _WT_ATTR[kind], copy)
            except AttributeError:
                pass


def _note_batch(tab):
    """racecheck.py sees pointers in argument lists; the batch's buffers sit in a device table - announce them."""
    note = getattr(lib, "note_batch", None)
    if note is not None:
        note(tab[0].data_ptr(), [(row[0], row[1]) for row in tab[1]])


def _wtag(w):
    return (w._version, getattr(w, "_sscg_epoch", _WEIGHT_EPOCH)[0], w.data_ptr())


_NO_COPY_SYNC = os.environ.get("SSCG_DBG_NO_COPY_SYNC") == "1"      # A/B aid: round 3's behaviour (tests/aids/fuzz_step.py shows the race)
_IN_REFRESH = [False]       # inside refresh_transposed_weights: the CALLER orders the readers (one event for the whole batch)


class _Copy(object):
    """A cached operand copy of a weight.  A copy built inside `refresh_transposed_weights` is ordered by its caller (one event for
    the batch: model._copies_ready).  A copy built LAZILY - by the first pass that needs it, on that pass's stream - carries its own
    event: a reader on another stream waits for it first (once per stream), and every reader stream is announced to the allocator
    (`record_stream`) when the copy is replaced, so that the block is not handed out while a reader is still in flight.  (Round 3
    had neither: at a model's first step - empty registry, nothing prebuilt - the copies were built by whichever of the two forward
    lanes reached a layer first and read by the other unsynchronised; racecheck.py reports exactly these launches.)"""
    __slots__ = ("tag", "t", "ev", "stream", "readers", "row", "nblk")

    def __init__(self, tag, build):
        self.tag = tag
        self.stream = _stream()
        self.t = build()
        self.readers = None
        self.ev = None
        if not _IN_REFRESH[0] and not _NO_COPY_SYNC:
            self.ev = torch.cuda.Event()
            self.ev.record()
            self.readers = set()

    def get(self):
        if self.ev is not None:
            h = _stream()
            if h != self.stream and h not in self.readers:
                torch.cuda.current_stream().wait_event(self.ev)
                self.readers.add(h)
        return self.t

    def retire(self):
        """The copy is about to be dropped: its block must outlive the readers on other streams."""
        if self.readers:
            dev = self.t.device
            for h in self.readers:
                self.t.record_stream(_stream_object(dev, h))


def _cached_copy(w, attr, build):
    tag = _wtag(w)
    ent = getattr(w, attr, None)
    if ent is None or ent.tag != tag:
        if ent is not None:
            ent.retire()
        ent = _Copy(tag, build)
        try:
            setattr(w, attr, ent)
        except AttributeError:
            pass
    return ent.get()


def _cached_wt(w, dtype=torch.float32):
    """Transposed copy of a weight, cached ON the tensor object (dies with it; a recycled address can never
    alias).  Valid while neither torch (`_version`) nor our optimiser (`_WEIGHT_EPOCH`) has rewritten it."""
    kind = "tx3" if dtype == "x3" else ("t16" if dtype == torch.bfloat16 else "t32")
    _note_user(w, kind)
    attr = {"tx3": "_sscg_wtx3", "t16": "_sscg_wt16", "t32": "_sscg_wt"}[kind]
    return _cached_copy(w, attr, lambda: weight_transposed(w, dtype))


def weight_bf16(w):
    """bf16 operand copy of a conv weight, same [K][R][S][C] layout.  A parameter owned by optim.FusedAdam reads the
    optimiser's bf16 shadow arena (rewritten by the Adam kernel itself: no per-step cast); any other weight (the frozen
    nets) gets a cached cast."""
    _note_user(w, "w16")
    opt = getattr(w, "_sscg_opt", None)
    opt = opt() if opt is not None else None
    if opt is not None:
        return opt.shadow_view(w)
    return _cached_copy(w, "_sscg_w16", lambda: cast(w, torch.bfloat16))


def _split3_copy(w):
    n = w.numel()
    t = torch.empty(3 * n, dtype=torch.bfloat16, device=w.device)
    src = w if w.is_contiguous(memory_format=CL) else w.contiguous(memory_format=CL)
    check(lib.sscg_split3(src.data_ptr(), t.data_ptr(), n, n, _stream()), "sscg_split3")
    return t


def weight_split(w):
    """(three-plane bf16 split copy of a conv weight in its own [K][R][S][C] layout, plane stride in elements): the operand of
    the split contraction.  A parameter owned by optim.FusedAdam reads the optimiser's split shadow arena (rewritten by the
    Adam kernel itself); any other weight (the frozen nets) gets a cached copy."""
    _note_user(w, "wx3")
    opt = getattr(w, "_sscg_opt", None)
    opt = opt() if opt is not None else None
    if opt is not None and hasattr(opt, "split_view"):
        return opt.split_view(w)
    return _cached_copy(w, "_sscg_wx3", lambda: _split3_copy(w)), w.numel()


def _acc_target(param):
    """Gradient arena slice of a parameter owned by optim.FusedAdam (None for a stock optimiser)."""
    acc = getattr(param, "_sscg_grad", None)
    if acc is not None:
        param._sscg_touched = True
    return acc


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (+ folded nn.ReflectionPad2d, + fused activation when no norm layer follows).

    `norm` = None, or the description of the normalisation layer that follows - (per_sample, eps, running_mean,
    running_var, momentum): the conv's epilogue then also produces that layer's batch statistics (arch/ops.py:40-57 "Conv +
    InstanceNorm / BatchNorm" blocks) and the function returns (y, mean, rstd); mean is None when the fusion does not apply
    to the geometry (the caller falls back to a statistics pass over y)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, pad_mode, act, slope, out_f32, norm):
        x = to_nhwc(x)
        mean = rstd = None
        if norm is not None:
            per_sample, eps, rmean, rvar, momentum = norm
            n, _, h, wd = x.shape
            p, q = conv_out_size(h, w.shape[2], stride, pad, dil), conv_out_size(wd, w.shape[3], stride, pad, dil)
            g, l, c = _glc_shape((n, w.shape[0], p, q), per_sample)
            if act == ACT_NONE:
                y, mean, rstd = conv2d_fwd_norm(x, w, bias, stride, pad, dil, pad_mode, out_f32, (g, l, c), eps, rmean, rvar, momentum)
            else:       # (no conv epilogue takes statistics behind an activation)
                y = conv2d_fwd(x, w, bias, stride, pad, dil, pad_mode, act, slope, out_f32)
        else:
            y = conv2d_fwd(x, w, bias, stride, pad, dil, pad_mode, act, slope, out_f32)
        ctx.cfg = (stride, pad, dil, pad_mode, act, slope)
        ctx.has_bias = bias is not None
        ctx.wref = w
        ctx.bref = bias
        if _will_backward(ctx, 1):
            _note_use(w, bias)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        # (mean, rstd) are non-differentiable outputs: left alone, autograd hands backward() two zero-filled tensors for them -
        # 850 fill launches per Cityscapes step
        ctx.set_materialize_grads(False)
        if norm is None:
            return y
        if mean is None:
            return y, None, None
        ctx.mark_non_differentiable(mean, rstd)
        return y, mean, rstd

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:          # nothing downstream used y
            return (None,) * 11
        x, w, y = ctx.saved_tensors
        stride, pad, dil, pad_mode, act, slope = ctx.cfg
        dy = to_nhwc(dy)
        if act != ACT_NONE:
            dy = act_bwd(dy, y, act, slope)
        dx, dw, db = _conv_backward(dy, x, w, ctx.wref, ctx.bref if ctx.has_bias else None, (stride, pad, dil, pad_mode),
                                    ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None, None, None, None, None


def _conv_backward(dy, x, w, wref, bref, geom, want_x, want_w, want_b):
    """Backward of y = conv(x, w) + bias for the gradient dy of the PRE-activation output: (dx, dw, db).  Weight / bias gradients of
    parameters owned by optim.FusedAdam go straight into its gradient arena on the parameter's side lane (dw, db then None)."""
    stride, pad, dil, pad_mode = geom
    dx = dw = db = None
    if want_x:
        if pad_mode == PAD_REFLECT:
            raise _lib.SscgError("input gradient through a reflection-padded conv: use ReflectPadFn + pad=0 conv")
        # x is the output of a conv -> norm -> activation unit that left its normalisation context on the tensor (ConvNormActFn):
        # this data gradient IS that unit's upstream gradient, so its epilogue takes the unit's backward sums and the reduction pass
        # over (dz, y) never runs.  The records travel on dx and are only honoured if dx arrives unchanged (`_version`): an
        # accumulation by the autograd engine (a tensor with two consumers) bumps it.
        # A fan-out (SplitFn) in front - a residual block's input feeds conv1 and the shortcut: the gradient the OTHER consumer already
        # left joins in this data gradient's store phase (`_Join`), the result is then the tensor's total gradient (and the sums above
        # apply to it); SplitFn.backward has nothing left to add.
        fuse = FUSE_BSUMS[0] and stride == 1 and x.dtype == dy.dtype and (x.dtype == torch.float32 or FUSE_BSUMS_BF16[0])
        join = getattr(x, "_sscg_join", None) if FUSE_JOIN[0] else None
        addend = None
        if join is not None:
            addend = join[0].take(join[1])
            info = join[0].norm if (fuse and addend is not None) else None
        else:
            info = getattr(x, "_sscg_norm", None) if fuse else None
        if info is not None and len(info) > 8 and info[8] and x.dtype != torch.float32:
            info = None             # (mask read off the unit's output: fp32 tensors only)
        if info is not None or addend is not None:
            dx, rec, joined = conv2d_dgrad_param(dy, wref, x.shape, w.shape, stride, pad, dil, out_dtype=x.dtype, bsums=info, addend=addend, z=x)
            if addend is not None and not joined:
                rec = None          # the sums saw a partial gradient
            if rec is not None:
                dx._sscg_bsums = (rec, dx._version)
            if join is not None:
                if joined:
                    join[0].folded = (join[1], dx)
                else:
                    join[0].deposit(join[1], dx)
        else:
            dx = conv2d_dgrad_param(dy, wref, x.shape, w.shape, stride, pad, dil, out_dtype=x.dtype)
            if join is not None:
                join[0].deposit(join[1], dx)
    wacc = _acc_target(wref) if want_w else None
    bacc = _acc_target(bref) if want_b else None
    n, k, p, q = dy.shape

    def arena_grads():      # accumulate straight into the optimiser's gradient arena
        if wacc is not None:
            conv2d_wgrad(x, dy, w.shape, stride, pad, dil, pad_mode, out=wacc, accumulate=True)
        if bacc is not None:
            colsum(n * p * q, k, dy, out=bacc, accumulate=True)

    if wacc is not None or bacc is not None:
        # nothing on the backward critical path reads these: run them beside the data-gradient chain
        run_on_side_stream(dy.device, (x, dy), arena_grads, lane=getattr(wref, "_sscg_lane", 0), defer=True)
    if want_w and wacc is None:
        dw = conv2d_wgrad(x, dy, w.shape, stride, pad, dil, pad_mode)
    if want_b and bacc is None:
        db = colsum(n * p * q, k, dy)
    if want_w:
        _note_done(wref, bref)
    return dx, dw, db


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d (arch/ops.py:55-56) as the data-gradient of the mirrored convolution.

    `w` is the torch ConvTranspose2d weight, logical [Cin, Cout, R, S], channels-last memory
    [Cin][R][S][Cout] - i.e. the [K][R][S][C] weight of a conv Cout->Cin."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, out_pad, act, slope, out_f32=False):
        x = to_nhwc(x)
        n, cin, h, wd = x.shape
        cin2, cout, r, s = w.shape
        oh = (h - 1) * stride - 2 * pad + (r - 1) + out_pad + 1
        ow = (wd - 1) * stride - 2 * pad + (s - 1) + out_pad + 1
        # mirrored conv: input [n, cout, oh, ow] -> output [n, cin, h, wd]
        if conv_out_size(oh, r, stride, pad, 1) != h or conv_out_size(ow, s, stride, pad, 1) != wd:
            raise _lib.SscgError("unsupported ConvTranspose2d geometry")
        # the transposed operand copy [Cout][R][S][Cin] comes from the per-parameter cache inside conv2d_dgrad
        y = conv2d_dgrad_param(x, w, (n, cout, oh, ow), (cin, cout, r, s), stride, pad, 1, bias, act, slope,
                               out_dtype=_out_dtype(out_f32))
        ctx.cfg = (stride, pad, act, slope)
        ctx.has_bias = bias is not None
        ctx.wref, ctx.bref = w, bias
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, slope = ctx.cfg
        dy = to_nhwc(dy)
        if act != ACT_NONE:
            dy = act_bwd(dy, y, act, slope)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # gradient wrt x = forward of the mirrored conv applied to dy
            dx = conv2d_fwd(dy, w, None, stride, pad, 1, out_f32=(x.dtype == torch.float32))
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        wacc = _acc_target(ctx.wref) if want_w else None
        bacc = _acc_target(ctx.bref) if want_b else None
        n, k, p, q = dy.shape

        def arena_grads():      # as Conv2dFn: a parameter's gradient kernels always run on ITS side lane (fixed accumulation order)
            if wacc is not None:
                # mirrored conv has input dy (as "x") and output-gradient x (as "dy")
                conv2d_wgrad(dy, x, w.shape, stride, pad, 1, out=wacc, accumulate=True)
            if bacc is not None:
                colsum(n * p * q, k, dy, out=bacc, accumulate=True)

        if wacc is not None or bacc is not None:
            run_on_side_stream(dy.device, (x, dy), arena_grads, lane=getattr(ctx.wref, "_sscg_lane", 0), defer=True)
        if want_w and wacc is None:
            dw = conv2d_wgrad(dy, x, w.shape, stride, pad, 1)
        if want_b and bacc is None:
            db = colsum(n * p * q, k, dy)
        return dx, dw, db, None, None, None, None, None, None


class NormActFn(torch.autograd.Function):
    """InstanceNorm2d / BatchNorm2d (+ residual add) + activation, one statistics pass + one apply pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, per_sample, training, momentum, eps, act, slope,
                pre_mean=None, pre_rstd=None):
        x = to_nhwc(x)
        if residual is not None:
            residual = to_nhwc(residual)
        use_batch_stats = training or running_mean is None
        if use_batch_stats and pre_mean is not None:
            mean, rstd = pre_mean, pre_rstd      # produced by the conv's epilogue (running statistics already advanced there)
        elif use_batch_stats:
            upd = training and running_mean is not None and per_sample is not True
            mean, rstd = norm_stats(x, per_sample, eps, running_mean if upd else None, running_var if upd else None, momentum)
        else:
            per_sample = False          # running statistics: one (mean, rstd) row whatever the grouping
            mean = running_mean.view(1, -1)
            rstd = rstd_from_var(running_var, eps).view(1, -1)
        y = norm_apply(x, mean, rstd, gamma, beta, residual, per_sample, act, slope)
        ctx.cfg = (per_sample, act, slope, use_batch_stats, residual is not None)
        ctx.gref, ctx.betaref = gamma, beta
        # y supplies the activation mask of the backward; without a residual (and for ReLU / LeakyReLU) the mask is the sign of
        # gamma * xhat + beta, which the backward kernels recompute from x: one tensor less to read in both of them
        need_y = act != ACT_NONE and (residual is not None or act not in (ACT_RELU, ACT_LRELU))
        ctx.save_for_backward(x, y if need_y else None, mean, rstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, gamma, beta = ctx.saved_tensors
        per_sample, act, slope, stats_grad, has_res = ctx.cfg
        dx, ret_g, ret_b, dres = _norm_backward(to_nhwc(dy), x, y, mean, rstd, gamma, beta, ctx.gref, ctx.betaref, per_sample, act, slope,
                                                stats_grad, gamma is not None and ctx.needs_input_grad[1],
                                                has_res and ctx.needs_input_grad[3])
        if not ctx.needs_input_grad[0]:
            dx = None
        return dx, ret_g, ret_b, dres, None, None, None, None, None, None, None, None, None, None


def _norm_backward(dy, x, y, mean, rstd, gamma, beta, gref, betaref, per_sample, act, slope, stats_grad, want_g, want_dres):
    """Backward of y = act(norm(x) [+ residual]): (dx, dgamma, dbeta, dres).  dgamma / dbeta of parameters owned by optim.FusedAdam are
    added into its gradient arena on the parameter's side lane (returned as None)."""
    dgamma = dbeta = dgb = None
    ret_g = ret_b = None
    gacc = bacc = None
    rec = getattr(dy, "_sscg_bsums", None)
    from_sums = (rec is not None and rec[1] == dy._version and stats_grad and act in (ACT_NONE, ACT_RELU, ACT_LRELU)
                 and (y is not None) == bool(rec[0][2]))        # (the sums' mask source: the unit's output iff a residual joined an activated unit)
    if want_g:
        gacc = _acc_target(gref)
        bacc = _acc_target(betaref)
        fin = rec[0][3] if (from_sums and len(rec[0]) > 3) else None
        if fin is not None:     # dgamma / dbeta were written by the data gradient's call, behind the coefficients
            dgb = fin[fin.numel() - 2 * gamma.numel():].view(2, gamma.numel())
        else:
            dgb = torch.empty((2, gamma.numel()), dtype=torch.float32, device=gamma.device)     # written (not accumulated) by the kernel: no zero fill
        dgamma, dbeta = dgb[0], dgb[1]
        if gacc is None or bacc is None:
            gacc = bacc = None
            ret_g, ret_b = dgamma, dbeta
    if from_sums:
        dx, dres = norm_bwd_from_sums(rec[0], dy, x, mean, rstd, gamma, beta, per_sample, act, slope, dgamma, dbeta, y=y, want_dres=want_dres)
    else:
        dx, dres = norm_bwd(dy, x, y, mean, rstd, gamma, per_sample, act, slope, stats_grad,
                            want_dres=want_dres, dgamma=dgamma, dbeta=dbeta, beta=beta, overwrite=True)
    if gacc is not None:
        # The sums were produced beside dx on this stream; their accumulation into the optimiser's arena runs on the
        # parameter's own side lane, like every other gradient of that parameter (two forward lanes may both reach
        # this layer: a read-modify-write of the arena slice from two streams would lose updates).  weight and bias of a
        # norm layer are neighbours in the arena: one launch adds both.
        n_c = gacc.numel()
        adjacent = bacc.data_ptr() == gacc.data_ptr() + 4 * n_c

        def arena_grads():
            if adjacent:
                check(lib.sscg_add(gacc.data_ptr(), dgb.data_ptr(), gacc.data_ptr(), F32, 2 * n_c, _stream()), "sscg_add")
            else:
                check(lib.sscg_add(gacc.data_ptr(), dgamma.data_ptr(), gacc.data_ptr(), F32, n_c, _stream()), "sscg_add")
                check(lib.sscg_add(bacc.data_ptr(), dbeta.data_ptr(), bacc.data_ptr(), F32, n_c, _stream()), "sscg_add")
        run_on_side_stream(dy.device, (dgb,), arena_grads, lane=getattr(gref, "_sscg_lane", 0), defer=True)
    return dx, ret_g, ret_b, dres


class ConvNormActFn(torch.autograd.Function):
    """The reference's fusion unit as ONE autograd node: conv -> InstanceNorm / BatchNorm (batch statistics, from the conv's
    epilogue where the library can fuse them) [+ residual] -> activation (arch/ops.py:40-57; Bottleneck conv+bn pairs,
    arch/generators.py:345-365).  The same kernels as Conv2dFn + NormActFn; one node instead of two halves the host's per-layer
    autograd cost (575 such pairs per step).  cfg = (stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, residual, running_mean, running_var, cfg):
        stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope = cfg
        x = to_nhwc(x)
        if residual is not None:
            residual = to_nhwc(residual)
        n, _, h, wd = x.shape
        p, q = conv_out_size(h, w.shape[2], stride, pad, dil), conv_out_size(wd, w.shape[3], stride, pad, dil)
        g, l, c = _glc_shape((n, w.shape[0], p, q), per_sample)
        y, mean, rstd = conv2d_fwd_norm(x, w, bias, stride, pad, dil, pad_mode, False, (g, l, c), eps, running_mean, running_var, momentum)
        if mean is None:
            upd = running_mean is not None and per_sample is not True
            mean, rstd = norm_stats(y, per_sample, eps, running_mean if upd else None, running_var if upd else None, momentum)
        z = norm_apply(y, mean, rstd, gamma, beta, residual, per_sample, act, slope)
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.wref, ctx.bref, ctx.gref, ctx.betaref = w, bias, gamma, beta
        if _will_backward(ctx, 1):
            _note_use(w, bias)
        need_z = act != ACT_NONE and (residual is not None or act not in (ACT_RELU, ACT_LRELU))
        ctx.save_for_backward(x, w, y, z if need_z else None, mean, rstd, gamma, beta)
        ctx.res_join = getattr(residual, "_sscg_join", None) if (residual is not None and FUSE_JOIN[0]) else None
        # (only for a forward that will see a backward: the attribute keeps y, mean, rstd alive as long as z lives - the frozen
        # generators, evaluation and validation would hold every pre-norm activation for nothing)
        if (FUSE_BSUMS[0] and _will_backward(ctx) and act in (ACT_NONE, ACT_RELU, ACT_LRELU)
                and (residual is None or FUSE_JOIN[0])):
            # for the consumer's data gradient (_conv_backward): what this unit's backward reduction needs besides dz; last entry: the
            # mask cannot be recomputed from y (a residual joined before the activation) - it is read off z, the consumer's own input
            z._sscg_norm = (y, mean, rstd, gamma, beta, (g, l, c), act, slope, need_z)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, y, z, mean, rstd, gamma, beta = ctx.saved_tensors
        stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope = ctx.cfg
        ni = ctx.needs_input_grad
        dy, ret_g, ret_b, dres = _norm_backward(to_nhwc(dz), y, z, mean, rstd, gamma, beta, ctx.gref, ctx.betaref, per_sample, act, slope,
                                                True, gamma is not None and ni[3], ctx.has_res and ni[5])
        if ctx.res_join is not None and dres is not None:
            ctx.res_join[0].deposit(ctx.res_join[1], dres)      # the shortcut's gradient: the block's conv1 may add it in its data gradient
        dx, dw, db = _conv_backward(dy, x, w, ctx.wref, ctx.bref if ctx.has_bias else None, (stride, pad, dil, pad_mode),
                                    ni[0], ni[1], ctx.has_bias and ni[2])
        return dx, dw, db, ret_g, ret_b, dres, None, None, None


class ConvNormActHeadFn(torch.autograd.Function):
    """PixelDiscriminator from its second conv on (arch/discriminators.py:70-75): Conv2d(ndf, 2 ndf, 1x1) -> norm (batch statistics
    from the conv's epilogue) -> LeakyReLU -> Conv2d(2 ndf, 1, 1x1), ONE autograd node.  The 2 ndf-channel map is written once (by
    the conv) and read once per direction of the tail: the head's output is formed while normalising (sscg_norm_head_fwd), and the
    backward rebuilds dy = dout * w3 and the activation from x in registers (sscg_norm_head_bwd) - the normalised map, the head's
    input gradient and the head's weight-gradient pass over it never exist in HBM.
    cfg = (stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, hw, hbias, running_mean, running_var, cfg):
        stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope = cfg
        x = to_nhwc(x)
        n, _, h, wd = x.shape
        p, q = conv_out_size(h, w.shape[2], stride, pad, dil), conv_out_size(wd, w.shape[3], stride, pad, dil)
        g, l, c = _glc_shape((n, w.shape[0], p, q), per_sample)
        y, mean, rstd = conv2d_fwd_norm(x, w, bias, stride, pad, dil, pad_mode, False, (g, l, c), eps, running_mean, running_var, momentum)
        if mean is None:
            upd = running_mean is not None and per_sample is not True
            mean, rstd = norm_stats(y, per_sample, eps, running_mean if upd else None, running_var if upd else None, momentum)
        hw32 = hw.detach().reshape(-1)
        if hw32.dtype != torch.float32:
            hw32 = hw32.float()
        out = norm_head_fwd(y, mean, rstd, gamma, beta, hw32, hbias, per_sample, act, slope)
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.has_hbias = hbias is not None
        ctx.wref, ctx.bref, ctx.gref, ctx.betaref, ctx.hwref, ctx.hbref = w, bias, gamma, beta, hw, hbias
        if _will_backward(ctx, 1):
            _note_use(w, bias)
        ctx.save_for_backward(x, w, y, mean, rstd, gamma, beta, hw32)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, y, mean, rstd, gamma, beta, hw32 = ctx.saved_tensors
        stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope = ctx.cfg
        ni = ctx.needs_input_grad
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            dout = dout.float()
        c = hw32.numel()
        want_g = gamma is not None and ni[3]
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device) if want_g else None
        dwb = torch.empty(c + 1, dtype=torch.float32, device=y.device)
        dy = norm_head_bwd(dout, hw32, y, mean, rstd, gamma, beta, per_sample, act, slope, True, dwb,
                           dgb[0] if want_g else None, dgb[1] if want_g else None)
        # head weight / bias, norm weight / bias: sums produced beside dy on this stream; their accumulation into the optimiser's
        # arena runs on each parameter's own side lane (as _norm_backward)
        ret_hw = ret_hb = ret_g = ret_b = None
        want_hw, want_hb = ni[5], ctx.has_hbias and ni[6]
        hwacc = _acc_target(ctx.hwref) if want_hw else None
        hbacc = _acc_target(ctx.hbref) if want_hb else None
        if want_hw and hwacc is None:
            ret_hw = dwb[:c].reshape(ctx.hwref.shape).to(ctx.hwref.dtype)
        if want_hb and hbacc is None:
            ret_hb = dwb[c:c + 1].clone()
        gacc = _acc_target(ctx.gref) if want_g else None
        bacc = _acc_target(ctx.betaref) if want_g else None
        if want_g and (gacc is None or bacc is None):
            gacc = bacc = None
            ret_g, ret_b = dgb[0], dgb[1]

        def arena_add(acc, src_ptr, count, ref, keep):      # every gradient of a parameter is accumulated on that parameter's lane
            def go():
                check(lib.sscg_add(acc.data_ptr(), src_ptr, acc.data_ptr(), F32, count, _stream()), "sscg_add")
            run_on_side_stream(y.device, (keep,), go, lane=getattr(ref, "_sscg_lane", 0), defer=True)
        if hwacc is not None:
            arena_add(hwacc, dwb.data_ptr(), c, ctx.hwref, dwb)
        if hbacc is not None:
            arena_add(hbacc, dwb.data_ptr() + 4 * c, 1, ctx.hbref, dwb)
        if gacc is not None:
            arena_add(gacc, dgb.data_ptr(), c, ctx.gref, dgb)
            arena_add(bacc, dgb.data_ptr() + 4 * c, c, ctx.betaref, dgb)
        dx, dw, db = _conv_backward(dy, x, w, ctx.wref, ctx.bref if ctx.has_bias else None, (stride, pad, dil, pad_mode),
                                    ni[0], ni[1], ctx.has_bias and ni[2])
        return dx, dw, db, ret_g, ret_b, ret_hw, ret_hb, None, None, None


class PixelDiscFn(torch.autograd.Function):
    """The whole PixelDiscriminator (arch/discriminators.py:66-80) as one autograd node: Conv2d(cin, 64, 1x1) -> LeakyReLU ->
    Conv2d(64, 2 ndf, 1x1) [+ the norm layer's batch statistics] in ONE launch (sscg_conv2d_front_fwd: the 64-channel map is formed
    in LDS and written only for a backward pass), then the fused tail of ConvNormActHeadFn (norm -> LeakyReLU -> Conv2d(2 ndf, 1, 1x1)
    while normalising).  The backward is the chain of the separate nodes: head / norm, second conv, LeakyReLU mask, first conv.
    cfg = (slope1, per_sample, eps, momentum, act, slope)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w, bias, gamma, beta, hw, hbias, running_mean, running_var, cfg):
        slope1, per_sample, eps, momentum, act, slope = cfg
        x = to_nhwc(x)
        n, _, h, wd = x.shape
        g, l, c = _glc_shape((n, w.shape[0], h, wd), per_sample)
        will = _will_backward(ctx)
        y, h1, cs = conv2d_front_fwd(x, w1, b1, slope1, w, bias, (g, l, c), want_h1=will)
        upd = running_mean is not None and per_sample is not True
        mean, rstd = norm_stats_from_conv(cs, (g, l, c), eps, running_mean if upd else None, running_var if upd else None, momentum)
        hw32 = hw.detach().reshape(-1)
        if hw32.dtype != torch.float32:
            hw32 = hw32.float()
        out = norm_head_fwd(y, mean, rstd, gamma, beta, hw32, hbias, per_sample, act, slope)
        ctx.cfg = cfg
        ctx.has_b1, ctx.has_bias, ctx.has_hbias = b1 is not None, bias is not None, hbias is not None
        ctx.w1ref, ctx.b1ref, ctx.wref, ctx.bref, ctx.gref, ctx.betaref, ctx.hwref, ctx.hbref = w1, b1, w, bias, gamma, beta, hw, hbias
        if _will_backward(ctx, 1):
            _note_use(w1, b1)
        if _will_backward(ctx, 3):
            _note_use(w, bias)
        if will:
            ctx.save_for_backward(x, w1, h1, w, y, mean, rstd, gamma, beta, hw32)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, h1, w, y, mean, rstd, gamma, beta, hw32 = ctx.saved_tensors
        slope1, per_sample, eps, momentum, act, slope = ctx.cfg
        ni = ctx.needs_input_grad
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            dout = dout.float()
        c = hw32.numel()
        want_g = gamma is not None and ni[5]
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device) if want_g else None
        dwb = torch.empty(c + 1, dtype=torch.float32, device=y.device)
        dy = norm_head_bwd(dout, hw32, y, mean, rstd, gamma, beta, per_sample, act, slope, True, dwb,
                           dgb[0] if want_g else None, dgb[1] if want_g else None)
        ret_hw = ret_hb = ret_g = ret_b = None
        want_hw, want_hb = ni[7], ctx.has_hbias and ni[8]
        hwacc = _acc_target(ctx.hwref) if want_hw else None
        hbacc = _acc_target(ctx.hbref) if want_hb else None
        if want_hw and hwacc is None:
            ret_hw = dwb[:c].reshape(ctx.hwref.shape).to(ctx.hwref.dtype)
        if want_hb and hbacc is None:
            ret_hb = dwb[c:c + 1].clone()
        gacc = _acc_target(ctx.gref) if want_g else None
        bacc = _acc_target(ctx.betaref) if want_g else None
        if want_g and (gacc is None or bacc is None):
            gacc = bacc = None
            ret_g, ret_b = dgb[0], dgb[1]

        def arena_add(acc, src_ptr, count, ref, keep):      # every gradient of a parameter is accumulated on that parameter's lane
            def go():
                check(lib.sscg_add(acc.data_ptr(), src_ptr, acc.data_ptr(), F32, count, _stream()), "sscg_add")
            run_on_side_stream(y.device, (keep,), go, lane=getattr(ref, "_sscg_lane", 0), defer=True)
        if hwacc is not None:
            arena_add(hwacc, dwb.data_ptr(), c, ctx.hwref, dwb)
        if hbacc is not None:
            arena_add(hbacc, dwb.data_ptr() + 4 * c, 1, ctx.hbref, dwb)
        if gacc is not None:
            arena_add(gacc, dgb.data_ptr(), c, ctx.gref, dgb)
            arena_add(bacc, dgb.data_ptr() + 4 * c, c, ctx.betaref, dgb)
        geom = (1, 0, 1, PAD_ZEROS)
        front = ni[0] or ni[1] or (ctx.has_b1 and ni[2])
        dh1, dw, db = _conv_backward(dy, h1, w, ctx.wref, ctx.bref if ctx.has_bias else None, geom, front, ni[3], ctx.has_bias and ni[4])
        dx = dw1 = db1 = None
        if front:
            dh1 = act_bwd(dh1, h1, ACT_LRELU, slope1)
            dx, dw1, db1 = _conv_backward(dh1, x, w1, ctx.w1ref, ctx.b1ref if ctx.has_b1 else None, geom, ni[0], ni[1], ctx.has_b1 and ni[2])
        return dx, dw1, db1, dw, db, ret_g, ret_b, ret_hw, ret_hb, None, None, None


def pixel_disc(x, w1, b1, slope1, w, bias, gamma, beta, hw, hbias, running_mean, running_var, per_sample, eps, momentum, act, slope):
    _CALLER_GRAD[0] = torch.is_grad_enabled()
    return PixelDiscFn.apply(x, w1, b1, w, bias, gamma, beta, hw, hbias, running_mean, running_var,
                             (slope1, per_sample, eps, momentum, act, slope))


def conv_norm_act_head(x, w, bias, stride, pad, dil, pad_mode, gamma, beta, hw, hbias, running_mean, running_var, per_sample, eps,
                       momentum, act, slope):
    """conv -> norm (batch statistics) -> activation -> 1x1 conv to one channel, as one node (PixelDiscriminator's tail)."""
    _CALLER_GRAD[0] = torch.is_grad_enabled()
    return ConvNormActHeadFn.apply(x, w, bias, gamma, beta, hw, hbias, running_mean, running_var,
                                   (stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope))


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _need_hip(x)
        if not (x.is_contiguous() or x.is_contiguous(memory_format=CL)):
            x = x.contiguous()
        y = act_fwd(x, act, slope)
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, slope = ctx.cfg
        if dy.stride() != y.stride():
            dy = to_nhwc(dy) if y.is_contiguous(memory_format=CL) else dy.contiguous()
        return act_bwd(dy, y, act, slope), None, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = to_nhwc(a), to_nhwc(b)
        return add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


FUSE_JOIN = [os.environ.get("SSCG_FUSE_JOIN", "1") != "0"]       # a fan-out's gradient sum inside the last consumer's data gradient


class _Join(object):
    """The fan-out of one activation over n consumers (SplitFn), seen from their backward passes.  A consumer announces the gradient it
    produced (`deposit`); a later consumer whose data gradient can add a tensor in its store phase takes it (`take`) and marks the
    fan-in as done (`folded`): SplitFn.backward then has nothing left to add.  Only for n == 2, only on the stream the deposit was
    produced on (the engine orders the consumers of one pass; a deposit from another lane would be read unordered)."""
    __slots__ = ("n", "norm", "slots", "streams", "folded", "broken")

    def __init__(self, n, norm):
        self.n, self.norm = n, norm
        self.slots, self.streams = [None] * n, [None] * n
        self.folded, self.broken = None, False

    def take(self, i):
        if self.broken or self.n != 2 or self.folded is not None or self.slots[i] is not None:
            return None
        g = self.slots[1 - i]
        if g is None or self.streams[1 - i] != _stream():
            return None
        return g

    def deposit(self, i, g):
        if self.slots[i] is not None or self.folded is not None:
            self.broken = True          # an alias with two consumers: leave everything to SplitFn.backward
        self.slots[i], self.streams[i] = g, _stream()

    def clear(self):
        self.slots, self.streams = [None] * self.n, [None] * self.n
        self.folded, self.broken = None, False


class SplitFn(torch.autograd.Function):
    """Explicit fan-out of an activation that two (or more) consumers read - a residual block's input, the generated
    image that feeds a generator, a discriminator and the L1 loss.  Forward returns aliases; backward sums the
    consumers' gradients with the HIP add kernel in a fixed order, instead of leaving the accumulation to autograd's
    own (torch) add kernel - unless the last consumer's data gradient already added the other one (`_Join`)."""

    @staticmethod
    def forward(ctx, x, n, join=None):
        ctx.join = join
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        join = ctx.join
        if join is not None and join.folded is not None:
            i, t = join.folded
            g = grads[i]
            if join.broken or g is None or g.data_ptr() != t.data_ptr() or g.shape != t.shape:
                raise _lib.SscgError("fan-in bookkeeping: the folded gradient did not arrive unchanged")
            join.clear()
            return t, None, None        # (t carries the backward sums of the unit in front, if its data gradient took them)
        if join is not None:
            join.clear()
        total = None
        for g in grads:
            if g is None:
                continue
            total = to_nhwc(g) if total is None else add(total, to_nhwc(g))
        return total, None, None


class BatchSplitFn(torch.autograd.Function):
    """k batches stacked along N -> k tensors (views).  Backward stacks the k gradients again (device copies)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        ctx.shape = x.shape
        return tuple(t for t in x.chunk(k, 0))

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.shape[0] // ctx.k
        ref = next(g for g in grads if g is not None)
        out = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device).contiguous(memory_format=CL)
        for i, g in enumerate(grads):
            if g is None:
                fill_(out[i * n:(i + 1) * n], 0.0)
            else:
                out[i * n:(i + 1) * n].copy_(g)
        return out, None


def split_batch(x, k):
    x = to_nhwc(x)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x.chunk(k, 0)
    return BatchSplitFn.apply(x, k)


def split(x, n=2):
    """n aliases of x whose gradients are summed by `sscg_add` (no-op outside a gradient-recording forward).
    CONTRACT: every alias has exactly ONE consumer (that is what the aliases are for: a tensor read by k nodes is split k ways).  With
    n == 2 the second consumer's data gradient may add the first one's gradient in its own store phase (`_Join`); an alias that is read
    twice after all breaks that bookkeeping - `_Join.deposit` notices the second deposit and leaves the sum to SplitFn.backward
    (`broken`), and where the fold has already happened SplitFn.backward raises rather than return a silently wrong sum."""
    if n < 2 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    join = _Join(n, getattr(x, "_sscg_norm", None)) if (FUSE_JOIN[0] and n == 2) else None
    outs = SplitFn.apply(x, n, join)
    if join is not None:
        for i, o in enumerate(outs):
            o._sscg_join = (join, i)
    return outs


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(0.5) in training mode.  The keep-mask is a pure function of (seed, element index)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = to_nhwc(x)
        ctx.cfg = (p, seed)
        return dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        return dropout(to_nhwc(dy), p, seed), None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        y, idx = maxpool_fwd(x)
        ctx.xshape = tuple(x.shape)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return maxpool_bwd(to_nhwc(dy), idx, ctx.xshape)


class MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) (utils.Vgg16: torchvision VGG16 features[4, 9, 16])."""

    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h // 2, w // 2, x.device, x.dtype)
        idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=x.device)
        check(lib.sscg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), _dt(x), n, h, w, c, _stream()), "sscg_maxpool2x2_fwd")
        ctx.xshape = (n, c, h, w)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = ctx.xshape
        dx = empty_nhwc(n, c, h, w, dy.device, dy.dtype)
        check(lib.sscg_maxpool2x2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), _dt(dy), n, h, w, c, _stream()), "sscg_maxpool2x2_bwd")
        return dx


def maxpool2x2(x):
    return MaxPool2Fn.apply(x)


class UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        x = to_nhwc(x)
        ctx.hw = (x.shape[2], x.shape[3])
        return upsample_fwd(x, oh, ow)

    @staticmethod
    def backward(ctx, dy):
        return upsample_bwd(to_nhwc(dy), *ctx.hw), None, None


class SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = softmax_fwd(to_nhwc(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return softmax_bwd(to_nhwc(dy), y)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(logits [N,C,H,W], labels [N,H,W]) - mean over N*H*W."""

    @staticmethod
    def forward(ctx, logits, labels):
        _need_hip(logits, f32_only=True)
        logits = to_nhwc(logits)
        labels = labels.contiguous()
        n, c, h, w = logits.shape
        if labels.numel() != n * h * w or labels.dtype != torch.int64:
            raise _lib.SscgError("labels must be int64 with N*H*W elements")
        loss = _scalar(logits.device)
        valid = _scalar(logits.device)      # pixels with a label in [0, C): the divisor of the mean (all of them in the reference)
        ws = _loss_ws(logits.device)
        check(lib.sscg_ce_fwd(logits.data_ptr(), labels.data_ptr(), n * h * w, c, loss.data_ptr(), valid.data_ptr(), ws.data_ptr(),
                              ws.numel(), _stream()), "sscg_ce_fwd")
        ctx.save_for_backward(logits, labels, valid)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, valid = ctx.saved_tensors
        n, c, h, w = logits.shape
        dx = torch.empty_like(logits, memory_format=CL)
        check(lib.sscg_ce_bwd(logits.data_ptr(), labels.data_ptr(), n * h * w, c, g.data_ptr(), 1.0, valid.data_ptr(),
                              dx.data_ptr(), _stream()), "sscg_ce_bwd")
        return dx, None


FUSE_HEAD = [os.environ.get("SSCG_FUSE_HEAD", "1") != "0"]      # interp -> {softmax, cross entropy} without the resized logits in memory


def _head_applies(x, oh, ow):
    """One block per source pixel gathers the output pixels around it: worth it when the map grows (DeepLab's 33x33 -> the crop)."""
    return (FUSE_HEAD[0] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] <= 64
            and oh * ow >= 16 * x.shape[2] * x.shape[3])


class UpsampleHeadFn(torch.autograd.Function):
    """interp (bilinear, align_corners=True) -> softmax2d and / or nn.CrossEntropyLoss, from the LOW-resolution logits (model.py:390-392,
    398, 401-402, 455).  Returns (softmax map or None, loss or None).  The forward of the cross-entropy branch already leaves the
    gradient with respect to the low-resolution logits (it depends on logits and labels only); the backward scales it."""

    @staticmethod
    def forward(ctx, x, oh, ow, labels, want_soft):
        _need_hip(x, f32_only=True)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, oh, ow, x.device) if want_soft else None
        loss = valid = dl = ws = None
        if labels is not None:
            labels = labels.contiguous()
            if labels.numel() != n * oh * ow or labels.dtype != torch.int64:
                raise _lib.SscgError("labels must be int64 with N*OH*OW elements")
            loss, valid = _scalar(x.device), _scalar(x.device)
            dl = empty_nhwc(n, c, h, w, x.device)
            ws = torch.empty(lib.sscg_upsample_head_workspace(n, h, w), dtype=torch.uint8, device=x.device)
        check(lib.sscg_upsample_head_fwd(x.data_ptr(), _ptr(labels), _ptr(y), _ptr(loss), _ptr(valid), _ptr(dl), n, h, w, c, oh, ow,
                                         _ptr(ws), ws.numel() if ws is not None else 0, _stream()), "sscg_upsample_head_fwd")
        ctx.geom = (oh, ow)
        ctx.save_for_backward(x, dl, valid)
        # an output nothing differentiates (the step reads lab_gt's softmax through .detach() only): backward gets None, not a
        # zero-filled [B, C, crop] map to gather over
        ctx.set_materialize_grads(False)
        return y, loss

    @staticmethod
    def backward(ctx, dy, g):
        x, dl, valid = ctx.saved_tensors
        oh, ow = ctx.geom
        n, c, h, w = x.shape
        if dy is None and (g is None or dl is None):
            return None, None, None, None, None
        if dy is not None:
            dy = to_nhwc(dy)
        use_ce = g is not None and dl is not None
        dx = empty_nhwc(n, c, h, w, x.device)
        check(lib.sscg_upsample_head_bwd(x.data_ptr(), _ptr(dy), _ptr(dl if use_ce else None), _ptr(g if use_ce else None),
                                         _ptr(valid if use_ce else None), dx.data_ptr(), n, h, w, c, oh, ow, _stream()),
              "sscg_upsample_head_bwd")
        return dx, None, None, None, None


def upsample_softmax_ce(x, size, labels=None, want_soft=True):
    """(softmax2d(interp(x)) or None, CrossEntropyLoss(interp(x), labels) or None) - fused when the resize grows the map, else the
    three separate passes."""
    oh, ow = int(size[0]), int(size[1])
    if _head_applies(x, oh, ow):
        return UpsampleHeadFn.apply(x, oh, ow, labels, want_soft)
    up = upsample_bilinear(x, size)
    return (softmax2d(up) if want_soft else None), (cross_entropy(up, labels) if labels is not None else None)


class MSEConstFn(torch.autograd.Function):
    """nn.MSELoss()(x, full_like(x, target)) - the LSGAN terms."""

    @staticmethod
    def forward(ctx, x, target):
        _need_hip(x, f32_only=True)
        if not (x.is_contiguous() or x.is_contiguous(memory_format=CL)):
            x = x.contiguous()
        loss = _scalar(x.device)
        ws = _loss_ws(x.device)
        check(lib.sscg_mse_const_fwd(x.data_ptr(), x.numel(), target, loss.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
              "sscg_mse_const_fwd")
        ctx.target = target
        ctx.save_for_backward(x)
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x, memory_format=torch.preserve_format)
        check(lib.sscg_mse_const_bwd(x.data_ptr(), x.numel(), ctx.target, g.data_ptr(), 1.0, dx.data_ptr(), _stream()),
              "sscg_mse_const_bwd")
        return dx, None


class L1Fn(torch.autograd.Function):
    """nn.L1Loss()(a, b); gradient flows to `a` only (b is data in the reference, model.py:461)."""

    @staticmethod
    def forward(ctx, a, b):
        _need_hip(a, f32_only=True)
        _need_hip(b, f32_only=True)
        a, b = to_nhwc(a), to_nhwc(b)
        loss = _scalar(a.device)
        ws = _loss_ws(a.device)
        check(lib.sscg_l1_fwd(a.data_ptr(), b.data_ptr(), a.numel(), loss.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
              "sscg_l1_fwd")
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a, memory_format=CL)
        check(lib.sscg_l1_bwd(a.data_ptr(), b.data_ptr(), a.numel(), g.data_ptr(), 1.0, da.data_ptr(), _stream()), "sscg_l1_bwd")
        return da, None


class MseFn(torch.autograd.Function):
    """nn.MSELoss()(a, b) between two tensors (utils.perceptual_loss, utils.py:205-206); gradients to both."""

    @staticmethod
    def forward(ctx, a, b):
        _need_hip(a, f32_only=True)
        _need_hip(b, f32_only=True)
        a, b = to_nhwc(a), to_nhwc(b)
        loss = _scalar(a.device)
        ws = _loss_ws(a.device)
        check(lib.sscg_mse_fwd(a.data_ptr(), b.data_ptr(), a.numel(), loss.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "sscg_mse_fwd")
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a, memory_format=CL)
        db = torch.empty_like(b, memory_format=CL) if ctx.needs_input_grad[1] else None
        check(lib.sscg_mse_bwd(a.data_ptr(), b.data_ptr(), a.numel(), g.data_ptr(), 1.0, da.data_ptr(), _ptr(db), _stream()), "sscg_mse_bwd")
        return da, db


def mse_loss(a, b):
    return MseFn.apply(a, b)


class WeightedSumFn(torch.autograd.Function):
    """sum_i w_i * term_i over 0-dim device scalars (gen_loss, discriminator_loss)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        n = len(terms)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in terms])
        ws = (C.c_float * n)(*[float(w) for w in weights])
        out = _scalar(terms[0].device)
        check(lib.sscg_weighted_sum(ptrs, ws, n, out.data_ptr(), _stream()), "sscg_weighted_sum")
        ctx.weights = [float(w) for w in weights]
        return out

    @staticmethod
    def backward(ctx, g):
        grads = []
        for w in ctx.weights:
            gi = _scalar(g.device)
            ptrs = (C.c_void_p * 1)(g.data_ptr())
            ws = (C.c_float * 1)(w)
            check(lib.sscg_weighted_sum(ptrs, ws, 1, gi.data_ptr(), _stream()), "sscg_weighted_sum")
            grads.append(gi)
        return (None, *grads)


# functional spellings
def conv2d(x, w, bias=None, stride=1, pad=0, dil=1, pad_mode=PAD_ZEROS, act=ACT_NONE, slope=0.0, out_f32=True):
    """out_f32 matters in bf16 mode only: True keeps the result fp32 (network heads), False makes it a bf16 activation."""
    _CALLER_GRAD[0] = torch.is_grad_enabled()
    return Conv2dFn.apply(x, w, bias, stride, pad, dil, pad_mode, act, slope, out_f32, None)


def backward(loss):
    """loss.backward() with the autograd engine on the CALLING thread.  Every node of these graphs launches kernels and returns
    (nothing to run in parallel on the host), while the hop to the engine's device thread costs ~10 % of a step's issue time
    (79 -> 71 ms at 64x64, batch 2).  Backward nodes still run on the stream of their forward (the engine's stream guards do
    not depend on the thread)."""
    with torch.autograd.set_multithreading_enabled(False):
        loss.backward()
    # weight / bias / norm gradients are queued per side lane in batches (run_on_side_stream(defer=True)): launch what is still
    # pending, so that a stream synchronisation after this call covers every gradient kernel of the pass (the arenas are complete
    # once the side lanes are joined: SideStream.join / FusedAdam.step)
    flush_side_work()


def conv_norm_act(x, w, bias, stride, pad, dil, pad_mode, gamma, beta, residual, running_mean, running_var, per_sample, eps, momentum,
                  act=ACT_NONE, slope=0.0):
    """conv -> norm (batch statistics) [+ residual] -> activation as one autograd node (training-mode normalisation only)."""
    _CALLER_GRAD[0] = torch.is_grad_enabled()
    return ConvNormActFn.apply(x, w, bias, gamma, beta, residual, running_mean, running_var,
                               (stride, pad, dil, pad_mode, per_sample, eps, momentum, act, slope))


def conv2d_norm_stats(x, w, bias, stride, pad, dil, pad_mode, norm):
    """Convolution whose epilogue also produces the statistics of the normalisation layer `norm` = (per_sample, eps,
    running_mean, running_var, momentum) that follows it.  Returns (y, mean, rstd); mean/rstd None = not fused."""
    _CALLER_GRAD[0] = torch.is_grad_enabled()
    return Conv2dFn.apply(x, w, bias, stride, pad, dil, pad_mode, ACT_NONE, 0.0, False, norm)


def conv_transpose2d(x, w, bias=None, stride=1, pad=0, out_pad=0, act=ACT_NONE, slope=0.0, out_f32=False):
    return ConvTranspose2dFn.apply(x, w, bias, stride, pad, out_pad, act, slope, out_f32)


class CatChannelsFn(torch.autograd.Function):
    """torch.cat([a, b], 1) of two channels-last tensors - the skip connection of UnetSkipConnectionBlock.forward
    (arch/generators.py:44).  Device copies only (no arithmetic); the gradient is the two channel slices."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = to_nhwc(a), to_nhwc(b)
        if a.dtype != b.dtype or a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
            raise _lib.SscgError("cat_channels: tensors of one dtype, batch and spatial size expected")
        n, ca, h, w = a.shape
        ctx.ca = ca
        y = empty_nhwc(n, ca + b.shape[1], h, w, a.device, a.dtype)
        y[:, :ca].copy_(a)
        y[:, ca:].copy_(b)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = to_nhwc(dy)
        return dy[:, :ctx.ca].contiguous(memory_format=CL), dy[:, ctx.ca:].contiguous(memory_format=CL)


def cat_channels(a, b):
    return CatChannelsFn.apply(a, b)


def instance_norm_act(x, act=ACT_NONE, slope=0.0, residual=None, eps=1e-5, stats=None):
    m, r = stats if stats is not None else (None, None)
    return NormActFn.apply(x, None, None, residual, None, None, True, True, 0.0, eps, act, slope, m, r)


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.0,
                   residual=None, groups=1, stats=None):
    """groups > 1: x holds `groups` batches stacked along N; each is normalised with its own statistics and the running
    statistics advance once per group, in order - what `groups` successive calls would compute, in one launch.
    stats = (mean, rstd) already produced by the preceding conv's epilogue (training mode)."""
    per = False if groups == 1 else int(groups)
    m, r = stats if stats is not None else (None, None)
    return NormActFn.apply(x, gamma, beta, residual, running_mean, running_var, per, training, momentum, eps, act, slope, m, r)


def upsample_bilinear(x, size):
    """nn.Upsample(size, mode='bilinear', align_corners=True) (model.py:390-392, 413-415).  A resize to the size the map already has
    (the ResnetGenerator outputs: model.py:390, :413) is the identity under align_corners=True - scale 1, every weight 1 or 0 - and so is
    its adjoint: no pass at all (it was a copy forward and a 3x3-candidate gather backward, 128 us each at 8 x 3 x 256 x 256)."""
    if x.dim() == 4 and x.shape[2] == int(size[0]) and x.shape[3] == int(size[1]):
        return x
    return UpsampleFn.apply(x, int(size[0]), int(size[1]))


def softmax2d(x):
    return SoftmaxFn.apply(x)


def cross_entropy(logits, labels):
    return CrossEntropyFn.apply(logits, labels)


def mse_const(x, target):
    return MSEConstFn.apply(x, float(target))


def l1_loss(a, b):
    return L1Fn.apply(a, b)


def weighted_sum(terms, weights):
    return WeightedSumFn.apply(list(weights), *terms)


if os.environ.get("SSCG_RACECHECK"):     # debug: model the autograd engine's stream hand-over for the ordering checker
    import sys as _sys
    from ._lib import dev_tool as _dev_tool
    _dev_tool("racecheck").wrap_functions(_sys.modules[__name__])

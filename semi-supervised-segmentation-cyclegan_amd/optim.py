"""Flat-arena Adam for the two optimisers of the step (model.py:286-287: Adam(lr, betas=(0.5, 0.999)) over
chain(Gis, Gsi) and chain(Di, Ds)).

MI355X-first layout: all trainable parameters of an optimiser live in ONE contiguous fp32 arena, with
matching arenas for the gradient and both moments.  Consequences:
  * `step()` is a single fused HIP launch over ~86 M elements (HBM-bound, 7 streams);
  * `zero_grad()` is one fill;
  * the data-parallel exchange is an RCCL all-reduce of one buffer (parallel.py), not one per tensor;
  * weight-gradient kernels accumulate straight into the arena (`param._sscg_grad`), so autograd never
    materialises or adds per-parameter gradient tensors.
`state_dict()` keeps torch.optim.Adam's format (per-parameter `step`, `exp_avg`, `exp_avg_sq`) for the
parameters that have received a gradient, so checkpoints interchange with the reference."""
import weakref

import torch

from . import functional as F


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.world_size = 1
        self._steps = 0
        self._build()

    def _build(self):
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not ps:
            raise ValueError("FusedAdam: no trainable parameters")
        dev = ps[0].device
        if dev.type != "cuda":
            raise F._lib.SscgError("FusedAdam needs parameters on the MI355X (call net.cuda() / pass gpu_ids first)")
        # every parameter starts on a 256-byte boundary: the conv loaders and the fused split reductions move 16 bytes
        # per lane, and a slice at an odd offset would split each of those accesses
        total = sum(self._padded(p.numel()) for p in ps)
        self.arena = torch.empty(total, dtype=torch.float32, device=dev)
        F.fill_(self.arena, 0.0)
        self.grad = torch.empty(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.empty(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.empty(total, dtype=torch.float32, device=dev)
        for t in (self.grad, self.exp_avg, self.exp_avg_sq):
            F.fill_(t, 0.0)
        self.trainable = ps
        self.arena16 = None    # bf16 shadow of `arena` (bf16 mode): the conv operand copies, rewritten by the Adam kernel itself
        self._shadow_ver = {}
        self.arena_x3 = None   # three-plane bf16 split of `arena` (split mode, conv_split.hip): planes `arena.numel()` elements apart
        self._split_ver = {}
        self._epoch = [0]      # bumped by step(): the cached transposed copies of these weights are stale
        self.slices = {}
        off = 0
        for p in ps:
            n = p.numel()
            self.slices[p] = (off, n)
            self._view(self.arena, p, off).copy_(p.data)     # one-time gather (torch copy: start-up plumbing)
            p.data = self._view(self.arena, p, off)
            g = self._view(self.grad, p, off)
            p._sscg_grad = g
            p._sscg_touched = False
            p._sscg_epoch = self._epoch
            p._sscg_opt = weakref.ref(self)
            p._sscg_lane = len(self.slices)      # side-stream lane of this parameter's gradient kernels (all of them: they accumulate)
            p.grad = g
            off += self._padded(n)

    @staticmethod
    def _padded(n):
        return (n + 63) // 64 * 64

    @staticmethod
    def _view(flat, p, off):
        """View of flat[off : off+numel] with p's logical shape and (dense) strides."""
        return torch.as_strided(flat, p.shape, p.stride(), off)

    # ---- bf16 shadow arena (BASELINE configs 3/5): conv weights as bf16 operands, fp32 master copy in `arena`
    def refresh_shadow(self):
        """(Re)build the whole shadow from the fp32 arena: first use, after a broadcast, after a checkpoint load."""
        if self.arena16 is None:
            self.arena16 = torch.empty(self.arena.numel(), dtype=torch.bfloat16, device=self.arena.device)
        F.check(F.lib.sscg_cast(self.arena.data_ptr(), F.F32, self.arena16.data_ptr(), F.BF16, self.arena.numel(), F._stream()),
                "sscg_cast")
        self._shadow_ver = {p: p._version for p in self.trainable}

    def shadow_view(self, p):
        """bf16 view of parameter p inside the shadow arena (same logical shape / strides as p)."""
        if self.arena16 is None:
            self.refresh_shadow()
        elif self._shadow_ver.get(p) != p._version:     # torch wrote p (load_state_dict, init): re-cast that slice
            off, n = self.slices[p]
            F.check(F.lib.sscg_cast(self.arena[off:off + n].data_ptr(), F.F32, self.arena16[off:off + n].data_ptr(), F.BF16, n,
                                    F._stream()), "sscg_cast")
            self._shadow_ver[p] = p._version
        return self._view(self.arena16, p, self.slices[p][0])

    # ---- split shadow (fp32-accurate contractions on the bf16 matrix cores): h + m + l planes of every parameter
    def refresh_split(self):
        n = self.arena.numel()
        if self.arena_x3 is None:
            self.arena_x3 = torch.empty(3 * n, dtype=torch.bfloat16, device=self.arena.device)
        F.check(F.lib.sscg_split3(self.arena.data_ptr(), self.arena_x3.data_ptr(), n, n, F._stream()), "sscg_split3")
        self._split_ver = {p: p._version for p in self.trainable}

    def split_view(self, p):
        """(plane 0 of parameter p inside the split arena - a flat bf16 view -, plane stride in elements)."""
        if self.arena_x3 is None:
            self.refresh_split()
        elif self._split_ver.get(p) != p._version:      # torch wrote p (load_state_dict, init): re-split that slice
            off, n = self.slices[p]
            F.check(F.lib.sscg_split3(self.arena[off:off + n].data_ptr(), self.arena_x3[off:off + n].data_ptr(), n, self.arena.numel(),
                                      F._stream()), "sscg_split3")
            self._split_ver[p] = p._version
        off, n = self.slices[p]
        return self.arena_x3[off:off + n], self.arena.numel()

    def ensure_operand_copies(self):
        """Bring the operand copy the current arithmetic mode reads (bf16 shadow / split planes) up to date on the CURRENT stream.
        The step calls this on the main lane before it forks: left to the first conv that asks (`shadow_view` / `split_view`), the
        whole-arena pass runs on whichever lane gets there first - at a model's first step that was the fork lane, with the main
        lane reading the planes unsynchronised (racecheck.py: 103 read-after-write reports per first step)."""
        mode = F.get_conv_precision()
        if F._NO_COPY_SYNC:
            return
        if mode == "bf16":
            if self.arena16 is None:
                self.refresh_shadow()
            else:
                for p in self.trainable:
                    if self._shadow_ver.get(p) != p._version:
                        self.shadow_view(p)
        elif mode == "f32s":
            if self.arena_x3 is None:
                self.refresh_split()
            else:
                for p in self.trainable:
                    if self._split_ver.get(p) != p._version:
                        self.split_view(p)

    def zero_grad(self, set_to_none=False):
        F.fill_(self.grad, 0.0)

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self._steps += 1
        F.SideStream.join(self.arena.device)     # weight gradients are accumulated on the side stream
        # the operand copy the current arithmetic mode reads is rewritten by the Adam kernel; a copy left over from another mode is
        # dropped (it would go stale silently; it is rebuilt from the arena on its next use)
        mode = F.get_conv_precision()
        if mode != "bf16" and self.arena16 is not None:
            self.arena16, self._shadow_ver = None, {}
        if mode != "f32s" and self.arena_x3 is not None:
            self.arena_x3, self._split_ver = None, {}
        F.adam_step(self.arena, self.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                    self._steps, 1.0 / self.world_size, shadow_bf16=self.arena16, shadow_split=self.arena_x3)
        F.bump_weight_epoch(self._epoch)

    def mark_touched(self):
        """Parameters whose gradient slice is non-zero have taken part in a backward pass (host check, used lazily by state_dict)."""
        for p in self.trainable:
            if not p._sscg_touched:
                off, n = self.slices[p]
                p._sscg_touched = bool(self.grad[off:off + n].abs().max().item() > 0) or p._sscg_touched

    def state_dict(self):
        # materialise torch.optim.Adam-style per-parameter state (views into the arenas) for touched params
        self.mark_touched()
        self.state.clear()
        if self._steps > 0:
            for p in self.trainable:
                if p._sscg_touched:
                    off, _ = self.slices[p]
                    self.state[p] = {"step": torch.tensor(float(self._steps)), "exp_avg": self._view(self.exp_avg, p, off),
                                     "exp_avg_sq": self._view(self.exp_avg_sq, p, off)}
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = 0
        for p, st in list(self.state.items()):
            if p in self.slices:
                off, _ = self.slices[p]
                self._view(self.exp_avg, p, off).copy_(st["exp_avg"])
                self._view(self.exp_avg_sq, p, off).copy_(st["exp_avg_sq"])
                p._sscg_touched = True
                steps = max(steps, int(float(st["step"])))
        self._steps = steps

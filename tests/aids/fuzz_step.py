#!/usr/bin/env python
"""Schedule fuzzing: the multi-stream, D-overlapped step must produce the SAME BITS as the serial one-stream step whatever the
streams' relative timing.  Random spin kernels are queued in front of random launches (racecheck.fuzz), a further stream is kept
busy, and (from the caller's environment) GPU_MAX_HW_QUEUES may force streams onto shared hardware queues.
SSCG_FUZZ_DELAY_FORK=<shader clocks>: additionally start every overlapped run with the fork lane asleep (the targeted case
for a model's first step).
usage: [GPU_MAX_HW_QUEUES=2] [SSCG_SIDE_LANES=3] [SSCG_FORCE_DP=1] python tests/aids/fuzz_step.py [seeds] [steps] [size] [batch] [dtype]
Prints one line per schedule and exits 1 on the first difference."""
import contextlib
import importlib
import io
import os
import sys

os.environ.setdefault("SSCG_FUZZ", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME  # noqa: E402
from oracle import fixtures as FX  # noqa: E402

md = importlib.import_module(PKG_NAME + ".model")
F = importlib.import_module(PKG_NAME + ".functional")
rc = importlib.import_module(PKG_NAME + "._lib").dev_tool("racecheck")
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
size = int(sys.argv[3]) if len(sys.argv) > 3 else 64
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda", 0)
dp = None
if os.environ.get("SSCG_FORCE_DP"):
    dp = importlib.import_module(PKG_NAME + ".parallel").DataParallel()
F.set_conv_precision(dtype)
sds = FX.semisup_state_dicts(21, torch.float32, "pool")
batches = [tuple(t.to(dev) for t in FX.step_batch("pool", s % 4, 21, size, size, batch)) for s in range(4)]


def poison(gib=16):
    """Every large block the next model gets is carved from memory full of NaN: a launch that reads a buffer before its producer
    has run then cannot pass by accident (without this, a fresh model's operand copies land on the blocks the previous model's
    identical copies just left - a racing reader finds the right bits)."""
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    x = torch.full((gib << 28,), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    del x


def run(serial, seed, busy=True):
    """`steps` steps from the keyed weights; returns (losses per step as raw bits, bits of both parameter arenas)."""
    poison()
    args = FX.make_args(dataset="voc2012", crop_height=size, crop_width=size, batch_size=batch, gpu_ids=[0], checkpoint_dir="/tmp/sscg_fz", as_written=True)
    args.overlap_d = not serial
    args.fork_forward = not serial
    F.SideStream.enabled = not serial
    with contextlib.redirect_stdout(io.StringIO()):
        m = md.semisuper_cycleGAN(args, data_parallel=None if serial else dp)
    for k, sd in sds.items():
        getattr(m, k).load_state_dict(sd, strict=True)
    np.random.seed(0)
    torch.manual_seed(0)
    torch.cuda.synchronize()
    rc.fuzz(seed, busy=busy and seed is not None)
    if not serial and os.environ.get("SSCG_FUZZ_DELAY_FORK"):      # targeted: the fork lane starts this run late
        with torch.cuda.stream(F.ForkStream.get(dev)):
            torch.cuda._sleep(int(os.environ["SSCG_FUZZ_DELAY_FORK"]))
    outs = []
    for s in range(steps):
        outs.append(m.step(*batches[s % 4]))          # no host synchronisation between the steps
    rc.fuzz(None)
    m.sync_losses()
    F.flush_side_work()
    torch.cuda.synchronize()
    F.SideStream.enabled = True
    losses = torch.stack([torch.stack([o[k] for k in md.LOSS_KEYS]) for o in outs]).view(torch.int32).cpu()
    state = [m.g_optimizer.arena.view(torch.int32).clone(), m.d_optimizer.arena.view(torch.int32).clone()]
    bn = torch.cat([b.detach().float().reshape(-1) for net in (m.Gis, m.Gsi) for b in net.buffers() if b.dtype.is_floating_point]).view(torch.int32).clone()
    finite = bool(torch.isfinite(losses.view(torch.float32)).all())
    del m
    return losses, state + [bn], finite


ref = run(True, None)
print("serial schedule: finite=%s" % ref[2], flush=True)
bad = 0
for name, seed in [("overlapped, no fuzz", None)] + [("overlapped, fuzz seed %d" % i, i) for i in range(seeds)]:
    rc.FUZZ["sleeps"] = 0
    got = run(False, seed)
    same_l = bool((got[0] == ref[0]).all())
    same_w = [bool((a == b).all()) for a, b in zip(got[1], ref[1])]
    ok = same_l and all(same_w)
    bad += 0 if ok else 1
    first = ""
    if not same_l:
        d = (got[0] != ref[0]).nonzero()[0].tolist()
        first = " first differing loss: step %d %s" % (d[0], md.LOSS_KEYS[d[1]])
    print("%s: %d sleeps, losses %s, G arena %s, D arena %s, BN state %s, finite=%s%s" % (
        name, rc.FUZZ["sleeps"], "same" if same_l else "DIFFER", *("same" if x else "DIFFER" for x in same_w), got[2], first), flush=True)
print("side lanes: priority %s, %d lanes" % (F.SideStream.priority, F.SideStream.lanes))
print("%d of %d schedules differ from the serial one" % (bad, seeds + 1))
sys.exit(1 if bad else 0)

"""Host-side logic of the drop-in layer (no GPU): CLI flags, state-dict ABI, pool / LR schedule / mIoU helpers."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_sub
from oracle import fixtures as FX
from oracle import nets

META = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "meta.json")))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_cli_flags_and_defaults_match_reference():
    sys.path.insert(0, ROOT)
    import main
    a = main.get_args([])
    # defaults of /root/reference main.py:12-44
    want = dict(epochs=400, decay_epoch=100, batch_size=2, lr=.0002, gpu_ids="0", crop_height=None, crop_width=None, lamda_img=0.5,
                lamda_gt=0.1, lamda_perceptual=0, lab_CE_weight=1, lab_MSE_weight=1, lab_perceptual_weight=0, adversarial_weight=1.0,
                discriminator_weight=1.0, training=False, testing=False, validation=False, model="supervised_model",
                results_dir="./results", validation_dir="./val_results", checkpoint_dir="./checkpoints/semisupervised_cycleGAN",
                dataset="voc2012", norm="instance", no_dropout=False, ngf=64, ndf=64, gen_net="deeplab", dis_net="fc_disc")
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = main.get_args(["--training", "False", "--dataset", "cityscapes"])   # type=bool quirk: any non-empty string is True
    assert b.training is True
    assert main.DEFAULT_CROP == {"voc2012": (320, 320), "acdc": (256, 256), "cityscapes": (512, 1024)}


def test_state_dict_keys_are_the_reference_abi():
    arch = load_sub("arch")
    cases = [
        (lambda: arch.define_Gen(3, 21, 64, "deeplab", "instance", False, []), nets.deeplab_spec(3, 21)),
        (lambda: arch.define_Gen(21, 3, 64, "resnet_9blocks", "instance", True, []), nets.resnet_gen_spec_full(21, 3, 64, 9, "instance", True)),
        (lambda: arch.define_Gen(3, 21, 64, "resnet_9blocks_softmax", "batch", False, []), nets.resnet_gen_spec_full(3, 21, 64, 9, "batch", False)),
        (lambda: arch.define_Dis(21, 64, "pixel", 3, "instance", []), nets.pixel_dis_spec(21)),
        (lambda: arch.define_Dis(3, 64, "n_layers", 3, "instance", []), nets.nlayer_dis_spec(3)),
        (lambda: arch.define_Dis(3, 64, "n_layers", 3, "batch", []), nets.nlayer_dis_spec(3, norm="batch")),
    ]
    for mk, spec in cases:
        m = quiet(mk)
        sd = m.state_dict()
        assert list(sd.keys()) == list(spec.keys())
        for k in sd:
            assert tuple(sd[k].shape) == tuple(spec[k][0]), k
        # keyed weights round-trip through load_state_dict(strict=True); conv weights stay channels-last
        m.load_state_dict({k: torch.zeros(s[0], dtype=torch.int64 if s[1] == "nbt" else torch.float32) for k, s in spec.items()}, strict=True)
        for p in m.parameters():
            if p.dim() == 4:
                assert p.is_contiguous(memory_format=torch.channels_last)


def test_deeplab_trainable_set_and_frozen_bn():
    arch = load_sub("arch")
    m = quiet(arch.define_Gen, 3, 21, 64, "deeplab", "instance", False, [])
    train = [k for k, p in m.named_parameters() if p.requires_grad]
    assert len(train) == 104 + 8          # 104 conv weights + 4 classifier (weight, bias) pairs
    assert all(".bn" not in k and not k.startswith("bn1") and ".downsample.1" not in k for k in train)
    arch.set_grad([m], False)
    assert not any(p.requires_grad for p in m.parameters())


def test_out_of_scope_generators_raise():
    import pytest
    arch = load_sub("arch")
    for name in ("enet", "lednet_128", "lednet_256"):
        with pytest.raises(NotImplementedError):
            arch.define_Gen(3, 3, 64, name, "instance", False, [])
    with pytest.raises(NotImplementedError):
        arch.define_Dis(3, 64, "nope", 3, "instance", [])


def test_unet_state_dict_keys_follow_the_reference_nesting():
    """UnetGenerator behind define_Gen('unet_128' / 'unet_256'); the keys are the checkpoint ABI
    (oracle.nets.unet_spec is checked key-for-key against the reference's own module by tests/golden/gen_golden.py)."""
    from oracle import nets
    arch = load_sub("arch")
    for name, downs, norm in (("unet_128", 7, "instance"), ("unet_256", 8, "batch")):
        with contextlib.redirect_stdout(io.StringIO()):
            m = arch.define_Gen(3, 2, 8, name, norm, False, [])
        spec = nets.unet_spec(3, 2, downs, 8, norm)
        sd = m.state_dict()
        assert list(sd.keys()) == list(spec.keys())
        assert all(tuple(sd[k].shape) == tuple(spec[k][0]) for k in spec)


def test_pool_lambda_lr_running_score():
    u = load_sub("utils")
    np.random.seed(0)
    pool = u.Sample_from_Pool(max_elements=3)
    assert [float(pool([np.float32(i)])[0]) for i in range(12)] == META["pool_trace_seed0_cap3"]
    lr = u.LambdaLR(400, 0, 100)
    for e, v in META["lambda_lr"].items():
        assert abs(lr.step(int(e)) - v) < 1e-15
    from oracle import weights as W
    for ds, C in (("voc2012", 21), ("cityscapes", 20), ("acdc", 4)):
        lt = W.randint(FX.SEED, "g5/lt/" + ds, (2, 16, 16), C).numpy()
        lp = W.randint(FX.SEED, "g5/lp/" + ds, (2, 16, 16), C).numpy()
        lp[0] = lt[0]
        rs = u.runningScore(C, ds)
        rs.update(lt, lp)
        sc, _ = rs.get_scores()
        assert abs(sc["Mean IoU : \t"] - META["miou_" + ds]["miou"]) < 1e-12
        assert abs(sc["Overall Acc: \t"] - META["miou_" + ds]["acc"]) < 1e-12


def test_pool_output_size_rule():
    F = load_sub("functional")
    for h, o in META["maxpool_ceil_sizes"].items():
        assert F.pool_out_size(int(h)) == o


def test_synthetic_loader_contract():
    d = load_sub("data")
    ld = d.SyntheticLoader(2, 21, 40, 48, 3, seed=1)
    batches = list(ld)
    assert len(batches) == 3 == len(ld)
    img, gt, names = batches[0]
    assert img.shape == (2, 3, 40, 48) and img.dtype == torch.float32 and float(img.min()) >= -1 and float(img.max()) <= 1
    assert gt.shape == (2, 1, 40, 48) and gt.dtype == torch.int64 and int(gt.min()) >= 0 and int(gt.max()) < 21
    assert len(names) == 2
    again = list(d.SyntheticLoader(2, 21, 40, 48, 3, seed=1))[0]
    assert torch.equal(again[0], img) and torch.equal(again[1], gt)


@pytest.mark.skipif(not os.path.isdir("/root/reference/arch"), reason="needs the reference checkout (build container only)")
def test_checkpoints_interchange_with_the_reference_modules(tmp_path):
    """a checkpoint written by the REFERENCE (its own modules' state_dict through its utils.save_checkpoint
    format, model.py:646-655) loads into this build's networks with strict=True, and a checkpoint written by this build loads
    into the reference's modules - for the DeepLab generators and the pixel / PatchGAN discriminators.  Runs where
    /root/reference exists (the build container); no tensor arithmetic, so no GPU is needed."""
    import subprocess
    import sys
    code = r'''
import sys, torch, warnings
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import arch
torch.manual_seed(7)
nets = {"Gis": arch.define_Gen(21, 3, 64, "deeplab", "instance", True, []), "Gsi": arch.define_Gen(3, 21, 64, "deeplab", "instance", True, []),
        "Di": arch.define_Dis(3, 64, "pixel", 3, "instance", []), "Ds": arch.define_Dis(21, 64, "n_layers", 3, "instance", []),
        "old": arch.define_Gen(21, 3, 64, "resnet_9blocks", "instance", True, [])}
if sys.argv[1] == "save":
    torch.save({"epoch": 3, "best_iou": 0.5, **{k: v.state_dict() for k, v in nets.items()}}, sys.argv[2])
else:
    ck = torch.load(sys.argv[2], map_location="cpu")
    for k, v in nets.items():
        v.load_state_dict(ck[k], strict=True)
    print("reference loaded", sorted(ck))
'''
    ref_ck, our_ck = str(tmp_path / "ref.ckpt"), str(tmp_path / "ours.ckpt")
    subprocess.run([sys.executable, "-c", code, "save", ref_ck], check=True, capture_output=True)
    arch = load_sub("arch")
    utils = load_sub("utils")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        ours = {"Gis": arch.define_Gen(21, 3, 64, "deeplab", "instance", True, []), "Gsi": arch.define_Gen(3, 21, 64, "deeplab", "instance", True, []),
                "Di": arch.define_Dis(3, 64, "pixel", 3, "instance", []), "Ds": arch.define_Dis(21, 64, "n_layers", 3, "instance", []),
                "old": arch.define_Gen(21, 3, 64, "resnet_9blocks", "instance", True, [])}
        ck = utils.load_checkpoint(ref_ck)
    assert ck["epoch"] == 3
    for k, net in ours.items():
        net.load_state_dict(ck[k], strict=True)                      # the reference's keys and shapes, nothing missing or extra
        for name, t in net.state_dict().items():
            assert torch.equal(t.cpu().contiguous(), ck[k][name].contiguous()), (k, name)
    utils.save_checkpoint({"epoch": 4, "best_iou": 0.6, **{k: v.state_dict() for k, v in ours.items()}}, our_ck)
    out = subprocess.run([sys.executable, "-c", code, "load", our_ck], check=True, capture_output=True, text=True).stdout
    assert "reference loaded" in out


def test_bench_cpu_baseline_leg_runs_and_reports(monkeypatch):
    """bench.py's cpu_baseline leg end to end on a tiny geometry (the driver's default bench run executes it at 256x256)."""
    sys.path.insert(0, ROOT)
    import bench
    threads = torch.get_num_threads()
    try:
        r = bench.cpu_baseline({"dataset": "voc2012", "C": 21, "H": 32, "W": 32})
    finally:
        torch.set_num_threads(threads)
    assert r["kind"] == "port" and r["unit"] == "img/s" and r["value"] > 0 and r["cores"] >= 1
    assert "32x32" in r["sample"] and "3 timed steps at the fastest setting" in r["sample"] and r["cpu_model"] in r["sample"]
    assert r["timed_steps"] >= 3 and r["warmup_steps"] == 1 and r["all_physical_cores"]["cores"] == r["physical_cores"]
    assert r["elided_dead_work"]["value"] > 0 and str(r["cores"]) in r["probe_seconds_per_step"]
    json.dumps(r)
    # on a stub clock: one warm-up step per thread count (32, 64, all 128 physical cores), three timed steps on all cores (9 s each:
    # inside the budget), three at the fastest setting (64), one elided step
    cost = {32: 10.0, 64: 6.0, 128: 9.0}
    clock = [0.0]
    calls = []

    def step(threads, as_written=True):
        calls.append(threads)
        clock[0] += cost[threads] * (1.0 if as_written else 0.9)
    monkeypatch.setattr(bench, "cpu_thread_candidates", lambda physical: [32, 64, 128])
    r = bench.cpu_baseline({"dataset": "voc2012", "C": 21, "H": 256, "W": 256}, step_fn=step, now=lambda: clock[0])
    assert calls == [32, 64, 128, 128, 128, 128, 64, 64, 64, 64] and r["cores"] == 64 and abs(r["seconds_per_step"] - 6.0) < 1e-9
    assert abs(r["value"] - 2 / 6.0) < 1e-4 and r["timed_steps"] == 3
    assert r["all_physical_cores"] == {"cores": 128, "value": round(2 / 9.0, 4), "seconds_per_step": 9.0, "timed_steps": 3, "warmup_steps": 1}
    # an over-subscribed all-cores step (52 s: BENCH_r02) is timed once more, not three times
    cost[128] = 52.0
    calls.clear()
    r = bench.cpu_baseline({"dataset": "voc2012", "C": 21, "H": 256, "W": 256}, step_fn=step, now=lambda: clock[0])
    assert calls == [32, 64, 128, 128, 64, 64, 64, 64] and r["cores"] == 64 and r["all_physical_cores"]["timed_steps"] == 1
    assert abs(r["all_physical_cores"]["seconds_per_step"] - 52.0) < 1e-9
    # all cores fastest: its three timed steps are the headline's
    cost[128] = 5.0
    calls.clear()
    r = bench.cpu_baseline({"dataset": "voc2012", "C": 21, "H": 256, "W": 256}, step_fn=step, now=lambda: clock[0])
    assert calls == [32, 64, 128, 128, 128, 128, 128] and r["cores"] == 128 and r["timed_steps"] == 3


def test_plan_sizes_follow_the_descriptor_tuning():
    """Tile-class / split overrides travel in sscg_conv_desc.tuning (no process-wide hook): a descriptor built under a forced
    tile class is a different descriptor with its own cached workspace size, and the library answers per descriptor."""
    import ctypes as C
    F = load_sub("functional")
    L = load_sub("_lib")
    mk = lambda: F.make_desc((16, 256, 33, 65), (256, 256, 3, 3), 1, 2, 2, xdt=L.BF16, wdt=L.BF16, ydt=L.BF16)
    d = mk()
    a = F._ws_bytes(d, "fwd")
    assert a == L.lib.sscg_conv2d_fwd_workspace(C.byref(d)) and F._ws_bytes(d, "fwd") == a
    old = F.tuning(tile_class=1)                    # 64x64 tiles: a different tail split, a different workspace
    try:
        d2 = mk()
        assert d2 is not d and d2.tuning == 2 and d.tuning == 0
        b = F._ws_bytes(d2, "fwd")
        assert b == L.lib.sscg_conv2d_fwd_workspace(C.byref(d2)) and b != a
    finally:
        F.TUNING[0], F.WGRAD_TUNING[0] = old
    assert mk() is d and F._ws_bytes(d, "fwd") == a
    assert not hasattr(L.lib, "sscg_debug_set_conv_cfg")


def test_bench_spawns_its_own_ranks_when_no_launcher_is_present(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE re-executes itself under torch.distributed.run (one rank per GPU, 127.0.0.1
    rendezvous, the caller's flags passed through); on a box with fewer GPUs the ranks share GPU 0 over gloo."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    import types

    def fake_run(cmd, env=None, **kw):
        seen.setdefault("cmds", []).append(cmd)
        seen["cmd"], seen["env"] = cmd, env
        # the first rendezvous port "was taken meanwhile": the launcher dies with EADDRINUSE and spawn_ranks tries another port
        if len(seen["cmds"]) == 1:
            return types.SimpleNamespace(returncode=1, stderr="RuntimeError: The server socket has failed to listen ... EADDRINUSE\n")
        return types.SimpleNamespace(returncode=0, stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--config", "3"])
    assert bench.spawn_ranks(4) == 0
    assert len(seen["cmds"]) == 2
    p0, p1 = (c[c.index("--master-port") + 1] for c in seen["cmds"])
    assert p0.isdigit() and p1.isdigit()
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--config", "3"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if torch.cuda.device_count() < 4:
        assert seen["env"]["SSCG_DP_SHARED_GPU"] == "1" and seen["env"]["SSCG_DP_BACKEND"] == "gloo"


def test_bench_line_is_the_last_line_on_stdout(tmp_path):
    """Under a process group the collective library prints through C stdio (RCCL's version banner): on a pipe that text left libc's
    buffer at exit(), BEHIND the JSON line (profiles/r06_experiments.txt item 23).  Two gloo ranks share one stdout pipe here, each
    with a banner in libc's buffer: the line is the last thing rank 0 or anyone else writes."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "stdout.txt"
    procs = []
    with open(out, "wb") as f:
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "aids", "emit_line_ranks.py")], stdout=f, env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
    lines = out.read_text().strip().splitlines()
    assert json.loads(lines[-1]) == {"metric": "m", "value": 1.0, "rank_count": 2}, lines
    assert sum(ln.startswith("{") for ln in lines) == 1
    # what was buffered BEFORE the line came out before it - on both ranks; rank 0's own later text would follow (there is none in bench.py)
    assert any(ln.startswith("BANNER of rank 0") for ln in lines[:-1]) and any(ln.startswith("BANNER of rank 1") for ln in lines[:-1])
    assert not any(ln.startswith("LATE text of rank 1") for ln in lines)


def test_pixel_discriminator_tail_is_recognised_and_lane_priority_follows_the_context(monkeypatch):
    """Host decisions of round 3 (no GPU): (1) `FusedSequential` hands the PixelDiscriminator's second conv -> norm -> LeakyReLU ->
    Conv2d(2 ndf, 1, 1x1) to the fused-tail node exactly when the library serves the width (a power of two in [16, 256]) and the norm
    uses batch statistics; a PatchGAN tail (4x4 head) is not matched.  (2) Side lanes are low priority only for a single process
    with >= 128 K pixels per batch; an environment override always wins."""
    ops = load_sub("arch.ops")
    disc = load_sub("arch.discriminators")
    F = load_sub("functional")
    for ndf, want in ((64, True), (8, True), (4, False), (24, False), (256, False)):      # 2 ndf = 128, 16, 8, 48, 512
        net = disc.PixelDiscriminator(3, ndf, norm_layer=ops.get_norm_layer("instance"), use_bias=True)
        m = list(net.dis_model)
        assert ops._is_pixel_head(m[2], m[3], m[4], m[5]) is want, ndf
        assert bool(F.norm_head_applies(2 * ndf)) is want
    bn = disc.PixelDiscriminator(3, 64, norm_layer=ops.get_norm_layer("batch"))
    m = list(bn.dis_model)
    assert ops._is_pixel_head(m[2], m[3], m[4], m[5])
    bn.eval()                                                                              # running statistics: the unfused path
    assert not ops._is_pixel_head(m[2], m[3], m[4], m[5])
    patch = list(disc.NLayerDiscriminator(3, 16, 2, norm_layer=ops.get_norm_layer("instance"), use_bias=True).dis_model)
    assert not any(isinstance(a, ops.Conv2d) and ops._is_norm(b) and ops._is_pixel_head(a, b, c, d)
                   for a, b, c, d in zip(patch, patch[1:], patch[2:], patch[3:]) if isinstance(c, ops._Act))
    monkeypatch.delenv("SSCG_SIDE_PRIORITY", raising=False)
    assert F.side_priority() == 1 and F.side_priority_for(8 * 256 * 256) == 1 and F.side_priority_for(2 * 64 * 64) == 0
    monkeypatch.setenv("SSCG_SIDE_PRIORITY", "0")
    assert F.side_priority() == 0 and F.side_priority_for(8 * 256 * 256) == 0
    monkeypatch.setenv("SSCG_SIDE_PRIORITY", "1")
    assert F.side_priority_for(2 * 64 * 64) == 1


def test_fan_in_bookkeeping_never_counts_a_gradient_twice(monkeypatch):
    """functional._Join / SplitFn.backward (the residual fan-in that joins in a data gradient's store phase): the second consumer takes
    the first one's gradient only on the stream it was produced on and only for a two-way fan-out; a gradient that was folded is
    returned alone, one that was not is summed; an alias with two consumers, or a folded gradient that arrives changed, is refused
    loudly rather than counted twice.  (Host logic only: the tensors stay on the CPU, the add is patched.)"""
    F = load_sub("functional")
    lane = [7]
    monkeypatch.setattr(F, "_stream", lambda: lane[0])
    monkeypatch.setattr(F, "to_nhwc", lambda t: t)
    monkeypatch.setattr(F, "add", lambda a, b: a + b)

    class Ctx(object):
        pass

    g0, g1 = torch.ones(2, 3), torch.full((2, 3), 2.0)
    j = F._Join(2, None)
    assert j.take(0) is None                      # nothing deposited yet
    j.deposit(1, g1)
    lane[0] = 8
    assert j.take(0) is None                      # deposited on another lane: would be read unordered
    lane[0] = 7
    assert j.take(0) is g1
    total = g0 + g1                               # what the joined data gradient returns
    j.folded = (0, total)
    ctx = Ctx(); ctx.join = j
    out = F.SplitFn.backward(ctx, total, g1)
    assert out[0] is total and out[1] is None and j.folded is None and j.slots == [None, None]      # nothing added again; state cleared
    # not folded: the plain sum, in consumer order
    j.deposit(0, g0); j.deposit(1, g1)
    out = F.SplitFn.backward(ctx, g0, g1)
    assert torch.equal(out[0], g0 + g1) and j.slots == [None, None]
    # three consumers: never folded
    j3 = F._Join(3, None)
    j3.deposit(1, g1)
    assert j3.take(0) is None
    # an alias with two consumers deposits twice: nothing may be taken any more
    j.deposit(1, g1); j.deposit(1, g1)
    assert j.broken and j.take(0) is None
    j.clear()
    # a folded gradient that arrives changed (e.g. accumulated by the engine) is an error, not a silent double count
    j.deposit(1, g1)
    j.folded = (0, total)
    with pytest.raises(Exception):
        F.SplitFn.backward(ctx, total.clone(), g1)

import sys, os, time, gc, contextlib, io, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
PKG = bench.PKG
md = importlib.import_module(PKG + ".model"); F = importlib.import_module(PKG + ".functional"); data = importlib.import_module(PKG + ".data")
import main as cli
def mk(h, w, b, ds="voc2012"):
    args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", ds, "--crop_height", str(h), "--crop_width", str(w), "--batch_size", str(b), "--checkpoint_dir", "/tmp/x"])
    args.gpu_ids, args.as_written, args.overlap_d = [0], True, True
    with contextlib.redirect_stdout(io.StringIO()):
        return md.semisuper_cycleGAN(args)
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
def probe(tag, C=21):
    small = mk(64, 64, 2)
    sl = list(data.SyntheticLoader(2, C, 64, 64, 24, 3, device=dev)); su = list(data.SyntheticLoader(2, C, 64, 64, 24, 4, device=dev))
    for i in range(4): small.step(sl[i][0], sl[i][1], su[i][0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(4, 24): small.step(sl[i][0], sl[i][1], su[i][0])
    th = (time.perf_counter() - t0) / 20; torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 20
    print(tag, "issue %.2f total %.2f ms" % (th * 1e3, ts * 1e3), flush=True)
for mode in ("f32x", "f32s"):
    F.set_conv_precision(mode); probe("fresh " + mode)
big = mk(256, 256, 2)
bl = list(data.SyntheticLoader(2, 21, 256, 256, 6, 1, device=dev)); bu = list(data.SyntheticLoader(2, 21, 256, 256, 6, 2, device=dev))
for i in range(6): big.step(bl[i][0], bl[i][1], bu[i][0])
torch.cuda.synchronize()
probe("after big f32s")
for mode in ("bf16", "f32x", "f32s"):
    F.set_conv_precision(mode)
    for i in range(3): big.step(bl[i][0], bl[i][1], bu[i][0])
torch.cuda.synchronize()
probe("after modes f32s")
gc.collect(); gc.freeze(); probe("after gc.freeze"); gc.disable(); probe("gc disabled")

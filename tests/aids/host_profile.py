"""cProfile of the host side of a step at a tiny geometry (64x64, batch 2: the step is purely launch-bound there)."""
import contextlib, cProfile, importlib, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import fixtures as FX
md = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.model")
data = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.data")
dev = torch.device("cuda:0")
args = FX.make_args(dataset="voc2012", crop_height=64, crop_width=64, batch_size=2, gpu_ids=[0], no_dropout=False, checkpoint_dir="/tmp/sscg_hp", as_written=True)
args.overlap_d = True
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
lab = list(data.SyntheticLoader(2, 21, 64, 64, 8, 1, device=dev))
unl = list(data.SyntheticLoader(2, 21, 64, 64, 8, 2, device=dev))
for i in range(2):
    m.step(lab[i][0], lab[i][1], unl[i][0])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(2, 6):
    m.step(lab[i][0], lab[i][1], unl[i][0])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

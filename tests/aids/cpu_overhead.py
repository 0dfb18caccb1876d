"""How long does the host need to *issue* one step (launch-side cost) vs the GPU to execute it?  (tuning aid)"""
import contextlib, importlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import fixtures as FX
md = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.model")
data = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.data")
dev = torch.device("cuda:0")
args = FX.make_args(dataset="voc2012", crop_height=int(os.environ.get("HH", "256")), crop_width=int(os.environ.get("HH", "256")), batch_size=int(os.environ.get("BB", "8")), gpu_ids=[0], no_dropout=False,
                    checkpoint_dir="/tmp/sscg_cpuo", as_written=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
lab = list(data.SyntheticLoader(int(os.environ.get("BB", "8")), 21, int(os.environ.get("HH", "256")), int(os.environ.get("HH", "256")), 6, 1, device=dev))
unl = list(data.SyntheticLoader(int(os.environ.get("BB", "8")), 21, int(os.environ.get("HH", "256")), int(os.environ.get("HH", "256")), 6, 2, device=dev))
for i in range(2):
    m.step(lab[i][0], lab[i][1], unl[i][0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2, 6):
    m.step(lab[i][0], lab[i][1], unl[i][0])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host issue time per step: %.1f ms; wall per step: %.1f ms" % (1e3 * t_issue / 4, 1e3 * t_all / 4))

#!/usr/bin/env python
"""First-step losses of the CPU oracle (oracle/step.py: pinned bit-exact against the reference's semisuper_cycleGAN.train by
gen_golden.py) in fp32 and fp64 for a set of keyed weight / input seeds -> tests/golden/g7_first_steps.json.

  ch0 .. ch5   VOC 21 classes, 64x64, batch 2: six independent keyed weight sets.  The GPU test runs the first step from each in
               both fp32 arithmetics of the build (exact fp32 MFMA, split contraction) and compares their distances to fp64
               (tests/test_nets_gpu.py::test_split_mode_is_as_close_to_fp64_as_exact_fp32_at_step_level);
  ds_cityscapes / ds_acdc   (+ the bf16-emulated fp64 step for ds_cityscapes: tests/test_bf16_gpu.py) the two first-step configurations of tests/test_step_gpu.py (were run live on the GPU box's host: 55 s each).

The oracle needs no reference checkout: this script runs anywhere (minutes of CPU); its output is data."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fixtures as FX  # noqa: E402
from oracle import step as ostep  # noqa: E402

CASES = [("ch%d" % i, "voc2012", 21, 64, 64) for i in range(6)] + [("ds_cityscapes", "cityscapes", 20, 64, 128), ("ds_acdc", "acdc", 4, 64, 64)]
ONLY_SKIP = set(sys.argv[1:])
PATH = os.path.join(ROOT, "tests", "golden", "g7_first_steps.json")
# `gen_first_steps.py only:<tag>[,<tag>]` recomputes the named entries and keeps the rest of the committed file (the oracle is
# deterministic: a full run reproduces them; the six-seed sweep is minutes of CPU)
ONLY = set(t for a in sys.argv[1:] if a.startswith("only:") for t in a[5:].split(","))
out = json.load(open(PATH)) if ONLY else {}
for tag, dataset, C, H, Wd in CASES:
    if ONLY and tag not in ONLY:
        continue
    l_img, l_gt, unl_img = FX.step_batch(tag, 0, C, H, Wd, 2)
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        np.random.seed(0)
        o = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, dt, tag), crop=(H, Wd))
        res[name] = {k: float(v) for k, v in o.step(l_img.to(dt), l_gt, unl_img.to(dt)).items()}
    out[tag] = {"dataset": dataset, "C": C, "H": H, "W": Wd, "B": 2, "oracle_f32": res["f32"], "oracle_f64": res["f64"]}
    if tag == "ds_cityscapes" and "emu" not in ONLY_SKIP:      # the yardstick of the bf16 first-step test: the fp64 step with every
        from oracle import nets as onets                        # tensor the build keeps in bf16 rounded at the same place
        np.random.seed(0)
        o = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, torch.float64, tag), crop=(H, Wd), q=onets.Bf16Emulation)
        out[tag]["oracle_f64_bf16_emulation"] = {k: float(v) for k, v in o.step(l_img.double(), l_gt, unl_img.double()).items()}
    print(tag, {k: "%.3e" % (abs(res["f32"][k] - res["f64"][k]) / abs(res["f64"][k])) for k in ostep.LOSS_KEYS}, flush=True)
# the step with the commented-out loss terms switched on (the build's --variants l1_cycle,lab_gt_dis): losses incl. the two extra
# terms, and the L2 norm of the generators' gradient (what the extra terms change besides their own value)
tag, C, H, Wd = "var", 21, 64, 64
l_img, l_gt, unl_img = FX.step_batch(tag, 0, C, H, Wd, 2)
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)) if (not ONLY or tag in ONLY) else ():
    np.random.seed(0)
    o = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, dt, tag), crop=(H, Wd), variants=("l1_cycle", "lab_gt_dis"), lamda_img=0.5)
    col = {}
    res[name] = {k: float(v) for k, v in o.step(l_img.to(dt), l_gt, unl_img.to(dt), collect=col).items()}
    res[name + "_gnorm"] = float(torch.sqrt(sum((g.double() ** 2).sum() for g in col["g_grads"] if g is not None)))
if res:
    out[tag] = {"dataset": "voc2012", "C": C, "H": H, "W": Wd, "B": 2, "variants": "l1_cycle,lab_gt_dis", "oracle_f32": res["f32"], "oracle_f64": res["f64"],
                "g_grad_norm_f32": res["f32_gnorm"], "g_grad_norm_f64": res["f64_gnorm"]}
    print(tag, out[tag]["g_grad_norm_f32"], out[tag]["g_grad_norm_f64"], flush=True)
# the step on the networks --gen_net / --dis_net name (the build's --honour_nets; the reference parses the flags, main.py:43-44, and
# never reads them): ResNet-9 generators (arch/generators.py:404-418) + PatchGAN discriminators (arch/discriminators.py:42-63) as
# the TRAINED nets, no dropout; first-step losses and both gradient norms
tag, C, H, Wd = "hn", 21, 64, 64
l_img, l_gt, unl_img = FX.step_batch(tag, 0, C, H, Wd, 2)
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)) if (not ONLY or tag in ONLY) else ():
    np.random.seed(0)
    o = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, dt, tag, "resnet_9blocks", "n_layers"), crop=(H, Wd), gen_net="resnet_9blocks", dis_net="n_layers")
    col = {}
    res[name] = {k: float(v) for k, v in o.step(l_img.to(dt), l_gt, unl_img.to(dt), collect=col).items()}
    res[name + "_gnorm"] = float(torch.sqrt(sum((g.double() ** 2).sum() for g in col["g_grads"] if g is not None)))
    res[name + "_dnorm"] = float(torch.sqrt(sum((g.double() ** 2).sum() for g in col["d_grads"] if g is not None)))
if res:
    out[tag] = {"dataset": "voc2012", "C": C, "H": H, "W": Wd, "B": 2, "gen_net": "resnet_9blocks", "dis_net": "n_layers",
                "oracle_f32": res["f32"], "oracle_f64": res["f64"], "g_grad_norm_f32": res["f32_gnorm"], "g_grad_norm_f64": res["f64_gnorm"],
                "d_grad_norm_f32": res["f32_dnorm"], "d_grad_norm_f64": res["f64_dnorm"]}
    print(tag, {k: "%.3e" % (abs(res["f32"][k] - res["f64"][k]) / abs(res["f64"][k])) for k in ostep.LOSS_KEYS}, flush=True)
with open(PATH, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)

#!/usr/bin/env python
"""Command line of the reference (main.py:10-87) driving the MI355X build: same flags, same defaults, same
dispatch on --training / --model.  Extra flags (never change a reference default): --synthetic_steps,
--as_written.  Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N main.py ...`
(one process per MI355X; gradients all-reduced with RCCL)."""
import importlib
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PKG = "semi-supervised-segmentation-cyclegan_amd"

# (flag, type, default) - verbatim from the reference, including its quirks: `type=bool` flags are true for
# ANY non-empty string and the loss weights are type=int with float defaults
FLAGS = [
    ("epochs", int, 400), ("decay_epoch", int, 100), ("batch_size", int, 2), ("lr", float, .0002), ("gpu_ids", str, "0"),
    ("crop_height", int, None), ("crop_width", int, None), ("lamda_img", int, 0.5), ("lamda_gt", int, 0.1),
    ("lamda_perceptual", int, 0), ("lab_CE_weight", int, 1), ("lab_MSE_weight", int, 1), ("lab_perceptual_weight", int, 0),
    ("adversarial_weight", int, 1.0), ("discriminator_weight", int, 1.0), ("training", bool, False), ("testing", bool, False),
    ("validation", bool, False), ("model", str, "supervised_model"), ("results_dir", str, "./results"),
    ("validation_dir", str, "./val_results"), ("checkpoint_dir", str, "./checkpoints/semisupervised_cycleGAN"),
    ("ngf", int, 64), ("ndf", int, 64), ("gen_net", str, "deeplab"), ("dis_net", str, "fc_disc"),
]
DEFAULT_CROP = {"voc2012": (320, 320), "acdc": (256, 256), "cityscapes": (512, 1024)}   # main.py:60-67


DATA_ROOTS = {'voc2012': './data/VOC2012', 'cityscapes': './data/Cityscape', 'acdc': './data/ACDC'}   # model.py:21-23


def get_args(argv=None):
    parser = ArgumentParser(description="cycleGAN PyTorch (MI355X-native build)")
    for name, typ, default in FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    parser.add_argument("--dataset", type=str, choices=["voc2012", "cityscapes", "acdc"], default="voc2012")
    parser.add_argument("--norm", type=str, default="instance", help="instance normalization or batch normalization")
    parser.add_argument("--no_dropout", action="store_true", help="no dropout for the generator")
    # build-only additions
    parser.add_argument("--synthetic_steps", type=int, default=8, help="iterations per epoch of the synthetic loaders")
    parser.add_argument("--as_written", type=int, default=1, help="1: also run the forwards whose outputs the reference never uses")
    parser.add_argument("--data", type=str, choices=["auto", "real", "synthetic"], default="auto",
                        help="real: the datasets under ./data (reference layout); synthetic: seeded random batches; auto: real if present")
    parser.add_argument("--dtype", type=str, choices=["f32", "f32x", "f32s", "bf16", "bf16c"], default="f32",
                        help="f32: the reference's dtype - fp32 tensors; heavy convolutions contract with the fp32-accurate 3-piece "
                             "split-bf16 scheme on the bf16 matrix cores (= f32s), everything else with the exact fp32 MFMA; f32x: exact "
                             "fp32 MFMA everywhere; bf16: bf16 activations / weight operands in HBM, fp32 master weights, statistics and "
                             "losses (BASELINE configs 3/5); bf16c: fp32 tensors, bf16 contractions")
    parser.add_argument("--honour_nets", type=int, default=0,
                        help="1: build the generators / discriminators --gen_net / --dis_net name (the reference ignores both flags)")
    parser.add_argument("--variants", type=str, default="",
                        help="comma list of what the reference has commented out / disabled: l1_cycle, lab_gt_dis, gauss_noise, "
                             "perceptual (weights --lamda_perceptual / --lab_perceptual_weight)")
    parser.add_argument("--vgg_weights", type=str, default=None,
                        help="torchvision VGG16 state dict for --variants perceptual (the reference downloads vgg16(pretrained=True))")
    parser.add_argument("--testing_gen", type=str, default="resnet_9blocks_softmax",
                        help="generator testing.py builds (the reference hard-codes resnet_9blocks_softmax, testing.py:40)")
    return parser.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    args.gpu_ids = [int(s) for s in args.gpu_ids.split(",") if int(s) >= 0]
    args.as_written = bool(args.as_written)
    args.overlap_d = True                # train() reads the losses after sync_losses()
    if args.crop_height is None and args.crop_width is None:
        args.crop_height, args.crop_width = DEFAULT_CROP[args.dataset]
    if args.gpu_ids:
        import torch
        torch.cuda.set_device(args.gpu_ids[0])          # arch/ops.py:31-34: everything lives on gpu_ids[0]
    md = importlib.import_module(PKG + ".model")
    importlib.import_module(PKG + ".functional").set_conv_precision(args.dtype)
    dp = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        par = importlib.import_module(PKG + ".parallel")
        dp = par.DataParallel()
        args.gpu_ids = [dp.device_index]
    if args.training:
        loaders = None
        root = DATA_ROOTS[args.dataset]
        if args.data == "real" or (args.data == "auto" and os.path.isdir(root)):
            import torch
            du = importlib.import_module(PKG + ".data_utils")
            loaders = du.build_loaders(args, roots=DATA_ROOTS, device=torch.device("cuda", args.gpu_ids[0]),
                                       rank=dp.rank if dp is not None else 0)
        else:
            print("no dataset under %s: training on synthetic batches (--synthetic_steps per epoch)" % root)
        if args.model == "semisupervised_cycleGAN":
            print("Training semi-supervised cycleGAN")
            md.semisuper_cycleGAN(args, data_parallel=dp).train(args, loaders=loaders)
        if args.model == "supervised_model":
            print("Training base model")
            md.supervised_model(args, data_parallel=dp).train(args, loaders=loaders)
    if args.testing:                                        # main.py:69-71
        print("Testing")
        import testing
        testing.test(args)
    if args.validation:                                     # main.py:72-74
        print("Validating")
        import validation
        validation.validation(args)


if __name__ == "__main__":
    main()

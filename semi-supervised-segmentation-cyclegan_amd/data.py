"""Synthetic batches with the tensor contract of the reference's data_utils:
img f32 [B,3,H,W] in [-1,1] (ToTensor + Normalize(.5,.5), data_utils/__init__.py:126-131), gt int64 [B,1,H,W]
in [0,C) as piecewise-constant blobs, name str.  The real datasets (PIL decode, VOC/Cityscapes/ACDC file
layouts) are outside this build's scope: there is no network for datasets here."""
import torch


class SyntheticLoader:
    """Yields `steps` batches of (img, gt, names), generated on the host from a seeded torch.Generator;
    rank-offset seeds give every data-parallel rank its own stream."""

    def __init__(self, batch, classes, height, width, steps, seed, block=16, device=None):
        self.batch, self.classes, self.h, self.w, self.steps, self.seed, self.block = batch, classes, height, width, steps, seed, block
        self.device = device

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        bh, bw = -(-self.h // self.block), -(-self.w // self.block)
        for s in range(self.steps):
            img = torch.rand(self.batch, 3, self.h, self.w, generator=g) * 2.0 - 1.0
            coarse = torch.randint(0, self.classes, (self.batch, 1, bh, bw), generator=g)
            gt = coarse.repeat_interleave(self.block, 2).repeat_interleave(self.block, 3)[:, :, :self.h, :self.w].contiguous()
            if self.device is not None:
                img, gt = img.to(self.device), gt.to(self.device)
            yield img, gt, ["synthetic_%d_%d" % (s, b) for b in range(self.batch)]


def synthetic_loaders(args, classes, steps=None, rank=0):
    """(labeled, unlabeled, val) loaders; seeds labeled=1, unlabeled=2, val=3 (+1000*rank)."""
    steps = steps if steps is not None else getattr(args, "synthetic_steps", 8)
    mk = lambda seed, n: SyntheticLoader(args.batch_size, classes, args.crop_height, args.crop_width, n, seed + 1000 * rank)
    return mk(1, steps), mk(2, steps), mk(3, 1)

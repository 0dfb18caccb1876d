"""Input pipeline with the wire format of the reference's `data_utils`:
img f32 [B,3|1,H,W] = Normalize(.5,.5)(ToTensor(CenterCrop(Resize(PIL)))), gt int64 [B,1,H,W], name str
(data_utils/__init__.py:117-187, dataloader.py:15-393).

Two modes of `get_transformation`:
  * host (default; drop-in): every transform runs on the CPU and returns what the reference's returns;
  * `device_finish=True` (what `main.py` uses): the PIL part (decode, Resize, CenterCrop - the same PIL calls
    torchvision's PIL backend makes) stays on the host, the sample crosses PCIe as uint8, and ToTensor + Normalize /
    ToLabel + Relabel / Cityscapes encode_segmap run batched on the MI355X (`sscg_image_u8_to_f32`,
    `sscg_label_lut`): a quarter of the host->device bytes and no per-sample float work on the host cores.
torchvision is not a dependency: Resize / CenterCrop / ToTensor / Normalize are restated below on PIL + numpy.
The reference's `augmentations.py` is never used (`augmentation=None` at every call site) and is not restated."""
import numpy as np
import torch
from PIL import Image

from .dataloader import VOCDataset, CityscapesDataset, ACDCDataset   # noqa: F401  (reference: `from data_utils import ...`)

BILINEAR, NEAREST = 2, 0     # PIL.Image filter ids (torchvision's `interpolation=` ints)


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class Resize:
    """torchvision.transforms.Resize on a PIL image with an (h, w) size: `img.resize((w, h), interpolation)`."""

    def __init__(self, size, interpolation=BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        if isinstance(self.size, int):                    # shorter side -> size, aspect kept
            w, h = img.size
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return img
            if w < h:
                return img.resize((self.size, int(self.size * h / w)), self.interpolation)
            return img.resize((int(self.size * w / h), self.size), self.interpolation)
        return img.resize(tuple(self.size[::-1]), self.interpolation)


class CenterCrop:
    """torchvision.transforms.CenterCrop on a PIL image: zero-pad if smaller, then crop at round((dim - crop) / 2)."""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        ch, cw = self.size
        w, h = img.size
        if cw > w or ch > h:
            left, top = (cw - w) // 2 if cw > w else 0, (ch - h) // 2 if ch > h else 0
            right, bottom = (cw - w + 1) // 2 if cw > w else 0, (ch - h + 1) // 2 if ch > h else 0
            padded = Image.new(img.mode, (w + left + right, h + top + bottom), 0)
            if img.mode == "P":
                padded.putpalette(img.getpalette())
            padded.paste(img, (left, top))
            img = padded
            w, h = img.size
            if cw == w and ch == h:
                return img
        top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
        return img.crop((left, top, left + cw, top + ch))


def _hwc_u8(pic):
    a = np.array(pic)
    if a.dtype != np.uint8:
        raise TypeError("8-bit images expected (mode %s)" % pic.mode)
    return a[:, :, None] if a.ndim == 2 else a


class ToTensor:
    """PIL (uint8) -> float32 CHW in [0, 1]."""

    def __call__(self, pic):
        return torch.from_numpy(np.ascontiguousarray(_hwc_u8(pic).transpose(2, 0, 1))).to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        m = torch.tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        s = torch.tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.sub(m).div(s)


class ToU8:
    """device_finish mode: PIL -> uint8 HWC tensor (ToTensor + Normalize happen on the MI355X)."""

    def __call__(self, pic):
        return torch.from_numpy(np.ascontiguousarray(_hwc_u8(pic)))


class ToLabel:
    """PIL label map -> int64 [1, H, W] (data_utils/__init__.py:52-58)."""

    def __call__(self, image):
        return torch.from_numpy(np.array(image)).long().unsqueeze(0)


class ToLabelU8:
    def __call__(self, image):
        a = np.array(image)
        if a.dtype != np.uint8:
            raise TypeError("8-bit label maps expected (mode %s)" % image.mode)
        return torch.from_numpy(np.ascontiguousarray(a))


class Relabel:
    """tensor[tensor == olabel] = nlabel (data_utils/__init__.py:34-50)."""

    def __init__(self, olabel, nlabel):
        self.olabel, self.nlabel = olabel, nlabel

    def __call__(self, tensor):
        assert tensor.dtype == torch.int64, 'tensor needs to be LongTensor'
        tensor[tensor == self.olabel] = self.nlabel
        return tensor


CITYSCAPES_VOID = [0, 1, 2, 3, 4, 5, 6, 9, 10, 14, 15, 16, 18, 29, 30, -1]
CITYSCAPES_VALID = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]


def cityscapes_encode(mask):
    """CityscapesDataset.encode_segmap (dataloader.py:260-267): void ids -> 250 -> 19, valid ids -> 0..18, in place and
    in the reference's order (the order matters for an in-place remap; see label_table)."""
    for v in CITYSCAPES_VOID:
        mask[mask == v] = 250
    for i, v in enumerate(CITYSCAPES_VALID):
        mask[mask == v] = i
    mask[mask == 250] = 19
    return mask


def label_table(dataset):
    """The 256-entry table `sscg_label_lut` applies: the host-side label transforms run on every possible uint8 id."""
    t = torch.arange(256, dtype=torch.int64)
    if dataset == 'voc2012':
        t = Relabel(255, 0)(t)
    elif dataset == 'cityscapes':
        t = cityscapes_encode(t)
    return t


def get_transformation(size, resize=False, dataset='voc2012', device_finish=False):
    """data_utils/__init__.py:117-187.  `size` = (height, width)."""
    assert dataset in ['voc2012', 'cityscapes', 'acdc'], 'The dataset name must be set correctly in the get_transformation function'
    nch = 1 if dataset == 'acdc' else 3
    head = [Resize(size), CenterCrop(size)] if resize else [CenterCrop(size)]
    head_gt = [Resize(size, interpolation=NEAREST), CenterCrop(size)] if resize else [CenterCrop(size)]
    if device_finish:
        return {'img': Compose(head + [ToU8()]), 'gt': Compose(head_gt + [ToLabelU8()]), 'device_finish': True,
                'lut': label_table(dataset), 'mean': [.5] * nch, 'std': [.5] * nch}
    tail_gt = [ToLabel()] + ([Relabel(255, 0)] if dataset == 'voc2012' else [])   # 255 (boundaries) -> 0
    return {'img': Compose(head + [ToTensor(), Normalize([.5] * nch, [.5] * nch)]), 'gt': Compose(head_gt + tail_gt)}


class DeviceLoader:
    """Wraps a torch DataLoader over a dataset built with `device_finish=True` transforms and yields what the step
    consumes - (img f32 [B,C,H,W] channels-last on the MI355X, gt int64 [B,1,H,W], names) - or (img, names) for the
    'test' split.  Batches cross PCIe as uint8 from pinned memory; the float conversion is one HIP launch per batch."""

    def __init__(self, loader, transformation, device):
        if not transformation.get('device_finish'):
            raise ValueError("DeviceLoader needs get_transformation(..., device_finish=True)")
        self.loader, self.device = loader, device
        self.lut = transformation['lut'].to(device)
        self.mean = torch.tensor(transformation['mean'], dtype=torch.float32, device=device)
        self.std = torch.tensor(transformation['std'], dtype=torch.float32, device=device)

    def __len__(self):
        return len(self.loader)

    def _up(self, t):
        t = t.contiguous()
        return (t.pin_memory() if self.device.type == "cuda" else t).to(self.device, non_blocking=True)

    def __iter__(self):
        from .. import functional as F
        for batch in self.loader:
            img = F.image_u8_to_f32(self._up(batch[0]), self.mean, self.std)
            if len(batch) == 2:
                yield img, batch[1]
            else:
                yield img, F.label_lut(self._up(batch[1]), self.lut), batch[2]


def build_loaders(args, roots=None, device=None, sets=('label', 'unlabel', 'val'), rank=0):
    """The datasets and DataLoaders of model.py:315-348 (ratios 0.1/0.1/0.5 for VOC, 0.5 otherwise; batch_size, shuffle and
    drop_last on all three, as written there), finished on `device` when one is given."""
    roots = roots or {'voc2012': './data/VOC2012', 'cityscapes': './data/Cityscape', 'acdc': './data/ACDC'}
    from torch.utils.data import DataLoader
    tr = get_transformation((args.crop_height, args.crop_width), resize=True, dataset=args.dataset, device_finish=device is not None)
    cls = {'voc2012': VOCDataset, 'cityscapes': CityscapesDataset, 'acdc': ACDCDataset}[args.dataset]
    out = []
    for name in sets:
        ratio = 0.5 if (args.dataset != 'voc2012' or name in ('val', 'test')) else 0.1
        ds = cls(root_path=roots[args.dataset], name=name, ratio=ratio, transformation=tr, augmentation=None)
        gen = torch.Generator().manual_seed(20260928 + 1000 * rank + len(out)) if rank else None   # ranks draw different batches
        ld = DataLoader(ds, batch_size=args.batch_size, shuffle=True, drop_last=True, generator=gen)
        out.append(DeviceLoader(ld, tr, device) if device is not None else ld)
    return out

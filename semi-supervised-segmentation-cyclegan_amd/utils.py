"""Hot-path helpers of the reference's utils.py (make_one_hot :314, Sample_from_Pool :278, cuda :221,
LambdaLR :434, runningScore :357, checkpoint I/O :265-273), re-expressed for a device-resident step:
nothing here moves activations to the host."""
import numpy as np
import torch

from . import functional as F

CLASSES = {"voc2012": 21, "cityscapes": 20, "acdc": 4}   # model.py:205-210, utils.py:335-342


def cuda(xs, gpu_id):
    """utils.cuda (utils.py:221-227): move a tensor / list of tensors to gpu_id[0] when a GPU exists."""
    if torch.cuda.is_available() and len(gpu_id) > 0:
        dev = torch.device("cuda", int(gpu_id[0]))
        if not isinstance(xs, (list, tuple)):
            return xs.to(dev, non_blocking=True)
        return [x.to(dev, non_blocking=True) for x in xs]
    return xs


def make_one_hot(labels, dataname, gpu_id=None):
    """Integer labels [N,1,H,W] -> fp32 one-hot [N,C,H,W] (utils.py:314-350), one HIP pass."""
    assert dataname in CLASSES, "dataset name should be one of the following: 'voc2012',given {}".format(dataname)
    return F.label_onehot(labels.long(), CLASSES[dataname])


class Sample_from_Pool(object):
    """History pool of Shrivastava et al. (utils.py:278-299): `max_elements` slots; once full, with p=0.5 a
    random stored item is returned and replaced.  Items stay on the device (the reference round-trips three
    activation batches through numpy every step, model.py:490-495); the host RNG calls (np.random.ranf /
    randint) are the reference's, so a seeded run makes the same decisions."""

    def __init__(self, max_elements=50):
        self.max_elements = max_elements
        self.cur_elements = 0
        self.items = []

    def __call__(self, in_items):
        out = []
        for item in in_items:
            if self.cur_elements < self.max_elements:
                self.items.append(item)
                self.cur_elements += 1
                out.append(item)
            elif np.random.ranf() > 0.5:
                idx = np.random.randint(0, self.max_elements)
                out.append(self.items[idx])
                self.items[idx] = item
            else:
                out.append(item)
        return out


class GaussianNoise(object):
    """utils.GaussianNoise (utils.py:116-140): x + sigma * x.detach() * N(0,1) in training mode - the (dead) noise branch of
    the discriminator step, model.py:486-488.  Device-resident: the reference moves the batch to the CPU first."""

    def __init__(self, sigma=0.1, is_relative_detach=True):
        self.sigma, self.is_relative_detach, self.training = sigma, is_relative_detach, True
        self._calls = 0

    def __call__(self, x):
        if self.training and self.sigma != 0:
            self._calls += 1
            seed = (torch.initial_seed() * 0x9E3779B1 + self._calls) & 0x7FFFFFFFFFFFFFF
            return F.gauss_noise(x.detach(), self.sigma, seed)
        return x


class _MaxPool2(torch.nn.Module):
    """nn.MaxPool2d(kernel_size=2, stride=2) (torchvision VGG16 features[4, 9, 16, 23])."""

    def forward(self, x):
        return F.maxpool2x2(x)


class Vgg16(torch.nn.Module):
    """utils.py:145-177: the first 23 layers of torchvision's VGG16 `features`, cut into four slices (relu1_2, relu2_2, relu3_3,
    relu4_3); state-dict keys `slice{1..4}.{features index}.{weight,bias}`.  The reference loads `vgg16(pretrained=True)`; there is
    no network here, so the weights are He-initialised unless `weights` names a torchvision VGG16 state dict (`features.N.*`)."""

    _LAYOUT = (("slice1", ((0, 3, 64), (2, 64, 64))),
               ("slice2", ((4, None, None), (5, 64, 128), (7, 128, 128))),
               ("slice3", ((9, None, None), (10, 128, 256), (12, 256, 256), (14, 256, 256))),
               ("slice4", ((16, None, None), (17, 256, 512), (19, 512, 512), (21, 512, 512))))

    def __init__(self, requires_grad=False, weights=None):
        super().__init__()
        from .arch import ops
        for name, layers in self._LAYOUT:
            seq = ops.FusedSequential()
            for idx, cin, cout in layers:
                if cin is None:
                    seq.add_module(str(idx), _MaxPool2())
                else:
                    conv = ops.Conv2d(cin, cout, 3, 1, 1)
                    conv.head = True          # the features stay fp32 in bf16 mode (they feed a loss)
                    with torch.no_grad():
                        conv.weight.copy_(torch.empty(conv.weight.shape).normal_(0.0, (2.0 / (cin * 9)) ** 0.5))
                    seq.add_module(str(idx), conv)
                    seq.add_module(str(idx + 1), ops.ReLU(True))
            setattr(self, name, seq)
        if weights is None:
            # the reference always loads the ImageNet weights (utils.py:149): without them the loss is finite but not the
            # reference's perceptual loss - say so, once per construction
            print("**perceptual loss: no --vgg_weights given, VGG16 is He-initialised (NOT the reference's pretrained features)**")
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            own = self.state_dict()
            for k in own:
                own[k] = sd["features." + k.split(".", 1)[1]]
            self.load_state_dict(own, strict=True)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        h1 = self.slice1(x)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        return {"relu1_2": h1, "relu2_2": h2, "relu3_3": h3, "relu4_3": h4}

    def relu2_2(self, x):
        return self.slice2(self.slice1(x))


_TRANS_MEAN = (0.485, 0.456, 0.406)
_TRANS_STD = (0.229, 0.224, 0.225)


def perceptual_loss(x, y, gpu_ids=None, vgg=None):
    """utils.py:181-208 as written: u = x / 2 + 1 / 2, then per channel u * std + mean (sic: the reference multiplies by the
    ImageNet std and adds the mean), both images through VGG16 up to relu2_2, nn.MSELoss between the features.
    `vgg`: a Vgg16 to reuse (the reference builds - and downloads - a new one on every call)."""
    dev = x.device
    if vgg is None:
        vgg = Vgg16(requires_grad=False).to(dev)
    a = torch.tensor([0.5 * s for s in _TRANS_STD], device=dev)
    b = torch.tensor([0.5 * s + m for s, m in zip(_TRANS_STD, _TRANS_MEAN)], device=dev)
    zero, one = torch.zeros(3, device=dev), torch.full((3,), 1.0 - 1e-5, device=dev)

    def prep(t):        # per-channel affine = the eval-mode BatchNorm kernel with mean 0, var + eps = 1
        return F.batch_norm_act(t, a, b, zero, one, False, 0.0, 1e-5)

    return F.mse_loss(vgg.relu2_2(prep(y)), vgg.relu2_2(prep(x)))


class LambdaLR():
    """Linear decay to zero after `decay_epoch` (utils.py:434-441)."""

    def __init__(self, epochs, offset, decay_epoch):
        self.epochs, self.offset, self.decay_epoch = epochs, offset, decay_epoch

    def step(self, epoch):
        return 1.0 - max(0, epoch + self.offset - self.decay_epoch) / (self.epochs - self.decay_epoch)


class runningScore(object):
    """Confusion-matrix mIoU (utils.py:357-412).  VOC ignores class 0, Cityscapes the last class, ACDC none."""

    def __init__(self, n_classes, dataset):
        self.n_classes, self.dataset = n_classes, dataset
        self.confusion_matrix = np.zeros((n_classes, n_classes))
        self._device_hist = None     # int64 [C, C] accumulated by sscg_confusion_hist; folded in by get_scores()

    def update_device(self, label_trues, label_preds):
        """Same counts as update(), for label / prediction tensors that already live on the MI355X: no
        device->host copy of the maps per batch (model.py:568-569 moves both to numpy)."""
        self._device_hist = F.confusion_hist(label_trues, label_preds, self.n_classes, self._device_hist)

    def _fold_device(self):
        if self._device_hist is not None:
            self.confusion_matrix += self._device_hist.cpu().numpy().astype(np.float64)
            self._device_hist = None

    def update(self, label_trues, label_preds):
        n = self.n_classes
        for lt, lp in zip(label_trues, label_preds):
            lt, lp = np.asarray(lt).ravel(), np.asarray(lp).ravel()
            keep = (lt >= 0) & (lt < n)
            self.confusion_matrix += np.bincount(n * lt[keep].astype(int) + lp[keep], minlength=n * n).reshape(n, n)

    def get_scores(self):
        self._fold_device()
        h, n = self.confusion_matrix, self.n_classes
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(h).sum() / h.sum()
            acc_cls = np.nanmean(np.diag(h) / h.sum(axis=1))
            sub = h[1:, 1:] if self.dataset == "voc2012" else (h[:n - 1, :n - 1] if self.dataset == "cityscapes" else h)
            iu = np.diag(sub) / (sub.sum(axis=1) + sub.sum(axis=0) - np.diag(sub))
        cls_iu = dict(zip(range(len(iu)), iu))
        return {"Overall Acc: \t": acc, "Mean Acc : \t": acc_cls, "Mean IoU : \t": np.nanmean(iu)}, cls_iu

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_classes, self.n_classes))
        self._device_hist = None


def save_checkpoint(state, save_path):
    torch.save(state, save_path)


def load_checkpoint(ckpt_path, map_location="cpu"):
    ckpt = torch.load(ckpt_path, map_location=map_location)
    print(" [*] Loading checkpoint from %s succeed!" % ckpt_path)
    return ckpt


def print_networks(nets, names):
    print("------------Number of Parameters---------------")
    for net, name in zip(nets, names):
        n = sum(p.numel() for p in net.parameters())
        print("[Network %s] Total number of parameters : %.3f M" % (name, n / 1e6))
    print("-----------------------------------------------")


# ---- inference-script helpers (utils.py:14-55 of the reference; the palette values are dataset constants)
def _pad_palette(p):
    return p + [0] * (256 * 3 - len(p))


palette = _pad_palette([0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0, 0, 0, 128, 128, 0, 128, 0, 128, 128,
                        128, 128, 128, 64, 0, 0, 192, 0, 0, 64, 128, 0, 192, 128, 0, 64, 0, 128, 192, 0, 128,
                        64, 128, 128, 192, 128, 128, 0, 64, 0, 128, 64, 0, 0, 192, 0, 128, 192, 0, 0, 64, 128])
cityscape_palette = _pad_palette([128, 64, 128, 244, 35, 232, 70, 70, 70, 102, 102, 156, 190, 153, 153, 153, 153, 153,
                                  250, 170, 30, 220, 220, 0, 107, 142, 35, 152, 251, 152, 0, 130, 180, 220, 20, 60,
                                  255, 0, 0, 0, 0, 142, 0, 0, 70, 0, 60, 100, 0, 80, 100, 0, 0, 230, 119, 11, 32])
acdc_palette = _pad_palette([0, 0, 0, 128, 64, 128, 70, 70, 70, 250, 170, 30])


def colorize_mask(mask, dataset):
    """One-channel class map (numpy) -> paletted PIL image (utils.py:41-55)."""
    from PIL import Image
    assert dataset in ('voc2012', 'cityscapes', 'acdc')
    new_mask = Image.fromarray(np.asarray(mask).astype(np.uint8)).convert('P')
    new_mask.putpalette({'voc2012': palette, 'cityscapes': cityscape_palette, 'acdc': acdc_palette}[dataset])
    return new_mask


def save_image(tensor, path):
    """torchvision.utils.save_image for one CHW image in [0, 1]: x*255 + 0.5, clamp, uint8, HWC."""
    from PIL import Image
    arr = tensor.detach().float().cpu().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(path)

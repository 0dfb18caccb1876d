"""TEST INFRASTRUCTURE (see oracle/__init__.py) - CPU restatement of one hot-loop iteration of
`semisuper_cycleGAN.train` (/root/reference model.py:370-552) and of the supervised step (model.py:120-143).

The reference does not expose a step function; this module factors one out of the 350-line
method, line for line in behaviour, on top of oracle/nets.py.  torch.optim.Adam is used as is - it is
the reference's own optimiser (model.py:286-287)."""
import copy

import numpy as np
import torch
import torch.nn.functional as TF

from . import nets

LOSS_KEYS = ("img_dis_loss", "gt_dis_loss", "cycle_img_dis_loss", "img_gen_loss", "gt_gen_loss", "img_cycle_loss",
             "gt_cycle_loss", "lab_loss_CE", "lab_loss_MSE")


def one_hot(labels, C, dtype):
    """utils.make_one_hot (utils.py:314-350)."""
    z = torch.zeros(labels.size(0), C, labels.size(2), labels.size(3), dtype=dtype)
    return z.scatter_(1, labels.long(), 1)


def argmax_one_hot(x, C):
    """model.py:435-437 / 509-511: `x.data.max(1)[1]` -> unsqueeze -> make_one_hot (no gradient)."""
    idx = x.detach().max(1)[1].unsqueeze(1)
    return one_hot(idx, C, x.dtype)


class Pool:
    """utils.Sample_from_Pool (utils.py:278-299): 50 slots, each holding whatever the caller passes
    (the reference passes whole batches, model.py:490-493).  Draws from numpy's global RNG like the reference."""

    def __init__(self, max_elements=50):
        self.max_elements = max_elements
        self.items = []

    def __call__(self, item):
        if len(self.items) < self.max_elements:
            self.items.append(item)
            return item
        if np.random.ranf() > 0.5:
            idx = np.random.randint(0, self.max_elements)
            old = copy.copy(self.items[idx])
            self.items[idx] = item
            return old
        return item


def lambda_lr(epoch, epochs, offset, decay_epoch):
    """utils.LambdaLR.step (utils.py:434-441)."""
    return 1.0 - max(0, epoch + offset - decay_epoch) / (epochs - decay_epoch)


def _trainable(sd, names_requiring_grad):
    ps = []
    for k, v in sd.items():
        if k in names_requiring_grad:
            v.requires_grad_(True)
            ps.append(v)
    return ps


def deeplab_trainable_keys(sd):
    """Conv weights + classifier biases; every BN affine is frozen (arch/generators.py:327-338,391-392,418-419)."""
    keys = []
    for k in sd:
        is_bn = (".bn" in k or k.startswith("bn1.") or ".downsample.1." in k)
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")) or is_bn:
            continue
        keys.append(k)
    return keys


class SemiSupOracle:
    """State + one step of the semi-supervised CycleGAN (model.py:203-311 ctor, :370-552 loop body)."""

    def __init__(self, n_classes, state_dicts, lr=2e-4, lab_CE_weight=1.0, lab_MSE_weight=1.0, adversarial_weight=1.0,
                 discriminator_weight=1.0, lamda_gt=0.1, norm="instance", use_dropout=False, crop=(64, 64), as_written=False, q=None,
                 variants=(), lamda_img=0.5, gen_net="deeplab", dis_net="pixel"):
        self.C = n_classes
        # the build's --honour_nets: model.py:215-222 hard-codes netG='deeplab' / netD='pixel' and never reads
        # --gen_net / --dis_net (main.py:43-44); with the flag honoured the SAME step runs on what define_Gen / define_Dis build for
        # those names (arch/generators.py:404-418 resnet_{6,9}blocks; arch/discriminators.py:42-63,83-92 n_layers).  Default = as written.
        self.gen_net, self.dis_net = gen_net, dis_net
        # loss terms the reference has commented out (the build's --variants), restated as the commented lines read:
        #   "l1_cycle"   model.py:453  img_cycle_loss = L1(recon_img, unl_img), weighted like the other image-cycle term (lamda_img)
        #   "lab_gt_dis" model.py:439,447  gt_label_gen_loss = MSE(Ds(lab_gt), ones), an adversarial term (adversarial_weight)
        self.variants = set(variants)
        self.lamda_img = lamda_img
        # q = nets.Bf16Emulation: the same step in the caller's dtype (fp64) with every tensor the bf16 build keeps in bf16 rounded
        # to bf16 at the same place - the yardstick of the bf16 parity tests (not the reference's arithmetic; None = exact)
        self.q = nets._NOQ if q is None else q
        self.as_written = as_written   # also run the reference's forwards whose outputs nothing reads (timing baseline only)
        self.sd = state_dicts  # dict: Gis, Gsi, Di, Ds, old_Gis, old_Gsi, old_Di -> flat state dicts (modified in place)
        self.w = dict(ce=lab_CE_weight, l1=lab_MSE_weight, adv=adversarial_weight, dis=discriminator_weight, gt=lamda_gt)
        self.norm = norm
        self.use_dropout = use_dropout
        self.crop = crop
        gen_keys = deeplab_trainable_keys if gen_net == "deeplab" else (lambda sd_: list(sd_.keys()))   # InstanceNorm carries no state
        g_params = _trainable(self.sd["Gis"], gen_keys(self.sd["Gis"])) + _trainable(self.sd["Gsi"], gen_keys(self.sd["Gsi"]))
        d_params = _trainable(self.sd["Di"], list(self.sd["Di"].keys())) + _trainable(self.sd["Ds"], list(self.sd["Ds"].keys()))
        self.g_params, self.d_params = g_params, d_params
        self.g_opt = torch.optim.Adam(g_params, lr=lr, betas=(0.5, 0.999))   # model.py:286
        self.d_opt = torch.optim.Adam(d_params, lr=lr, betas=(0.5, 0.999))   # model.py:287
        self.pools = (Pool(), Pool(), Pool())  # new_img_fake_sample, img_fake_sample, gt_fake_sample (model.py:350-352)

    def interp(self, x):
        """nn.Upsample(crop, mode='bilinear', align_corners=True) (model.py:268)."""
        return TF.interpolate(x, size=self.crop, mode="bilinear", align_corners=True)

    def _old_g(self, name, x, tanh):
        return nets.resnet_generator(self.sd[name], x, 9, tanh, self.norm, self.use_dropout, q=self.q)

    def _dis(self, name, x):
        if self.dis_net == "n_layers" and name != "old_Di":                 # old_Di stays netD='pixel' (model.py:229-230)
            return nets.nlayer_discriminator(self.sd[name], x, 3, self.norm, q=self.q)
        return nets.pixel_discriminator(self.sd[name], x, self.norm, q=self.q)

    def _gen(self, name, x):
        """Gis / Gsi as the step calls them (model.py:385-387,408-410)."""
        if self.gen_net == "deeplab":
            return nets.deeplab(self.sd[name], x, True, q=self.q)
        n_blocks = 9 if "9blocks" in self.gen_net else 6
        return nets.resnet_generator(self.sd[name], x, n_blocks, not self.gen_net.endswith("_softmax"), self.norm, self.use_dropout, q=self.q)

    def step(self, l_img, l_gt, unl_img, collect=None):
        C, sd, w, q = self.C, self.sd, self.w, self.q
        mse = lambda x, t: ((x - t) ** 2).mean()  # nn.MSELoss against ones/zeros (model.py:441-446,514-528)
        # ---------------- generator step (model.py:376-474)
        self.g_opt.zero_grad()
        for p in self.d_params:
            p.requires_grad_(False)                                        # set_grad(..., False) :379
        lab = l_gt.squeeze(1)
        fake_img = self.interp(self._gen("Gis", one_hot(l_gt, C, l_img.dtype)))   # :385,390
        fake_gt = self.interp(self._gen("Gsi", unl_img))                            # :386,391
        lab_gt = self.interp(self._gen("Gsi", l_img))                               # :387,392
        lab_loss_CE = TF.cross_entropy(lab_gt, lab)                                              # :398
        lab_gt = torch.softmax(lab_gt, 1)                                                        # :401
        fake_gt = torch.softmax(fake_gt, 1)                                                      # :402
        recon_img = self.interp(self._gen("Gis", fake_gt))                          # :408,413
        with torch.no_grad():
            self._gen("Gis", lab_gt.detach())        # :409 result unused, but advances Gis' BN running stats
        recon_gt = self.interp(self._gen("Gsi", fake_img))                          # :410,415
        with torch.no_grad():
            resnet_fake_gt = torch.softmax(self._old_g("old_Gsi", unl_img, False), 1)            # :418,421
            resnet_recon_img = self._old_g("old_Gis", resnet_fake_gt, True)                      # :422
            # :419-420,423 (old_Gsi(l_img) -> old_Gis) feed nothing and hold no state: elided unless the CPU baseline asks
            # for the step exactly as written (same work as the GPU headline); the results are dropped either way
            if self.as_written:
                resnet_lab_gt = torch.softmax(self._old_g("old_Gsi", l_img, False), 1)               # :419-420
                self._old_g("old_Gis", resnet_lab_gt, True)                                          # :423
        fake_img_dis = self._dis("Di", fake_img)                                                 # :431
        resnet_fake_img_dis = self._dis("old_Di", recon_img)                                     # :432
        fake_gt_dis = self._dis("Ds", argmax_one_hot(fake_gt, C))                                # :435-438
        img_gen_loss = mse(fake_img_dis, 1.0)                                                    # :445
        gt_gen_loss = mse(fake_gt_dis, 1.0)                                                      # :446
        img_cycle_loss = mse(resnet_fake_img_dis, 1.0)                                           # :452
        gt_cycle_loss = TF.cross_entropy(recon_gt, lab)                                          # :455
        lab_loss_MSE = (fake_img - l_img).abs().mean()                                           # :461 (an L1 loss)
        full = w["ce"] * lab_loss_CE + w["l1"] * lab_loss_MSE                                    # :464
        unsup = w["adv"] * (img_gen_loss + gt_gen_loss) + img_cycle_loss + gt_cycle_loss * w["gt"]  # :466
        extras = {}
        if "l1_cycle" in self.variants:
            extras["img_cycle_l1"] = (recon_img - unl_img).abs().mean()                          # :453 (commented out)
            unsup = unsup + self.lamda_img * extras["img_cycle_l1"]
        if "lab_gt_dis" in self.variants:
            extras["gt_label_gen_loss"] = mse(self._dis("Ds", lab_gt), 1.0)                      # :439,:447 (commented out)
            unsup = unsup + w["adv"] * extras["gt_label_gen_loss"]
        (full + unsup).backward()                                                                # :472
        if collect is not None:
            collect["g_grads"] = [None if p.grad is None else p.grad.detach().clone() for p in self.g_params]
            collect["fake_img"], collect["fake_gt"], collect["recon_img"] = fake_img.detach(), fake_gt.detach(), recon_img.detach()
        self.g_opt.step()                                                                        # :474
        # ---------------- discriminator step (model.py:477-542)
        for p in self.d_params:
            p.requires_grad_(True)                                                               # :481
        self.d_opt.zero_grad()
        recon_img_p = self.pools[0](recon_img.detach())                                          # :490
        fake_img_p = self.pools[1](fake_img.detach())                                            # :491
        fake_gt_p = self.pools[2](fake_gt.detach())                                              # :493
        unl_img_dis = self._dis("Di", unl_img)                                                   # :499
        fake_img_dis = self._dis("Di", fake_img_p)                                               # :500
        resnet_recon_img_dis = self._dis("old_Di", resnet_recon_img)                             # :501
        resnet_fake_img_dis = self._dis("old_Di", recon_img_p)                                   # :502
        real_gt_dis = self._dis("Ds", one_hot(l_gt, C, l_img.dtype))                             # :506-507
        fake_gt_dis = self._dis("Ds", argmax_one_hot(fake_gt_p, C))                              # :509-512
        img_dis_loss = (mse(unl_img_dis, 1.0) + mse(fake_img_dis, 0.0)) * 0.5                    # :521-522,531
        gt_dis_loss = (mse(real_gt_dis, 1.0) + mse(fake_gt_dis, 0.0)) * 0.5                      # :523-524,532
        cycle_img_dis_loss = mse(resnet_recon_img_dis, 1.0) + mse(resnet_fake_img_dis, 0.0)      # :527-528,534
        (w["dis"] * (img_dis_loss + gt_dis_loss) + cycle_img_dis_loss).backward()                # :538-539
        if collect is not None:
            collect["d_grads"] = [None if p.grad is None else p.grad.detach().clone() for p in self.d_params]
        self.d_opt.step()                                                                        # :542
        vals = (img_dis_loss, gt_dis_loss, cycle_img_dis_loss, img_gen_loss, gt_gen_loss, img_cycle_loss, gt_cycle_loss,
                lab_loss_CE, lab_loss_MSE)
        out = {k: float(v.detach()) for k, v in zip(LOSS_KEYS, vals)}                             # scalars of :548-550
        out.update({k: float(v.detach()) for k, v in extras.items()})
        return out


    # ------------------------------------------------------------------ teacher-forced second pass
    def first_pass(self, l_img, l_gt, unl_img, want_lab=True):
        """The first-pass outputs the second pass consumes (model.py:385-387,390-392,401-402), without a graph:
        fake_img = interp(Gis(onehot(l_gt))), fake_gt = softmax(interp(Gsi(unl_img))), lab_gt = softmax(interp(Gsi(l_img)))."""
        C, sd, q = self.C, self.sd, self.q
        with torch.no_grad():
            fake_img = self.interp(self._gen("Gis", one_hot(l_gt, C, l_img.dtype)))     # :385,390
            fake_gt = torch.softmax(self.interp(self._gen("Gsi", unl_img)), 1)           # :386,391,402
            lab_gt = torch.softmax(self.interp(self._gen("Gsi", l_img)), 1) if want_lab else None   # :387,392,401
        return fake_img, fake_gt, lab_gt

    def second_pass(self, fake_img, fake_gt, l_gt, unl_img):
        """The part of the step that sits TWO DeepLab passes deep, fed GIVEN first-pass outputs (teacher forcing: the chaos of the
        first pass cannot compound, so each quantity below is one DeepLab pass deep and can be held to north_star's 1e-3):
          recon_img = interp(Gis(fake_gt)) (model.py:408,413) -> old_Di (:432) -> img_cycle_loss (:452)
          recon_gt  = interp(Gsi(fake_img)) (:410,415)        -> gt_cycle_loss (:455)
          cycle_img_dis_loss = MSE(old_Di(old_Gis(softmax(old_Gsi(unl_img)))), 1) + MSE(old_Di(recon_img), 0) (:418-422,501-502,527-528,534;
          the pool hands back the current item while it fills, :490)
        Returns the three losses and d(img_cycle_loss)/d(fake_gt), d(gt_cycle_loss)/d(fake_img).  tests/test_oracle_golden.py checks
        that, fed the step's own first-pass outputs, this reproduces the pinned `step`'s three chained losses."""
        sd, q = self.sd, self.q
        mse = lambda x, t: ((x - t) ** 2).mean()
        fake_gt = fake_gt.detach().clone().requires_grad_(True)
        fake_img = fake_img.detach().clone().requires_grad_(True)
        recon_img = self.interp(self._gen("Gis", fake_gt))                               # :408,413
        recon_gt = self.interp(self._gen("Gsi", fake_img))                               # :410,415
        img_cycle_loss = mse(self._dis("old_Di", recon_img), 1.0)                                          # :432,452
        gt_cycle_loss = TF.cross_entropy(recon_gt, l_gt.squeeze(1))                                        # :455
        d_fake_gt, = torch.autograd.grad(img_cycle_loss, fake_gt)
        d_fake_img, = torch.autograd.grad(gt_cycle_loss, fake_img)
        with torch.no_grad():
            resnet_fake_gt = torch.softmax(self._old_g("old_Gsi", unl_img, False), 1)                      # :418,421
            resnet_recon_img = self._old_g("old_Gis", resnet_fake_gt, True)                                # :422
            cycle_img_dis_loss = mse(self._dis("old_Di", resnet_recon_img), 1.0) + mse(self._dis("old_Di", recon_img.detach()), 0.0)
        return dict(img_cycle_loss=float(img_cycle_loss.detach()), gt_cycle_loss=float(gt_cycle_loss.detach()),
                    cycle_img_dis_loss=float(cycle_img_dis_loss), d_fake_gt=d_fake_gt, d_fake_img=d_fake_img,
                    recon_img=recon_img.detach())


class SupervisedOracle:
    """supervised_model step (model.py:120-143): DeepLab Gsi + CE + Adam(0.9, 0.999)."""

    def __init__(self, n_classes, gsi_sd, lr=2e-4, crop=(128, 128)):
        self.C, self.sd, self.crop = n_classes, gsi_sd, crop
        self.params = _trainable(gsi_sd, deeplab_trainable_keys(gsi_sd))
        self.opt = torch.optim.Adam(self.params, lr=lr, betas=(0.9, 0.999))   # model.py:69

    def step(self, l_img, l_gt):
        self.opt.zero_grad()
        out = nets.deeplab(self.sd, l_img, True)
        out = TF.interpolate(out, size=self.crop, mode="bilinear", align_corners=True)
        loss = TF.cross_entropy(out, l_gt.squeeze(1))
        loss.backward()
        self.opt.step()
        return float(loss.detach())


def running_score(conf, dataset):
    """utils.runningScore.get_scores (utils.py:375-409): (overall acc, mean acc, mIoU) from a confusion matrix.
    VOC drops class 0, Cityscapes drops the last class, ACDC keeps all."""
    hist = np.asarray(conf, dtype=np.float64)
    n = hist.shape[0]
    acc = np.diag(hist).sum() / hist.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        sub = hist[1:, 1:] if dataset == "voc2012" else (hist[:n - 1, :n - 1] if dataset == "cityscapes" else hist)
        iu = np.diag(sub) / (sub.sum(axis=1) + sub.sum(axis=0) - np.diag(sub))
    return acc, acc_cls, np.nanmean(iu), iu


def confusion(label_true, label_pred, n_class):
    """utils.runningScore._fast_hist (utils.py:363-369)."""
    lt, lp = np.asarray(label_true).ravel(), np.asarray(label_pred).ravel()
    mask = (lt >= 0) & (lt < n_class)
    return np.bincount(n_class * lt[mask].astype(int) + lp[mask], minlength=n_class ** 2).reshape(n_class, n_class)

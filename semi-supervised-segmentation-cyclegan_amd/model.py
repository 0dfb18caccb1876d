"""Training drivers of the reference (model.py) on the MI355X kernels.

`semisuper_cycleGAN` mirrors /root/reference model.py:202-662 - same constructor argument (`args` from
main.py), same networks / losses / optimisers / checkpoint keys - but factors the 350-line `train` method
into `step(l_img, l_gt, unl_img)` (one iteration of model.py:370-552) plus a thin epoch loop.  Differences
that are deliberate and MI355X-first:
  * every tensor of the step stays in HBM: the image pools hold device tensors (the reference round-trips
    three activation batches through numpy per step, model.py:490-495) and the nine losses are device
    scalars that are only read back when they are logged;
  * forwards whose outputs the reference never uses (model.py:409) run without an autograd graph;
  * gradients accumulate into flat arenas (optim.FusedAdam) that RCCL all-reduces in one piece per
    optimiser under data parallelism (parallel.py).
`supervised_model` mirrors model.py:33-199 (BASELINE config 1)."""
import itertools
import os

import torch

from . import functional as F
from . import utils
from . import arch
from .arch import define_Dis, define_Gen, set_grad
from .optim import FusedAdam
from .utils import CLASSES, make_one_hot

_DBG_TOPWAIT = [x for x in os.environ.get("SSCG_DBG_TOPWAIT", "").split(",") if x]
# where the host ISSUES the frozen generators' pass (model.py:418-423: it feeds the D step only) and the unused Gis(lab_gt) pass (:409):
# "early" = the reference's program order, "mid" = behind the trainable generators' forwards, "late" = behind the backward's launches
_PHASES = os.environ.get("SSCG_PHASE_EVENTS") == "1"      # diagnostic: timed events around the passes of a step (tools/phases.py)
_FROZEN_AT = os.environ.get("SSCG_FROZEN_AT", "early")
_UNUSED_AT = os.environ.get("SSCG_UNUSED_AT", "early")
for _name, _v in (("SSCG_FROZEN_AT", _FROZEN_AT), ("SSCG_UNUSED_AT", _UNUSED_AT)):
    if _v not in ("early", "mid", "late"):      # (a typo would silently skip a pass: BN running statistics diverge from model.py:409)
        raise ValueError("%s=%r: expected early, mid or late" % (_name, _v))


def _check_bf16_widths(args):
    """--dtype bf16: the bf16 conv kernels move 64-channel k-tiles, so every activation between a network's first conv and its head
    must have a multiple of 64 channels - true for the reference's default widths (--ngf 64 --ndf 64), not for e.g. --ngf 32."""
    if F.get_conv_precision() == "bf16":
        for name in ("ngf", "ndf"):
            v = getattr(args, name, 64)
            if v % 64:
                raise ValueError("--dtype bf16 needs --%s to be a multiple of 64 (got %d): bf16 activations are tiled 64 channels at a "
                                 "time; use --dtype f32 for narrower networks" % (name, v))


def _select_device(args):
    """The reference puts everything on gpu_ids[0] (arch/ops.py:31-34, utils.py:221-227).  Kernels are launched on the
    CURRENT device's stream, so that device is made current once, here - `--gpu_ids 1` then works as in the reference."""
    if getattr(args, "gpu_ids", None):
        torch.cuda.set_device(int(args.gpu_ids[0]))


LOSS_KEYS = ("img_dis_loss", "gt_dis_loss", "cycle_img_dis_loss", "img_gen_loss", "gt_gen_loss", "img_cycle_loss",
             "gt_cycle_loss", "lab_loss_CE", "lab_loss_MSE")


class semisuper_cycleGAN(object):
    def __init__(self, args, data_parallel=None):
        self.args = args
        _select_device(args)
        _check_bf16_widths(args)
        self.n_channels = CLASSES[args.dataset]                     # model.py:205-210
        C, ids = self.n_channels, args.gpu_ids
        drop = not args.no_dropout
        # construction order = the reference's (model.py:215-230)
        # The reference hard-codes 'deeplab' / 'pixel' and never reads --gen_net / --dis_net (model.py:215-222).  Opt-in
        # (SURVEY 8(f) N4): --honour_nets 1 builds what the flags name; --variants enables loss terms that are commented
        # out in the reference (l1_cycle: model.py:453; lab_gt_dis: :439,:447).  Defaults reproduce the reference.
        honour = bool(getattr(args, "honour_nets", 0))
        gen = args.gen_net if honour else 'deeplab'
        dis = args.dis_net if honour else 'pixel'
        self.variants = set(v for v in str(getattr(args, "variants", "") or "").split(",") if v)
        unknown = self.variants - {"l1_cycle", "lab_gt_dis", "gauss_noise", "perceptual"}
        if unknown:
            raise ValueError("unknown --variants %s" % sorted(unknown))
        # model.py:281,486-488: `if torch.rand(1) < 0.0` never fires, but draws one number from torch's CPU generator per
        # step (the stream DataLoader shuffling also draws from).  The draw is kept; the variant raises the threshold to 1.
        self.gauss_noise = utils.GaussianNoise(sigma=0.2)
        self.noise_prob = 1.0 if "gauss_noise" in self.variants else 0.0
        self.Gis = define_Gen(input_nc=C, output_nc=3, ngf=args.ngf, netG=gen, norm=args.norm, use_dropout=drop, gpu_ids=ids)
        self.Gsi = define_Gen(input_nc=3, output_nc=C, ngf=args.ngf, netG=gen, norm=args.norm, use_dropout=drop, gpu_ids=ids)
        self.Di = define_Dis(input_nc=3, ndf=args.ndf, netD=dis, n_layers_D=3, norm=args.norm, gpu_ids=ids)
        self.Ds = define_Dis(input_nc=C, ndf=args.ndf, netD=dis, n_layers_D=3, norm=args.norm, gpu_ids=ids)
        self.old_Gis = define_Gen(input_nc=C, output_nc=3, ngf=args.ngf, netG='resnet_9blocks', norm=args.norm, use_dropout=drop, gpu_ids=ids)
        self.old_Gsi = define_Gen(input_nc=3, output_nc=C, ngf=args.ngf, netG='resnet_9blocks_softmax', norm=args.norm, use_dropout=drop, gpu_ids=ids)
        self.old_Di = define_Dis(input_nc=3, ndf=args.ndf, netD='pixel', n_layers_D=3, norm=args.norm, gpu_ids=ids)
        if args.dataset == 'voc2012':                               # model.py:251-257
            try:
                ck = utils.load_checkpoint('./ckpt_for_Arnab_loss.ckpt')
                self.old_Gis.load_state_dict(ck['Gis'])
                self.old_Gsi.load_state_dict(ck['Gsi'])
            except Exception:
                print('**There is an error in loading the ckpt_for_Arnab_loss**')
        utils.print_networks([self.Gis, self.Gsi, self.Di, self.Ds], ['Gis', 'Gsi', 'Di', 'Ds'])

        self.crop = (args.crop_height, args.crop_width)             # nn.Upsample(..., align_corners=True), model.py:268
        self.running_metrics_val = utils.runningScore(C, args.dataset)
        self.as_written = getattr(args, "as_written", True)         # keep the reference's unused forwards
        self.fork_forward = getattr(args, "fork_forward", True)     # two stream lanes for the trainable generator passes
        self.stack_gsi = getattr(args, "stack_gsi", True)           # the two independent Gsi passes as one grouped-BN pass
        # D step on its own stream: it then overlaps the NEXT step's generator forwards (which read no discriminator
        # weight until :431).  Opt-in, because the three discriminator losses a step returns are then produced on that
        # stream: callers read them after `sync_losses()` (train() and bench.py do).
        self.overlap_d = bool(getattr(args, "overlap_d", False))
        self.dp = data_parallel

        self.g_optimizer = FusedAdam(itertools.chain(self.Gis.parameters(), self.Gsi.parameters()), lr=args.lr, betas=(0.5, 0.999))
        self.d_optimizer = FusedAdam(itertools.chain(self.Di.parameters(), self.Ds.parameters()), lr=args.lr, betas=(0.5, 0.999))
        if self.dp is not None:
            self.dp.attach(self.g_optimizer, self.d_optimizer, [self.Gis, self.Gsi, self.Di, self.Ds, self.old_Gis, self.old_Gsi, self.old_Di])
        lam = utils.LambdaLR(args.epochs, 0, args.decay_epoch).step
        self.g_lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.g_optimizer, lr_lambda=lam)
        self.d_lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.d_optimizer, lr_lambda=lam)

        self.pools = [utils.Sample_from_Pool() for _ in range(3)]   # new_img_fake / img_fake / gt_fake, model.py:350-352

        if not os.path.isdir(args.checkpoint_dir):
            os.makedirs(args.checkpoint_dir, exist_ok=True)
        try:                                                        # model.py:298-311
            ck = utils.load_checkpoint('%s/latest_semisuper_cycleGAN.ckpt' % (args.checkpoint_dir))
            self.start_epoch = ck['epoch']
            for k in ('Di', 'Ds', 'Gis', 'Gsi'):
                getattr(self, k).load_state_dict(ck[k])
            self.d_optimizer.load_state_dict(ck['d_optimizer'])
            self.g_optimizer.load_state_dict(ck['g_optimizer'])
            self.best_iou = ck['best_iou']
        except Exception:
            print(' [*] No checkpoint!')
            self.start_epoch = 0
            self.best_iou = -100

    # ------------------------------------------------------------------------------------------ one iteration
    def _mark(self, name, stream=None):
        """Diagnostic (SSCG_PHASE_EVENTS=1): a timed event on `stream` (default: the current one) + the host's clock, under `name`."""
        if not _PHASES:
            return
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self.phase_marks.append((name, ev, time.perf_counter()))

    def interp(self, x):
        return F.upsample_bilinear(x, self.crop)

    def step(self, l_img, l_gt, unl_img):
        """One G step + one D step (model.py:376-542).  Returns the nine losses as 0-dim device tensors."""
        a, C = self.args, self.n_channels
        self.phase_marks = []
        self._mark("step start")
        F.set_side_priority(F.side_priority_for(l_img.shape[0] * l_img.shape[2] * l_img.shape[3]))      # (the process's first step decides)
        if self.overlap_d and _DBG_TOPWAIT:     # bisection aid: these streams wait for the previous D step before anything of this step
            F.debug_wait_for_d(l_img.device, _DBG_TOPWAIT)
        # ---- generators (model.py:376-474)
        set_grad([self.Di, self.Ds, self.old_Di], False)
        set_grad([self.old_Gsi, self.old_Gis], False)
        self.g_optimizer.zero_grad()
        if self.dp is not None:
            self.dp.begin_backward(self.g_optimizer)     # SSCG_DP_BUCKETS: buckets of the gradient arena go out as the backward fills them
        # the generators' operand copies (bf16 shadow / split planes) exist and are current before any lane reads them: a no-op
        # after the first step (the Adam kernel rewrites them); the discriminators' follow at :431, behind their own update
        self.g_optimizer.ensure_operand_copies()
        labels = l_gt.reshape(l_gt.shape[0], l_gt.shape[2], l_gt.shape[3])          # l_gt.squeeze(1)
        onehot_gt = make_one_hot(l_gt, a.dataset, a.gpu_ids)

        # The frozen generators depend only on the input images and feed only the D step: their forward runs
        # on the side stream, concurrently with the DeepLab forwards/backwards below.
        def frozen_branch():
            with torch.no_grad():
                if self.as_written:
                    # :418-423 runs old_Gsi/old_Gis on unl_img (used) and on l_img (results never used): both batches
                    # in one pass (per-sample InstanceNorm; batch_groups keeps a BatchNorm variant equivalent too)
                    self._mark("frozen: start")
                    with arch.batch_groups(2):
                        both = self.old_Gis(F.softmax2d(self.old_Gsi(torch.cat([unl_img, l_img], 0))))
                    self._mark("frozen: end")
                    return both[:unl_img.shape[0]]
                fake = F.softmax2d(self.old_Gsi(unl_img))                            # :418,421
                return self.old_Gis(fake)                                            # :422
        resnet_recon_img = F.run_on_side_stream(l_img.device, (unl_img, l_img), frozen_branch) if _FROZEN_AT == "early" else None
        dev = l_img.device
        fork = F.SideStream.enabled and self.fork_forward
        self._wait_operand_copies(torch.cuda.current_stream(dev))
        if fork:
            # Two lanes: Gis(onehot) -> Gsi(fake_img) on the fork stream, Gsi(unl) -> Gsi(l_img) -> Gis(fake_gt) here.
            # The reference's order of BN running-stat updates (Gis: :385 before :408; Gsi: :386, :387 before :410) is kept
            # with two events; autograd runs every backward op on its forward's stream, so the backward chains fork too.
            # (the generators' copies only: the discriminators' update of the previous step may still be in flight on
            # the D stream - rebuilding THEIR copies here would read half-updated weights and free copies still being read)
            F.refresh_transposed_weights((p for net in (self.Gis, self.Gsi) for p in net.parameters() if p.dim() == 4),
                                         all_users=False)
            main, lane = torch.cuda.current_stream(dev), F.ForkStream.get(dev)
            lane.wait_stream(main)
            with torch.cuda.stream(lane):
                self._mark("fork Gis(onehot): start")
                fake_img = self.interp(self.Gis(onehot_gt))                          # :385,390
                self._mark("fork Gis(onehot): end")
                gis_first = torch.cuda.Event()
                gis_first.record(lane)
            onehot_gt.record_stream(lane)
        else:
            fake_img = self.interp(self.Gis(onehot_gt))                              # :385,390
        fake_img_all = None
        if self.stack_gsi:
            # Gsi(unl_img) and Gsi(l_img) (:386-387) as ONE pass over both batches: every BatchNorm normalises the two
            # halves separately and advances its running statistics twice, in order (arch.batch_groups) - the same
            # arithmetic, on convolutions with twice the rows and half the launches.
            self._mark("main Gsi(unl, l): start")
            with arch.batch_groups(2):
                both = self.Gsi(torch.cat([unl_img, l_img], 0))
            self._mark("main Gsi(unl, l): end")
            fake_logits, lab_logits = F.split_batch(both, 2)
        else:
            fake_logits = self.Gsi(unl_img)                                          # :386
            lab_logits = self.Gsi(l_img)                                             # :387
        if fork:
            gsi_second = torch.cuda.Event()
            gsi_second.record(main)
        # :391-392 (interp), :398 (CE of the resized logits), :401-402 (their softmax) from the low-resolution logits in one pass each:
        # the resized [B, C, crop] logits are never written (functional.UpsampleHeadFn)
        lab_gt, lab_loss_CE = F.upsample_softmax_ce(lab_logits, self.crop, labels)
        fake_gt, _ = F.upsample_softmax_ce(fake_logits, self.crop)
        if fork:
            main.wait_event(gis_first)
        # (Gis(onehot_gt) and Gis(fake_gt) as ONE grouped-BatchNorm pass on the main lane was built in round 4 and measured slower than
        # the two lanes; deleted in round 6)
        self._mark("main Gis(fake_gt): start")
        recon_img = self.interp(self.Gis(fake_gt))                               # :408,413
        self._mark("main Gis(fake_gt): end")
        # :409 - output unused by the reference, but it advances Gis' BN running stats (after those of the :408 pass
        # above, which the side stream waits for).  Nothing reads the result: it runs beside the critical path and is
        # joined before the optimiser touches Gis' weights.
        lab_det = lab_gt.detach()

        def unused_pass():
            with torch.no_grad():
                self._mark("side Gis(lab_gt) unused: start")
                self.Gis(lab_det)
                self._mark("side Gis(lab_gt) unused: end")
        if _UNUSED_AT == "early":
            F.run_on_side_stream(l_img.device, (lab_det,), unused_pass)
        if fork:
            with torch.cuda.stream(lane):
                lane.wait_event(gsi_second)
                fake_img_all = fake_img
                fake_img, fake_img_d, fake_img_l1 = F.split(fake_img, 3)             # consumers: Gsi, Di, L1
                self._mark("fork Gsi(fake_img): start")
                recon_logits = self.Gsi(fake_img)                                    # :410
                self._mark("fork Gsi(fake_img): end")
            main.wait_stream(lane)
            fake_img.record_stream(main)
            recon_logits.record_stream(main)
        else:
            fake_img_all = fake_img
            fake_img, fake_img_d, fake_img_l1 = F.split(fake_img, 3)                 # consumers: Gsi, Di, L1
            recon_logits = self.Gsi(fake_img)                                        # :410
        extra_terms, extra_weights, extras = [], [], {}
        if "l1_cycle" in self.variants:                                              # :453 (commented out in the reference)
            recon_img, recon_img_l1 = F.split(recon_img, 2)
            extras["img_cycle_l1"] = F.l1_loss(recon_img_l1, unl_img)
            extra_terms.append(extras["img_cycle_l1"])
            extra_weights.append(a.lamda_img)
        if "perceptual" in self.variants:                                            # :454,:462 (commented out; weights main.py:23,28)
            if getattr(self, "vgg", None) is None:
                self.vgg = utils.Vgg16(requires_grad=False, weights=getattr(a, "vgg_weights", None)).to(dev)
            recon_img, recon_img_p = F.split(recon_img, 2)
            fake_img_l1, fake_img_p = F.split(fake_img_l1, 2)
            extras["img_cycle_loss_perceptual"] = utils.perceptual_loss(recon_img_p, unl_img, a.gpu_ids, self.vgg)
            extras["lab_loss_perceptual"] = utils.perceptual_loss(fake_img_p, l_img, a.gpu_ids, self.vgg)
            extra_terms += [extras["img_cycle_loss_perceptual"], extras["lab_loss_perceptual"]]
            extra_weights += [a.lamda_perceptual, a.lab_perceptual_weight]
        if "lab_gt_dis" in self.variants:                                            # :439,:447 (commented out)
            extras["gt_label_gen_loss"] = F.mse_const(self.Ds(lab_gt), 1.0)
            extra_terms.append(extras["gt_label_gen_loss"])
            extra_weights.append(a.adversarial_weight)
        if _UNUSED_AT == "mid":
            F.run_on_side_stream(l_img.device, (lab_det,), unused_pass)
        if _FROZEN_AT == "mid":
            resnet_recon_img = F.run_on_side_stream(l_img.device, (unl_img, l_img), frozen_branch)
        if self.overlap_d:      # the previous step's discriminator update (on the D stream) must have landed
            torch.cuda.current_stream(dev).wait_stream(F.d_stream(dev))
        self.d_optimizer.ensure_operand_copies()
        fake_img_dis = self.Di(fake_img_d)                                           # :431
        resnet_fake_img_dis = self.old_Di(recon_img)                                 # :432
        fake_gt_onehot, _ = F.argmax_onehot(fake_gt.detach())                        # :435-437 (no gradient path)
        fake_gt_dis = self.Ds(fake_gt_onehot)                                        # :438
        img_gen_loss = F.mse_const(fake_img_dis, 1.0)                                # :445
        gt_gen_loss = F.mse_const(fake_gt_dis, 1.0)                                  # :446
        img_cycle_loss = F.mse_const(resnet_fake_img_dis, 1.0)                       # :452
        _, gt_cycle_loss = F.upsample_softmax_ce(recon_logits, self.crop, labels, want_soft=False)    # :415 (interp), :455
        lab_loss_MSE = F.l1_loss(fake_img_l1, l_img)                                 # :461
        # :464-468  gen_loss = CE_w*CE + MSE_w*L1 + adv_w*(img_gen + gt_gen) + img_cycle + lamda_gt*gt_cycle
        gen_loss = F.weighted_sum(
            [lab_loss_CE, lab_loss_MSE, img_gen_loss, gt_gen_loss, img_cycle_loss, gt_cycle_loss] + extra_terms,
            [a.lab_CE_weight, a.lab_MSE_weight, a.adversarial_weight, a.adversarial_weight, 1.0, a.lamda_gt] + extra_weights)
        self._mark("main: backward start")
        if _PHASES:     # when each pass's backward STARTS (the gradient at its output arrives): host clock + an event on the stream the engine runs it on
            for name, t in (("bwd: d recon_logits (fork Gsi(fake_img) backward starts)", recon_logits), ("bwd: d recon_img (main Gis(fake_gt) backward starts)", recon_img),
                            ("bwd: d fake_img summed (fork Gis(onehot) backward starts)", fake_img_all), ("bwd: d fake_gt (main head + Gsi(unl, l) backward starts)", fake_gt),
                            ("bwd: d lab_logits / fake_logits reached", fake_logits)):
                if t is not None and t.requires_grad:
                    t.register_hook(lambda g, name=name: self._mark(name))
        F.backward(gen_loss)                                                         # :472
        self._mark("main: backward issued / main's part done")
        if _UNUSED_AT == "late":
            F.run_on_side_stream(l_img.device, (lab_det,), unused_pass)
        if _FROZEN_AT == "late":
            resnet_recon_img = F.run_on_side_stream(l_img.device, (unl_img, l_img), frozen_branch)
        F.ForkStream.join(l_img.device)
        F.SideStream.join(l_img.device)            # side stream: weight gradients + frozen generators are complete
        self._mark("main: all lanes joined")
        resnet_recon_img.record_stream(torch.cuda.current_stream(l_img.device))
        g_works = None
        if self.dp is not None:
            g_works = self.dp.sync_grads_async(self.g_optimizer)     # overlaps the D step; the update is applied below
        else:
            self.g_optimizer.step()                                                  # :474
            # the transposed weight copies the next step's data gradients read: rebuilt beside the discriminator step.  The
            # list holds the discriminators' copies too (stale since the previous D update): every stream that reads a copy
            # - the discriminator step below, the next step's lanes - first waits for this event.
            F.run_on_side_stream(l_img.device, (), F.refresh_transposed_weights)
            if F.SideStream.enabled:
                self._copies_ready = torch.cuda.Event()
                self._copies_ready.record(F.SideStream.get(dev, 0))

        # ---- discriminators (model.py:477-542)
        if self.overlap_d:
            main_s, d_s = torch.cuda.current_stream(dev), F.d_stream(dev)
            d_s.wait_stream(main_s)
            self._wait_operand_copies(d_s)
            for t in (recon_img, fake_img, fake_gt, unl_img, onehot_gt, resnet_recon_img):
                t.record_stream(d_s)
            with torch.cuda.stream(d_s):
                d_vals = self._d_step(a, recon_img, fake_img, fake_gt, unl_img, onehot_gt, resnet_recon_img)
            if self.dp is not None:
                self.dp.wait(g_works)
                self.g_optimizer.step()                                              # :474 (deferred past the all-reduce)
                self._refresh_generator_copies(dev)
        else:
            self._wait_operand_copies(torch.cuda.current_stream(dev))
            d_vals = self._d_step(a, recon_img, fake_img, fake_gt, unl_img, onehot_gt, resnet_recon_img)
            if self.dp is not None:
                self.dp.wait(g_works)
                self.g_optimizer.step()                                              # :474 (deferred: no D-step op reads G weights)
                self._refresh_generator_copies(dev)
        self._mark("main: step issued (G update queued)")
        if self.overlap_d:
            self._mark("D stream: D step done", F.d_stream(dev))
        vals = d_vals + (img_gen_loss, gt_gen_loss, img_cycle_loss, gt_cycle_loss, lab_loss_CE, lab_loss_MSE)
        out = {k: v.detach() for k, v in zip(LOSS_KEYS, vals)}
        out.update({k: v.detach() for k, v in extras.items()})
        return out

    def second_pass(self, fake_img, fake_gt, l_gt, unl_img):
        """Diagnostic twin of the part of `step` that sits two DeepLab passes deep, on GIVEN first-pass outputs (teacher forcing,
        SURVEY App. D.3): the same modules, fused head and loss kernels as step() uses for model.py:408,410,413,415,432,452,455 and for the
        discriminator step's :418-422,501-502,527-528,534 - but no optimiser, no pools.  Returns img_cycle_loss, gt_cycle_loss,
        cycle_img_dis_loss (device scalars), d(img_cycle_loss)/d(fake_gt), d(gt_cycle_loss)/d(fake_img) and recon_img.  The parity
        tests feed it the fp64 oracle's first-pass outputs, so that the three losses the step can only bound statistically (the
        first pass's fp32 noise is amplified by the second) are held to 1e-3 here."""
        set_grad([self.Di, self.Ds, self.old_Di, self.old_Gsi, self.old_Gis], False)
        self.g_optimizer.zero_grad()
        self.g_optimizer.ensure_operand_copies()
        self.d_optimizer.ensure_operand_copies()
        labels = l_gt.reshape(l_gt.shape[0], l_gt.shape[2], l_gt.shape[3])
        x_gt = fake_gt.detach().clone().requires_grad_(True)
        x_img = fake_img.detach().clone().requires_grad_(True)
        recon_img = self.interp(self.Gis(x_gt))                                          # :408,413
        img_cycle_loss = F.mse_const(self.old_Di(recon_img), 1.0)                        # :432,452
        _, gt_cycle_loss = F.upsample_softmax_ce(self.Gsi(x_img), self.crop, labels, want_soft=False)    # :410,415,455
        F.backward(F.weighted_sum([img_cycle_loss, gt_cycle_loss], [1.0, 1.0]))          # (each term reaches one of the two inputs only)
        F.SideStream.join(l_gt.device)
        with torch.no_grad():
            resnet_recon_img = self.old_Gis(F.softmax2d(self.old_Gsi(unl_img)))          # :418,421,422
            r_c = F.mse_const(self.old_Di(resnet_recon_img), 1.0)                        # :501,527
            f_c = F.mse_const(self.old_Di(recon_img.detach()), 0.0)                      # :502,528 (the pool returns the current item)
            cycle_img_dis_loss = F.weighted_sum([r_c, f_c], [1.0, 1.0])                  # :534
        self.g_optimizer.zero_grad()
        return dict(img_cycle_loss=img_cycle_loss.detach(), gt_cycle_loss=gt_cycle_loss.detach(), cycle_img_dis_loss=cycle_img_dis_loss,
                    d_fake_gt=x_gt.grad, d_fake_img=x_img.grad, recon_img=recon_img.detach())

    def _refresh_generator_copies(self, dev):
        """Data parallel: the generator update lands AFTER the discriminator step was queued, so the operand copies of the
        GENERATORS' weights are rebuilt here, on side lane 0, off the next step's critical path (the discriminators' copies are
        rebuilt lazily by their own step: its stream may still be reading the old ones)."""
        gen_w = [p for net in (self.Gis, self.Gsi) for p in net.parameters() if p.dim() == 4]
        F.run_on_side_stream(dev, (), lambda: F.refresh_transposed_weights(gen_w, all_users=False))
        if F.SideStream.enabled:
            self._copies_ready = torch.cuda.Event()
            self._copies_ready.record(F.SideStream.get(dev, 0))

    def _wait_operand_copies(self, stream):
        """Order `stream` behind the side lane that rebuilt the transposed / bf16 operand copies of the weights."""
        ev = getattr(self, "_copies_ready", None)
        if ev is not None:
            stream.wait_event(ev)

    def sync_losses(self):
        """Make the current stream wait for the discriminator stream (overlap_d): call before reading a step's losses."""
        if self.overlap_d:
            dev = torch.device("cuda", self.args.gpu_ids[0])
            torch.cuda.current_stream(dev).wait_stream(F.d_stream(dev))

    def _d_step(self, a, recon_img, fake_img, fake_gt, unl_img, onehot_gt, resnet_recon_img):
        l_img = unl_img
        set_grad([self.Di, self.Ds], True)
        set_grad([self.old_Di], self.as_written)   # old_Di is in no optimiser: its wgrad only exists in the as-written graph
        self.d_optimizer.zero_grad()
        if torch.rand(1) < self.noise_prob:                                          # :486 (threshold 0.0 in the reference)
            fake_img = self.gauss_noise(fake_img.detach())                           # :487
            fake_gt = self.gauss_noise(fake_gt.detach())                             # :488
        recon_img_p = self.pools[0]([recon_img.detach()])[0]                         # :490
        fake_img_p = self.pools[1]([fake_img.detach()])[0]                           # :491
        fake_gt_p = self.pools[2]([fake_gt.detach()])[0]                             # :493
        if self.overlap_d:      # a pool may hand back a tensor of an earlier step (allocated on the main stream)
            for t in (recon_img_p, fake_img_p, fake_gt_p):
                t.record_stream(torch.cuda.current_stream(t.device))
        unl_img_dis = self.Di(unl_img)                                               # :499
        fake_img_dis = self.Di(fake_img_p)                                           # :500
        resnet_recon_img_dis = self.old_Di(resnet_recon_img)                         # :501
        resnet_fake_img_dis = self.old_Di(recon_img_p)                               # :502
        real_gt_dis = self.Ds(onehot_gt)                                             # :506-507
        fake_gt_onehot, _ = F.argmax_onehot(fake_gt_p)                               # :509-511
        fake_gt_dis = self.Ds(fake_gt_onehot)                                        # :512
        r_i, f_i = F.mse_const(unl_img_dis, 1.0), F.mse_const(fake_img_dis, 0.0)     # :521-522
        r_g, f_g = F.mse_const(real_gt_dis, 1.0), F.mse_const(fake_gt_dis, 0.0)      # :523-524
        r_c, f_c = F.mse_const(resnet_recon_img_dis, 1.0), F.mse_const(resnet_fake_img_dis, 0.0)  # :527-528
        img_dis_loss = F.weighted_sum([r_i, f_i], [0.5, 0.5])                        # :531
        gt_dis_loss = F.weighted_sum([r_g, f_g], [0.5, 0.5])                         # :532
        cycle_img_dis_loss = F.weighted_sum([r_c, f_c], [1.0, 1.0])                  # :534
        dis_loss = F.weighted_sum([img_dis_loss, gt_dis_loss, cycle_img_dis_loss],
                                  [a.discriminator_weight, a.discriminator_weight, 1.0])  # :538
        F.backward(dis_loss)                                                         # :539
        F.SideStream.join(l_img.device)
        if self.dp is not None:
            self.dp.sync_grads(self.d_optimizer)
        self.d_optimizer.step()                                                      # :542
        return (img_dis_loss, gt_dis_loss, cycle_img_dis_loss)

    # ------------------------------------------------------------------------------------------ evaluation (model.py:555-574)
    @torch.no_grad()
    def evaluate(self, val_loader):
        self.Gsi.eval()
        self.Gis.eval()
        self.running_metrics_val.reset()
        for val_img, val_gt, _ in val_loader:
            val_img, val_gt = utils.cuda([val_img, val_gt], self.args.gpu_ids)
            outputs = F.softmax2d(self.interp(self.Gsi(val_img)))
            self.running_metrics_val.update_device(val_gt.squeeze(1), F.argmax_index(outputs))   # :566-569 without the host round trip
        score, class_iou = self.running_metrics_val.get_scores()
        self.Gsi.train()
        self.Gis.train()
        return score["Mean IoU : \t"], class_iou

    # ------------------------------------------------------------------------------------------ epoch loop
    def train(self, args, loaders=None, max_steps=None, log_every=1, writer=None):
        """Epoch loop of model.py:359-660.  `loaders` = (labeled, unlabeled, val) iterables yielding the
        reference's dataset tuples (img f32[B,3,H,W], gt i64[B,1,H,W], name); None builds synthetic ones
        (the real datasets / transforms of data_utils are outside this build's scope)."""
        if loaders is None:
            from .data import synthetic_loaders
            loaders = synthetic_loaders(args, self.n_channels, rank=self.dp.rank if self.dp is not None else 0)
        labeled_loader, unlabeled_loader, val_loader = loaders
        rank0 = self.dp is None or self.dp.rank == 0
        done = 0
        history = []
        for epoch in range(self.start_epoch, args.epochs):
            if rank0:
                print('learning rate = %.7f' % self.g_optimizer.param_groups[0]['lr'])
            self.Gsi.train()
            self.Gis.train()
            n_it = min(len(labeled_loader), len(unlabeled_loader))
            for i, ((l_img, l_gt, _), (unl_img, _, _)) in enumerate(zip(labeled_loader, unlabeled_loader)):
                l_img, unl_img, l_gt = utils.cuda([l_img, unl_img, l_gt], args.gpu_ids)
                losses = self.step(l_img, l_gt, unl_img)
                done += 1
                if (i % log_every == 0) and rank0:
                    self.sync_losses()
                    vals = torch.stack([losses[k] for k in LOSS_KEYS]).cpu().tolist()   # the only host sync of the step
                    rec = dict(zip(LOSS_KEYS, vals))
                    history.append(rec)
                    print("Epoch: (%3d) (%5d/%5d) | Dis Loss:%.2e | Unlab Gen Loss:%.2e | Lab Gen loss:%.2e" % (
                        epoch, i + 1, n_it, rec["img_dis_loss"] + rec["gt_dis_loss"],
                        args.adversarial_weight * (rec["img_gen_loss"] + rec["gt_gen_loss"]) + rec["img_cycle_loss"] + rec["gt_cycle_loss"] * args.lamda_gt,
                        args.lab_CE_weight * rec["lab_loss_CE"] + args.lab_MSE_weight * rec["lab_loss_MSE"]))
                    if writer is not None:
                        it = len(labeled_loader) * epoch + i
                        writer.add_scalars('Dis Loss', {k: rec[k] for k in LOSS_KEYS[0:3]}, it)
                        writer.add_scalars('Unlabelled Loss', {k: rec[k] for k in LOSS_KEYS[3:7]}, it)
                        writer.add_scalars('Labelled Loss', {k: rec[k] for k in LOSS_KEYS[7:9]}, it)
                if max_steps is not None and done >= max_steps:
                    return history
            self.sync_losses()                       # the last discriminator update is visible to what follows on this stream
            if val_loader is not None:
                miou, class_iou = self.evaluate(val_loader)
                if rank0:
                    print("The mIoU for the epoch is: ", miou)
                if miou >= self.best_iou and rank0:                                 # model.py:641-655
                    self.best_iou = miou
                    utils.save_checkpoint({'epoch': epoch + 1, 'Di': self.Di.state_dict(), 'Ds': self.Ds.state_dict(),
                                           'Gis': self.Gis.state_dict(), 'Gsi': self.Gsi.state_dict(),
                                           'd_optimizer': self.d_optimizer.state_dict(), 'g_optimizer': self.g_optimizer.state_dict(),
                                           'best_iou': self.best_iou, 'class_iou': class_iou},
                                          '%s/latest_semisuper_cycleGAN.ckpt' % (args.checkpoint_dir))
            self.g_lr_scheduler.step()                                              # model.py:659-660
            self.d_lr_scheduler.step()
        return history


class supervised_model(object):
    """DeepLab Gsi + CrossEntropy + Adam(0.9, 0.999) (model.py:33-199; BASELINE config 1)."""

    def __init__(self, args, data_parallel=None):
        self.args = args
        _select_device(args)
        _check_bf16_widths(args)
        self.n_channels = CLASSES[args.dataset]
        self.Gsi = define_Gen(input_nc=3, output_nc=self.n_channels, ngf=args.ngf, netG='deeplab', norm=args.norm,
                              use_dropout=not args.no_dropout, gpu_ids=args.gpu_ids)
        utils.print_networks([self.Gsi], ['Gsi'])
        self.crop = (args.crop_height, args.crop_width)
        self.gsi_optimizer = FusedAdam(self.Gsi.parameters(), lr=args.lr, betas=(0.9, 0.999))   # model.py:69
        self.dp = data_parallel
        if self.dp is not None:
            self.dp.attach_one(self.gsi_optimizer, [self.Gsi])
        self.running_metrics_val = utils.runningScore(self.n_channels, args.dataset)
        if not os.path.isdir(args.checkpoint_dir):
            os.makedirs(args.checkpoint_dir, exist_ok=True)
        try:
            ck = utils.load_checkpoint('%s/latest_supervised_model.ckpt' % (args.checkpoint_dir))
            self.start_epoch = ck['epoch']
            self.Gsi.load_state_dict(ck['Gsi'])
            self.gsi_optimizer.load_state_dict(ck['gsi_optimizer'])
            self.best_iou = ck['best_iou']
        except Exception:
            print(' [*] No checkpoint!')
            self.start_epoch = 0
            self.best_iou = -100

    def step(self, l_img, l_gt):
        """model.py:120-143."""
        self.gsi_optimizer.zero_grad()
        _, loss = F.upsample_softmax_ce(self.Gsi(l_img), self.crop, l_gt.reshape(l_gt.shape[0], l_gt.shape[2], l_gt.shape[3]),
                                        want_soft=False)
        F.backward(loss)
        if self.dp is not None:
            F.SideStream.join(l_img.device)
            self.dp.sync_grads(self.gsi_optimizer)
        self.gsi_optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def evaluate(self, val_loader):
        """model.py:145-162.  The reference interpolates to a hard-coded 512x512 (`interp_val`, model.py:63,152), which only
        works for a 512x512 crop; the crop size is used here, as the semi-supervised driver does."""
        self.Gsi.eval()
        self.running_metrics_val.reset()
        for val_img, val_gt, _ in val_loader:
            val_img, val_gt = utils.cuda([val_img, val_gt], self.args.gpu_ids)
            outputs = F.softmax2d(F.upsample_bilinear(self.Gsi(val_img), self.crop))
            self.running_metrics_val.update_device(val_gt.squeeze(1), F.argmax_index(outputs))
        score, class_iou = self.running_metrics_val.get_scores()
        self.running_metrics_val.reset()
        self.Gsi.train()
        return score["Mean IoU : \t"], class_iou

    def train(self, args, loaders=None, max_steps=None):
        rank = self.dp.rank if self.dp is not None else 0
        if loaders is None:
            from .data import synthetic_loaders
            loaders = synthetic_loaders(args, self.n_channels, rank=rank)
        labeled_loader = loaders[0]
        val_loader = loaders[2] if len(loaders) > 2 else None
        history, done = [], 0
        for epoch in range(self.start_epoch, args.epochs):
            self.Gsi.train()
            for i, (l_img, l_gt, _) in enumerate(labeled_loader):
                l_img, l_gt = utils.cuda([l_img, l_gt], args.gpu_ids)
                loss = float(self.step(l_img, l_gt))
                history.append(loss)
                if rank == 0:
                    print("Epoch: (%3d) (%5d/%5d) | Crossentropy Loss:%.2e" % (epoch, i + 1, len(labeled_loader), loss))
                done += 1
                if max_steps is not None and done >= max_steps:
                    return history
            if val_loader is not None:                                              # model.py:145-197
                miou, class_iou = self.evaluate(val_loader)
                if rank == 0:
                    print("The mIoU for the epoch is: ", miou)
                if miou >= self.best_iou and rank == 0:
                    self.best_iou = miou
                    utils.save_checkpoint({'epoch': epoch + 1, 'Gsi': self.Gsi.state_dict(),
                                           'gsi_optimizer': self.gsi_optimizer.state_dict(), 'best_iou': self.best_iou,
                                           'class_iou': class_iou}, '%s/latest_supervised_model.ckpt' % (self.args.checkpoint_dir))
        return history

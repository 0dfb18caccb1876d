#!/usr/bin/env python
"""Throughput of the hot path: full G+D training steps of the semi-supervised CycleGAN on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]
    (N > 1 without a launcher: bench.py starts its own N ranks through torch.distributed.run)

--config 2 (default) = BASELINE.json configs[1], the configuration the metric is quoted on: VOC2012 21-class, 256x256,
    semisupervised_cycleGAN, batch 8 per GPU, fp32 tensors.  `--dtype f32` (the default of this configuration) contracts the
    heavy convolutions with the fp32-accurate split contraction on the bf16 matrix cores (conv_split.hip; at or below the exact
    fp32 MFMA kernel's error against fp64 on every shape of the step); `--dtype f32x` is the exact fp32 MFMA, reported beside
    the headline as `f32_exact`.
--config 3 = BASELINE.json configs[2]: Cityscapes 20-class, 256x512, batch 16 per GPU, bf16 (bf16 activations and conv
    weight operands in HBM, fp32 master weights / statistics / losses).
--config 4 / 5 = the per-rank workloads of configs[3] (config 2 on each of 8 ranks) and configs[4] (Cityscapes 512x1024, global
    batch 32 = batch 4 per rank, bf16): meant for `--gpus 8`.
Random-init weights (the reference's N(0,0.02) init), synthetic image/label batches already resident in HBM.
One step = one iteration of /root/reference model.py:370-552 as written (all seven networks, both optimisers).
Weak scaling: every rank runs the per-GPU batch; `value` = N * batch * K / (max-over-ranks wall time of K steps).

Besides the contract line this prints, in the same JSON object:
  roofline     - the implicit-GEMM conv kernels measured live with HIP events on the launch stream during one
                 extra (untimed) step: algorithmic FLOP / kernel time against the MFMA peak of the arithmetic
                 (exact fp32: 157.3 TFLOP/s; bf16: 2500 TFLOP/s dense; split fp32: six bf16 piece products per fp32 product,
                 i.e. 2500 / 6 = 416.7 TFLOP/s of algorithmic work);
  cpu_baseline - oracle/ (the CPU restatement of the reference) timed on this box's host cores on a bounded
                 sample (rank 0, N = 1 only);
  secondary figures, never the headline: elided_dead_work (the step without the reference's unused forwards), bf16 and f32_exact
                 (the fp32 configuration in the bf16 arithmetic / with exact fp32 MFMA contractions), host_bound_case (64x64,
                 batch 2: the host's issue cost of a step; measured by `bench.py --host-bound-only` in a process of its own),
                 host_issue_ms_per_step.
"""
import argparse
import contextlib
import importlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "semi-supervised-segmentation-cyclegan_amd"

# /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 256 CU x 2.4 GHz; dense bf16 MFMA (no sparsity)
PEAK = {"f32": 157.3, "f32x": 157.3, "f32s": 157.3, "bf16": 2500.0, "bf16c": 2500.0}
SPLIT_PEAK = 2500.0 / 6.0       # algorithmic TFLOP/s ceiling of the split contraction: six bf16 MFMA products per fp32 product
# conv FLOP of one as-written step per labeled/unlabeled pair (BASELINE.md section 2 / SURVEY 8(d), forward hooks on every conv)
CONFIGS = {
    2: dict(dataset="voc2012", C=21, H=256, W=256, B=8, dtype="f32", tflop_per_pair=1.983,
            label="VOC2012 21-class 256x256 semisupervised_cycleGAN as-written G+D step"),
    3: dict(dataset="cityscapes", C=20, H=256, W=512, B=16, dtype="bf16", tflop_per_pair=3.914,
            label="Cityscapes 20-class 256x512 semisupervised_cycleGAN as-written G+D step"),
    # the per-rank workloads of the two 8-GPU configurations (run them with --gpus 8): configs[3] = config 2 on every rank
    # (global batch 64), configs[4] = Cityscapes 512x1024, global batch 32 = batch 4 per rank, bf16
    4: dict(dataset="voc2012", C=21, H=256, W=256, B=8, dtype="f32", tflop_per_pair=1.983,
            label="VOC2012 21-class 256x256 semisupervised_cycleGAN as-written G+D step"),
    5: dict(dataset="cityscapes", C=20, H=512, W=1024, B=4, dtype="bf16", tflop_per_pair=15.387,
            label="Cityscapes 20-class 512x1024 semisupervised_cycleGAN as-written G+D step"),
}
DTYPE_TEXT = {"f32": "fp32", "f32x": "fp32 tensors, exact fp32 MFMA contractions (v_mfma_f32_32x32x2_f32)", "bf16": "bf16 (bf16 activations + conv weight operands in HBM, fp32 accumulate / master weights / norm statistics / losses)",
              "bf16c": "bf16 conv contractions (fp32 tensors)",
              "f32s": "fp32 tensors, 3-piece split-bf16 contraction, fp32-accurate (six exact bf16 piece products per fp32 product on "
                      "the bf16 matrix cores, fp32 accumulation - forward, data and weight gradients; 21 / 20-channel stems and heads zero-padded onto the same "
                      "contraction, 1- / 3-channel ends on the exact fp32 MFMA)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2, help="BASELINE.json configuration (1-based index)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-elided", action="store_true")
    ap.add_argument("--no-small", action="store_true", help="skip the 64x64 batch-2 host-bound figure")
    ap.add_argument("--no-unblocked", action="store_true", help="skip host_issue_unblocked_ms (a few dry-run steps)")
    ap.add_argument("--host-bound-only", action="store_true", help="(internal) run the 64x64 batch-2 case alone and print its JSON object")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the configuration's)")
    ap.add_argument("--dtype", choices=["f32", "f32x", "f32s", "bf16", "bf16c"], default=None, help="default: the configuration's")
    ap.add_argument("--no-bf16", action="store_true", help="config 2: skip the secondary bf16 / exact-fp32 figures")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    dtype = a.dtype or cfg["dtype"]
    C, H, W = cfg["C"], cfg["H"], cfg["W"]
    bsz = a.batch or cfg["B"]

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a.gpus))          # `python bench.py --gpus N` launches its own N ranks
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a %d-rank job" % (a.gpus, world))
    md = importlib.import_module(PKG + ".model")
    F = importlib.import_module(PKG + ".functional")
    par = importlib.import_module(PKG + ".parallel")
    data = importlib.import_module(PKG + ".data")

    # SSCG_FORCE_DP=1 exercises the RCCL code path (init, broadcast, all-reduce) on a single rank
    dp = par.DataParallel() if (world > 1 or os.environ.get("SSCG_FORCE_DP")) else None
    rank = dp.rank if dp else 0
    local = dp.device_index if dp else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # before anything is built: per-rank batch >= 2, one physical device per rank, LOCAL_RANK = the device in use
    # (collective; a refusal costs seconds, not a warmed-up model)
    par.preflight(dp, bsz, shared_gpu_ok=bool(os.environ.get("SSCG_DP_SHARED_GPU")))

    import main as cli                   # the product CLI's own defaults (oracle/ is used by the cpu_baseline leg only)
    args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", cfg["dataset"], "--crop_height", str(H), "--crop_width", str(W),
                         "--batch_size", str(bsz), "--checkpoint_dir", "/tmp/sscg_bench_ckpt_%d" % rank, "--epochs", "400",
                         "--decay_epoch", "100", "--dtype", dtype])
    args.gpu_ids, args.as_written = [local], True
    args.overlap_d = os.environ.get("SSCG_OVERLAP_D", "1") == "1"   # the D step overlaps the next step's generator forwards
    torch.manual_seed(0)
    F.set_conv_precision(dtype)
    arith = F.get_conv_precision()          # "f32" names fp32 TENSORS; this is the contraction it resolves to ("f32" exact / "f32s" split)
    # The reference's own default is batch 2 (main.py:14) on small crops: there the step is bound by the host's issue rate, not by
    # the GPU.  One line beside the headline, single rank only: the same step at 64x64, batch 2, same dtype, in a PROCESS OF ITS OWN
    # (`bench.py --host-bound-only`) - what a `main.py --batch_size 2` run sees.  Inside this process the large model's registries
    # and allocator pools cost the small case 10-20 ms per step, and a process keeps ONE set of side lanes (functional.set_side_priority):
    # the small case wants them at normal priority, the configuration's own step at low priority.
    if a.host_bound_only:
        sargs = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", cfg["dataset"], "--crop_height", "64", "--crop_width", "64",
                              "--batch_size", "2", "--checkpoint_dir", "/tmp/sscg_bench_ckpt_small", "--dtype", dtype])
        sargs.gpu_ids, sargs.as_written, sargs.overlap_d = [local], True, args.overlap_d
        with contextlib.redirect_stdout(io.StringIO()):
            small = md.semisuper_cycleGAN(sargs)
        sl = list(data.SyntheticLoader(2, C, 64, 64, 14, 3, device=dev))
        su = list(data.SyntheticLoader(2, C, 64, 64, 14, 4, device=dev))
        for i in range(4):
            small.step(sl[i][0], sl[i][1], su[i][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 14):
            small.step(sl[i][0], sl[i][1], su[i][0])
        th = (time.perf_counter() - t0) / 10
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 10
        print(json.dumps({"workload": "the same step at 64x64, batch 2 (main.py:14 default batch), in a process of its own", "ms_per_step": round(1e3 * ts, 2),
                          "host_issue_ms_per_step": round(1e3 * th, 2), "value": round(2 / ts, 2), "unit": "img/s"}))
        return
    host_bound_case = None
    if world == 1 and not a.no_small:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-bound-only", "--config", str(a.config), "--dtype", dtype],
                               capture_output=True, text=True, timeout=900)
            host_bound_case = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:      # a secondary figure must not take the headline with it
            host_bound_case = {"error": "%s: %s" % (type(e).__name__, e)}

    with contextlib.redirect_stdout(io.StringIO()):
        model = md.semisuper_cycleGAN(args, data_parallel=dp)

    # synthetic batches, resident in HBM before the clock starts (rank-offset seeds)
    nb = a.warmup + a.steps + 1
    lab = list(data.SyntheticLoader(bsz, C, H, W, nb, 1 + 1000 * rank, device=dev))
    unl = list(data.SyntheticLoader(bsz, C, H, W, nb, 2 + 1000 * rank, device=dev))

    def run(i):
        return model.step(lab[i][0], lab[i][1], unl[i][0])

    def timed(first, count):
        if dp:
            dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for i in range(first, first + count):
            out = run(i)
        host[0] = (time.perf_counter() - t0) / count      # the host thread is free again: issue cost of a step (incl. back-pressure)
        torch.cuda.synchronize()
        own[0] = (time.perf_counter() - t0) / count       # this rank's own time, before it waits for the others
        if dp:
            dp.barrier()
        return par.max_over_ranks(time.perf_counter() - t0), out

    host, own = [0.0], [0.0]

    # The three image pools (utils.Sample_from_Pool, model.py:350-352) keep the last 50 fake images each: for a run's first 50 steps every
    # step retains (3 + 3 + C) * B * H * W elements more than the one before - by the reference's design, and inside any timed region
    # shorter than that.  The blocks are handed to torch's caching allocator HERE (allocated at the pool items' exact sizes and freed
    # again), so that filling the pools takes cached blocks and no step of the timed region calls hipMalloc for them.
    esz = 2 if dtype == "bf16" else 4
    warm = [torch.empty(bsz * ch * H * W * esz, dtype=torch.uint8, device=dev) for _ in range(50) for ch in (3, 3, C)]
    del warm
    for i in range(a.warmup):
        run(i)
    # device memory must not grow inside the timed region (hipMalloc serialises the device; at N > 1 it also stalls the ranks that
    # wait in the all-reduce): torch's caching allocator holds every buffer of the step (the library's workspaces are torch
    # tensors too), so `reserved` before and after the K steps says whether the region allocated.  A region that grew is timed again
    # (once: the allocator's pools are warm then); the line carries what happened in either pass.
    torch.cuda.synchronize()
    reserved0 = torch.cuda.memory_reserved(dev)
    dt, losses = timed(a.warmup, a.steps)
    grew = torch.cuda.memory_reserved(dev) - reserved0
    retimed, grew2, first_ms = False, 0, None
    if dp is not None and world > 1:
        grew = int(par.max_over_ranks(float(grew)))       # every rank takes the same branch
    if grew > 0:
        first_ms = 1e3 * dt / a.steps
        reserved0 = torch.cuda.memory_reserved(dev)
        dt, losses = timed(a.warmup, a.steps)
        retimed = True
        grew2 = torch.cuda.memory_reserved(dev) - reserved0
        if dp is not None and world > 1:
            grew2 = int(par.max_over_ranks(float(grew2)))
        if grew2 > 0 and rank == 0:
            print("bench.py: device memory still grew inside the timed region (+%d bytes reserved in the second pass of %d steps); the "
                  "figure stands, `timed_region_alloc` says so" % (grew2, a.steps), file=sys.stderr)
    if os.environ.get("SSCG_PHASE_EVENTS") == "1" and rank == 0:
        # diagnostic: one more step with timed events around its passes (model.py _mark) - when each pass ran on the GPU and when
        # the host issued it, without a tracer's overhead
        torch.cuda.synchronize()
        run(a.warmup + a.steps)
        torch.cuda.synchronize()
        m0 = model.phase_marks[0]
        for name, ev, th in model.phase_marks:
            print("%8.2f ms gpu  %8.2f ms host  %s" % (m0[1].elapsed_time(ev), (th - m0[2]) * 1e3, name), file=sys.stderr)
    finite = all(bool(torch.isfinite(v)) for v in losses.values())
    peak = PEAK[dtype]
    value = world * bsz * a.steps / dt
    out = {
        "metric": "training images/sec (G+D step) at %dx%d" % (H, W), "value": round(value, 4), "unit": "img/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        **({"shared_gpu": True} if os.environ.get("SSCG_DP_SHARED_GPU") else {}),
        "config": {"workload": "%s, batch=%d per GPU, %s" % (cfg["label"], bsz, DTYPE_TEXT[arith if dtype == "f32" else dtype]),
                   "baseline_config": a.config,
                   "global_batch": world * bsz, "image_unit": "one labeled + one unlabeled %dx%d image" % (H, W),
                   "parallelism": "dp%d" % world, "losses_finite": finite},
        "step_conv_tflops": round(world * bsz * cfg["tflop_per_pair"] * a.steps / dt, 2),
        "step_frac_of_mfma_peak": round(bsz * cfg["tflop_per_pair"] * a.steps / dt / peak, 4),
        "mfma_peak_tflops": peak,
        "host_issue_ms_per_step": round(1e3 * host[0], 2),
        "timed_region_alloc": {"reserved_growth_bytes_first_pass": int(grew), "retimed": retimed, "reserved_growth_bytes_second_pass": int(grew2),
                               **({"first_pass_ms_per_step": round(first_ms, 3)} if first_ms is not None else {}),
                               "reserved_bytes": int(torch.cuda.memory_reserved(dev))},
    }
    if host_bound_case is not None:
        out["host_bound_case"] = host_bound_case
    if dp:
        # the evidence beside `n_gpus` (= WORLD_SIZE of the environment): what the process group saw - one physical device per rank,
        # the collective library's version, every rank's own step time (collective call: every rank takes part)
        census = par.rank_census(1e3 * own[0], host_issue_ms=1e3 * host[0], affinity=dp.affinity)
        census["gradient_buckets"] = par.dp_buckets(world)
        out["rccl"] = census
        if census["distinct_devices"] != census["world_size"] and not os.environ.get("SSCG_DP_SHARED_GPU"):
            raise SystemExit("bench.py: %d ranks on %d distinct devices %s - not a %d-GPU measurement (SSCG_DP_SHARED_GPU=1 marks the "
                             "one-GPU test rig)" % (census["world_size"], census["distinct_devices"], census["devices"], census["world_size"]))

    # The host thread's issue cost of one step with NO back-pressure from the device: the same step with every kernel launch of the
    # library turned into a no-op (sscg_set_dry_run; torch's own copies / events still run).  `host_issue_ms_per_step` above includes
    # the time the thread spends blocked on a full launch queue; the difference is the back-pressure.
    if not a.no_unblocked:
        lib = importlib.import_module(PKG + "._lib").lib
        torch.cuda.synchronize()
        pools = [(list(p.items), p.cur_elements) for p in model.pools]       # a dry step stores never-written tensors in the image pools
        # ... advances both optimisers' step counters (bias correction) beside no-op Adam kernels, and draws from numpy's / torch's CPU
        # generators (pool decisions, model.py:486): all of it is put back, so that the figures measured after this block run the same
        # trajectory as without it.  Operand copies built lazily DURING the dry steps hold garbage; the first real optimiser step
        # (the next `run`) moves the weights' epoch and rebuilds them before any kernel reads them.
        opt_steps = [(o, o._steps) for o in (model.g_optimizer, model.d_optimizer)]
        import numpy as np
        np_state, torch_state = np.random.get_state(), torch.get_rng_state()
        lib.sscg_set_dry_run(1)
        try:
            run(a.warmup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.warmup, a.warmup + min(a.steps, 4)):
                run(i)
            out["host_issue_unblocked_ms"] = round(1e3 * (time.perf_counter() - t0) / min(a.steps, 4), 2)
            torch.cuda.synchronize()
        finally:
            lib.sscg_set_dry_run(0)
            for p, (items, n) in zip(model.pools, pools):
                p.items, p.cur_elements = items, n
            for o, n in opt_steps:
                o._steps = n
            np.random.set_state(np_state)
            torch.set_rng_state(torch_state)

    # secondary figure (BASELINE.md section 2 / SURVEY 8(d)): the same step without the forwards whose outputs the
    # reference never uses (old_Gsi(l_img) -> old_Gis, model.py:419-420,423) and without old_Di's never-applied wgrad
    if not a.no_elided:
        model.as_written = False
        run(a.warmup + a.steps)
        dte, _ = timed(a.warmup, a.steps)
        model.as_written = True
        out["elided_dead_work"] = {"value": round(world * bsz * a.steps / dte, 4), "unit": "img/s", "ms_per_step": round(1e3 * dte / a.steps, 3),
                                   "note": "not the headline: skips the forwards with unused outputs + old_Di's unused wgrad"}

    # secondary figure at config 2: the same as-written step in bf16 (the arithmetic of configs 3/5); never the fp32 headline
    if dtype == "f32" and not a.no_bf16:
        F.set_conv_precision("bf16")
        run(a.warmup + a.steps)
        dtb, lb = timed(a.warmup, a.steps)
        F.set_conv_precision(dtype)
        out["bf16"] = {"value": round(world * bsz * a.steps / dtb, 4), "unit": "img/s", "ms_per_step": round(1e3 * dtb / a.steps, 3),
                       "losses_finite": all(bool(torch.isfinite(v)) for v in lb.values()),
                       "note": "not the headline (this configuration is fp32): " + DTYPE_TEXT["bf16"]}

    # beside the headline: the same fp32 step with the OTHER contraction of fp32 tensors (exact fp32 MFMA when the headline runs the
    # split contraction, and the other way round)
    if dtype == "f32" and not a.no_bf16:
        other = "f32x" if arith == "f32s" else "f32s"
        F.set_conv_precision(other)
        run(a.warmup + a.steps)
        dts, ls = timed(a.warmup, a.steps)
        F.set_conv_precision(dtype)
        out["f32_exact" if other == "f32x" else "f32_split"] = {
            "value": round(world * bsz * a.steps / dts, 4), "unit": "img/s", "ms_per_step": round(1e3 * dts / a.steps, 3),
            "losses_finite": all(bool(torch.isfinite(v)) for v in ls.values()), "note": "not the headline: " + DTYPE_TEXT[other]}

    if not a.no_roofline:
        # per-kernel timing needs the kernels one at a time: the side stream (concurrent weight gradients /
        # frozen generators) is switched off for this extra, untimed step only.  EVERY rank runs the step (it contains
        # the gradient all-reduces); rank 0 reports.
        F.SideStream.enabled = False
        torch.cuda.synchronize()
        run(a.warmup + a.steps)          # untimed: on one stream the per-stream workspaces are requested (and grown: hipMalloc) anew
        torch.cuda.synchronize()
        with F.ConvProfile() as prof:
            run(a.warmup + a.steps)
        summ = prof.summary()
        F.SideStream.enabled = True
    if rank == 0 and not a.no_roofline:
        fam = "bf16" if dtype == "bf16" else ("split" if arith == "f32s" else "f32")       # kernel family that dominates this configuration
        kname = {"bf16": "conv16_kernel (implicit-GEMM conv forward + data-gradient on bf16 LDS tiles, v_mfma_f32_32x32x16_bf16)",
                 "split": "convs_kernel (implicit-GEMM conv forward + data-gradient, fp32 tensors, 3-piece split-bf16 contraction: six "
                          "v_mfma_f32_32x32x16_bf16 per fp32 product block, fp32-accurate)",
                 "f32": "conv_kc_kernel (implicit-GEMM conv forward + data-gradient, v_mfma_f32_32x32x2_f32)"}[fam]
        if fam == "split":
            peak = SPLIT_PEAK
        kc = {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0}
        for kind in ("fwd", "dgrad"):
            s = summ.get((kind, fam))
            if s:
                for f in kc:
                    kc[f] += s[f]
        tf = kc["flops"] / (kc["ms"] * 1e-3) / 1e12 if kc["ms"] > 0 else 0.0
        out["roofline"] = {
            "kernel": kname, "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(tf / peak, 4), "traffic": None,
            "launches_per_step": kc["launches"], "avg_launch_us": round(1e3 * kc["ms"] / max(kc["launches"], 1), 2),
            "flop_per_launch_avg": round(kc["flops"] / max(kc["launches"], 1)),
            "algorithmic_bytes_per_launch_avg": round(kc["bytes"] / max(kc["launches"], 1)),
            "conv_ms_per_step": {"%s/%s" % k: round(v["ms"], 2) for k, v in summ.items()},
            "conv_tflops": {"%s/%s" % k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in summ.items() if v["ms"] > 0},
        }
        if fam == "split":
            out["roofline"]["peak_note"] = ("algorithmic fp32 FLOP against the dense bf16 MFMA peak (2500 TFLOP/s) / 6 piece products per fp32 "
                                            "product; executed matrix-core rate = 6 x achieved")
            out["roofline"]["frac_of_fp32_mfma_peak"] = round(tf / PEAK["f32"], 4)
            out["roofline"]["executed_bf16_mfma_tflops"] = round(6.0 * tf, 1)
        # the whole step's conv FLOP (exact + split kernels, weight gradients included) against the fp32 MFMA peak stays in
        # step_frac_of_mfma_peak above: that is the figure north_star's "fraction of the conv roofline" asks for
        out["roofline"].update(pmc_traffic(fam))
        # the 3x3 family north_star singles out (3x3 convs of the dominant family only)
        f3 = m3 = 0.0
        for (kind, fm), v in summ.items():
            if fm != fam:
                continue
            for key, (n, fl, ms) in v["shapes"].items():
                if " r3 " in key:
                    f3 += fl
                    m3 += ms
        if m3 > 0:
            out["roofline"]["conv3x3_tflops"] = round(f3 / (m3 * 1e-3) / 1e12, 2)
            out["roofline"]["conv3x3_frac"] = round(f3 / (m3 * 1e-3) / 1e12 / peak, 4)
        if os.environ.get("SSCG_BENCH_SHAPES"):
            rows = []
            for (kind, fm), v in summ.items():
                for key, (n, fl, ms) in v["shapes"].items():
                    rows.append((ms, kind + "/" + fm, key, n, fl / (ms * 1e-3) / 1e12))
            rows.sort(reverse=True)
            with open(os.environ["SSCG_BENCH_SHAPES"], "w") as f:
                for ms, kind, key, n, tfl in rows:
                    f.write("%8.3f ms  %-10s %-40s x%-3d %6.1f TF/s\n" % (ms, kind, key, n, tfl))

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)

    emit_line(out, rank, bool(dp))


def emit_line(out, rank, in_group):
    """The JSON line is the LAST line this job writes to stdout.  RCCL prints a version banner through C stdio when its first
    communicator comes up; on a pipe that text sits in libc's buffer until exit() - behind anything Python printed (measured:
    the banner followed the line, profiles/r06_experiments.txt item 23).  Every rank empties both buffers; the other ranks do so
    BEFORE the last barrier and point fd 1 at /dev/null for whatever their teardown may still print; rank 0 tears its process
    group down, empties the buffers once more and prints.  (tests/test_host_logic.py runs this over gloo with a banner in libc's buffer.)"""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    if in_group:
        import torch.distributed as dist
        if rank != 0:
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        dist.barrier()
        dist.destroy_process_group()
        sys.stdout.flush()
        libc.fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


def spawn_ranks(n):
    """Re-execute this command line under torch.distributed.run with one rank per GPU (what the driver's own launcher
    does).  A box with fewer than n GPUs (the 1-GPU test box) runs the ranks on GPU 0 over gloo - the same control flow,
    not a scaling figure; the JSON line then says "shared_gpu": true."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import torch
        have = torch.cuda.device_count()
    except Exception:
        have = 0
    if have < n:
        env.update(SSCG_DP_SHARED_GPU="1", SSCG_DP_BACKEND="gloo")
    rc = 1
    for attempt in range(3):          # the port was free a moment ago; somebody may have taken it since (EADDRINUSE in the launcher): try another
        if attempt:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(r.stderr)
        rc = r.returncode
        if rc == 0 or not ("EADDRINUSE" in r.stderr or "address already in use" in r.stderr.lower()):
            break
        print("bench.py: rendezvous port %d was taken (attempt %d), retrying on another" % (port, attempt + 1), file=sys.stderr)
    return rc


def pmc_traffic(fam):
    """HBM bytes per launch of the dominant kernel.  The live bench cannot collect PMC counters itself (rocprofv3 --pmc runs
    in its own passes), so this figure is read from a committed profile - and only when that profile was taken over THIS
    kernel family (its kernel names are checked), with its source named in the line.  Units and the gfx950 correction as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: counters are KiB; FETCH_SIZE under-reports wide coalesced reads by 2x."""
    want = {"bf16": "conv16_kernel", "split": "convs_kernel", "f32": "conv_kc_kernel"}[fam]
    for name in ("r06_pmc_per_kernel_%s.json" % fam, "r05_pmc_per_kernel_%s.json" % fam, "r04_pmc_per_kernel_%s.json" % fam, "r03_pmc_per_kernel_%s.json" % fam, "r02_pmc_per_kernel_%s.json" % fam):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        tot_b = tot_n = busy = act = 0.0
        for kn, cs in d.items():
            if want in kn and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                n = cs["FETCH_SIZE"]["launches"]
                tot_b += (2.0 * cs["FETCH_SIZE"]["sum"] + cs["WRITE_SIZE"]["sum"]) * 1024.0
                tot_n += n
                if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
                    busy += cs["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"]
                    act += cs["GRBM_GUI_ACTIVE"]["sum"]
        if tot_n == 0:
            continue
        return {"traffic": round(tot_b / tot_n), "traffic_unit": "HBM bytes per %s launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)" % want,
                "traffic_source": "profiles/%s (offline rocprofv3 --pmc passes over this command; not measured by this run)" % name,
                "mfma_util_pmc": round(busy / max(act / 8.0 * 1024.0, 1.0), 4)}   # GRBM_GUI_ACTIVE: summed over 8 XCDs; busy cycles: over 1024 SIMDs
    return {"traffic": None, "traffic_source": "no committed PMC profile covers %s" % want}


def cpu_sample_text(cfg, torch_version, threads, probes, physical, logical, model, n_timed, n_all):
    return ("as-written G+D step (incl. model.py:419-420,423), %s %d-class %dx%d, batch 2, torch %s CPU fp32 on %s (%d physical / %d logical "
            "cores), SURVEY 8(d): one warm-up step at each of %s threads (the probes), then %d timed steps at the fastest setting (%d "
            "threads) = `value`, and %d timed step(s) on ALL physical cores = `all_physical_cores` (oneDNN on these 33x33 maps "
            "over-subscribes past 32-64 threads: both figures are in the line); OMP_PROC_BIND / NUMA policy left at the box's defaults" % (
                cfg["dataset"], cfg["C"], cfg["H"], cfg["W"], torch_version, model, physical, logical, "/".join(str(t) for t in probes),
                n_timed, threads, n_all))


def cpu_thread_candidates(physical):
    """Thread counts to probe: 32, 64 and all physical cores (oneDNN on 33x33 maps over-subscribes long before 128 threads:
    BENCH_r02 measured 52 s/step on 128 threads against 10.5 s on 64)."""
    c = [t for t in (32, 64) if t < physical]
    return c + [physical]


CPU_TIMED_STEPS = 3            # >= 3 timed steps after 1 warm-up
CPU_ALL_CORES_BUDGET_S = 20.0  # an all-cores step slower than this is timed ONCE more (three of them would take the leg past three minutes)


def cpu_baseline(cfg, step_fn=None, now=time.perf_counter):
    """oracle/ = CPU restatement of the reference step (validated bit-exact against the reference's losses by
    tests/golden/gen_golden.py), timed on this box's host cores: the as-written step at the configuration's geometry with
    batch 2.  One warm-up step at each candidate thread count - 32, 64, all physical cores - then CPU_TIMED_STEPS timed
    steps at the fastest setting (`value`, `cores`) and 1-3 timed steps on all physical cores (`all_physical_cores`).
    `step_fn(threads)` runs one step (tests inject a stub)."""
    import torch
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    C, H, W = cfg["C"], cfg["H"], cfg["W"]
    bs = 2
    state = {"i": 0}
    if step_fn is None:
        import numpy as np
        from oracle import fixtures as FX
        from oracle import step as ostep
        o = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, torch.float32, "bench"), crop=(H, W), as_written=True)
        np.random.seed(0)

        def step_fn(threads, as_written=True):
            torch.set_num_threads(threads)
            o.as_written = as_written
            l_img, l_gt, unl_img = FX.step_batch("bench", state["i"], C, H, W, bs)
            state["i"] += 1
            o.step(l_img, l_gt, unl_img)

    def timed_step(threads, **kw):
        t0 = now()
        step_fn(threads, **kw)
        return now() - t0

    probes = cpu_thread_candidates(physical)
    probe_s = [timed_step(t) for t in probes]                      # (each is the warm-up of its thread count)
    allc = probes[-1]
    n_all = CPU_TIMED_STEPS if probe_s[-1] <= CPU_ALL_CORES_BUDGET_S else 1
    all_times = [timed_step(allc) for _ in range(n_all)]
    best = probes[min(range(len(probes)), key=lambda i: probe_s[i])]
    times = all_times if (best == allc and n_all == CPU_TIMED_STEPS) else [timed_step(best) for _ in range(CPU_TIMED_STEPS)]
    dt = sum(times) / len(times)
    dta = sum(all_times) / len(all_times)
    # the same step without the reference's unused forwards (model.py:419-420,423), one timed step, for the side-by-side
    dte = timed_step(best, as_written=False)
    return {"value": round(bs / dt, 4), "unit": "img/s", "cores": best, "kind": "port",
            "sample": cpu_sample_text(cfg, torch.__version__, best, probes, physical, logical, model, len(times), n_all),
            "seconds_per_step": round(dt, 2), "timed_steps": len(times), "warmup_steps": 1,
            "probe_seconds_per_step": {str(t): round(v, 2) for t, v in zip(probes, probe_s)},
            "all_physical_cores": {"cores": allc, "value": round(bs / dta, 4), "seconds_per_step": round(dta, 2), "timed_steps": n_all, "warmup_steps": 1},
            "physical_cores": physical, "cpu_model": model,
            "elided_dead_work": {"value": round(bs / dte, 4), "seconds_per_step": round(dte, 2)}}


if __name__ == "__main__":
    main()

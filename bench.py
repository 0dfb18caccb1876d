#!/usr/bin/env python
"""Throughput of the hot path: full G+D training steps of the semi-supervised CycleGAN on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: VOC2012 21-class, 256x256, semisupervised_cycleGAN, batch 8 per GPU, fp32,
random-init weights (the reference's N(0,0.02) init), synthetic image/label batches already resident in HBM.
One step = one iteration of /root/reference model.py:370-552 as written (all seven networks, both optimisers).
Weak scaling: every rank runs batch 8; `value` = N * 8 * K / (max-over-ranks wall time of K steps).

Besides the contract line this prints, in the same JSON object:
  roofline     - the implicit-GEMM conv kernels measured live with HIP events on the launch stream during one
                 extra (untimed) step: algorithmic FLOP / kernel time against the fp32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline - oracle/ (the CPU restatement of the reference) timed on this box's host cores on a bounded
                 sample (rank 0, N = 1 only).
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

PEAK_F32_MFMA_TFLOPS = 157.3            # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
STEP_TFLOP_PER_PAIR = 1.983             # conv FLOP of one as-written step per labeled/unlabeled pair @VOC 256x256
C, H, W, B = 21, 256, 256, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-elided", action="store_true")
    ap.add_argument("--batch", type=int, default=B)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="arithmetic of the conv contractions; BASELINE config 2 (the bench config) is fp32")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16-contraction figure")
    a = ap.parse_args()

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
    local = dp.local_rank if dp else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    bsz = a.batch

    import main as cli                   # the product CLI's own defaults (oracle/ is used by the cpu_baseline leg only)
    args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", "voc2012", "--crop_height", str(H), "--crop_width", str(W),
                         "--batch_size", str(bsz), "--checkpoint_dir", "/tmp/sscg_bench_ckpt_%d" % rank, "--epochs", "400",
                         "--decay_epoch", "100", "--dtype", a.dtype])
    args.gpu_ids, args.as_written = [local], True
    args.overlap_d = os.environ.get("SSCG_OVERLAP_D", "1") == "1"   # the D step overlaps the next step's generator forwards
    torch.manual_seed(0)
    F.set_conv_precision(a.dtype)
    with contextlib.redirect_stdout(io.StringIO()):
        model = md.semisuper_cycleGAN(args, data_parallel=dp)

    # synthetic batches, resident in HBM before the clock starts (rank-offset seeds)
    nb = a.warmup + a.steps + 1
    lab = list(data.SyntheticLoader(bsz, C, H, W, nb, 1 + 1000 * rank, device=dev))
    unl = list(data.SyntheticLoader(bsz, C, H, W, nb, 2 + 1000 * rank, device=dev))

    def run(i):
        return model.step(lab[i][0], lab[i][1], unl[i][0])

    for i in range(a.warmup):
        run(i)
    if dp:
        dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        losses = run(i)
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    dt = time.perf_counter() - t0
    dt = par.max_over_ranks(dt)
    finite = all(bool(torch.isfinite(v)) for v in losses.values())

    value = world * bsz * a.steps / dt
    out = {
        "metric": "training images/sec (G+D step) at 256x256", "value": round(value, 4), "unit": "img/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        **({"shared_gpu": True} if os.environ.get("SSCG_DP_SHARED_GPU") else {}),
        "config": {"workload": "VOC2012 21-class 256x256 semisupervised_cycleGAN as-written G+D step, batch=%d per GPU, %s" % (
                       bsz, "fp32" if a.dtype == "f32" else "bf16 conv contractions (fp32 accumulate, fp32 tensors/norms/Adam)"),
                   "global_batch": world * bsz, "image_unit": "one labeled + one unlabeled 256x256 image", "parallelism": "dp%d" % world,
                   "losses_finite": finite},
        "step_conv_tflops": round(world * bsz * STEP_TFLOP_PER_PAIR * a.steps / dt, 2),
        "step_frac_of_f32_mfma_peak": round(bsz * STEP_TFLOP_PER_PAIR * a.steps / dt / PEAK_F32_MFMA_TFLOPS, 4),
    }

    # secondary figure (BASELINE.md section 2 / SURVEY 8(d)): the same step without the forwards whose outputs the
    # reference never uses (old_Gsi(l_img) -> old_Gis, model.py:419-420,423) and without old_Di's never-applied wgrad
    if not a.no_elided:
        model.as_written = False
        run(a.warmup + a.steps)
        if dp:
            dp.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.warmup, a.warmup + a.steps):
            run(i)
        torch.cuda.synchronize()
        if dp:
            dp.barrier()
        dte = par.max_over_ranks(time.perf_counter() - t1)
        model.as_written = True
        out["elided_dead_work"] = {"value": round(world * bsz * a.steps / dte, 4), "unit": "img/s", "ms_per_step": round(1e3 * dte / a.steps, 3),
                                   "note": "not the headline: skips 53.25 GMAC/pair of forwards with unused outputs + 1.1 GMAC/pair of unused wgrad"}

    # secondary figure: the same as-written step with the heavy convolutions contracting in bf16 (fp32 accumulation,
    # tensors still fp32 in HBM) - the arithmetic BASELINE configs 3/5 name; never the headline of this fp32 config
    if a.dtype == "f32" and not a.no_bf16:
        F.set_conv_precision("bf16")
        run(a.warmup + a.steps)
        if dp:
            dp.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(a.warmup, a.warmup + a.steps):
            lb = run(i)
        torch.cuda.synchronize()
        if dp:
            dp.barrier()
        dtb = par.max_over_ranks(time.perf_counter() - t2)
        F.set_conv_precision("f32")
        out["bf16_contractions"] = {"value": round(world * bsz * a.steps / dtb, 4), "unit": "img/s", "ms_per_step": round(1e3 * dtb / a.steps, 3),
                                    "losses_finite": all(bool(torch.isfinite(v)) for v in lb.values()),
                                    "note": "not the headline (BASELINE config 2 is fp32): conv operands rounded to bf16 in LDS->MFMA, fp32 accumulate"}

    if not a.no_roofline:
        # per-kernel timing needs the kernels one at a time: the side stream (concurrent weight gradients /
        # frozen generators) is switched off for this extra, untimed step only.  EVERY rank runs the step (it contains
        # the gradient all-reduces); rank 0 reports.
        F.SideStream.enabled = False
        torch.cuda.synchronize()
        with F.ConvProfile() as prof:
            run(a.warmup + a.steps)
        summ = prof.summary()
        F.SideStream.enabled = True
    if rank == 0 and not a.no_roofline:
        kc = {"flops": 0.0, "ms": 0.0, "launches": 0}
        for kind in ("fwd", "dgrad"):
            if kind in summ:
                for f in kc:
                    kc[f] += summ[kind][f]
        tf = kc["flops"] / (kc["ms"] * 1e-3) / 1e12 if kc["ms"] > 0 else 0.0
        out["roofline"] = {
            "kernel": "conv_kc_kernel (implicit-GEMM conv forward + data-gradient, v_mfma_f32_32x32x2_f32)",
            "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
            "launches_per_step": kc["launches"], "avg_launch_us": round(1e3 * kc["ms"] / max(kc["launches"], 1), 2),
            "flop_per_launch_avg": round(kc["flops"] / max(kc["launches"], 1)),
            "conv_ms_per_step": {k: round(v["ms"], 2) for k, v in summ.items()},
            "conv_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in summ.items() if v["ms"] > 0},
        }
        out["roofline"].update(pmc_traffic())
        # the 3x3 family north_star singles out (3x3 convs only)
        f3 = m3 = 0.0
        for kind, v in summ.items():
            for key, (n, fl, ms) in v["shapes"].items():
                if " r3 " in key:
                    f3 += fl
                    m3 += ms
        if m3 > 0:
            out["roofline"]["conv3x3_tflops"] = round(f3 / (m3 * 1e-3) / 1e12, 2)
            out["roofline"]["conv3x3_frac"] = round(f3 / (m3 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        if os.environ.get("SSCG_BENCH_SHAPES"):
            rows = []
            for kind, v in summ.items():
                for key, (n, fl, ms) in v["shapes"].items():
                    rows.append((ms, kind, key, n, fl / (ms * 1e-3) / 1e12))
            rows.sort(reverse=True)
            with open(os.environ["SSCG_BENCH_SHAPES"], "w") as f:
                for ms, kind, key, n, tfl in rows:
                    f.write("%8.3f ms  %-5s %-40s x%-3d %6.1f TF/s\n" % (ms, kind, key, n, tfl))

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()

    if rank == 0:
        print(json.dumps(out))
    if dp:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01q_pmc_per_kernel.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same bench command).  Units and the gfx950
    correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: counters are KiB; FETCH_SIZE under-reports wide
    coalesced reads by 2x.  The live bench cannot collect PMC itself, hence the file."""
    path = os.path.join(ROOT, "profiles", "r01q_pmc_per_kernel.json")
    if not os.path.exists(path):
        return {"traffic": None}
    d = json.load(open(path))
    tot_b = tot_n = busy = act = 0.0
    for name, cs in d.items():
        if "conv_kc_kernel" in name and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            n = cs["FETCH_SIZE"]["launches"]
            tot_b += (2.0 * cs["FETCH_SIZE"]["sum"] + cs["WRITE_SIZE"]["sum"]) * 1024.0
            tot_n += n
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
                busy += cs["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"]
                act += cs["GRBM_GUI_ACTIVE"]["sum"]
    return {"traffic": round(tot_b / max(tot_n, 1)), "traffic_unit": "HBM bytes per conv_kc launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": "profiles/r01q_pmc_per_kernel.json",
            "mfma_util_pmc": round(busy / max(act / 8.0 * 1024.0, 1.0), 4)}   # GRBM_GUI_ACTIVE is summed over the 8 XCDs, busy cycles over 1024 SIMDs


def cpu_baseline():
    """oracle/ = CPU restatement of the reference step (validated bit-exact against the reference's losses by
    tests/golden/gen_golden.py), timed on this box's host cores: one step at the bench geometry with batch 2."""
    import numpy as np
    import torch
    from oracle import fixtures as FX
    from oracle import step as ostep
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    bs = 2
    sds = FX.semisup_state_dicts(C, torch.float32, "bench")
    o = ostep.SemiSupOracle(C, sds, crop=(H, W))
    l_img, l_gt, unl_img = FX.step_batch("bench", 0, C, H, W, bs)
    np.random.seed(0)
    t0 = time.perf_counter()
    o.step(l_img, l_gt, unl_img)
    dt = time.perf_counter() - t0
    return {"value": round(bs / dt, 4), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": "1 full G+D step (no warm-up), VOC 21-class 256x256, batch 2, torch %s CPU fp32, %d threads of %d host cores"
                      % (torch.__version__, threads, cores), "seconds": round(dt, 2)}


if __name__ == "__main__":
    main()

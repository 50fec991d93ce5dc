#!/usr/bin/env python3
"""Headline benchmark: interpolated frames/s of a TRAIN STEP of FinalBidirectionAttenfusion.

BASELINE.json metric: "interpolated frames/sec (train step) GoPro 256x256 11+1"; workload =
configs[1]: GoPro 11+1 blur-VFI, batch 8 per GPU, 256x256, T=23, img_chn=26, fp32, full
step (zero_grad -> forward -> Charbonnier -> BPTT -> clip 0.01 -> AdamW), synthetic inputs
already resident in HBM.

One process per GPU (the reference launches `torch.distributed.launch --nproc_per_node=N`,
/root/reference/README.md:138, basicsr/utils/dist_util.py:11-30):
  * under a launcher (WORLD_SIZE set, e.g. `python -m torch.distributed.run --nproc-per-node N bench.py
    --gpus N`) this process is one rank; WORLD_SIZE must equal --gpus;
  * without one, `python bench.py --gpus N` (N > 1) re-executes itself under torch.distributed.run with
    N ranks on 127.0.0.1; fewer visible devices than ranks is an error, never a silent 1-rank run.
N > 1 uses RCCL (backend 'nccl') for the gradient all-reduce.

--scaling weak (default): --batch samples PER GPU whatever N is;
--scaling strong: --batch is the GLOBAL batch, sharded (batch/N per GPU: config 2's batch 8 -> B=1 per GPU at
  8 GPUs, the reference recipe's `batch_size_per_gpu: 1`).
With N > 1 the weak run also times the strong-scaling shard of the same global batch afterwards and reports
it as `"strong": {...}` in the same JSON line.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      the dominant kernel's achieved fp32 TFLOP/s (algorithmic FLOPs / HIP-event time
                of its launches in one instrumented step) against the 157.3 TFLOP/s fp32 MFMA peak;
  cpu_baseline  the CPU oracle (oracle/refid_oracle.py, kind "port") timed on this box's host
                cores for train steps at B=1 of the same workload (rank 0, N=1 only).

Input hand-over (the reference's hot loop is prefetcher.next() -> feed_data -> optimize_parameters, train.py:217-232, with
`prefetch_mode: cuda` copying batch k+1 host -> device on a side stream during step k, data/prefetch_dataloader.py:84-125):
--h2d prefetch (default) keeps the synthetic batches in PINNED HOST memory and runs refid_amd.data.CUDAPrefetcher + feed_data
inside the timed loop; step k's inputs are resident in HBM when step k starts (the copy rode under step k-1), and the line
reports how long the compute stream had to wait for copies (`h2d_ms_exposed`, per step).  --h2d resident feeds once outside
the loop (the rounds 1-3 protocol).

--mode infer --config 4|5: inference throughput of BASELINE configs[3] (HighREV 7-skip sharp-VFI, 512x512, T=7, img_chn 6) and
configs[4] (15-skip, 1224x1632, T=15, the reference's 512x512 tile grid through refid_amd.tiling, tiles sharded over the ranks,
PSNR/SSIM validation tail on the GPU); one "step" = one sample through `test()` (twoImage_event_recurrent_model.py:312-330).
The default command (no --mode) stays the configs[1] train step.

--dry-run --backend gloo: launcher / rendezvous / gradient-sync / timing protocol only, on CPU tensors (no
model step; `value` is null).  Used by tests/test_bench_launcher.py; never a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
METRIC = "interpolated frames/sec (train step) GoPro 256x256 11+1"


def synthetic_batch(B, T, H, W, img_chn, seed, device):
    """frames/gt U[0,1); voxels 85 % exact zeros, rest N(0,1) rounded to 1/8,
    clipped to +-4; the 20 deblur-voxel channels of x are voxel-like."""
    g = torch.Generator(device=device).manual_seed(seed)

    def vox(*shape):
        v = torch.clamp(torch.round(torch.randn(*shape, generator=g, device=device) * 8) / 8, -4, 4)
        return torch.where(torch.rand(*shape, generator=g, device=device) < 0.85, torch.zeros_like(v), v)

    x = torch.rand(B, img_chn, H, W, generator=g, device=device)
    if img_chn == 26:
        x[:, 3:13] = vox(B, 10, H, W)
        x[:, 16:26] = vox(B, 10, H, W)
    ev = vox(B, T, 2, H, W)
    gt = torch.rand(B, T, 3, H, W, generator=g, device=device)
    return x, ev, gt


def options(args):
    mode = getattr(args, "mode", "train")                  # (tools/profile_step.py passes its own minimal namespace)
    if mode == "infer":
        args.img_chn = INFER_CONFIGS[args.config]["img_chn"]
    return {
        "name": "bench", "is_train": mode == "train", "num_gpu": 1,
        "network_g": dict(type="FinalBidirectionAttenfusion", img_chn=args.img_chn, ev_chn=2, num_encoders=3,
                          base_num_channels=32, num_block=1, num_residual_blocks=2, compute_dtype=args.dtype),
        "path": {"pretrain_network_g": None},
        "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                  "scheduler": dict(type="TrueCosineAnnealingLR", T_max=200000, eta_min=1e-7),
                  "pixel_opt": dict(type="CharbonnierLoss", loss_weight=1, reduction="mean")},
        "val": {"max_minibatch": 2},
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """SURVEY.md 8(d) / BASELINE.md 3 'CPU baseline timing': the oracle's S2 train step (fwd + bwd + clip + AdamW) on
    the host cores, fp32, B=1, ALL T frames, warm-up at size, 3 timed steps, median = `value`.  The thread count is
    swept first on a short sample (the first --cpu-frames frames, one step each) because the oracle's small per-step
    convolutions oversubscribe a many-core host; the full-length steps run at the best count.  (The short sample
    flatters the CPU: the oracle's e[:, t] select-backward zero-fill is O(T^2), SURVEY A0 -- it is only reported
    in `sample`.)  --cpu-full 0 restores the short sample as `value` (quick local runs)."""
    from oracle import refid_oracle as O
    torch.manual_seed(0)
    P = O.make_params(args.img_chn, mode="init", seed=0)
    for k in P:
        if k.endswith((".beta", ".gamma")):
            P[k] = torch.randn_like(P[k]) * 0.1
    Tc = max(1, min(args.T, args.cpu_frames))
    x, ev, gt = O.make_inputs(1, Tc, args.size, args.size, args.img_chn, seed=1, mode="rng")

    def one():
        Pc = {k: v.clone() for k, v in P.items()}
        t0 = time.perf_counter()
        O.train_step(Pc, O.TrainState(Pc), x, ev, gt)
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    cands = sorted({n for n in (8, 16, 32, 64, 128) if n <= ncpu} | {min(ncpu, 8)})
    t_begin = time.perf_counter()
    torch.set_num_threads(cands[0])
    one()                                                    # warm-up at size (allocator, thread pool)
    sweep = {}
    for n in cands:
        torch.set_num_threads(n)
        sweep[n] = one()
        if time.perf_counter() - t_begin > args.cpu_budget * 0.5:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    short = f"thread sweep on the first {Tc} frames { {n: round(Tc / t, 3) for n, t in sweep.items()} } frames/s"
    Tf = Tc
    if args.cpu_full and Tc < args.T:
        Tf = args.T
        x, ev, gt = O.make_inputs(1, Tf, args.size, args.size, args.img_chn, seed=1, mode="rng")
        one()                                                # warm-up at the full length (allocator)
    times = []
    for _ in range(3):
        times.append(one())
        if time.perf_counter() - t_begin > args.cpu_budget and times:
            break
    med = sorted(times)[len(times) // 2]
    return {"value": round(Tf / med, 4), "unit": "frames/s", "cores": best, "kind": "port",
            "sample": f"oracle train step (fwd+bwd+clip+AdamW), B=1, {Tf} of T={args.T} frames, "
                      f"{args.size}x{args.size}, fp32; {short}; then warm-up + median of "
                      f"{len(times)} timed steps at {best} threads ({med:.2f} s/step); host: {ncpu} logical CPUs, "
                      f"{_cpu_model()}"}


def roofline_from_profile(prof, step_seconds, dtype, unit_note, traffic_lookup=True):
    """Dominant kernel (largest accumulated HIP-event time in one instrumented single-stream pass) against its roof.
    traffic_lookup: True = the PMC passes of the configs[1] train step (profiles/r*_pmc_traffic[_bf16].json); a string =
    that workload's own passes (profiles/r*_pmc_traffic_<string>.json, tools/profile_infer.sh); False = report null."""
    agg = {}
    for name, fl, e0, e1, _shape, nb in prof:
        a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
        a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1; a[3] += nb
    name, (fl, sec, cnt, nbytes) = max(agg.items(), key=lambda kv: kv[1][1])
    # HBM traffic of the same kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs, tools/pmc_traffic.py); bench.py cannot collect hardware counters itself
    traffic = None
    import glob
    suffix = {"fp32": "", "bf16": "_bf16"}.get(dtype) if traffic_lookup else None
    if isinstance(traffic_lookup, str):
        suffix = "_" + traffic_lookup if dtype == "fp32" else None
    pats = [] if suffix is None else sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")),
                                            reverse=True)
    for path in pats:
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        rec = table.get(name)
        if rec:
            traffic = rec["hbm_bytes_per_launch"]
            break
        inst = [v for k, v in table.items() if k.startswith(name + "<")]      # template instances of the same kernel
        if inst and not name.startswith("conv_split_kernel<"):
            traffic = int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in inst) / sum(v["launches"] for v in inst))
            break
        if name.startswith("conv_split_kernel<"):
            # the event timing groups the split tile by its product count; rocprof names the template instances
            # <MT, NT, planes, KS, mode>: launch-weighted mean over the instances with that many planes
            planes = {"1": 1, "3": 2, "6": 3}[name[len("conv_split_kernel<"):-1].split(",")[0]]
            rows = [v for k, v in table.items() if k.startswith("conv_split_kernel<") and
                    int(k[len("conv_split_kernel<"):-1].split(",")[2]) == planes]
            if rows:
                traffic = int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows) /
                              sum(v["launches"] for v in rows))
                break
    # Winograd F(2x2,3x3) executes 16/36 of the direct convolution's multiplies on the matrix
    # pipe: price the kernel against the MFMA roof with the FLOPs it actually issues, and keep
    # the SURVEY 8(d) direct-convolution figure beside it.
    executed = fl * (16.0 / 36.0) if "wino" in name else fl
    if "wino24_down" in name:
        executed = fl * (12.0 / 16.0)                            # conv_down through its parity phases: 4 x 3 MFMA-units per pixel x 16 taps
    elif "wino24" in name:
        executed = fl * (12.0 / 36.0)                            # Winograd over 2x4 tiles: 24 MFMA-units per 8 pixels x 9 taps
    peak = FP32_MFMA_PEAK_TFLOPS
    ach = executed / sec / 1e12
    bound, unit = "mfma", "TFLOP/s"
    if "wino6" in name:
        # Winograd-domain products as six bf16 MFMAs each (exact three-plane operand split), or -- "<.., true>", round 6 --
        # three fp16 MFMAs each (two-plane operands): issued FLOPs on the bf16 / fp16 matrix pipe = direct x 16/36 x 6 (3)
        executed = fl * (16.0 / 36.0) * (3.0 if "true" in name else 6.0)
        ach = executed / sec / 1e12
    if "split" in name:
        # split-bf16 tile: `terms` bf16 MFMAs per direct-conv multiply (+ 1/9 for the zero tenth tap of its tap pairs)
        # ("<3, fp16>": three fp16 products on two-plane operands; the stride-2 modes it serves have no zero tap)
        executed = fl * int(name.split("<")[1].split(">")[0].split(",")[0]) * (1.0 if "fp16" in name else 10.0 / 9.0)
        ach = executed / sec / 1e12
    if "bf16" in name or "split" in name or "wino6" in name:
        # bf16 matrix-core operands: the dense bf16 MFMA peak is 2.5 PFLOP/s (MI355X_MICROARCH.md); such a
        # tile streams fp32 tensors and is priced against whichever roof it is closer to
        peak = 2500.0
        gbs = nbytes / sec / 1e9
        if gbs / 8000.0 > ach / peak:
            bound, unit, ach, peak = "hbm", "GB/s", gbs, 8000.0
    conv_fl = sum(v[0] for v in agg.values()); conv_t = sum(v[1] for v in agg.values())
    # the comparable figures next to `frac` (which prices the FLOPs the kernel issues on the pipe it runs on):
    #   frac_fp32_equiv  the multiplies the kernel's algorithm needs (Winograd: direct x 16/36; 2x4 tiles: x 12/36; direct
    #                    tiles: all of them), each counted ONCE, against the fp32 matrix peak -- "what fp32 MFMA tile is this
    #                    kernel worth" (above 1: faster than any fp32-pipe kernel of the same algorithm could be)
    #   hbm_frac         ALGORITHMIC bytes per second against the HBM peak;   waste = PMC traffic / algorithmic bytes
    mult = fl * (16.0 / 36.0) if "wino" in name else fl
    if "wino24_down" in name:
        mult = fl * (12.0 / 16.0)
    elif "wino24" in name:
        mult = fl * (12.0 / 36.0)
    alg_bytes = nbytes / cnt
    return {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
            "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC, profiles/)",
            "frac_fp32_equiv": round(mult / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "hbm_frac": round(nbytes / sec / 8e12, 4),
            "waste": round(traffic / alg_bytes, 3) if traffic else None,
            "algorithmic_bytes_per_launch": round(nbytes / cnt), "kernel": name, "launches": cnt,
            "avg_launch_us": round(sec / cnt * 1e6, 2),
            "direct_conv_equivalent_tflops": round(fl / sec / 1e12, 2),
            "note": unit_note,
            "all_gemm_kernels": {"direct_conv_equivalent_tflops": round(conv_fl / conv_t / 1e12, 2),
                                 "share_of_step": round(conv_t / step_seconds, 3)}}


INFER_CONFIGS = {
    # BASELINE.json configs[3]: options/test/GoPro|HighREV Test_Final_7skip.yml shape (SURVEY.md 8d "Config 4")
    4: dict(name="HighREV 7-skip sharp-VFI inference", H=512, W=512, T=7, img_chn=6, crop=None,
            metric="interpolated frames/sec (inference) HighREV 7-skip 512x512 T=7"),
    # configs[4]: 15-skip at the sensor resolution, the reference's tile grid (val.grids / crop_size 512), PSNR/SSIM tail
    5: dict(name="HighREV 15-skip sharp-VFI tiled inference + PSNR/SSIM", H=1224, W=1632, T=15, img_chn=6, crop=512,
            metric="interpolated frames/sec (tiled inference + PSNR/SSIM) HighREV 15-skip 1632x1224 T=15"),
}


def infer_cpu_baseline(args, cfg):
    """The oracle's forward on the host cores: config 4 whole (B=1, 512x512, T=7); config 5 on ONE of its 12 tiles
    (512x512, T=15) -- frames/s scaled by 1/12 since every tile costs the same."""
    from oracle import refid_oracle as O
    torch.manual_seed(0)
    P = O.make_params(cfg["img_chn"], mode="init", seed=0)
    ncpu = os.cpu_count() or 1
    nthr = min(ncpu, 16)                                    # the train-step sweep's optimum on this host class
    torch.set_num_threads(nthr)
    side = 512
    x, ev, _ = O.make_inputs(1, cfg["T"], side, side, cfg["img_chn"], seed=1, mode="rng")
    with torch.no_grad():
        O.forward(P, x[..., :128, :128].contiguous(), ev[:, :2, :, :128, :128].contiguous())       # warm-up (allocator, threads)
        times = []
        t_begin = time.perf_counter()
        for _ in range(3):
            t0 = time.perf_counter()
            O.forward(P, x, ev)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > args.cpu_budget * 0.5:
                break
    med = sorted(times)[len(times) // 2]
    tiles = 1
    if cfg["crop"]:
        from refid_amd.tiling import grid_indices
        tiles = len(grid_indices(cfg["H"], cfg["W"], cfg["crop"])[0])
    return {"value": round(cfg["T"] / (med * tiles), 4), "unit": "frames/s", "cores": nthr, "kind": "port",
            "sample": f"oracle forward (no_grad), B=1, 512x512, T={cfg['T']}, fp32, median of {len(times)} at {nthr} threads "
                      f"({med:.2f} s per pass)" + (f"; one of the {tiles} tiles, value = T / ({tiles} x pass)" if tiles > 1 else "") +
                      f"; host: {ncpu} logical CPUs, {_cpu_model()}"}


def infer_main(args, model, dev, rank, world, use_dist, sync):
    """--mode infer: `test()` (eval, no_grad, no BPTT stash) of one sample per step; config 5 through the tile grid with the
    tiles rank-strided over the ranks and the PSNR / SSIM validation tail (GPU kernels, one small D2H) in the step."""
    from refid_amd import engine as _engine, ops
    from refid_amd.metrics import calculate_psnr_frames, calculate_ssim_frames
    from refid_amd.tiling import grid_indices, tiled_forward
    cfg = INFER_CONFIGS[args.config]
    H, W, T = cfg["H"], cfg["W"], cfg["T"]
    g = torch.Generator(device=dev).manual_seed(7 + args.config)
    x = torch.rand(1, 2, 3, H, W, generator=g, device=dev)
    gt = torch.rand(1, T, 3, H, W, generator=g, device=dev)
    v = torch.clamp(torch.round(torch.randn(1, T, 2, H, W, generator=g, device=dev) * 8) / 8, -4, 4)
    ev = torch.where(torch.rand(1, T, 2, H, W, generator=g, device=dev) < 0.85, torch.zeros_like(v), v)
    del v
    net = model.net_g
    tiles = len(grid_indices(H, W, cfg["crop"])[0]) if cfg["crop"] else 1
    metrics = {}

    def one():
        if cfg["crop"] is None:
            model.feed_data({"lq": x, "voxel": ev})
            model.test()                                        # twoImage_event_recurrent_model.py:312-330
            return model.output
        net.eval()
        out = tiled_forward(net, x, ev, crop=cfg["crop"], max_minibatch=args.max_minibatch, rank=rank, world=world)
        net.train()
        if rank == 0:                                           # validation runs on rank 0 only (:348-355)
            metrics["psnr"] = calculate_psnr_frames(out[0], gt[0])
            metrics["ssim"] = calculate_ssim_frames(out[0], gt[0])
        return out

    for _ in range(args.warmup):
        one()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    roof = None
    if not args.no_roofline:
        pipeline, _engine.PIPELINE = _engine.PIPELINE, False
        if rank == 0:
            ops.PROFILE = []
        one()
        torch.cuda.synchronize()
        _engine.PIPELINE = pipeline
        if rank == 0:
            prof, ops.PROFILE = ops.PROFILE, None
            roof = roofline_from_profile(prof, dt / args.steps, args.dtype,
                                         "per-kernel timing from one extra single-stream inference pass; traffic: this "
                                         "workload's own FETCH_SIZE / WRITE_SIZE passes (tools/profile_infer.sh), fp32 only",
                                         traffic_lookup=f"infer_config{args.config}")
    if use_dist:
        torch.distributed.barrier()
    if rank == 0:
        frames = T * args.steps
        out = {"metric": cfg["metric"], "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
               "ms_per_frame": round(dt / frames * 1e3, 3), "higher_is_better": True,
               "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
               "dtype": {"fp32": "f32", "bf16x3": "f32 tensors, 3 bf16 products", "bf16": "bf16"}[args.dtype], "data": "synthetic",
               "rccl_ranks": world if use_dist else 0,
               "config": {"workload": f"{cfg['name']}, B=1, {W}x{H}, T={T}, img_chn={cfg['img_chn']}, {args.dtype} "
                                      f"(BASELINE configs[{args.config - 1}])" +
                                      (f", {tiles} tiles of {cfg['crop']}x{cfg['crop']} (max_minibatch {args.max_minibatch}), "
                                       f"rank-strided over {world} rank(s)" if cfg["crop"] else ""),
                          "global_batch": 1, "parallelism": f"tiles over {world} rank(s)" if cfg["crop"] else "replica"}}
        if metrics:
            out["config"]["psnr_mean_dB"] = round(sum(metrics["psnr"]) / len(metrics["psnr"]), 4)
            out["config"]["ssim_mean"] = round(sum(metrics["ssim"]) / len(metrics["ssim"]), 6)
        if roof is not None:
            out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = infer_cpu_baseline(args, cfg)
        _emit(json.dumps(out))
    if use_dist:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (weak) / global batch (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--T", type=int, default=23)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--img-chn", dest="img_chn", type=int, default=26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", dest="cpu_frames", type=int, default=3, help="frames in the CPU-baseline sample")
    ap.add_argument("--cpu-budget", dest="cpu_budget", type=float, default=90.0,
                    help="soft wall-clock bound (s) of the CPU-baseline leg")
    ap.add_argument("--cpu-full", dest="cpu_full", type=int, default=1,
                    help="1: time the CPU baseline on all T frames (the contract's B=1, T=23 configuration); 0: on the "
                         "--cpu-frames sample only")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--h2d", choices=["prefetch", "resident"], default="prefetch",
                    help="prefetch: pinned host batches, side-stream host->device copy of step k+1 during step k, feed_data in "
                         "the timed loop (the reference's CUDAPrefetcher); resident: feed once before the loop")
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--config", type=int, choices=[4, 5], default=4,
                    help="--mode infer: 4 = BASELINE configs[3] (512x512, T=7), 5 = configs[4] (1224x1632, T=15, tiled)")
    ap.add_argument("--max-minibatch", dest="max_minibatch", type=int, default=2,
                    help="--mode infer --config 5: tiles per forward call (val.max_minibatch)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N>1 weak runs: skip the extra strong-scaling timing")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step from captured hipGraphs (auto: while B*H*W <= 2*256*256 per GPU)")
    ap.add_argument("--dtype", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="fp32 = BASELINE configs[1] (headline); bf16 = config-3 style compute (bf16 MFMA operands)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo: --dry-run only")
    ap.add_argument("--dry-run", dest="dry_run", action="store_true",
                    help="launcher / collective / timing protocol only (CPU tensors, no model step, value=null)")
    args = ap.parse_args(argv)
    if args.backend == "gloo" and not args.dry_run:
        ap.error("--backend gloo is only meaningful with --dry-run (the HIP path has no CPU fallback)")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.scaling == "strong" and args.batch % args.gpus:
        ap.error(f"--scaling strong: global batch {args.batch} is not divisible by --gpus {args.gpus}")
    return args


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) of this same script."""
    if not args.dry_run:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible: "
                             "refusing to run fewer ranks than requested")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class _DryModel:
    """Stand-in step for --dry-run: the product's GradSync plan over a CPU arena of the real parameter inventory."""

    def __init__(self, args):
        from refid_amd.dist import GradSync
        from refid_amd.engine import ParamArena, param_shapes
        self.arena = ParamArena(param_shapes(args.img_chn), torch.device("cpu"))
        self.sync = GradSync(self.arena.flat_g, self.arena.offsets) if torch.distributed.is_initialized() else None

    def feed_data(self, data):
        pass

    def update_learning_rate(self, it):
        pass

    def optimize_parameters(self, it):
        self.arena.flat_g.fill_(1.0)
        if self.sync is not None:
            self.sync("early")
            self.sync("late")
            world = torch.distributed.get_world_size()
            if float(self.arena.flat_g[0]) != float(world) or float(self.arena.flat_g[-1]) != float(world):
                raise RuntimeError("dry-run gradient all-reduce returned a wrong sum")

    def get_current_log(self):
        return {"l_pix": 0.0}


_JSON_FD = None


def _claim_stdout():
    """stdout carries ONE JSON line and nothing else.  RCCL prints a banner (HIP / ROCm version, host, library path) on the
    C-level stdout when its first communicator comes up, and C stdio flushes it at exit -- i.e. AFTER the JSON line when stdout
    is a pipe.  So file descriptor 1 is pointed at stderr for everything native (and for stray Python prints), and the JSON
    line goes to a private duplicate of the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    if _JSON_FD is None:
        print(line, flush=True)
    else:
        os.write(_JSON_FD, (line + "\n").encode())


def main(argv=None, claim_stdout=False):
    """claim_stdout: only the process that IS the benchmark (`python bench.py`, every spawned rank) redirects file descriptor
    1; an in-process caller (tests, tools importing bench) keeps its stdout and gets the JSON line through print()."""
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))
    if claim_stdout and not args.dry_run:
        _claim_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); "
                         "they must match (one process per GPU)")
    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: local rank {local_rank} has no device ({torch.cuda.device_count()} visible): "
                             "one GPU per rank is required")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    pinned = []
    if world > 1 and os.environ.get("REFID_PIN_CORES", "1") != "0":
        # every rank keeps its enqueue thread (and OMP_NUM_THREADS helpers) on the cores next to its GPU
        from refid_amd.dist import pin_to_local_cores
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        pinned = pin_to_local_cores(local_rank, lw, numa_nodes=[-1] * lw if args.dry_run else None)
    use_dist = world > 1 or os.environ.get("REFID_FORCE_GRADSYNC") == "1"    # 1-rank RCCL dry run of the N>1 path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def sync():
        if use_dist:
            torch.distributed.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    if args.dry_run:
        model = _DryModel(args)
    else:
        from refid_amd import ops
        from refid_amd.train import TwoImageEventRecurrentRestorationModel
        torch.manual_seed(1234)                        # same init on every rank (then broadcast anyway)
        model = TwoImageEventRecurrentRestorationModel(options(args))
        with torch.no_grad():                          # released-checkpoint-like: beta/gamma ~ N(0, 0.1^2)
            for k, p in model.net_g.named_parameters():
                if k.endswith((".beta", ".gamma")):
                    p.normal_(0.0, 0.1)
        model.net_g.notify_params_changed()

    it = 0
    h2d = {"exposed_ms": None}

    class _HostBatches:
        """Endless re-iterable of synthetic batches that live in pinned host memory (what a DataLoader with
        pin_memory: true hands to the prefetcher); two distinct batches alternate."""

        def __init__(self, per_gpu_batch):
            self.batches = []
            for k in range(2):
                x, ev, gt = synthetic_batch(per_gpu_batch, args.T, args.size, args.size, args.img_chn, 100 + rank + 1000 * k, dev)
                self.batches.append({"lq": x.cpu().pin_memory(), "voxel": ev.cpu().pin_memory(), "gt": gt.cpu().pin_memory()})
                del x, ev, gt

        def __iter__(self):
            k = 0
            while True:
                yield dict(self.batches[k % 2])
                k += 1

    def timed(per_gpu_batch, steps, warmup):
        """W warm-up steps, then K steps bracketed by barrier + device sync; MAX over ranks."""
        nonlocal it
        pre = None
        if not args.dry_run:
            if args.h2d == "prefetch":
                from refid_amd.data import CUDAPrefetcher
                pre = CUDAPrefetcher(_HostBatches(per_gpu_batch), device=dev, time_waits=True)
            else:
                x, ev, gt = synthetic_batch(per_gpu_batch, args.T, args.size, args.size, args.img_chn, 100 + rank, dev)
                model.feed_data({"lq": x, "voxel": ev, "gt": gt})
            # auto (default): replayed from hipGraphs while B H W <= 2 x 256 x 256 per GPU (round 6: B=1 83.1 vs 85.9-87.5 ms
            # eager; the 8-GPU strong-scaling shard), eager launches above (B=8: the GPU is the limiter)
            if hasattr(model, "set_graph_mode"):
                model.set_graph_mode({"on": True, "off": False, "auto": "auto"}[args.graph])

        def one():
            nonlocal it
            it += 1
            if pre is not None:
                model.feed_data(pre.next())       # train.py:217-232: prefetcher.next() -> feed_data -> optimize_parameters
            model.update_learning_rate(it)
            model.optimize_parameters(it)

        for _ in range(warmup):
            one()
        sync()
        if pre is not None:
            pre.exposed_ms()                      # (clears the warm-up's records)
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        sync()
        dt = time.perf_counter() - t0
        if pre is not None:
            h2d["exposed_ms"] = round(pre.exposed_ms() / max(1, steps), 4)
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    if args.mode == "infer":
        return infer_main(args, model, dev, rank, world, use_dist, sync)

    per_gpu = args.batch if args.scaling == "weak" else args.batch // world
    dt = timed(per_gpu, args.steps, args.warmup)
    replayed = bool(getattr(model, "graph_on", False))          # did the timed steps run from hipGraphs? (--graph auto / on)
    loss = model.get_current_log()["l_pix"]

    roof = None
    if not args.no_roofline and not args.dry_run:
        # one extra, instrumented step (every rank runs it: the step contains collectives; only rank 0
        # records): HIP events around every conv-tile / wgrad launch on the launch stream; the dominant
        # kernel = the kernel with the largest accumulated time
        # For a per-kernel duration that means something against the roofline the instrumented step runs every kernel
        # alone on one stream (no forward wavefront; the weight gradients are on the main stream by default anyway).
        from refid_amd import engine as _engine
        overlap, _engine.OVERLAP_WGRAD = _engine.OVERLAP_WGRAD, False
        pipeline, _engine.PIPELINE = _engine.PIPELINE, False
        if hasattr(model, "set_graph_mode"):
            model.set_graph_mode(False)
        if rank == 0:
            ops.PROFILE = []
        it += 1
        model.update_learning_rate(it)
        model.optimize_parameters(it)
        torch.cuda.synchronize()
        _engine.OVERLAP_WGRAD = overlap
        _engine.PIPELINE = pipeline
        if rank == 0:
            prof, ops.PROFILE = ops.PROFILE, None
            roof = roofline_from_profile(prof, dt / args.steps, args.dtype,
                                         "per-kernel timing from one extra single-stream step (kernels run alone, HIP events on the "
                                         "launch stream); rocprof counterpart: profiles/*_nooverlap_kernel_stats.csv")

    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_strong_leg and args.batch % world == 0:
        # the same GLOBAL batch as the 1-GPU configuration, sharded over the ranks (B = batch/N per GPU)
        sdt = timed(args.batch // world, args.steps, args.warmup)
        strong = {"global_batch": args.batch, "per_gpu_batch": args.batch // world,
                  "value": None if args.dry_run else round(args.batch * args.T * args.steps / sdt, 3), "unit": "frames/s",
                  "ms_per_step": round(sdt / args.steps * 1e3, 2)}
    if use_dist:
        torch.distributed.barrier()

    if rank == 0:
        gbatch = per_gpu * world
        frames = gbatch * args.T * args.steps
        out = {
            "metric": METRIC, "value": None if args.dry_run else round(frames / dt, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "f32 tensors/accumulation, 3x3 conv products as 3 bf16 MFMAs (2^-16)",
                                          "bf16": "bf16"}[args.dtype], "data": "synthetic",
            "rccl_ranks": world if (use_dist and args.backend == "nccl") else (0 if not use_dist else None),
            "arithmetic": {"fp32": "fp32 tensors, operands and accumulation; 3x3: Winograd F(2x2,3x3), transforms in fp32; its "
                                   "transform-domain products: every fp32 operand as TWO fp16 numbers h = rne16(v), l = rne16(v - h) "
                                   "(22 significand bits), three fp16 MFMAs per product (hh + hl + lh, ~2^-22 per product, below the "
                                   "fp32 accumulation's own error), fp32 accumulation; fp16's range is bridged by exact power-of-two "
                                   "scales (weights per tensor, activations per Winograd tile and online along K), so any finite fp32 "
                                   "magnitude keeps the same relative error (tests/test_hip_conv.py::test_wino_f16_accuracy_class_and_"
                                   "dynamic_range: same distance from float64 as the fp32 MFMA tile; REFID_WINO_F16=0: six bf16 products "
                                   "on exact three-plane operands, round 5's form; REFID_WINO6=0: fp32 MFMA); conv_down (4x4 stride 2) "
                                   "fwd/dgrad: six bf16 MFMAs per product on exact three-bf16-plane operands (REFID_DOWN_SPLIT=0: fp32 "
                                   "MFMA); 3x3 "
                                   "weight gradients: fp32 MFMA in the Winograd domain over 2x4 tiles of the output gradient (F(3,2) x "
                                   "F(3,4), transforms and accumulation fp32; 3e-6 .. 6e-6 of a tensor's scale from the float64 gradient, "
                                   "test_wgrad_f4_accuracy_class; REFID_WGRAD_F4=0: 2x2 tiles); conv_down's weight gradient: the same kernel on the "
                                   "input's four parity phases (REFID_WGRAD_DOWN_F4=0: direct fp32 tile)",
                           "bf16x3": "fp32 tensors and accumulation; 3x3 / 4x4 forward and input-gradient products as three bf16 MFMAs "
                                     "(2^-16 per product); weight gradients fp32",
                           "bf16": "bf16 MFMA operands (forward, input and weight gradients), fp32 tensors / accumulation / "
                                   "optimizer"}[args.dtype],
            "config": {"workload": f"GoPro 11+1 blur-VFI train step, batch {per_gpu}/GPU"
                                   f"{' (global batch ' + str(gbatch) + ' sharded)' if args.scaling == 'strong' else ''}, "
                                   f"{args.size}x{args.size}, T={args.T}, img_chn={args.img_chn}, {args.dtype}" +
                                   (" (BASELINE configs[1])" if args.dtype == "fp32" and args.T == 23 and gbatch == 8 * (world if args.scaling == "weak" else 1) else ""),
                       "global_batch": gbatch, "parallelism": f"dp{world}", "loss": round(loss, 6)},
        }
        if not args.dry_run:
            out["launch"] = "hipGraph replay" if replayed else "eager launches"
            if getattr(model, "graph_fallback", None):
                out["launch"] += f" (graph capture failed, fell back: {model.graph_fallback})"
            out["h2d"] = ("prefetched, in timed region (pinned host batch -> HBM on a side stream during the previous step; "
                          "feed_data inside the loop)") if args.h2d == "prefetch" else "resident (fed once before the timed region)"
            out["h2d_ms_exposed"] = h2d["exposed_ms"]
        if pinned:
            out["rank0_cpu_affinity"] = f"{len(pinned)} cores ({pinned[0]}-{pinned[-1]})"
        gs = getattr(model, "grad_sync", None)
        if gs is not None and getattr(gs, "calls", 0):
            out["gradsync_host_ms_per_step"] = {k: round(v / gs.calls * 1e3, 3) for k, v in gs.host_s.items()}
        if args.dry_run:
            out["dry_run"] = True
            out["backend"] = args.backend
        if roof is not None:
            out["roofline"] = roof
        if strong is not None:
            out["strong"] = strong
        if not args.no_cpu_baseline and world == 1 and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(args)
        _emit(json.dumps(out))
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(claim_stdout=True)

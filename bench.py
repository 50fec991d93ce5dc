#!/usr/bin/env python3
"""Headline benchmark: interpolated frames/s of a TRAIN STEP of FinalBidirectionAttenfusion.

BASELINE.json metric: "interpolated frames/sec (train step) GoPro 256x256 11+1"; workload =
configs[1]: GoPro 11+1 blur-VFI, batch 8 per GPU, 256x256, T=23, img_chn=26, fp32, full
step (zero_grad -> forward -> Charbonnier -> BPTT -> clip 0.01 -> AdamW), synthetic inputs
already resident in HBM.  One process per GPU; N>1 is launched by torch.distributed.run and
uses RCCL (backend 'nccl') for the gradient all-reduce; per-GPU batch is fixed (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      the dominant kernel's achieved fp32 TFLOP/s (algorithmic FLOPs / HIP-event time
                of its launches in one instrumented step) against the 157.3 TFLOP/s fp32 MFMA peak;
  cpu_baseline  the CPU oracle (oracle/refid_oracle.py, kind "port") timed on this box's host
                cores for ONE train step at B=1 of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


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
    return {
        "name": "bench", "is_train": True, "num_gpu": 1,
        "network_g": dict(type="FinalBidirectionAttenfusion", img_chn=args.img_chn, ev_chn=2, num_encoders=3,
                          base_num_channels=32, num_block=1, num_residual_blocks=2, compute_dtype=args.dtype),
        "path": {"pretrain_network_g": None},
        "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                  "scheduler": dict(type="TrueCosineAnnealingLR", T_max=200000, eta_min=1e-7),
                  "pixel_opt": dict(type="CharbonnierLoss", loss_weight=1, reduction="mean")},
        "val": {"max_minibatch": 2},
    }


def cpu_baseline(args):
    """One oracle train step on the host cores, B=1 of the same workload (bounded sample)."""
    from oracle import refid_oracle as O
    torch.manual_seed(0)
    P = O.make_params(args.img_chn, mode="init", seed=0)
    for k in P:
        if k.endswith((".beta", ".gamma")):
            P[k] = torch.randn_like(P[k]) * 0.1
    st = O.TrainState(P)
    xw, ew, gw = O.make_inputs(1, 2, 32, 32, args.img_chn, mode="rng")          # thread-pool warm-up
    O.train_step({k: v.clone() for k, v in P.items()}, O.TrainState(P), xw, ew, gw)
    # bounded sample: B=1 and the first Tc of the T frames (cost is linear in frames: 2T recurrent
    # steps + one image branch), so frames/s is directly comparable
    Tc = min(args.T, args.cpu_frames)
    x, ev, gt = O.make_inputs(1, Tc, args.size, args.size, args.img_chn, seed=1, mode="rng")
    t0 = time.perf_counter()
    O.train_step(P, st, x, ev, gt)
    dt = time.perf_counter() - t0
    return {"value": round(Tc / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 oracle train step (fwd+bwd+clip+AdamW), B=1, {Tc} of T={args.T} frames, "
                      f"{args.size}x{args.size}, fp32, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--T", type=int, default=23)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--img-chn", dest="img_chn", type=int, default=26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", dest="cpu_frames", type=int, default=5, help="frames in the CPU-baseline sample")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = BASELINE configs[1] (headline); bf16 = config-3 style compute (bf16 MFMA operands)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("REFID_FORCE_GRADSYNC") == "1"    # 1-rank RCCL dry run of the N>1 path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from refid_amd import ops
    from refid_amd.train import TwoImageEventRecurrentRestorationModel

    torch.manual_seed(1234)                        # same init on every rank (then broadcast anyway)
    model = TwoImageEventRecurrentRestorationModel(options(args))
    with torch.no_grad():                          # released-checkpoint-like: beta/gamma ~ N(0, 0.1^2)
        for k, p in model.net_g.named_parameters():
            if k.endswith((".beta", ".gamma")):
                p.normal_(0.0, 0.1)
    model.net_g.notify_params_changed()
    x, ev, gt = synthetic_batch(args.batch, args.T, args.size, args.size, args.img_chn, 100 + rank, dev)
    model.feed_data({"lq": x, "voxel": ev, "gt": gt})

    def sync():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        it += 1
        model.update_learning_rate(it)
        model.optimize_parameters(it)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        model.update_learning_rate(it)
        model.optimize_parameters(it)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    loss = model.get_current_log()["l_pix"]

    roof = None
    if not args.no_roofline:
        # one extra, instrumented step (every rank runs it: the step contains collectives; only rank 0
        # records): HIP events around every conv-tile / wgrad launch on the launch stream; the dominant
        # kernel = the kernel with the largest accumulated time
        # The timed steps run the weight-gradient kernels on a side stream, concurrently with the dgrad chain;
        # for a per-kernel duration that means something against the roofline the instrumented step runs
        # every kernel alone on one stream (same as REFID_OVERLAP_WGRAD=0).
        from refid_amd import engine as _engine
        overlap, _engine.OVERLAP_WGRAD = _engine.OVERLAP_WGRAD, False
        if rank == 0:
            ops.PROFILE = []
        it += 1
        model.update_learning_rate(it)
        model.optimize_parameters(it)
        torch.cuda.synchronize()
        _engine.OVERLAP_WGRAD = overlap
    if not args.no_roofline and rank == 0:
        prof, ops.PROFILE = ops.PROFILE, None
        agg = {}
        for name, fl, e0, e1, _shape, nb in prof:
            a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
            a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1; a[3] += nb
        name, (fl, sec, cnt, nbytes) = max(agg.items(), key=lambda kv: kv[1][1])
        # HBM traffic of the same kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, tools/pmc_traffic.py); bench.py cannot collect hardware counters itself
        traffic = None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                rec = json.load(open(path)).get(name)
            except (OSError, ValueError):
                rec = None
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
                break
        # Winograd F(2x2,3x3) executes 16/36 of the direct convolution's multiplies on the matrix
        # pipe: price the kernel against the MFMA roof with the FLOPs it actually issues, and keep
        # the SURVEY 8(d) direct-convolution figure beside it.
        executed = fl * (16.0 / 36.0) if "wino" in name else fl
        peak = FP32_MFMA_PEAK_TFLOPS
        if args.dtype == "bf16" and "wino" not in name and "wgrad" not in name:
            peak = 2500.0                  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
        ach = executed / sec / 1e12
        conv_fl = sum(v[0] for v in agg.values()); conv_t = sum(v[1] for v in agg.values())
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC, profiles/)",
                "algorithmic_bytes_per_launch": round(nbytes / cnt), "kernel": name, "launches": cnt,
                "avg_launch_us": round(sec / cnt * 1e6, 2),
                "direct_conv_equivalent_tflops": round(fl / sec / 1e12, 2),
                "note": "per-kernel timing from one extra single-stream step (kernels run alone; the timed steps "
                        "overlap wgrad kernels on a side stream); rocprof counterpart: profiles/*_nooverlap_kernel_stats.csv",
                "all_gemm_kernels": {"direct_conv_equivalent_tflops": round(conv_fl / conv_t / 1e12, 2),
                                     "share_of_step": round(conv_t / (dt / args.steps), 3)}}
    if use_dist:
        torch.distributed.barrier()

    if rank == 0:
        frames = args.batch * args.T * world * args.steps
        out = {
            "metric": "interpolated frames/sec (train step) GoPro 256x256 11+1", "value": round(frames / dt, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"GoPro 11+1 blur-VFI train step, batch {args.batch}/GPU, {args.size}x{args.size}, "
                                   f"T={args.T}, img_chn={args.img_chn}, {args.dtype}" +
                                   (" (BASELINE configs[1])" if args.dtype == "fp32" and args.T == 23 else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "loss": round(loss, 6)},
        }
        if roof is not None:
            out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the hot path: EnCodec-24k, 8 codebooks, encode + decode of 64 x 10 s per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = Codec.sig_to_toks + Codec.toks_to_sig over one synthetic batch that is already resident
in HBM (BASELINE.json configs[1]; weights: seeded synthetic checkpoint -- no pretrained weights
exist offline).  Clips are independent units: every rank encodes/decodes its own 64 clips
(weak scaling); with N > 1 the only collective is the RCCL all_gather of the token ids, and it is
inside the timed step.  Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline     -- dominant kernel (by HIP-event time measured in the timed steps) against the pipe that bounds it:
                  the dense bf16 MFMA peak (2.5 PF) for the split-operand tap-GEMM (6 bf16 partial products per fp32
                  product, tap_gemm6.h; AC_GEMM=fp32 selects the exact-product kernels, bounded by the 157.3 TF fp32
                  MFMA peak) or HBM (8 TB/s)
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference) timed on this host's
                  cores on a bounded sample of the same workload.  Baseline only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
SPLIT_TERMS = 6                 # tap_gemm6.h: bf16 partial products executed per fp32 product
PEAK_HBM_GBS = 8000.0          # same guide, "HBM3E peak BW" (spec; 6.29 TB/s achievable)
# algorithmic work per audio-second of encode+decode (EnCodec-24k, K=8)
FLOP_PER_AUDIO_S = 6.12e9
LAYER_BYTES_PER_AUDIO_S = 117.6e6


def cpu_baseline_mimi(cfg, sd, sig_cpu, clips=8):
    """Same protocol with the Mimi oracle (oracle/mimi_oracle.py)."""
    from oracle import mimi_oracle as O  # checker/baseline only -- never on the product path

    W = O.cast_weights(sd)
    x = sig_cpu[:clips]
    ncpu = os.cpu_count() or 2
    cands = sorted({t for t in (16, 32, 64, ncpu // 2) if 1 <= t <= ncpu}) or [ncpu]
    best, best_t = None, None
    with torch.inference_mode():
        for t in cands:
            torch.set_num_threads(t)
            O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x[:1, :48000]))
            t0 = time.perf_counter()
            O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, best_t = dt, t
    audio_s = x.shape[0] * x.shape[1] / cfg.sampling_rate
    return {
        "value": round(audio_s / best, 2), "unit": "audio-s/s", "cores": best_t, "kind": "port",
        "sample": f"{clips} clips x {x.shape[1] / cfg.sampling_rate:.0f} s of the same batch, fp32 torch-CPU Mimi oracle; "
                  f"best of thread counts {cands} (one timed run each after a warm-up)",
    }


def cpu_baseline_dac(cfg, sd, sig_cpu, clips=2):
    """Same protocol with the DAC oracle (oracle/dac_oracle.py); 2 clips: DAC is ~100 GMAC per audio-second."""
    from oracle import dac_oracle as O  # checker/baseline only -- never on the product path

    W = O.cast_weights(sd)
    x = sig_cpu[:clips]
    ncpu = os.cpu_count() or 2
    cands = sorted({t for t in (32, 64, ncpu // 2) if 1 <= t <= ncpu}) or [ncpu]
    best, best_t = None, None
    with torch.inference_mode():
        for t in cands:
            torch.set_num_threads(t)
            O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x[:1, :44100], None, 9))
            t0 = time.perf_counter()
            O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x, None, 9))
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, best_t = dt, t
    audio_s = x.shape[0] * x.shape[1] / cfg.sampling_rate
    return {
        "value": round(audio_s / best, 2), "unit": "audio-s/s", "cores": best_t, "kind": "port",
        "sample": f"{clips} clips x {x.shape[1] / cfg.sampling_rate:.0f} s of the same batch, fp32 torch-CPU DAC oracle; "
                  f"best of thread counts {cands} (one timed run each after a warm-up)",
    }


def cpu_baseline(cfg, sd, sig_cpu, clips=8):
    """Oracle (kind 'port': torch-CPU restatement of the reference, oracle/encodec_oracle.py) on the
    host cores.  torch's CPU convs do not scale to every core of a 2-socket host on 8 clips, so a
    few thread counts are tried (short warm-up each) and the best one is reported with its count."""
    from oracle import encodec_oracle as O  # checker/baseline only -- never on the product path

    W = O.fold_weight_norm(sd)
    x = sig_cpu[:clips]
    ncpu = os.cpu_count() or 2
    cands = sorted({t for t in (16, 32, 64, ncpu // 2) if 1 <= t <= ncpu}) or [ncpu]
    best, best_t = None, None
    with torch.inference_mode():
        for t in cands:
            torch.set_num_threads(t)
            O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x[:1, :48000]))  # warm-up (thread pool, allocator)
            t0 = time.perf_counter()
            toks = O.sig_to_toks(cfg, W, x)
            O.toks_to_sig(cfg, W, toks)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, best_t = dt, t
    audio_s = x.shape[0] * x.shape[1] / cfg.sampling_rate
    return {
        "value": round(audio_s / best, 2),
        "unit": "audio-s/s",
        "cores": best_t,
        "kind": "port",
        "sample": f"{clips} clips x {x.shape[1] / cfg.sampling_rate:.0f} s of the same batch, fp32 torch-CPU oracle; "
                  f"best of thread counts {cands} (one timed run each after a warm-up)",
    }


def measured_traffic(kernel_name, unit, codec="encodec", batch=64):
    """HBM bytes per launch of `kernel_name` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 FETCH x2 correction: tools/collect_traffic.py).  bench.py cannot
    run the profiler on itself, so the newest profiles/r*_traffic.json is quoted; null if absent."""
    import glob

    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")) if (codec in os.path.basename(f)) or (codec == "encodec" and not any(c in os.path.basename(f) for c in ("mimi", "dac"))))
    import re

    def batch_ok(f):   # r1_mimi_b32_traffic.json was collected at 32 clips per GPU; no tag = the default 64
        m = re.search(r"_b(\d+)_", os.path.basename(f))
        return (int(m.group(1)) if m else 64) == batch

    files = [f for f in files if batch_ok(f)]
    if not files:
        return None
    for f in reversed(files):   # newest summary that knows this kernel
        try:
            k = json.load(open(f))["kernels"].get(kernel_name)
        except Exception:
            continue
        if k and k["hbm_bytes_per_launch"] is not None:
            return round(k["hbm_bytes_per_launch"])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--codec", choices=["encodec", "mimi", "dac"], default="encodec",
                    help="encodec = BASELINE.json configs[1] (the contract's default); mimi = configs[3] shape (SURVEY.md §8 f3); "
                         "dac = configs[2] (DAC 44.1 kHz, 9 codebooks; use --batch 256)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from audiocodecs_amd import DAC, Encodec, Mimi, checkpoint, prng
    from audiocodecs_amd.config import DAC_44KHZ, ENCODEC_24KHZ, MIMI_24KHZ
    from audiocodecs_amd.sharding import gather_tokens

    mimi = args.codec != "encodec"   # "not the headline codec": whole-path fractions from the kernels' own counts
    cfg = {"mimi": MIMI_24KHZ, "dac": DAC_44KHZ, "encodec": ENCODEC_24KHZ}[args.codec]
    label = {"mimi": "Mimi-24k", "dac": "DAC-44.1k", "encodec": "EnCodec-24k"}[args.codec]
    ncb = 9 if args.codec == "dac" else 8
    B, T = args.batch, int(round(args.seconds * cfg.sampling_rate))
    if args.codec == "mimi":
        sd = checkpoint.synthetic_mimi_state_dict(cfg, seed=0)
        codec = Mimi(cfg.sampling_rate, num_codebooks=8, state_dict=sd).eval()
    elif args.codec == "dac":
        sd = checkpoint.synthetic_dac_state_dict(cfg, seed=0)
        codec = DAC(cfg.sampling_rate, cfg.sampling_rate, num_codebooks=9, state_dict=sd, config=cfg).eval()
    else:
        sd = checkpoint.synthetic_state_dict(cfg, seed=0)
        codec = Encodec(cfg.sampling_rate, num_codebooks=8, state_dict=sd).eval()
    # sig = 0.1*N(0,1), repo PRNG seed 123; each rank draws its own shard
    sig_cpu = torch.from_numpy((prng.normal(123, f"bench.sig.rank{rank}", (B, T)) * 0.1).astype(np.float32))
    sig = sig_cpu.cuda()

    def step():
        toks = codec.sig_to_toks(sig)
        if dist is not None:
            gather_tokens(toks)
        return codec.toks_to_sig(toks)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        stats = codec.profile_kernels(lambda: [step() for _ in range(args.steps)])
        fence()
        dt = time.perf_counter() - t0
        # unprofiled repeat (no per-kernel events on the stream) as a side figure
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_plain = time.perf_counter() - t1

    if dist is not None:
        tt = torch.tensor([dt, dt_plain], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_plain = tt.tolist()

    audio_s = world * B * T / cfg.sampling_rate * args.steps
    if rank == 0:
        ms = dt / args.steps * 1e3
        stats.sort(key=lambda s: -s[2])
        name, launches, tot_ms, flops, nbytes = stats[0]
        avg_us = tot_ms / launches * 1e3
        ai = flops / max(nbytes, 1.0)
        if name.startswith("tap_gemm6"):
            # split-operand GEMM (tap_gemm6.h): every fp32 product is 6 bf16 MFMA partial products, so the kernel's
            # roofline is the dense bf16 MFMA peak; `achieved` counts the bf16 flops it actually executes
            eq = flops / (tot_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(SPLIT_TERMS * eq, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "pipe": f"bf16 MFMA, {SPLIT_TERMS} partial products per fp32 product", "fp32_equivalent_tflops": round(eq, 2)}
        elif ai > PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
            roof = {"bound": "mfma", "achieved": round(flops / (tot_ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(nbytes / (tot_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["traffic"] = measured_traffic(name, roof["unit"], args.codec, B)
        roof["kernel"] = name
        roof["launches_per_step"] = launches / args.steps
        roof["avg_launch_us"] = round(avg_us, 2)
        roof["share_of_step"] = round(tot_ms / (dt * 1e3), 4)
        out = {
            "metric": f"encode+decode audio-sec/s, {label} {ncb}cb",
            "value": round(audio_s / dt, 1),
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if os.environ.get("AC_GEMM") == "fp32" else "f32 (GEMMs and LSTM products: operands split exactly into 3 bf16 terms, 6 partial products, fp32 accumulate)",
            "data": f"synthetic (0.1*N(0,1) clips, seeded synthetic weights of the {label} architecture)",
            "config": {"workload": f"{label} {ncb} codebooks, encode+decode, {B} clips x {args.seconds:g} s per GPU, resident in HBM",
                       "clips_per_gpu": B, "seconds_per_clip": args.seconds, "parallelism": f"clip-sharded x{world}"},
            "rtf": round(dt / audio_s, 7),
            "x_realtime_per_gpu": round(audio_s / dt / world, 1),
            "whole_path": {
                # EnCodec: SURVEY.md §8(d) per-audio-second figures; Mimi: the kernels' own algorithmic counts
                "mfma_fp32_frac": round((sum(s[3] for s in stats) if mimi else FLOP_PER_AUDIO_S * audio_s / world) / dt / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                "hbm_layer_boundary_frac": round((sum(s[4] for s in stats) if mimi else LAYER_BYTES_PER_AUDIO_S * audio_s / world) / dt / (PEAK_HBM_GBS * 1e9), 4),
                "ms_per_step_without_kernel_events": round(dt_plain / args.steps * 1e3, 3),
            },
            "roofline": roof,
            "kernels": [
                {"name": s[0], "launches_per_step": s[1] / args.steps, "ms_per_step": round(s[2] / args.steps, 3),
                 "tflops": round(s[3] / (s[2] * 1e-3) / 1e12, 2), "gbs": round(s[4] / (s[2] * 1e-3) / 1e9, 1)}
                for s in stats
            ],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = {"mimi": cpu_baseline_mimi, "dac": cpu_baseline_dac, "encodec": cpu_baseline}[args.codec](cfg, sd, sig_cpu)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
                  the dense fp16 MFMA peak (2.5 PF) for the split-operand tap-GEMM (3 fp16 partial products per fp32
                  product, split16.h; --precision fp32_bf16x3: 6 bf16 products, tap_gemm6.h; fp32_exact: the exact-product
                  kernels, bounded by the 157.3 TF fp32 MFMA peak) or HBM (8 TB/s)
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference) timed on this host's
                  cores on a bounded sample of the same workload.  Baseline only.
and, at N = 1, outside the timed region (all skipped by --no-parity):
  parity                 -- token exact-match and decode RMS error on the reference-generated fixture
  exact_fp32_ms_per_step -- the same step with every product an IEEE fp32 product (precision="fp32_exact"), so the
                            split16 figure never travels without its exact-product twin
  other_configs          -- BASELINE.json configs 3-5 at their per-GPU sizes (DAC 256 x 10 s, Mimi 128 x 10 s,
                            WavTokenizer 64 x 10 s): value, ms_per_step and parity gate of a short (1 + 3 step) run each
Nothing inside an `if rank == 0` block issues a collective (tests/test_bench_contract.py checks the source for it).
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
SPLIT_TERMS = 6                 # tap_gemm6.h, three bf16 planes: partial products executed per fp32 product
SPLIT16_TERMS = 3               # split16.h, two fp16 planes (the default arithmetic): partial products per fp32 product
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: dense fp16 MFMA peak = the bf16 one
# bare v_mfma_f32_32x32x16_f16 stream on random operands (tools/ubench/mfma_f16_probe.hip): 1.58-1.68 PF at 1.54-1.69 GHz
SUSTAINED_F16_MFMA_TFLOPS = 1680.0
PEAK_HBM_GBS = 8000.0          # same guide, "HBM3E peak BW" (spec; 6.29 TB/s achievable)
# What the bf16 matrix pipe SUSTAINS on random operands: the bare v_mfma_f32_32x32x16_bf16 stream keeps the pipe 100 % busy and
# power management drops the shader clock to 1.66-1.8 GHz (measured in-kernel, profiles/r2_tapgemm_variants.md): 1.69-1.83 PF
SUSTAINED_BF16_MFMA_TFLOPS = 1760.0
# algorithmic work per audio-second of encode+decode (EnCodec-24k, K=8)
FLOP_PER_AUDIO_S = 6.12e9
LAYER_BYTES_PER_AUDIO_S = 117.6e6


def host_cpu():
    """(model name, physical cores, hardware threads) of this host from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    threads = os.cpu_count() or 1
    return model, (len(cores) or threads), threads


def cpu_baseline(codec_name, cfg, sd, sig_cpu, ncb):
    """The CPU oracle (kind "port": torch-CPU restatement of the reference's path, oracle/*.py -- checker/baseline only,
    never on the product path) timed on this host's cores, BASELINE.md section 3 protocol: 2 warm-ups + 3 timed runs of
    encode+decode on a bounded sample of the same batch, median reported.  torch's CPU convolutions do not scale to
    every core of a 2-socket host on a few clips, so the thread count is picked first by a short scan (1 clip x 2 s each)
    and reported as `cores` next to the host's physical core count and CPU model."""
    if codec_name == "mimi":
        from oracle import mimi_oracle as O
        W = O.cast_weights(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
        clips = 4
    elif codec_name == "dac":
        from oracle import dac_oracle as O
        W = O.cast_weights(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x, None, ncb))
        clips = 1   # ~100 GMAC per audio-second
    elif codec_name == "wavtokenizer":
        from oracle import wavtokenizer_oracle as O
        W = O.cast_weights(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
        clips = 4
    else:
        from oracle import encodec_oracle as O
        W = O.fold_weight_norm(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
        clips = 4
    sr = cfg.sampling_rate
    x = sig_cpu[:clips]
    model, phys, threads = host_cpu()
    cands = sorted({t for t in (8, 16, 32, 64, phys) if 1 <= t <= threads}) or [threads]
    scan = {}
    with torch.inference_mode():
        probe = x[:1, : 2 * sr]
        for t in cands:
            torch.set_num_threads(t)
            run(probe)
            t0 = time.perf_counter()
            run(probe)
            scan[t] = time.perf_counter() - t0
        best_t = min(scan, key=scan.get)
        torch.set_num_threads(best_t)
        for _ in range(2):
            run(x[:1])
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(x)
            times.append(time.perf_counter() - t0)
    audio_s = x.shape[0] * x.shape[1] / sr
    med = sorted(times)[1]
    return {
        "value": round(audio_s / med, 2), "unit": "audio-s/s", "cores": best_t, "kind": "port",
        "host_physical_cores": phys, "host_hw_threads": threads, "cpu_model": model,
        "runs_s": [round(t, 3) for t in times],
        "sample": f"{clips} clips x {x.shape[1] / sr:.0f} s of the same batch, fp32 torch-CPU oracle, encode+decode; "
                  f"2 warm-ups + 3 timed runs (median); thread count {best_t} chosen from {cands} by a 2 s probe",
    }


def parity_gate(codec_name, codec):
    """the two parity figures that travel with every throughput number, computed outside the timed
    region on the committed reference-generated fixture `full_noise_b2` (tests/golden/*.npz: the reference wrapper's own
    tokens and waveform for a seeded input; DAC: the stand-in's, WavTokenizer: the unpinned oracle's): token exact-match
    rate of sig_to_toks and RMS error of toks_to_sig on the fixture's tokens."""
    tests = os.path.join(ROOT, "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    mod = {"encodec": "golden_cases", "mimi": "mimi_cases", "dac": "dac_cases", "wavtokenizer": "wavtok_cases"}[codec_name]
    cases = __import__(mod)
    gdir = os.path.join(tests, "golden")
    z = np.load(os.path.join(gdir, f"{codec_name}_golden.npz"))
    case = next(c for c in cases.CASES if c["name"] == "full_noise_b2")
    inp = cases.make_input(case, gdir)
    gold = z["full_noise_b2.toks"].astype(np.int64)
    toks = codec.sig_to_toks(inp["sig"].cuda()).cpu().numpy()
    rec = codec.toks_to_sig(torch.from_numpy(gold).cuda()).cpu().numpy()
    err = rec.reshape(-1)[:: cases.REC_STRIDE].astype(np.float64) - z["full_noise_b2.rec_strided"]
    margin = z["full_noise_b2.margin64"]
    safe = np.cumprod(margin > 1e-4, axis=-1).astype(bool)
    pinned = {"encodec": "reference wrapper (transformers EncodecModel)", "mimi": "reference wrapper (transformers MimiModel)",
              "dac": "stand-in transformers.DacModel (reference backend not on disk: parity unpinned)",
              "wavtokenizer": "oracle only (reference backend not on disk: parity unpinned)"}[codec_name]
    return {
        "fixture": "full_noise_b2", "pinned_to": pinned,
        "token_exact_match": round(float((toks == gold).mean()), 6), "tokens": int(gold.size),
        "token_mismatches_outside_fp64_near_ties": int(((toks != gold) & safe).sum()),
        "decode_rms_err": float(np.sqrt(np.mean(err ** 2))), "decode_rms_bar": 1e-4,
    }


def mfma16_terms(kernel_name, mode):
    """16-bit MFMA partial products a kernel executes per fp32 product it stands for (0: not on the 16-bit matrix pipe)."""
    split = kernel_name.startswith(("tap_gemm6", "lstm_persist6", "lstm_persist16", "rb_fused6", "rb128_fused6", "thin_conv6"))
    if kernel_name.startswith(("enc_front", "dec_tail", "enc_mid", "dec_mid", "lstm_persist16")):
        return SPLIT16_TERMS
    if not split:
        return 0
    if kernel_name.rstrip().endswith(", 1>"):
        return 1
    if kernel_name.rstrip().endswith(", 2>"):
        return SPLIT16_TERMS
    return SPLIT_TERMS if mode != "bf16" else 1


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
            return round(k["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(f)
    return None


CODEC_LABEL = {"mimi": "Mimi-24k", "dac": "DAC-44.1k", "encodec": "EnCodec-24k", "wavtokenizer": "WavTokenizer-24k-40tok"}
CODEC_NCB = {"dac": 9, "wavtokenizer": 1, "mimi": 8, "encodec": 8}


def build_codec(name, precision=None):
    """(codec, cfg, state dict) of one of the four wrappers on seeded synthetic weights of the full architecture."""
    from audiocodecs_amd import DAC, Encodec, Mimi, WavTokenizer, checkpoint
    from audiocodecs_amd.config import DAC_44KHZ, ENCODEC_24KHZ, MIMI_24KHZ, WAVTOK_40

    cfg = {"mimi": MIMI_24KHZ, "dac": DAC_44KHZ, "encodec": ENCODEC_24KHZ, "wavtokenizer": WAVTOK_40}[name]
    if name == "mimi":
        sd = checkpoint.synthetic_mimi_state_dict(cfg, seed=0)
        codec = Mimi(cfg.sampling_rate, num_codebooks=8, state_dict=sd, precision=precision).eval()
    elif name == "dac":
        sd = checkpoint.synthetic_dac_state_dict(cfg, seed=0)
        codec = DAC(cfg.sampling_rate, cfg.sampling_rate, num_codebooks=9, state_dict=sd, config=cfg, precision=precision).eval()
    elif name == "wavtokenizer":
        sd = checkpoint.synthetic_wavtok_state_dict(cfg, seed=0)
        codec = WavTokenizer(cfg.sampling_rate, state_dict=sd, arch=cfg, precision=precision).eval()
    else:
        sd = checkpoint.synthetic_state_dict(cfg, seed=0)
        codec = Encodec(cfg.sampling_rate, num_codebooks=8, state_dict=sd, precision=precision).eval()
    return codec, cfg, sd


def drop_codec(codec):
    """Release a codec's device memory (weights + workspace) before the next one is built."""
    import gc

    for nat in list(getattr(codec, "_natives", {}).values()):
        nat.ws = None
        if getattr(nat, "h", None):
            nat.lib.ac_destroy(nat.h)
            nat.h = None
    codec._natives.clear()
    gc.collect()
    torch.cuda.empty_cache()


def short_run(name, batch, seconds, steps, warmup, precision=None):
    """One of the other BASELINE.json configs as a short driver-timed run of the same step (encode + decode of `batch` clips
    resident in HBM), with its parity gate: value, ms_per_step and the gate travel in the headline JSON line (`other_configs`)."""
    from audiocodecs_amd import prng

    codec, cfg, sd = build_codec(name, precision)
    T = int(round(seconds * cfg.sampling_rate))
    sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
    with torch.no_grad():
        for _ in range(warmup):
            codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gate = parity_gate(name, codec)
    del sig
    drop_codec(codec)
    audio_s = batch * T / cfg.sampling_rate * steps
    return {"workload": f"{CODEC_LABEL[name]} {CODEC_NCB[name]} codebooks, encode+decode, {batch} clips x {seconds:g} s on 1 GPU, resident in HBM",
            "value": round(audio_s / dt, 1), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
            "parity": gate}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--codec", choices=["encodec", "mimi", "dac", "wavtokenizer"], default="encodec",
                    help="encodec = BASELINE.json configs[1] (the contract's default); mimi = configs[3] shape (SURVEY.md §8 f3); "
                         "dac = configs[2] (DAC 44.1 kHz, 9 codebooks; use --batch 256); wavtokenizer = configs[4] (40 tok/s, 64 clips per GPU)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--precision", choices=["fp32", "fp32_exact", "bf16", "fp32_bf16x3"], default=None,
                    help="arithmetic of the GEMM-shaped kernels: default = fp32 fidelity (split-operand; the parity arithmetic, what `value` is "
                         "quoted for); bf16 = OPT-IN reduced precision, a reported side mode with its own parity figures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates and every other untimed extra (profiling passes: keeps the kernel trace to the timed workload)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE.json configs 3-5 (DAC, Mimi, WavTokenizer) that travel in the same JSON line")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32-product twin of the headline figure")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dist = None
    # under torch.distributed.run (RANK in the environment) the collective path runs even at world size 1, so that a 1-GPU
    # box can exercise exactly what the N > 1 launches do (RCCL init, barrier, token all_gather, MAX-reduce of the timings)
    if world > 1 or (os.environ.get("RANK") is not None and os.environ.get("MASTER_PORT") is not None):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from audiocodecs_amd import prng
    from audiocodecs_amd.sharding import gather_tokens

    mimi = args.codec != "encodec"   # "not the headline codec": whole-path fractions from the kernels' own counts
    label, ncb = CODEC_LABEL[args.codec], CODEC_NCB[args.codec]
    codec, cfg, sd = build_codec(args.codec, args.precision)
    B, T = args.batch, int(round(args.seconds * cfg.sampling_rate))
    # sig = 0.1*N(0,1), repo PRNG seed 123; each rank draws its own shard
    sig_cpu = torch.from_numpy((prng.normal(123, f"bench.sig.rank{rank}", (B, T)) * 0.1).astype(np.float32))
    sig = sig_cpu.cuda()

    def step():
        toks = codec.sig_to_toks(sig)
        if dist is not None:
            gather_tokens(toks, force=True)
        return codec.toks_to_sig(toks)

    def local_step():   # the same work without the collective: for rank-local diagnostics outside the timed region
        return codec.toks_to_sig(codec.sig_to_toks(sig))

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
    mode_ = args.precision or {"fp32": "fp32_exact", "bf16": "bf16", "bf16x3": "fp32_bf16x3"}.get(os.environ.get("AC_GEMM", ""), "fp32")
    if rank == 0:
        ms = dt / args.steps * 1e3
        stats.sort(key=lambda s: -s[2])
        # The dominant kernel.  tap_gemm6_kernel is ONE kernel template launched in several tile arrangements (1 x 4 waves
        # over 128 or 256 columns, 1 x 8 waves, ...; rocprof lists each instantiation on its own row): the arrangements are
        # taken together when ranking, and listed one by one (with their own average launch time, the figure to compare
        # with rocprofv3's rows) under roofline.arrangements.
        fam = {}
        for s_ in stats:
            f_ = s_[0].split("<")[0] if s_[0].startswith("tap_gemm6") else s_[0]
            a_ = fam.setdefault(f_, [f_, 0, 0.0, 0.0, 0.0, []])
            a_[1] += s_[1]; a_[2] += s_[2]; a_[3] += s_[3]; a_[4] += s_[4]; a_[5].append(s_)
        top = max(fam.values(), key=lambda a_: a_[2])
        name, launches, tot_ms, flops, nbytes, members = top
        if len(members) == 1:
            name = members[0][0]
        avg_us = tot_ms / launches * 1e3
        ai = flops / max(nbytes, 1.0)
        split_kernel = name.startswith(("tap_gemm6", "lstm_persist6", "rb_fused6", "rb128_fused6", "thin_conv6", "enc_front", "dec_tail", "enc_mid", "dec_mid"))
        if name.startswith("tap_gemm6") and mode_ == "bf16":
            tf = flops / (tot_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "pipe": "bf16 MFMA, one product per operand pair"}
        elif split_kernel:
            # split-operand kernels (tap_gemm6.h arithmetic): every fp32 product is 6 bf16 MFMA partial products, so the kernel's
            # roofline is the dense bf16 MFMA peak; `achieved` counts the bf16 flops it actually executes
            eq = flops / (tot_ms * 1e-3) / 1e12
            s16 = all(m_[0].rstrip().endswith(", 2>") or m_[0].startswith(("enc_front", "dec_tail", "enc_mid", "dec_mid")) for m_ in members)     # split16.h kernels (template argument NP = 2; the fused chains exist in that arithmetic only)
            terms, cap = (SPLIT16_TERMS, SUSTAINED_F16_MFMA_TFLOPS) if s16 else (SPLIT_TERMS, SUSTAINED_BF16_MFMA_TFLOPS)
            roof = {"bound": "mfma", "achieved": round(terms * eq, 1), "peak": PEAK_F16_MFMA_TFLOPS if s16 else PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "pipe": (f"fp16 MFMA, {terms} partial products per fp32 product (two fp16 planes per operand, split16.h)" if s16 else
                             f"bf16 MFMA, {terms} partial products per fp32 product"), "fp32_equivalent_tflops": round(eq, 2),
                    "fp32_equivalent_peak": round((PEAK_F16_MFMA_TFLOPS if s16 else PEAK_BF16_MFMA_TFLOPS) / terms, 1),
                    "power_capped_peak": cap, "frac_of_power_capped_peak": round(terms * eq / cap, 4)}
        elif ai > PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
            roof = {"bound": "mfma", "achieved": round(flops / (tot_ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(nbytes / (tot_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        if len(members) > 1:
            trs = [(m_[1], measured_traffic(m_[0], roof["unit"], args.codec, B)) for m_ in members]
            tr = None
            if all(t_ for _, t_ in trs):
                tr = (round(sum(n_ * t_[0] for n_, t_ in trs) / sum(n_ for n_, _ in trs)), trs[0][1][1])
            roof["arrangements"] = [{"kernel": m_[0], "launches_per_step": m_[1] / args.steps, "avg_launch_us": round(m_[2] / m_[1] * 1e3, 2),
                                     "fp32_equivalent_tflops": round(m_[3] / (m_[2] * 1e-3) / 1e12, 2)} for m_ in members]
        else:
            tr = measured_traffic(name, roof["unit"], args.codec, B)
        if name.startswith("lstm_persist"):
            roof["note"] = "sequential recurrence: latency-bound (one exchange of h per time step), not a throughput kernel"
        roof["traffic"] = tr[0] if tr else None
        # PMC counters cannot be collected from inside this process: the figure is the one of the committed rocprofv3
        # --pmc passes over this same command (a different run / box); null when no such summary knows the kernel
        roof["traffic_source"] = tr[1] if tr else None
        # the same launches against the HBM roofline: algorithmic bytes (inputs read once, outputs
        # written once, weights once) per launch / measured launch time
        roof["algorithmic_bytes_per_launch"] = round(nbytes / launches)
        roof["hbm_gbs"] = round(nbytes / (tot_ms * 1e-3) / 1e9, 1)
        roof["hbm_frac"] = round(nbytes / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        roof["kernel"] = name if len(members) == 1 else f"{name}<...> ({len(members)} tile arrangements of one kernel)"
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
            "dtype": {"fp32_exact": "f32", "bf16": "bf16 operands / f32 accumulate in the tap-GEMMs (OPT-IN side mode, not the parity arithmetic); everything else f32-faithful"}.get(
                mode_, "f32 (GEMM-shaped kernels and LSTM products: operands as two scaled fp16 planes, 3 partial products, fp32 accumulate -- split16.h)"
                       if mode_ == "fp32" else
                       "f32 (GEMMs and LSTM products: operands split exactly into 3 bf16 terms, 6 partial products, fp32 accumulate)"),
            "data": f"synthetic (0.1*N(0,1) clips, seeded synthetic weights of the {label} architecture)",
            "config": {"workload": f"{label} {ncb} codebooks, encode+decode, {B} clips x {args.seconds:g} s per GPU, resident in HBM",
                       "clips_per_gpu": B, "seconds_per_clip": args.seconds, "parallelism": f"clip-sharded x{world}"},
            "rtf": round(dt / audio_s, 7),
            "x_realtime_per_gpu": round(audio_s / dt / world, 1),
            "whole_path": {
                # (a) HBM: layer-boundary bytes of the UNFUSED layer stack (117.6 MB per audio-second for EnCodec; the
                #     kernels' own algorithmic counts for the other codecs) per second against 8 TB/s -- fused chains move fewer bytes
                #     than this model, so the fraction measures time, not traffic;
                # (b) matrix pipe: the 16-bit MFMA flops the split-operand kernels EXECUTE (3 partial products per fp32 product in
                #     split16 arithmetic, 6 with three bf16 planes, 1 in the opt-in bf16 mode) per second against the 2.5 PF dense peak.
                "hbm_layer_boundary_frac": round((sum(s[4] for s in stats) if mimi else LAYER_BYTES_PER_AUDIO_S * audio_s / world) / dt / (PEAK_HBM_GBS * 1e9), 4),
                "executed_mfma16_tflops": round(sum(s[3] * mfma16_terms(s[0], mode_) for s in stats) / dt / 1e12, 1),
                "executed_mfma16_frac": round(sum(s[3] * mfma16_terms(s[0], mode_) for s in stats) / dt / (PEAK_F16_MFMA_TFLOPS * 1e12), 4),
                "ms_per_step_without_kernel_events": round(dt_plain / args.steps * 1e3, 3),
            },
            "roofline": roof,
            "kernels": [
                {"name": s[0], "launches_per_step": s[1] / args.steps, "ms_per_step": round(s[2] / args.steps, 3),
                 "tflops": round(s[3] / (s[2] * 1e-3) / 1e12, 2), "gbs": round(s[4] / (s[2] * 1e-3) / 1e9, 1)}
                for s in stats
            ],
        }
        extras = not args.no_parity
        with torch.no_grad():
            if extras:
                try:   # shader clock under the tap-GEMMs (power cap): 2 collective-free steps outside the timed region (rank 0 only)
                    import ctypes as _C
                    nat = next(iter(codec._natives.values()))
                    mhz = _C.c_double(0.0)
                    nat.lib.ac_debug_clock(nat.h, 1, _C.byref(mhz))
                    local_step(); local_step()
                    nat.lib.ac_debug_clock(nat.h, 0, _C.byref(mhz))
                    out["roofline"]["tap_gemm6_shader_clock_mhz"] = round(mhz.value, 0)
                except Exception:  # diagnostics only
                    out["roofline"]["tap_gemm6_shader_clock_mhz"] = None
            out["parity"] = parity_gate(args.codec, codec) if extras else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.codec, cfg, sd, sig_cpu, ncb)
        if world == 1 and extras and mode_ == "fp32" and not args.no_exact:
            # the split16 figure always travels with its exact-product twin: the same step with every product an IEEE fp32 product
            # (precision="fp32_exact": tap_gemm4 / rb_fused / lstm_persist kernels), 1 warm-up + 3 steps outside the timed region
            try:
                drop_codec(codec)
                ex, _, _ = build_codec(args.codec, "fp32_exact")
                with torch.no_grad():
                    ex.toks_to_sig(ex.sig_to_toks(sig))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        ex.toks_to_sig(ex.sig_to_toks(sig))
                    torch.cuda.synchronize()
                    out["exact_fp32_ms_per_step"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
                    out["exact_fp32_parity"] = parity_gate(args.codec, ex)
                drop_codec(ex)
            except Exception as e:
                out["exact_fp32_ms_per_step"] = None
                out["exact_fp32_error"] = repr(e)[:200]
        if world == 1 and extras and args.codec == "encodec" and not args.no_other_configs:
            # BASELINE.json configs 3-5 at their per-GPU sizes: short runs of the same step, driver-visible in this line
            out["other_configs"] = {}
            for nm, bt, st_ in (("dac", 256, 3), ("mimi", 128, 3), ("wavtokenizer", 64, 3)):
                try:
                    out["other_configs"][nm] = short_run(nm, bt, 10.0, st_, 1, args.precision)
                except Exception as e:
                    out["other_configs"][nm] = {"error": repr(e)[:300]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

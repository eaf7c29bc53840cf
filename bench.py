#!/usr/bin/env python3
"""Benchmark of the hot path: EnCodec-24k, 8 codebooks, encode + decode of 64 x 10 s per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
     --master-port P bench.py --gpus N ... -- or bare: `python bench.py --gpus N` then starts that launcher itself as a child
     process before touching the GPU and forwards its output and exit code)

A step = Codec.sig_to_toks + Codec.toks_to_sig over one synthetic batch that is already resident
in HBM (BASELINE.json configs[1]; weights: seeded synthetic checkpoint -- no pretrained weights
exist offline).  Clips are independent units: every rank encodes/decodes its own 64 clips
(weak scaling); with N > 1 the only collective is the RCCL all_gather of the token ids, and it is
inside the timed step.  Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline     -- dominant kernel (by HIP-event time measured in the timed steps) against the pipe that bounds it:
                  the dense fp16 MFMA peak (2.5 PF) for the split-operand tap-GEMM (3 fp16 partial products per fp32
                  product, split16.h; --precision fp32_exact: the exact-product kernels, bounded by the 157.3 TF fp32 MFMA
                  peak) or HBM (8 TB/s)
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference) timed on this host's
                  cores on a bounded sample of the same workload.  Baseline only.
and, at N = 1, outside the timed region (all skipped by --no-parity):
  parity                 -- token exact-match and decode RMS error on the reference-generated fixture
  exact_fp32_ms_per_step -- the same step with every product an IEEE fp32 product (precision="fp32_exact"), so the
                            split16 figure never travels without its exact-product twin
  other_configs          -- BASELINE.json configs 3-5 at their per-GPU sizes (DAC 256 x 10 s, Mimi 128 x 10 s,
                            WavTokenizer 64 x 10 s): value, ms_per_step, roofline (dominant kernel family) and parity gate of a
                            short run each (DAC 1 + 3 steps of 2.1 s, Mimi 2 + 8, WavTokenizer 2 + 10)
  latency                -- the reference's own measurement regime (batch 1, its profiler's clip lengths): encode + decode of
                            B = 1 / 8 clips of 1 / 10 / 32 s, ms per call, RTF, top-3 kernels (other_configs.{mimi,dac}.batch1_latency: 1 x 1 s eager
                            and through the wrappers' opt-in hipGraph replay, graph=True; the LSTM codecs decline it, audiocodecs_amd/codec.py)
Nothing inside an `if rank == 0` block issues a collective (tests/test_bench_contract.py checks the source for it), and the
flow from the warm-ups to the JSON line is `run()`, which tests/test_bench_flow_gloo.py executes at world size 2 over gloo.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
SPLIT16_TERMS = 3               # split16.h, two fp16 planes (the default arithmetic): partial products per fp32 product
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: dense fp16 / bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
# bare v_mfma_f32_32x32x16_f16 stream on random operands (tools/ubench/mfma_f16_probe.hip): 1.58-1.68 PF at 1.54-1.69 GHz
SUSTAINED_F16_MFMA_TFLOPS = 1680.0
PEAK_HBM_GBS = 8000.0          # same guide, "HBM3E peak BW" (spec; 6.29 TB/s achievable)
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
    never on the product path) timed on this host's cores, BASELINE.md section 3 protocol: B = min(config B, 8) clips of the
    same batch per call, fp32, torch.inference_mode(), 2 warm-ups + 3 timed runs of encode+decode, median reported.
    Threads: section 3 says all physical cores of the GPU host -- that run is made and reported (`all_physical_cores`), but
    torch's CPU convolutions do not scale to every core of a 2-socket host on a few clips, so a short scan (1 clip x 2 s per
    candidate) also picks the fastest thread count and `value` / `cores` are that faster configuration's.  Both are printed.
    The sample is bounded (DAC: 1 clip, ~100 GMAC per audio-second) so that the default bench.py run finishes in minutes."""
    if codec_name == "mimi":
        from oracle import mimi_oracle as O
        W = O.cast_weights(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
        clips = 8
    elif codec_name == "dac":
        from oracle import dac_oracle as O
        W = O.cast_weights(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x, None, ncb))
        clips = 1   # ~100 GMAC per audio-second
    elif codec_name == "wavtokenizer":
        from oracle import wavtokenizer_oracle as O
        W = O.cast_weights(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
        clips = 8
    else:
        from oracle import encodec_oracle as O
        W = O.fold_weight_norm(sd)
        run = lambda x: O.toks_to_sig(cfg, W, O.sig_to_toks(cfg, W, x))
        clips = 8
    sr = cfg.sampling_rate
    x = sig_cpu[:clips]
    model, phys, threads = host_cpu()
    cands = sorted({t for t in (8, 16, 32, 64, phys) if 1 <= t <= threads}) or [threads]
    scan = {}

    def timed(n_threads, warm, reps):
        torch.set_num_threads(n_threads)
        for _ in range(warm):
            run(x[:1])
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            run(x)
            ts.append(time.perf_counter() - t0)
        return ts

    with torch.inference_mode():
        probe = x[:1, : 2 * sr]
        for t in cands:
            torch.set_num_threads(t)
            run(probe)
            t0 = time.perf_counter()
            run(probe)
            scan[t] = time.perf_counter() - t0
        best_t = min(scan, key=scan.get)
        times = timed(best_t, 2, 3)
        allc = None
        if phys != best_t and phys <= threads:
            ta = timed(phys, 1, 1)
            allc = {"cores": phys, "value": round(x.shape[0] * x.shape[1] / sr / ta[0], 2), "runs_s": [round(ta[0], 3)]}
    audio_s = x.shape[0] * x.shape[1] / sr
    med = sorted(times)[1]
    return {
        "value": round(audio_s / med, 2), "unit": "audio-s/s", "cores": best_t, "kind": "port",
        "host_physical_cores": phys, "host_hw_threads": threads, "cpu_model": model,
        "runs_s": [round(t, 3) for t in times], "all_physical_cores": allc,
        "sample": f"{x.shape[0]} clips x {x.shape[1] / sr:.0f} s of the same batch, fp32 torch-CPU oracle, encode+decode; "
                  f"2 warm-ups + 3 timed runs (median); thread count {best_t} chosen from {cands} by a 2 s probe; "
                  f"all {phys} physical cores (BASELINE.md section 3): one run, reported beside it",
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


_SHAPE_SUFFIX = re.compile(r"\s+B\d+ M\d+ N\d+ K\d+.*$")
SPLIT_FAMILIES = ("tap_gemm6_kernel", "tap_gemm8_kernel", "rb_fused6_kernel", "rb128_fused6_kernel", "thin_conv6_kernel", "lstm_persist16_kernel",
                  "enc_front_kernel", "dec_tail_kernel", "rb_stream6_kernel", "rb_stream6m_kernel", "rb_stream128m_kernel", "enc_stream_kernel", "dec_stream_kernel", "rvq_encode16_kernel", "attention16_kernel", "dac_unit6_kernel", "rb_fused6_head_kernel")


def parse_kernel(name):
    """(family, [template arguments], shape suffix) of a kernel name as ac_profile_end or rocprofv3 prints it: `void`, the `ac::`
    namespace, the argument list and the AC_PROF_DETAIL shape suffix (" B64 M30000 N256 K256 J2 s1") are stripped."""
    n = re.sub(r"^void\s+", "", name.strip()).replace("ac::", "")
    n = re.sub(r"\(.*\)\s*$", "", n)
    m = _SHAPE_SUFFIX.search(n)
    suffix = m.group(0).strip() if m else ""
    if m:
        n = n[: m.start()]
    m = re.match(r"^([A-Za-z_0-9]+)\s*<(.*)>\s*$", n)
    if not m:
        return n.strip(), [], suffix
    return m.group(1), [a.strip() for a in m.group(2).split(",")], suffix


def canonical_kernel(name):
    """One spelling per instantiation: the tap-GEMM's slab-halo template argument (rocprofv3 prints `..., 2, 7>` / `..., 2, 56>`,
    the library's own records `..., 2>` / `..., 2, dil>`) becomes nothing / `dil`; everything else is family<arguments>."""
    fam, args, _ = parse_kernel(name)
    if fam == "tap_gemm6_kernel" and len(args) >= 6:
        args = args[:5] + ([] if args[5] == "7" else ["dil"])
    if fam == "tap_gemm8_kernel":      # rocprofv3: <WGM, WGN, WMT, WN, row mode, one tap>; the library's records: <WGM, WGN, WMT, WN, 2>
        args = args[:4]
    return f"{fam}<{', '.join(args)}>" if args else fam


def kernel_planes(name):
    """Operand planes of a split-operand kernel (template argument NP of the kernels that have one: the FIFTH of the tap-GEMM,
    the last of the fused blocks): 2 = split16.h (two fp16 planes, 3 partial products per fp32 product); None = a kernel that does
    not run on the 16-bit matrix pipe."""
    fam, args, _ = parse_kernel(name)
    if fam not in SPLIT_FAMILIES:
        return None
    if fam == "tap_gemm8_kernel":
        return 2        # split16 only (its template arguments after the tile form are flags, not a plane count)
    if fam == "tap_gemm6_kernel":
        return int(args[4]) if len(args) >= 5 and args[4].isdigit() else 2
    if fam in ("rb_fused6_kernel", "rb128_fused6_kernel", "thin_conv6_kernel"):
        return int(args[-1]) if args and args[-1].isdigit() else 2
    return 2        # the kernels that exist in split16 arithmetic only


def mfma16_terms(kernel_name, mode=None):
    """16-bit MFMA partial products a kernel executes per fp32 product it stands for (0: not on the 16-bit matrix pipe)."""
    np_ = kernel_planes(kernel_name)
    return 0 if np_ is None else {2: SPLIT16_TERMS}.get(np_, SPLIT16_TERMS)


def measured_traffic(kernel_name, unit, codec="encodec", batch=64):
    """HBM bytes per launch of `kernel_name` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 FETCH x2 correction: tools/collect_traffic.py).  bench.py cannot
    run the profiler on itself, so the newest profiles/r*_traffic.json that knows the kernel is quoted (names compared in
    their canonical spelling); null if absent."""
    import glob

    def rank_of(f):   # r5ae_... after r5z_... after r4b_... after r4a_... (run tags count a, b, .. z, aa, ab, ..)
        m = re.match(r"r(\d+)([a-z]*)", os.path.basename(f))
        return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (0, 0, "")

    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))
                    if (codec in os.path.basename(f)) or (codec == "encodec" and not any(c in os.path.basename(f) for c in ("mimi", "dac", "wavtok")))), key=rank_of)

    def batch_ok(f):   # r1_mimi_b32_traffic.json was collected at 32 clips per GPU; no tag = the bench default of that codec
        m = re.search(r"_b(\d+)_", os.path.basename(f))
        if m and codec == "dac" and int(m.group(1)) == 39 and batch >= 39:
            return True   # DAC walks its batch in chunks of 39 clips (dac_path.hip): a launch of a 256-clip step IS a 39-clip launch
        return (int(m.group(1)) == batch) if m else True

    want = canonical_kernel(kernel_name)
    for f in reversed([f for f in files if batch_ok(f)]):   # newest summary that knows this kernel
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        # (several instantiations can share a canonical name -- tap_gemm8's row-mode / one-tap / request-placement flags: launch-weighted mean)
        hits = [(k.get("launches", 1), k["hbm_bytes_per_launch"]) for nm, k in ks.items()
                if canonical_kernel(nm) == want and k.get("hbm_bytes_per_launch") is not None]
        if hits:
            return round(sum(n_ * b_ for n_, b_ in hits) / sum(n_ for n_, _ in hits)), "profiles/" + os.path.basename(f)
    return None


def measured_step_traffic(codec="encodec", batch=64):
    """HBM bytes ONE STEP moved, all kernels, from the newest committed PMC summary of that codec (same source and batch rule as
    measured_traffic): (fetch bytes, write bytes, file) or None.  DAC: the summary is of one 39-clip chunk -- scaled to the batch."""
    import glob

    def rank_of(f):
        m = re.match(r"r(\d+)([a-z]*)", os.path.basename(f))
        return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (0, 0, "")

    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))
                    if (codec in os.path.basename(f)) or (codec == "encodec" and not any(c in os.path.basename(f) for c in ("mimi", "dac", "wavtok")))), key=rank_of)
    default_batch = {"encodec": 64, "mimi": 128, "wavtokenizer": 64, "dac": 256}.get(codec)
    for f in reversed(files):
        m = re.search(r"_b(\d+)_", os.path.basename(f))
        scale = 1.0
        if not m and batch != default_batch:
            continue            # (an untagged summary is of the codec's default bench batch)
        if m:
            if codec == "dac" and int(m.group(1)) == 39 and batch >= 39:
                scale = batch / 39.0
            elif int(m.group(1)) != batch:
                continue
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        fe = sum(k.get("launches", 1) * k["fetch_bytes_per_launch_corrected"] for k in ks.values() if k.get("hbm_bytes_per_launch") is not None)
        wr = sum(k.get("launches", 1) * k["write_bytes_per_launch"] for k in ks.values() if k.get("hbm_bytes_per_launch") is not None)
        if fe + wr > 0:
            return fe * scale, wr * scale, "profiles/" + os.path.basename(f)
    return None


CODEC_LABEL = {"mimi": "Mimi-24k", "dac": "DAC-44.1k", "encodec": "EnCodec-24k", "wavtokenizer": "WavTokenizer-24k-40tok"}
CODEC_NCB = {"dac": 9, "wavtokenizer": 1, "mimi": 8, "encodec": 8}


def build_codec(name, precision=None, graph=False):
    """(codec, cfg, state dict) of one of the four wrappers on seeded synthetic weights of the full architecture."""
    from audiocodecs_amd import DAC, Encodec, Mimi, WavTokenizer, checkpoint
    from audiocodecs_amd.config import DAC_44KHZ, ENCODEC_24KHZ, MIMI_24KHZ, WAVTOK_40

    cfg = {"mimi": MIMI_24KHZ, "dac": DAC_44KHZ, "encodec": ENCODEC_24KHZ, "wavtokenizer": WAVTOK_40}[name]
    if name == "mimi":
        sd = checkpoint.synthetic_mimi_state_dict(cfg, seed=0)
        codec = Mimi(cfg.sampling_rate, num_codebooks=8, state_dict=sd, precision=precision, graph=graph).eval()
    elif name == "dac":
        sd = checkpoint.synthetic_dac_state_dict(cfg, seed=0)
        codec = DAC(cfg.sampling_rate, cfg.sampling_rate, num_codebooks=9, state_dict=sd, config=cfg, precision=precision, graph=graph).eval()
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


def roofline_of(stats, dt, steps, codec_name, batch, exact):
    """The `roofline` object of the contract for one timed region.  stats = [(kernel name, launches, total ms, flops, bytes)] from
    the HIP events the library brackets every launch with on the stream it launches on (ac_profile_begin / _end), dt = wall
    seconds of the region.  The dominant kernel is the family with the largest event time: tap_gemm6_kernel is ONE kernel
    template launched in several tile arrangements (rocprofv3 lists each instantiation on its own row) -- the arrangements are
    ranked together and listed one by one under `arrangements` with their own average launch time.
      split-operand kernels (split16.h): bound = the dense fp16 MFMA peak; `achieved` counts the fp16 MFMA flops EXECUTED
                                         (3 partial products per fp32 product);
      exact-product kernels            : the fp32 MFMA peak when the arithmetic intensity is above the ridge, else HBM;
      everything else                  : HBM, algorithmic bytes (inputs once, outputs once per flavour, weights once) / time."""
    stats = sorted(stats, key=lambda s_: -s_[2])
    fam = {}
    for s_ in stats:
        f_ = parse_kernel(s_[0])[0] if parse_kernel(s_[0])[0] in ("tap_gemm6_kernel", "tap_gemm8_kernel") else s_[0]
        f_ = "tap_gemm" if f_ in ("tap_gemm6_kernel", "tap_gemm8_kernel") else f_
        a_ = fam.setdefault(f_, [f_, 0, 0.0, 0.0, 0.0, []])
        a_[1] += s_[1]; a_[2] += s_[2]; a_[3] += s_[3]; a_[4] += s_[4]; a_[5].append(s_)
    name, launches, tot_ms, flops, nbytes, members = max(fam.values(), key=lambda a_: a_[2])
    if len(members) == 1:
        name = members[0][0]
    ai = flops / max(nbytes, 1.0)
    terms = {mfma16_terms(m_[0]) for m_ in members}
    if terms == {SPLIT16_TERMS} and not exact:
        eq = flops / (tot_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(SPLIT16_TERMS * eq, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "pipe": f"fp16 MFMA, {SPLIT16_TERMS} partial products per fp32 product (two fp16 planes per operand, split16.h)",
                "fp32_equivalent_tflops": round(eq, 2), "fp32_equivalent_peak": round(PEAK_F16_MFMA_TFLOPS / SPLIT16_TERMS, 1),
                "sustained_bare_mfma_stream": SUSTAINED_F16_MFMA_TFLOPS}
    elif ai > PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
        roof = {"bound": "mfma", "achieved": round(flops / (tot_ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": round(nbytes / (tot_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    if len(members) > 1:
        trs = [(m_[1], measured_traffic(m_[0], roof["unit"], codec_name, batch)) for m_ in members]
        tr = None
        if all(t_ for _, t_ in trs):
            tr = (round(sum(n_ * t_[0] for n_, t_ in trs) / sum(n_ for n_, _ in trs)), sorted({t_[1] for _, t_ in trs})[-1])
        roof["arrangements"] = [{"kernel": m_[0], "launches_per_step": m_[1] / steps, "avg_launch_us": round(m_[2] / m_[1] * 1e3, 2),
                                 "fp32_equivalent_tflops": round(m_[3] / (m_[2] * 1e-3) / 1e12, 2)} for m_ in members]
    else:
        tr = measured_traffic(name, roof["unit"], codec_name, batch)
    if name.startswith("lstm_persist"):
        roof["note"] = "sequential recurrence: latency-bound (one exchange of h per time step), not a throughput kernel"
    # PMC counters cannot be collected from inside this process: the figure is the one of the committed rocprofv3 --pmc passes
    # over this same command (a different run / box); null when no summary knows the kernel
    roof["traffic"] = tr[0] if tr else None
    roof["traffic_source"] = tr[1] if tr else None
    roof["algorithmic_bytes_per_launch"] = round(nbytes / launches)
    roof["hbm_gbs"] = round(nbytes / (tot_ms * 1e-3) / 1e9, 1)
    roof["hbm_frac"] = round(nbytes / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    roof["kernel"] = name if len(members) == 1 else f"{name}<...> ({len(members)} tile arrangements of one kernel family)"
    roof["launches_per_step"] = launches / steps
    roof["avg_launch_us"] = round(tot_ms / launches * 1e3, 2)
    roof["share_of_step"] = round(tot_ms / (dt * 1e3), 4)
    return roof


def whole_path(stats, dt, layer_bytes, codec=None, batch=None, steps=1):
    """Whole-step fractions: (a) layer-boundary bytes of the UNFUSED layer stack per second against 8 TB/s (fused chains move fewer
    bytes than this model, so the fraction measures time, not traffic); (b) the fp16 MFMA flops the split16 kernels EXECUTE per
    second against the 2.5 PF dense peak; (c) the HBM bytes a step really moved (FETCH_SIZE / WRITE_SIZE counter passes of the same
    workload, all kernels: measured_step_traffic) over this run's step time."""
    ex = sum(s_[3] * mfma16_terms(s_[0]) for s_ in stats)
    out = {"hbm_layer_boundary_frac": round(layer_bytes / dt / (PEAK_HBM_GBS * 1e9), 4),
           "executed_mfma16_tflops": round(ex / dt / 1e12, 1), "executed_mfma16_frac": round(ex / dt / (PEAK_F16_MFMA_TFLOPS * 1e12), 4)}
    tr = measured_step_traffic(codec, batch) if codec else None
    if tr:
        per_step_s = dt / steps
        out.update({"measured_hbm_gb_per_step": round((tr[0] + tr[1]) / 1e9, 2), "measured_hbm_fetch_gb_per_step": round(tr[0] / 1e9, 2),
                    "measured_hbm_gbs": round((tr[0] + tr[1]) / per_step_s / 1e9, 1), "measured_hbm_frac": round((tr[0] + tr[1]) / per_step_s / (PEAK_HBM_GBS * 1e9), 4),
                    "measured_hbm_source": tr[2]})
    return out


def kernel_rows(stats, steps, top=None):
    rows = [{"name": s_[0], "launches_per_step": s_[1] / steps, "ms_per_step": round(s_[2] / steps, 3),
             "tflops": round(s_[3] / (s_[2] * 1e-3) / 1e12, 2), "gbs": round(s_[4] / (s_[2] * 1e-3) / 1e9, 1)}
            for s_ in sorted(stats, key=lambda s_: -s_[2])]
    return rows[:top] if top else rows


def short_run(name, batch, seconds, steps, warmup, precision=None):
    """One of the other BASELINE.json configs as a short driver-timed run of the same step (encode + decode of `batch` clips
    resident in HBM), with its parity gate and its own roofline (dominant kernel family of a separately event-timed pass): value,
    ms_per_step, the gate and the roofline travel in the headline JSON line (`other_configs`)."""
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
        t1 = time.perf_counter()
        stats = codec.profile_kernels(lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
        torch.cuda.synchronize()
        dt_prof = time.perf_counter() - t1
        gate = parity_gate(name, codec)
        mhz = None
        try:   # shader clock under the tap-GEMMs (the box-to-box spread of this pool is ~5 % of clock: every figure travels with its own)
            import ctypes as _C
            nat = next(iter(codec._natives.values()))
            m_ = _C.c_double(0.0)
            nat.lib.ac_debug_clock(nat.h, 1, _C.byref(m_))
            codec.toks_to_sig(codec.sig_to_toks(sig))
            torch.cuda.synchronize()
            nat.lib.ac_debug_clock(nat.h, 0, _C.byref(m_))
            mhz = round(m_.value, 0)
        except Exception:  # diagnostics only
            pass
        b1 = None
        if name in ("mimi", "dac"):     # batch 1 x 1 s, eager against the wrapper's hipGraph replay (median of 5 after 3 calls: the first captures)
            b1 = {}
            one = sig[:1, : cfg.sampling_rate].contiguous()
            for label, c_ in (("eager_ms", codec), ("graph_ms", build_codec(name, precision, graph=True)[0])):
                for _ in range(3):
                    c_.toks_to_sig(c_.sig_to_toks(one))
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    t2 = time.perf_counter()
                    c_.toks_to_sig(c_.sig_to_toks(one))
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t2)
                b1[label] = round(sorted(ts)[2] * 1e3, 3)
                if c_ is not codec:
                    drop_codec(c_)
    del sig
    drop_codec(codec)
    audio_s = batch * T / cfg.sampling_rate * steps
    return {"workload": f"{CODEC_LABEL[name]} {CODEC_NCB[name]} codebooks, encode+decode, {batch} clips x {seconds:g} s on 1 GPU, resident in HBM",
            "value": round(audio_s / dt, 1), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
            "tap_gemm_shader_clock_mhz": mhz,
            "roofline": roofline_of(stats, dt_prof, 1, name, batch, precision == "fp32_exact"),
            "whole_path": whole_path(stats, dt_prof, sum(s_[4] for s_ in stats), name, batch),
            "top_kernels": kernel_rows(stats, 1, top=4), "parity": gate, "batch1_latency": b1}


def latency_regime(codec, cfg):
    """The regime the reference itself measures (downstream/hparams/tasks/sr.yaml:28 `test_batch_size: 1`; downstream/test_sr.py:379-391
    profiles shapes (1, sr * {1, 2, 4, 8, 16, 32})): encode + decode of B = 1 and B = 8 clips of 1 / 10 / 32 s, resident in HBM --
    ms per call (median of 5 after 2 warm-ups, host-timed around a device sync as the reference does), RTF, and the three
    kernels that take the most time (from a separate event-timed call)."""
    from audiocodecs_amd import prng

    out = []
    sr = cfg.sampling_rate
    with torch.no_grad():
        for B in (1, 8):
            for sec in (1, 10, 32):
                sig = torch.from_numpy((prng.normal(321, f"bench.latency.{B}.{sec}", (B, sec * sr)) * 0.1).astype(np.float32)).cuda()
                for _ in range(2):
                    codec.toks_to_sig(codec.sig_to_toks(sig))
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    toks = codec.sig_to_toks(sig)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    codec.toks_to_sig(toks)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0, t1 - t0))
                ts.sort()
                tot, enc = ts[len(ts) // 2]
                stats = codec.profile_kernels(lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
                torch.cuda.synchronize()
                ksum = sum(s_[2] for s_ in stats)
                out.append({"clips": B, "seconds": sec, "ms_per_call": round(tot * 1e3, 3), "encode_ms": round(enc * 1e3, 3),
                            "decode_ms": round((tot - enc) * 1e3, 3), "rtf": round(tot / (B * sec), 6),
                            "kernel_event_ms": round(ksum, 3), "launches": int(sum(s_[1] for s_ in stats)),
                            "top3": [{"name": s_[0], "ms": round(s_[2], 3)} for s_ in sorted(stats, key=lambda s_: -s_[2])[:3]]})
                del sig
    return out


def timed_steps(step, fence, steps, warmup, profile):
    """The contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by fence() (barrier + device
    synchronise) on both sides -- once with the library's per-kernel events armed (`profile(fn)` returns their statistics: the
    roofline figures) and once more without them: the run `value` comes from.  Returns (dt with events, dt plain, statistics).
    Every rank runs this; nothing here is rank-conditional."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = profile(lambda: [step() for _ in range(steps)])
    fence()
    dt = time.perf_counter() - t0
    fence()
    t1 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt_plain = time.perf_counter() - t1
    return dt, dt_plain, stats


def max_over_ranks(dist, values, device):
    """MAX over the ranks of a list of floats (every rank calls this; identity without a process group)."""
    if dist is None:
        return list(values)
    tt = torch.tensor(list(values), device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return tt.tolist()


def contract_line(label, ncb, world, steps, warmup, dt, audio_s, B, seconds, dtype, extra):
    """The ONE JSON object rank 0 prints (keys of the task statement's contract first)."""
    out = {
        "metric": f"encode+decode audio-sec/s, {label} {ncb}cb",
        "value": round(audio_s / dt, 1),
        "unit": "audio-s/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": f"synthetic (0.1*N(0,1) clips, seeded synthetic weights of the {label} architecture)",
        "config": {"workload": f"{label} {ncb} codebooks, encode+decode, {B} clips x {seconds:g} s per GPU, resident in HBM",
                   "clips_per_gpu": B, "seconds_per_clip": seconds, "parallelism": f"clip-sharded x{world}"},
        "rtf": round(dt / audio_s, 7),
        "x_realtime_per_gpu": round(audio_s / dt / world, 1),
    }
    out.update(extra)
    return out


def run(args, codec, cfg, sd, sig, sig_cpu, rank, world, dist, device):
    """Everything after set-up, for any codec object with sig_to_toks / toks_to_sig / profile_kernels and any process group (RCCL on
    the GPU; tests/test_bench_flow_gloo.py drives it at world size 2 over gloo with a stub codec on CPU tensors).
    Collectives: the token all_gather inside step(), the barrier inside fence(), the MAX-reduce of the timings, the closing
    barrier -- all outside every `if rank == 0` block."""
    from audiocodecs_amd.sharding import gather_tokens

    label, ncb = CODEC_LABEL[args.codec], CODEC_NCB[args.codec]
    B, T = sig.shape[0], sig.shape[1]
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)

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
        sync()

    with torch.no_grad():
        dt_ev, dt, stats = timed_steps(step, fence, args.steps, args.warmup, codec.profile_kernels)
    # `value` / `ms_per_step` = the PLAIN pass (K steps between the fences, nothing else in the region); dt_ev = the same K steps with the
    # library's per-kernel HIP events armed (the `roofline` / `kernels` figures come from those events; 1 - 2 % slower)
    dt_ev, dt = max_over_ranks(dist, [dt_ev, dt], device)

    audio_s = world * B * T / cfg.sampling_rate * args.steps
    exact = (args.precision or ("fp32_exact" if os.environ.get("AC_GEMM", "") == "fp32" else "fp32")) == "fp32_exact"
    if rank == 0:
        other = args.codec != "encodec"   # not the headline codec: layer-boundary bytes from the kernels' own algorithmic counts
        wp = whole_path(stats, dt_ev, sum(s_[4] for s_ in stats) if other else LAYER_BYTES_PER_AUDIO_S * audio_s / world, args.codec, B, args.steps)
        wp["ms_per_step_with_kernel_events"] = round(dt_ev / args.steps * 1e3, 3)
        roof = roofline_of(stats, dt_ev, args.steps, args.codec, B, exact)
        dtype = ("f32 (every product an IEEE fp32 product: the exact-product kernels)" if exact else
                 "f32 (GEMM-shaped kernels and LSTM products: operands as two scaled fp16 planes, 3 partial products, fp32 accumulate -- split16.h)")
        out = contract_line(label, ncb, world, args.steps, args.warmup, dt, audio_s, B, args.seconds, dtype,
                            {"whole_path": wp, "roofline": roof, "kernels": kernel_rows(stats, args.steps)})
        extras = not args.no_parity and device.type == "cuda"
        with torch.no_grad():
            if extras and args.codec == "encodec":
                # a speech-like input beside the noise batch (amplitude-modulated multi-sines of tests/golden_cases.py
                # `tones`: ELU / argmax timing is data-independent, this guards against degenerate-token fast paths).  Same step, 5 steps.
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    from golden_cases import tones
                    sp = tones(4321, B, T).to(device)
                    for _ in range(2):
                        codec.toks_to_sig(codec.sig_to_toks(sp))
                    sync()
                    t_sp = time.perf_counter()
                    for _ in range(5):
                        codec.toks_to_sig(codec.sig_to_toks(sp))
                    sync()
                    out["speech_like_ms_per_step"] = round((time.perf_counter() - t_sp) / 5 * 1e3, 3)
                    out["speech_like_input"] = "amplitude-modulated five-tone mixtures (tests/golden_cases.py tones, seed 4321), same clips x seconds"
                    del sp
                except Exception as e:  # diagnostics only
                    out["speech_like_ms_per_step"] = None
                    out["speech_like_error"] = repr(e)[:200]
            if extras:
                try:   # shader clock under the tap-GEMMs (power cap): 2 collective-free steps outside the timed region (rank 0 only)
                    import ctypes as _C
                    nat = next(iter(codec._natives.values()))
                    mhz = _C.c_double(0.0)
                    nat.lib.ac_debug_clock(nat.h, 1, _C.byref(mhz))
                    local_step(); local_step()
                    nat.lib.ac_debug_clock(nat.h, 0, _C.byref(mhz))
                    out["roofline"]["tap_gemm_shader_clock_mhz"] = round(mhz.value, 0)
                except Exception:  # diagnostics only
                    out["roofline"]["tap_gemm_shader_clock_mhz"] = None
            out["parity"] = parity_gate(args.codec, codec) if extras else None
            if extras and args.codec == "encodec" and world == 1 and not args.no_latency:
                try:
                    out["latency"] = latency_regime(codec, cfg)
                except Exception as e:
                    out["latency"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline and sd is not None:
            out["cpu_baseline"] = cpu_baseline(args.codec, cfg, sd, sig_cpu, ncb)
        if world == 1 and extras and not exact and not args.no_exact:
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
            for nm, bt, st_, wu_ in (("dac", 256, 5, 1), ("mimi", 128, 8, 2), ("wavtokenizer", 64, 10, 2)):
                try:
                    out["other_configs"][nm] = short_run(nm, bt, 10.0, st_, wu_, args.precision)
                except Exception as e:
                    out["other_configs"][nm] = {"error": repr(e)[:300]}
        print(json.dumps(ordered_line(out)), flush=True)
    if dist is not None:
        dist.barrier()
    return 0


def ordered_line(out):
    """The JSON line with what a reader of its TAIL needs last: the long arrays (`kernels`, `latency`, the full `other_configs`) go to the
    FRONT of the line behind the contract's fields, and a compact `summary` -- every config's value, ms per step, shader clock, roofline
    fraction and traffic ratio, the headline's both step figures -- closes it (round-5 verdict item 7: DAC's value fell off the front
    of the driver's tail)."""
    def brief(d):
        if not isinstance(d, dict) or "error" in d:
            return d
        r = d.get("roofline") or {}
        t, a = r.get("traffic"), r.get("algorithmic_bytes_per_launch")
        return {"value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"),
                "tap_gemm_shader_clock_mhz": d.get("tap_gemm_shader_clock_mhz", r.get("tap_gemm_shader_clock_mhz")),
                "roofline_frac": r.get("frac"), "roofline_kernel": r.get("kernel"),
                "traffic_over_algorithmic": round(t / a, 3) if t and a else None,
                "parity_ok": None if d.get("parity") is None else bool(d["parity"].get("token_mismatches_outside_fp64_near_ties", 1) == 0 and d["parity"].get("decode_rms_err", 1.0) <= d["parity"].get("decode_rms_bar", 1e-4))}
    heavy = ("kernels", "latency", "other_configs")
    line = {k: v for k, v in out.items() if k not in heavy}
    summary = {"headline": brief(out)}
    summary["headline"]["ms_per_step_plain_pass"] = out.get("ms_per_step")
    summary["headline"]["ms_per_step_with_kernel_events"] = (out.get("whole_path") or {}).get("ms_per_step_with_kernel_events")
    for nm, d in (out.get("other_configs") or {}).items():
        summary[nm] = brief(d)
    ordered = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        if k in line:
            ordered[k] = line.pop(k)
    for k in heavy:
        if k in out:
            ordered[k] = out[k]
    ordered.update(line)
    ordered["summary"] = summary
    return ordered


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--codec", choices=["encodec", "mimi", "dac", "wavtokenizer"], default="encodec",
                    help="encodec = BASELINE.json configs[1] (the contract's default); mimi = configs[3] shape (SURVEY.md §8 f3); "
                         "dac = configs[2] (DAC 44.1 kHz, 9 codebooks; use --batch 256); wavtokenizer = configs[4] (40 tok/s, 64 clips per GPU)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--precision", choices=["fp32", "fp32_exact"], default=None,
                    help="arithmetic of the GEMM-shaped kernels: default = fp32 fidelity (split16: the parity arithmetic, what `value` is "
                         "quoted for); fp32_exact = every product an IEEE fp32 product")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL on the GPUs (the measurement); gloo = tests only: the same launch + run() flow on CPU with a stub codec")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates and every other untimed extra (profiling passes: keeps the kernel trace to the timed workload)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE.json configs 3-5 (DAC, Mimi, WavTokenizer) that travel in the same JSON line")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32-product twin of the headline figure")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 / B = 8 latency table (the reference's own measurement regime)")
    return ap.parse_args(argv)


def launch_ranks(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process and hand back its exit code; the
    child's stdout (rank 0's one JSON line) and stderr pass straight through.  The parent has not touched the GPU when it gets
    here (main() calls this before any HIP / torch.cuda call) and never does: on this pool a process that has initialised the
    GPU must not exec or be replaced, and a parent holding a context would also sit on rank 0's device."""
    import socket
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool's host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = 1
    for attempt in range(2):
        # A port found free by bind-and-close can be taken before the launcher binds it (concurrent bench / test runs on one host): the
        # rendezvous then fails within seconds, before any rank has printed -- one retry on a fresh port (round-5 advisor finding).
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(r.stderr)
        rc = r.returncode
        in_use = "address already in use" in r.stderr.lower() or "eaddrinuse" in r.stderr.lower()
        if rc == 0 or not (in_use and time.perf_counter() - t0 < 60.0):
            break
    return rc


def main_gloo_stub(args, rank, world):
    """--backend gloo (tests only): the same flow -- launcher, process group, run(), one JSON line -- on CPU tensors over gloo with
    the stub codec of tests/bench_flow_child.py standing in for the HIP library.  Measures nothing; it exists so that the N > 1
    launch path of THIS file can be executed where there is no GPU (tests/test_bench_launcher_gloo.py)."""
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bench_flow_child import StubCodec
    from audiocodecs_amd.config import ENCODEC_24KHZ as cfg

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sig = torch.zeros(args.batch, int(round(args.seconds * cfg.sampling_rate)))
    args.no_cpu_baseline = True
    rc = run(args, StubCodec(rank), cfg, None, sig, sig, rank, world, dist, torch.device("cpu"))
    dist.destroy_process_group()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and os.environ.get("RANK") is None:
        # not under a launcher: spawn the N ranks ourselves (BEFORE anything touches the GPU) and forward the child's exit code
        return launch_ranks(argv, args.gpus)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a launcher")
    if args.backend == "gloo":
        return main_gloo_stub(args, rank, world)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # under torch.distributed.run (RANK in the environment) the collective path runs even at world size 1, so that a 1-GPU
    # box can exercise exactly what the N > 1 launches do (RCCL init, barrier, token all_gather, MAX-reduce of the timings)
    if world > 1 or (os.environ.get("RANK") is not None and os.environ.get("MASTER_PORT") is not None):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from audiocodecs_amd import prng

    codec, cfg, sd = build_codec(args.codec, args.precision)
    B, T = args.batch, int(round(args.seconds * cfg.sampling_rate))
    # sig = 0.1*N(0,1), repo PRNG seed 123; each rank draws its own shard
    sig_cpu = torch.from_numpy((prng.normal(123, f"bench.sig.rank{rank}", (B, T)) * 0.1).astype(np.float32))
    sig = sig_cpu.cuda()
    run(args, codec, cfg, sd, sig, sig_cpu, rank, world, dist, device)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    _rc = main()
    if _rc:            # (success returns from the script normally: under rocprofv3 an explicit sys.exit(0) reaches the HIP runtime's exit handlers
        sys.exit(_rc)  #  before the profiler's finalisation, which then never writes its output -- round-5 evidence run)

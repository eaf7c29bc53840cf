"""Round 5: batch-1 call latency of the four codecs, eager against the wrappers' opt-in hipGraph replay (graph=True).  Median of 7 calls of
encode + decode, host-timed around a device sync like the reference does (downstream/test_sr.py:56-59,82-86)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiocodecs_amd import DAC, Encodec, Mimi, WavTokenizer, checkpoint, prng
from audiocodecs_amd.config import DAC_44KHZ, ENCODEC_24KHZ, MIMI_24KHZ, WAVTOK_40

def build(name, graph):
    if name == "mimi":
        return Mimi(24000, num_codebooks=8, state_dict=SD[name], graph=graph).eval(), 24000
    if name == "dac":
        return DAC(44100, 44100, num_codebooks=9, state_dict=SD[name], config=DAC_44KHZ, graph=graph).eval(), 44100
    if name == "wavtokenizer":
        return WavTokenizer(24000, state_dict=SD[name], arch=WAVTOK_40, graph=graph).eval(), 24000
    return Encodec(24000, num_codebooks=8, state_dict=SD[name], graph=graph).eval(), 24000

SD = {"encodec": checkpoint.synthetic_state_dict(ENCODEC_24KHZ, 0), "mimi": checkpoint.synthetic_mimi_state_dict(MIMI_24KHZ, 0),
      "dac": checkpoint.synthetic_dac_state_dict(DAC_44KHZ, 0), "wavtokenizer": checkpoint.synthetic_wavtok_state_dict(WAVTOK_40, 0)}
for name in ("mimi", "dac", "encodec", "wavtokenizer"):
    row = []
    for graph in (False, True):
        codec, sr = build(name, graph)
        for B, sec in ((1, 1), (1, 10), (8, 1)):
            sig = torch.from_numpy((prng.normal(5, f"lat.{B}.{sec}", (B, sec * sr)) * 0.1).astype(np.float32)).cuda()
            with torch.no_grad():
                for _ in range(3): codec.toks_to_sig(codec.sig_to_toks(sig))
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter(); codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row.append((graph, B, sec, sorted(ts)[3] * 1e3))
        n_graphs = len(codec._graphs)
        del codec; torch.cuda.empty_cache()
    print(name, " ".join(f"[{'graph' if g else 'eager'} {B}x{s}s {ms:.3f} ms]" for g, B, s, ms in row), f"graphs captured: {n_graphs}", flush=True)

"""Experiment: do two half-batches on two HIP streams overlap (one's LSTM launch chain under the other's
conv kernels)?  Compares 64 clips on one stream with 2 x 32 clips on two streams (two handles)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audiocodecs_amd import Encodec, checkpoint, prng
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg

sd = checkpoint.synthetic_state_dict(cfg, 0)
c0 = Encodec(24000, state_dict=sd).eval()
c1 = Encodec(24000, state_dict=sd).eval()
c2 = Encodec(24000, state_dict=sd).eval()
sig = torch.from_numpy((prng.normal(5, "x", (64, 240000)) * 0.1).astype(np.float32)).cuda()
a, b = sig[:32].contiguous(), sig[32:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def full():
    t = c0.sig_to_toks(sig); return c0.toks_to_sig(t)

def split(offset_decode=False):
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev); ta = c1.sig_to_toks(a); ra = c1.toks_to_sig(ta)
    with torch.cuda.stream(s2):
        s2.wait_event(ev); tb = c2.sig_to_toks(b); rb = c2.toks_to_sig(tb)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    return ra, rb

tb_prev = [c2.sig_to_toks(b)]

def split_staggered():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev); ta = c1.sig_to_toks(a); ra = c1.toks_to_sig(ta)
    with torch.cuda.stream(s2):
        s2.wait_event(ev); rb = c2.toks_to_sig(tb_prev[0]); tb_prev[0] = c2.sig_to_toks(b)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    return ra, rb

for name, fn in (("one stream, 64 clips", full), ("two streams, 2 x 32 clips", split), ("two streams, phases staggered", split_staggered)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step")

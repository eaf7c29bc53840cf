"""Round 6: the persistent LSTM's step timeline (developer library, AC_LSTM_DBG=32: 100 MHz stamps of slice 0 / thread 0 of every role at
steps 100 .. 103; points: 1 operand arrived, 2 partial sums ready, 3 barrier passed, 6 gates done, 7 published, 4 hand-over, 5 projection done).
Usage: AUDIOCODECS_AMD_LIB=audiocodecs_amd/libaudiocodecs_amd_dev.so AC_LSTM_DBG=32 python tools/experiments/r6u_lstm_trace.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
codec, cfg, sd = bench.build_codec("encodec")
B = int(os.environ.get("CLIPS", "64"))
sig = torch.from_numpy((prng.normal(123, "bench.sig.encodec", (B, 240000)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(3):
        codec.sig_to_toks(sig)
torch.cuda.synchronize()

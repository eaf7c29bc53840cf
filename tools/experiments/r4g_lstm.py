"""Round 4: what the persistent LSTM's time step costs by number of clips (live lanes per 16-clip group, number of active XCD roles)
and with layer 1 switched off (lstm_dbg bit 0: layer 0 free-running).  EnCodec encoder, 10 s clips."""
import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd._native import debug_set
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
from golden_cases import noise
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
c = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
c.sig_to_feats(noise(5, 1, 24000).cuda())
ref = {}
for dbg in (0, 1):
    debug_set(c, "lstm_dbg", dbg)
    for B in [int(a) for a in sys.argv[1:]] or (1, 8, 16, 17, 32, 33, 48, 64):
        sig = noise(5, B, 240000).cuda()
        try:
            for _ in range(2): c.sig_to_feats(sig)
            st = c.profile_kernels(lambda: [c.sig_to_feats(sig) for _ in range(5)])
        except Exception as e:
            print("dbg", dbg, "B", B, "raised", str(e)[:100]); continue
        for s in st:
            if "lstm_persist" in s[0]: print(f"lstm_dbg={dbg} B={B:3d} {s[0]:32s} {s[2]/5:.3f} ms per launch -> {s[2]/5/750*1e3:.2f} us/step", flush=True)

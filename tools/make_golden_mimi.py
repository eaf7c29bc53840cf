#!/usr/bin/env python3
"""Generate tests/golden/mimi_golden.npz by running THE REFERENCE WRAPPER (build container only).

    python tools/make_golden_mimi.py

What runs: ``audiocodecs.mimi.Mimi`` imported from /root/reference (shims in tools/reference_shim.py)
on top of transformers' MimiModel holding OUR seeded synthetic weights
(audiocodecs_amd.checkpoint.synthetic_mimi_state_dict).  Stored per case: the reference's own
``sig_to_toks`` / ``toks_to_sig`` / ``sig_to_feats`` / ``toks_to_qfeats`` results, ``embs()`` in both
`latent` settings (strided), and for the tiny config every module output captured by forward hooks.
Inputs are NOT stored (re-drawn by `mimi_cases.make_input`).  `margin64` (fp64 oracle gap between the
best and second-best codeword, relative) is auxiliary -- the near-tie audit of the parity tests.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from audiocodecs_amd import checkpoint  # noqa: E402
from audiocodecs_amd.config import MIMI_24KHZ, MIMI_TINY  # noqa: E402
from mimi_cases import CASES, REC_STRIDE, make_input  # noqa: E402
from oracle import mimi_oracle as O  # noqa: E402
from reference_shim import load_reference_mimi  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ACT_FULL_MAX, ACT_STRIDE, EMBS_STRIDE = 16384, 13, 499


def _out(o):
    return (o[0] if isinstance(o, tuple) else o).detach().clone()


def main():
    import transformers

    torch.set_num_threads(8)
    out = {}
    meta = {
        "transformers": transformers.__version__,
        "torch": torch.__version__,
        "reference": "lucadellalib/audiocodecs v0.0.2 audiocodecs/mimi.py (Mimi.sig_to_toks/toks_to_sig)",
        "rec_stride": REC_STRIDE,
        "act_full_max": ACT_FULL_MAX,
        "act_stride": ACT_STRIDE,
        "embs_stride": EMBS_STRIDE,
        "cases": {},
    }
    models = {}
    for case in CASES:
        name, cfg_name, seed = case["name"], case["cfg"], case["weights_seed"]
        cfg = {"full": MIMI_24KHZ, "tiny": MIMI_TINY}[cfg_name]
        key = (cfg_name, seed)
        if key not in models:
            sd = checkpoint.synthetic_mimi_state_dict(cfg, seed=seed)
            models[key] = (sd, O.cast_weights(sd, torch.float64))
        sd, W64 = models[key]
        Mimi = load_reference_mimi(sd, cfg)
        K = case.get("K", 8)
        ref = Mimi(sample_rate=24000, num_codebooks=K).eval()
        ref_proj = Mimi(sample_rate=24000, num_codebooks=K, latent=False).eval()
        meta["attn_implementation"] = ref.model.config._attn_implementation

        inp = make_input(case, GOLD)
        info = {"K": K, "cfg": cfg_name, "weights_seed": seed}
        with torch.no_grad():
            if case["kind"] == "decode":
                toks = inp["toks"]
                rec = ref.toks_to_sig(toks)
            else:
                sig, length = inp["sig"], inp.get("length")
                acts, hooks = {}, []
                if case.get("taps"):
                    m = ref.model
                    named = [(f"enc{i}", l) for i, l in enumerate(m.encoder.layers)]
                    named += [(f"dec{i}", l) for i, l in enumerate(m.decoder.layers)]
                    named += [(f"enctr{i}", l) for i, l in enumerate(m.encoder_transformer.layers)]
                    named += [(f"dectr{i}", l) for i, l in enumerate(m.decoder_transformer.layers)]
                    named += [("downsample", m.downsample), ("upsample", m.upsample)]
                    for nm, layer in named:
                        hooks.append(layer.register_forward_hook(lambda mod, a, o, nm=nm: acts.__setitem__(nm, _out(o))))
                toks = ref.sig_to_toks(sig, length)
                rec = ref.toks_to_sig(toks)
                for h in hooks:
                    h.remove()
                feats = ref.sig_to_feats(sig, length)  # [B,N,hidden]
                out[f"{name}.feats_strided"] = feats.numpy().reshape(-1)[::REC_STRIDE].copy()
                if case.get("taps"):
                    out[f"{name}.feats"] = feats.numpy()
                    for nm, v in acts.items():  # big activations: every 13th element (flattened)
                        a = v.numpy().reshape(-1)
                        out[f"{name}.act.{nm}"] = a[:: (1 if a.size <= ACT_FULL_MAX else ACT_STRIDE)].copy()
                        info.setdefault("act_shapes", {})[nm] = list(v.shape)
                    out[f"{name}.rec_full"] = rec.numpy()[:, ::1 if rec.numel() <= ACT_FULL_MAX else ACT_STRIDE].copy()
                _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, K, True)
                out[f"{name}.margin64"] = m64.numpy().astype(np.float32)
                info["min_margin64"] = float(m64.min())
                out[f"{name}.toks"] = toks.numpy().astype(np.int16)
            qf = ref.toks_to_qfeats(toks)  # [B,N,hidden]
            out[f"{name}.qfeats_strided"] = qf.numpy().reshape(-1)[::REC_STRIDE].copy()
            if name in ("full_noise_b2", "tiny_taps", "full_T4800_K1"):
                out[f"{name}.embs_latent_strided"] = ref.embs().numpy().reshape(-1)[::EMBS_STRIDE].copy()
                out[f"{name}.embs_proj_strided"] = ref_proj.embs().numpy().reshape(-1)[::EMBS_STRIDE].copy()
                info["embs_shapes"] = [list(ref.embs().shape), list(ref_proj.embs().shape)]
        rec_np = rec.numpy()
        out[f"{name}.rec_strided"] = rec_np.reshape(-1)[::REC_STRIDE].copy()
        info.update(
            rec_shape=list(rec_np.shape),
            rec_rms=float(np.sqrt(np.mean(rec_np.astype(np.float64) ** 2))),
            rec_sha256=hashlib.sha256(rec_np.tobytes()).hexdigest(),
            toks_shape=list(toks.shape),
        )
        meta["cases"][name] = info
        print(name, info, flush=True)

    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLD, "mimi_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()

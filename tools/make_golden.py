#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE WRAPPER (build container only).

    python tools/make_golden.py            # rewrites tests/golden/encodec_golden.npz (+ example.wav)

What runs: ``audiocodecs.encodec.Encodec`` imported from /root/reference (shims in
tools/reference_shim.py) on top of transformers' EncodecModel holding OUR seeded synthetic weights
(audiocodecs_amd.checkpoint.synthetic_state_dict).  Expected outputs stored here are the
reference's own ``sig_to_toks`` / ``toks_to_sig`` results (and, for the tiny config, every module
output captured by forward hooks).  Inputs are NOT stored: they are re-drawn from the repo PRNG by
`golden_cases.make_input` on both sides (example.wav, a data file of the reference, is copied).

Besides the reference outputs each case carries `margin64`: the per-token relative gap between the
best and the second-best codeword computed by the fp64 oracle -- the near-tie audit the parity
tests use (SURVEY.md §7 hard part 2).  It is auxiliary (not a reference output) and labelled so.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from audiocodecs_amd import checkpoint  # noqa: E402
from audiocodecs_amd.config import ENCODEC_24KHZ, TINY  # noqa: E402
from golden_cases import CASES, REC_STRIDE, make_input  # noqa: E402
from oracle import encodec_oracle as O  # noqa: E402
from reference_shim import load_reference_encodec  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    import transformers

    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    shutil.copyfile("/root/reference/audiocodecs/example.wav", os.path.join(GOLD, "example.wav"))

    out = {}
    meta = {
        "transformers": transformers.__version__,
        "torch": torch.__version__,
        "reference": "lucadellalib/audiocodecs v0.0.2 audiocodecs/encodec.py (Encodec.sig_to_toks/toks_to_sig)",
        "rec_stride": REC_STRIDE,
        "cases": {},
    }
    models = {}
    for case in CASES:
        name, cfg_name, seed = case["name"], case["cfg"], case["weights_seed"]
        cfg = {"full": ENCODEC_24KHZ, "tiny": TINY}[cfg_name]
        key = (cfg_name, seed)
        if key not in models:
            sd = checkpoint.synthetic_state_dict(cfg, seed=seed)
            hf_kwargs = {} if cfg_name == "full" else dict(num_filters=cfg.num_filters, hidden_size=cfg.hidden_size)
            models[key] = (sd, hf_kwargs, O.fold_weight_norm(sd, torch.float64))
        sd, hf_kwargs, W64 = models[key]
        Encodec = load_reference_encodec(sd, hf_kwargs)
        K = case.get("K", 8)
        ref = Encodec(sample_rate=24000, orig_sample_rate=24000, num_codebooks=K).eval()

        inp = make_input(case, GOLD)
        info = {"K": K, "cfg": cfg_name, "weights_seed": seed}
        with torch.no_grad():
            if case["kind"] == "decode":
                toks = inp["toks"]
                rec = ref.toks_to_sig(toks)
            else:
                sig, length = inp["sig"], inp.get("length")
                acts = {}
                hooks = []
                if case.get("taps"):
                    for part in ("encoder", "decoder"):
                        for i, layer in enumerate(getattr(ref.model, part).layers):
                            hooks.append(
                                layer.register_forward_hook(
                                    lambda m, a, o, nm=f"{part[:3]}{i}": acts.__setitem__(nm, o.detach().clone())
                                )
                            )
                toks = ref.sig_to_toks(sig, length)
                rec = ref.toks_to_sig(toks)
                for h in hooks:
                    h.remove()
                feats = ref.sig_to_feats(sig, length)  # [B,N,H]: encoder output, reference's own API
                out[f"{name}.feats_strided"] = feats.numpy().reshape(-1)[::REC_STRIDE].copy()
                if case.get("taps"):
                    out[f"{name}.feats"] = feats.numpy()
                    for nm, v in acts.items():
                        out[f"{name}.act.{nm}"] = v.numpy()
                    out[f"{name}.rec_full"] = rec.numpy()
                # auxiliary: fp64 oracle margins
                _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None if length is None else length.double(), K, True)
                out[f"{name}.margin64"] = m64.numpy().astype(np.float32)
                info["min_margin64"] = float(m64.min())
                out[f"{name}.toks"] = toks.numpy().astype(np.int16)
        rec_np = rec.numpy()
        out[f"{name}.rec_strided"] = rec_np.reshape(-1)[::REC_STRIDE].copy()
        info.update(
            rec_shape=list(rec_np.shape),
            rec_rms=float(np.sqrt(np.mean(rec_np.astype(np.float64) ** 2))),
            rec_sha256=hashlib.sha256(rec_np.tobytes()).hexdigest(),
            toks_shape=list(toks.shape),
        )
        meta["cases"][name] = info
        print(name, info)

    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLD, "encodec_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()

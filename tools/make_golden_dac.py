#!/usr/bin/env python3
"""Generate tests/golden/dac_golden.npz from the STAND-IN for the reference's DAC backend.

    python tools/make_golden_dac.py

`audiocodecs.dac.DAC` (/root/reference/audiocodecs/dac.py) cannot be imported here: its backend
`descript-audio-codec` is not installed (dac.py:44-48 raises ImportError).  What runs instead is the
same-architecture third-party `transformers.DacModel` holding OUR seeded synthetic weights, called the way
the wrapper calls `dac.DAC`:
    _sig_to_toks   (dac.py:96-99)   model.encode(sig[:, None], n_quantizers=K)          -> audio_codes
    _sig_to_qfeats (dac.py:117-119) same call                                            -> quantized_representation
    _sig_to_feats  (dac.py:109-111) model.encoder(sig[:, None])
    _toks_to_sig   (dac.py:126-129) model.quantizer.from_codes(toks.movedim(-1,-2))[0] -> model.decoder(...)[:, 0]
    embs           (dac.py:63-90)   codebooks / out_proj(codebooks)
The fixtures therefore pin oracle and HIP path to the stand-in only ("parity unpinned" w.r.t. the reference).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from audiocodecs_amd import checkpoint  # noqa: E402
from audiocodecs_amd.config import DAC_44KHZ, DAC_TINY  # noqa: E402
from dac_cases import CASES, REC_STRIDE, make_input  # noqa: E402
from oracle import dac_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ACT_FULL_MAX, ACT_STRIDE, EMBS_STRIDE = 16384, 13, 499


def hf_config(cfg):
    from transformers import DacConfig as HFConfig

    return HFConfig(
        encoder_hidden_size=cfg.encoder_hidden_size, downsampling_ratios=list(cfg.downsampling_ratios),
        decoder_hidden_size=cfg.decoder_hidden_size, upsampling_ratios=list(cfg.upsampling_ratios),
        n_codebooks=cfg.n_codebooks, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
        sampling_rate=cfg.sampling_rate,
    )


def main():
    import transformers
    from transformers import DacModel

    torch.set_num_threads(8)
    out = {}
    meta = {
        "transformers": transformers.__version__, "torch": torch.__version__,
        "source": "STAND-IN transformers.DacModel called as audiocodecs/dac.py calls dac.DAC (descript-audio-codec absent)",
        "rec_stride": REC_STRIDE, "act_full_max": ACT_FULL_MAX, "act_stride": ACT_STRIDE, "embs_stride": EMBS_STRIDE,
        "cases": {},
    }
    models = {}
    for case in CASES:
        name, cfg_name, seed = case["name"], case["cfg"], case["weights_seed"]
        cfg = {"full": DAC_44KHZ, "tiny": DAC_TINY}[cfg_name]
        key = (cfg_name, seed)
        if key not in models:
            sd = checkpoint.synthetic_dac_state_dict(cfg, seed=seed)
            m = DacModel(hf_config(cfg)).eval()
            m.load_state_dict(sd, strict=True)
            models[key] = (m, O.cast_weights(sd, torch.float64))
        m, W64 = models[key]
        K = case["K"]
        inp = make_input(case, GOLD)
        info = {"K": K, "cfg": cfg_name, "weights_seed": seed}
        with torch.no_grad():
            if case["kind"] == "decode":
                toks = inp["toks"]
            else:
                sig = inp["sig"]
                acts, hooks = {}, []
                if case.get("taps"):
                    named = [("encoder.conv1", m.encoder.conv1), ("encoder.conv2", m.encoder.conv2),
                             ("decoder.conv1", m.decoder.conv1), ("decoder.conv2", m.decoder.tanh)]
                    for part, blocks in (("encoder", m.encoder.block), ("decoder", m.decoder.block)):
                        for i, blk in enumerate(blocks):
                            for u in (1, 2, 3):
                                named.append((f"{part}.block.{i}.res_unit{u}", getattr(blk, f"res_unit{u}")))
                            named.append((f"{part}.block.{i}.conv1", blk.conv1) if part == "encoder" else (f"{part}.block.{i}.conv_t1", blk.conv_t1))
                    for nm, layer in named:
                        hooks.append(layer.register_forward_hook(lambda mod, a, o, nm=nm: acts.__setitem__(nm, o.detach().clone())))
                enc = m.encode(sig[:, None], n_quantizers=K)
                toks = enc.audio_codes.movedim(-1, -2).contiguous()        # [B,N,K]
                qfeats = enc.quantized_representation.movedim(-1, -2)       # [B,N,H]
                feats = m.encoder(sig[:, None]).movedim(-1, -2)             # [B,N,H]
                feats_lat = m.quantizer.quantizers[0].in_proj(m.encoder(sig[:, None])).movedim(-1, -2)
                out[f"{name}.feats_strided"] = feats.numpy().reshape(-1)[::REC_STRIDE].copy()
                out[f"{name}.feats_latent"] = feats_lat.numpy().reshape(-1)[::7].copy()
                out[f"{name}.qfeats_fwd_strided"] = qfeats.numpy().reshape(-1)[::REC_STRIDE].copy()
                _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, K, "descript", True)
                out[f"{name}.margin64"] = m64.numpy().astype(np.float32)
                info["min_margin64"] = float(m64.min())
                out[f"{name}.toks"] = toks.numpy().astype(np.int16)
            z_q = m.quantizer.from_codes(toks.movedim(-1, -2))[0]
            out[f"{name}.qfeats_codes_strided"] = z_q.movedim(-1, -2).numpy().reshape(-1)[::REC_STRIDE].copy()
            rec = m.decoder(z_q)[:, 0]
            if case.get("taps"):
                for h in hooks:
                    h.remove()
                for nm, v in acts.items():
                    a = v.numpy().reshape(-1)
                    out[f"{name}.act.{nm}"] = a[:: (1 if a.size <= ACT_FULL_MAX else ACT_STRIDE)].copy()
                    info.setdefault("act_shapes", {})[nm] = list(v.shape)
            if name in ("full_noise_b2", "tiny_taps"):
                cbs = torch.stack([q.codebook.weight for q in m.quantizer.quantizers[:K]])
                proj = torch.stack([q.out_proj(q.codebook.weight[:, :, None])[..., 0] for q in m.quantizer.quantizers[:K]])
                out[f"{name}.embs_latent_strided"] = cbs.numpy().reshape(-1)[::EMBS_STRIDE].copy()
                out[f"{name}.embs_proj_strided"] = proj.numpy().reshape(-1)[::EMBS_STRIDE].copy()
                info["embs_shapes"] = [list(cbs.shape), list(proj.shape)]
        rec_np = rec.numpy()
        out[f"{name}.rec_strided"] = rec_np.reshape(-1)[::REC_STRIDE].copy()
        info.update(
            rec_shape=list(rec_np.shape), rec_rms=float(np.sqrt(np.mean(rec_np.astype(np.float64) ** 2))),
            rec_sha256=hashlib.sha256(rec_np.tobytes()).hexdigest(), toks_shape=list(toks.shape),
        )
        meta["cases"][name] = info
        print(name, {k: v for k, v in info.items() if k != "act_shapes"}, flush=True)

    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLD, "dac_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()

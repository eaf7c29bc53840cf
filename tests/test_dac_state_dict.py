"""descript-audio-codec checkpoint names -> loader names (audiocodecs_amd.dac.state_dict_from_descript).

The mapping is written from the published module structure of dac.model.dac.DAC (nn.Sequential indices, old-style
weight-norm `weight_g` / `weight_v`); no real checkpoint is available offline, so this test only pins its
self-consistency: a synthetic checkpoint re-expressed in descript's naming converts back bit-exactly."""
import torch

from audiocodecs_amd import checkpoint
from audiocodecs_amd.config import DAC_TINY
from audiocodecs_amd.dac import state_dict_from_descript


def to_descript(sd, cfg):
    nb, nu = len(cfg.downsampling_ratios), len(cfg.dilations)
    ren = {"encoder.conv1": "encoder.block.0", "encoder.snake1": f"encoder.block.{nb + 1}", "encoder.conv2": f"encoder.block.{nb + 2}",
           "decoder.conv1": "decoder.model.0", "decoder.snake1": f"decoder.model.{nb + 1}", "decoder.conv2": f"decoder.model.{nb + 2}"}
    unit = {"snake1": 0, "conv1": 1, "snake2": 2, "conv2": 3}
    for i in range(nb):
        for u in range(nu):
            for nm, j in unit.items():
                ren[f"encoder.block.{i}.res_unit{u + 1}.{nm}"] = f"encoder.block.{i + 1}.block.{u}.block.{j}"
                ren[f"decoder.block.{i}.res_unit{u + 1}.{nm}"] = f"decoder.model.{i + 1}.block.{u + 2}.block.{j}"
        ren[f"encoder.block.{i}.snake1"] = f"encoder.block.{i + 1}.block.{nu}"
        ren[f"encoder.block.{i}.conv1"] = f"encoder.block.{i + 1}.block.{nu + 1}"
        ren[f"decoder.block.{i}.snake1"] = f"decoder.model.{i + 1}.block.0"
        ren[f"decoder.block.{i}.conv_t1"] = f"decoder.model.{i + 1}.block.1"
    out = {}
    for k, v in sd.items():
        prefix, leaf = k.rsplit(".", 1)
        new = ren.get(prefix, prefix)
        if leaf == "weight" and "codebook" not in prefix:           # weight-normed convs: g = |v|, v = w
            g = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            out[new + ".weight_g"] = g
            out[new + ".weight_v"] = v.clone()
        else:
            out[f"{new}.{leaf}"] = v
    return out


def test_descript_names_round_trip():
    cfg = DAC_TINY
    sd = checkpoint.synthetic_dac_state_dict(cfg, seed=3)
    back = state_dict_from_descript(to_descript(sd, cfg), cfg)
    assert set(back) == set(sd)
    for k in sd:
        if k.endswith(".weight") and "codebook" not in k:
            torch.testing.assert_close(back[k], sd[k], rtol=2e-7, atol=0)   # g * v / |v| with g = |v|
        else:
            assert torch.equal(back[k], sd[k]), k

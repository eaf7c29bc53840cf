"""Collects the parity figures the GPU tests measure (per fixture: token exact-match rate, near-tie tokens excused,
tokens outside the near-tie band that differ, waveform / feature RMS error) and writes them at session end:
`parity_report.json` (repo root; also `gpurun_out/parity_report.json` when that directory exists) plus ONE summary line
on the terminal, so the numbers DESIGN.md quotes are in the test log and not only in swallowed prints."""
from __future__ import annotations

import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROWS: dict = {}


def record(codec: str, case: str, **fields) -> None:
    row = _ROWS.setdefault(f"{codec}/{case}", {})
    for k, v in fields.items():
        row[k] = v.item() if hasattr(v, "item") else v


def tokens(codec: str, case: str, toks, gold, margin=None, tau=None):
    """toks / gold: numpy int arrays [B,N,K]; margin: fp64 relative margins (same shape) or None.
    Returns (mismatches, bad, excused): `bad` = mismatches outside the near-tie band (must be 0)."""
    import numpy as np

    mism = int((toks != gold).sum())
    bad, excused = mism, 0
    if margin is not None:
        safe = np.cumprod(margin > tau, axis=-1).astype(bool)
        bad = int(((toks != gold) & safe).sum())
        excused = int((~safe).sum())
    record(codec, case, tokens=int(gold.size), token_mismatches=mism, token_exact_rate=1.0 - mism / max(1, gold.size),
           near_tie_tokens=excused, mismatches_outside_near_ties=bad,
           min_margin64=(float(margin.min()) if margin is not None and margin.size else None))
    return mism, bad, excused


def summary() -> dict:
    rows = list(_ROWS.values())
    tok = [r for r in rows if "tokens" in r]
    rms = [r["waveform_rms_err"] for r in rows if r.get("waveform_rms_err") is not None]
    return {
        "cases": len(rows),
        "tokens_compared": sum(r["tokens"] for r in tok),
        "token_mismatches": sum(r["token_mismatches"] for r in tok),
        "mismatches_outside_near_ties": sum(r["mismatches_outside_near_ties"] for r in tok),
        "near_tie_tokens": sum(r["near_tie_tokens"] for r in tok),
        "worst_waveform_rms_err": max(rms) if rms else None,
    }


def write() -> str | None:
    if not _ROWS:
        return None
    doc = {"summary": summary(), "cases": _ROWS}
    paths = [os.path.join(ROOT, "parity_report.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(doc, f, indent=1, sort_keys=True)
        except OSError:
            pass
    s = doc["summary"]
    return (f"PARITY {s['cases']} cases: {s['tokens_compared']} tokens compared, {s['token_mismatches']} differ "
            f"({s['mismatches_outside_near_ties']} outside fp64 near-ties; {s['near_tie_tokens']} tokens sit in near-tie frames); "
            f"worst waveform RMS error {s['worst_waveform_rms_err']}")

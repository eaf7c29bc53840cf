#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: separate passes (TCC slots), values in KiB
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024), and on gfx950 FETCH_SIZE under-counts wide coalesced
streaming reads by exactly 2x -> doubled here.  Usage:
    python tools/collect_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv \
                                    gpurun_out/pmc_write/w_counter_collection.csv > profiles/rN_traffic.json
    [steps_in_trace]   (bench.py --steps 1 --warmup 1 runs the step 3 times: warm-up, timed, unprofiled repeat)
Only the LAST step of the trace is counted per kernel name (warm caches)."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = collections.OrderedDict()
    for r in rows:
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*\)$", "", name).replace("ac::", "")
        out.setdefault(name, []).append(float(r["Counter_Value"]))
    return out


def main(fetch_csv, write_csv, steps=3):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for name in f:
        fl, wl = f[name], w.get(name, [])
        n = max(1, len(fl) // steps)         # launches of this kernel in one step
        fl, wl = fl[-n:], wl[-n:]
        res[name] = {
            "launches": n,
            "fetch_bytes_per_launch_corrected": sum(fl) * 1024 * 2 / n,
            "write_bytes_per_launch": (sum(wl) * 1024 / n) if wl else None,
            "hbm_bytes_per_launch": (sum(fl) * 2 + sum(wl)) * 1024 / n if wl else None,
        }
    json.dump({"note": "FETCH_SIZE doubled (gfx950 wide-read correction); KiB -> bytes; per launch, last step of the trace",
               "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)

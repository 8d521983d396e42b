#!/usr/bin/env python
"""Per-kernel sums of a `rocprofv3 --pmc ... --kernel-trace -f csv` run (the *_counter_collection.csv file).

usage: pmc_summary.py <counter_collection.csv> [out.json]

Prints one JSON object {kernel: {launches, <counter>: sum, ...}} and the derived figures used in profiles/:
  * MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): GRBM_GUI_ACTIVE is summed over the 8 XCDs of the
    MI355X, SQ_VALU_MFMA_BUSY_CYCLES over its 1024 SIMDs (256 CUs x 4) -- the round-1 file divided by the 8-XCD sum and under-reported
    the busy fraction 8x;
  * fp64 MFMA flop = SQ_INSTS_VALU_MFMA_F64 [per-wave instruction count] x 2048 for v_mfma_f64_16x16x4_f64;
  * HBM read bytes = 2 * FETCH_SIZE * 1024 (gfx950: the counter ticks 64-byte requests but rocprofv3 scales it as 32-byte ones, see the
    HBM section of /opt/skills/guides/MI355X_MICROARCH.md as applied in profiles/r01_cfg2_pmc_traffic.json), write bytes = WRITE_SIZE * 1024.
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:64]


def main(path, out=None):
    acc = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    res = {}
    for k, c in acc.items():
        e = {"launches": len(seen[k])}
        e.update({n: v for n, v in c.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_INSTS_VALU_MFMA_F64" in c:
            e["mfma_f64_flop_per_launch"] = c["SQ_INSTS_VALU_MFMA_F64"] * 2048.0 / max(len(seen[k]), 1)
        if "FETCH_SIZE" in c:
            e["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / max(len(seen[k]), 1)
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / max(len(seen[k]), 1)
        res[k] = e
    txt = json.dumps(res, indent=1, sort_keys=True)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

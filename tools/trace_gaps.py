#!/usr/bin/env python
"""Timeline summary of a `rocprofv3 --kernel-trace -f csv` run: where the wall clock between the first and the last dispatch goes.

Prints (and optionally writes) per kernel: calls, busy time, and the IDLE time that precedes it (gap between the end of the previous
dispatch on the device and its own start), then the (previous kernel -> kernel) pairs with the largest accumulated gaps.  Used to
tell launch-/host-bound phases (gap >> duration) from kernel-bound ones.

usage: trace_gaps.py <kernel_trace.csv> [out.txt] [--skip-first N | --tail F]   (--tail 0.5: only the last half of the dispatches)
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:48]


def main(path, out=None, skip=0):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[skip:] if skip >= 0 else rows[int(len(rows) * (1.0 + skip)):]
    if not rows:
        print("empty trace")
        return
    busy = defaultdict(int)
    gap = defaultdict(int)
    calls = defaultdict(int)
    pair_gap = defaultdict(int)
    pair_n = defaultdict(int)
    prev_end, prev_name = rows[0][0], "<start>"
    overlap = 0
    for s, e, nme in rows:
        calls[nme] += 1
        busy[nme] += e - s
        g = s - prev_end
        if g > 0:
            gap[nme] += g
            pair_gap[(prev_name, nme)] += g
            pair_n[(prev_name, nme)] += 1
        else:
            overlap += -g
        if e > prev_end:
            prev_end, prev_name = e, nme
    span = max(e for _, e, _ in rows) - rows[0][0]
    tb, tg = sum(busy.values()), sum(gap.values())
    lines = ["span %.3f ms, kernel busy %.3f ms (%.1f %%), idle gaps %.3f ms (%.1f %%), overlap %.3f ms, %d dispatches"
             % (span / 1e6, tb / 1e6, 100.0 * tb / span, tg / 1e6, 100.0 * tg / span, overlap / 1e6, len(rows)),
             "%-48s %8s %10s %10s %10s %10s" % ("KERNEL", "CALLS", "BUSY_ms", "AVG_us", "GAP_ms", "GAP_AVG_us")]
    for nme in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
        lines.append("%-48s %8d %10.3f %10.3f %10.3f %10.3f" % (nme, calls[nme], busy[nme] / 1e6, busy[nme] / 1e3 / calls[nme],
                                                                gap[nme] / 1e6, gap[nme] / 1e3 / calls[nme]))
    lines.append("")
    lines.append("largest accumulated gaps (previous kernel -> kernel):")
    for (a, b) in sorted(pair_gap, key=lambda k: -pair_gap[k])[:25]:
        lines.append("  %-40s -> %-40s n=%6d total %9.3f ms avg %8.2f us" % (a, b, pair_n[(a, b)], pair_gap[(a, b)] / 1e6,
                                                                           pair_gap[(a, b)] / 1e3 / pair_n[(a, b)]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sk = 0
    for flag in ("--skip-first", "--tail"):
        if flag in sys.argv:
            val = sys.argv[sys.argv.index(flag) + 1]
            args = [a for a in args if a != val]
            sk = int(val) if flag == "--skip-first" else -float(val)
    main(args[0], args[1] if len(args) > 1 else None, sk)

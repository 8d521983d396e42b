"""Median duration of the real (non-gated) symmetric-product launches in a rocprofv3 kernel trace.  usage: gemm_lab_summary.py <dir> <label>"""
import csv, glob, sys
import numpy as np
rows = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_symm_gemm" in n:
            key = n[n.find("k_symm_gemm"):n.find("(")]
            rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(rows.items()):
    v = np.array(v); real = v[v > 20000]
    print("%-10s %-28s launches %4d real %4d  median %.1f us  min %.1f  p90 %.1f" % (sys.argv[2], k, len(v), len(real), np.median(real) / 1e3 if len(real) else 0,
          real.min() / 1e3 if len(real) else 0, np.percentile(real, 90) / 1e3 if len(real) else 0))

#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r03/trace_cfg5 -- python $R/bench.py --workload cfg5 --steps 12 --warmup 4 --no-cpu-baseline --no-float32 > /dev/null 2>&1
cd $R
f=$(find gpurun_out/r03/trace_cfg5 -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "symm_gemm" in r["Name"] or "k_cg_" in r["Name"]:
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ",""); n=n[:n.find("(")]
        print(n, r["Calls"], round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2))
PY
rm -rf gpurun_out/r03/trace_cfg5

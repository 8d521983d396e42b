#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/r03/trace_cfg5 -- python $R/bench.py --workload cfg5 --steps 12 --warmup 4 --no-cpu-baseline --no-float32 > /dev/null 2>&1
cd $R
f=$(find gpurun_out/r03/trace_cfg5 -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f gpurun_out/r03/cfg5_timeline_v4.txt --tail 0.35 | head -34
rm -rf gpurun_out/r03/trace_cfg5

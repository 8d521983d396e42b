mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/t4.log; cat gpurun_out/t4.log
timeout 400 python bench.py 2>gpurun_out/bench2.err | tail -1 > gpurun_out/bench2.json; cat gpurun_out/bench2.json; tail -3 gpurun_out/bench2.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r1 -o cfg2 -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --exact-launches > /root/repo/gpurun_out/prof_r1.log 2>&1
tail -2 /root/repo/gpurun_out/prof_r1.log
find /root/repo/gpurun_out/prof_r1 -name "*stats*" | head

mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$c -o cfg2 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --exact-launches > /root/repo/gpurun_out/pmc_$c.log 2>&1
tail -1 /root/repo/gpurun_out/pmc_$c.log | cut -c1-200
ls /root/repo/gpurun_out/pmc_$c/
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r1b -o cfg2 -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --exact-launches > /root/repo/gpurun_out/prof_r1b.log 2>&1
ls /root/repo/gpurun_out/prof_r1b/
cd /root/repo
timeout 400 python bench.py 2>gpurun_out/bench3.err | tail -1 > gpurun_out/bench3.json; cut -c1-600 gpurun_out/bench3.json

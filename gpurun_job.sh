mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_sharding.py -m gpu -q -x 2>&1 | tail -30) > gpurun_out/t10.log; cat gpurun_out/t10.log

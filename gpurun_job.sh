mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q 2>&1 | tail -40) > gpurun_out/t14.log; cat gpurun_out/t14.log

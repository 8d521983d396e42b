mkdir -p gpurun_out
(timeout 500 python -m pytest tests/test_gpu_parity_psd.py -m gpu -q -x 2>&1 | tail -40) > gpurun_out/t5.log; cat gpurun_out/t5.log

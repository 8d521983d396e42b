mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_batch.py -m gpu -q -x -s 2>&1 | tail -40) > gpurun_out/t9.log; cat gpurun_out/t9.log

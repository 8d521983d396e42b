mkdir -p gpurun_out
(timeout 800 python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/t8.log; cat gpurun_out/t8.log

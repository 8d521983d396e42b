mkdir -p gpurun_out
(timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/t12.log; cat gpurun_out/t12.log
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1

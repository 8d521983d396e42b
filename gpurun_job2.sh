#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_cg_fold.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for rep in 1 2 3; do
for v in 0 1; do
  COSMO_HIP_CG_GRAPH=$v timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 graph=$v', round(d['value'],2), 'f32', round(d['float32']['value'],2) if isinstance(d.get('float32'),dict) else d.get('float32'))"
done
done
COSMO_HIP_CG_GRAPH_LEN=8 timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 len8', round(d['value'],2), 'f32', round(d['float32']['value'],2) if isinstance(d.get('float32'),dict) else d.get('float32'))"
COSMO_HIP_CG_GRAPH_LEN=32 timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 len32', round(d['value'],2), 'f32', round(d['float32']['value'],2) if isinstance(d.get('float32'),dict) else d.get('float32'))"

#!/bin/bash
# quick correctness + timing loop for the generator kernels (run on the GPU box from the repo root)
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_generator.py tests/test_gpu_chain.py tests/test_gpu_full_size.py tests/test_gpu_pa_critic.py -x -q --timeout 120 --timeout-method thread > gpurun_out/quick_tests.log 2>&1 || { tail -n 30 gpurun_out/quick_tests.log; exit 1; }
tail -n 2 gpurun_out/quick_tests.log
for w in "$@"; do
  timeout -k 10 120 python -u bench.py --workload $w --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w', 'ms/step %.3f'%d['ms_per_step'], 'kernel_ms %.4f'%r.get('kernel_ms',0), 'frac %.3f'%r['frac'], 'chain_frac', r.get('chain_frac'))" || exit 1
done

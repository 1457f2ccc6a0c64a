#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -n 40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -n 2 gpurun_out/r3_gpu_tests.log
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/r3_bench_nocpu.json 2> gpurun_out/r3_bench_nocpu.err || { tail -n 20 gpurun_out/r3_bench_nocpu.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_nocpu.json').read().strip().splitlines()[-1])
print('stage2 ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))
for k,v in d['extra'].items():
    r=v['roofline']; print(k, 'ms/step %.3f'%v['ms_per_step'], r['bound'], 'frac %.3f'%r['frac'], 'kernel_ms %.4f'%r.get('kernel_ms',0), 'fp32eq', round(r.get('fp32_equivalent',{}).get('ratio',0),3))
PY

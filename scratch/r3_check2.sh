#!/bin/bash
# round 3, second GPU call: spectral tests, kernel timings of the product build and two variants, gloo run with a device synchronise
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_spectral_quant.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3_t2.log 2>&1
rc=$?; tail -n 6 gpurun_out/r3_t2.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
timeout -k 10 200 python -u scratch/tk.py reps=20 >> gpurun_out/r3_tk_base.log 2>&1 || { tail -n 20 gpurun_out/r3_tk_base.log; exit 1; }
DPD_HIP_LIBRARY=$PWD/scratch/lib_q88w4.so timeout -k 10 100 python -u scratch/tk.py q88 reps=20 >> gpurun_out/r3_tk_base.log 2>&1 || { tail -n 20 gpurun_out/r3_tk_base.log; exit 1; }
DPD_HIP_LIBRARY=$PWD/scratch/lib_specnarrow.so timeout -k 10 100 python -u scratch/tk.py spec reps=20 >> gpurun_out/r3_tk_base.log 2>&1 || { tail -n 20 gpurun_out/r3_tk_base.log; exit 1; }
done
cat gpurun_out/r3_tk_base.log
DPD_BENCH_REHEARSAL=1 DPD_GLOO_DEVICE_TENSORS=2 DPD_BENCH_WATCHDOG=50 timeout -k 10 120 python -u bench.py --gpus 2 --steps 30 --warmup 2 --rows 1024 --no-extras --no-cpu-baseline > gpurun_out/r3_gloo_sync.log 2>&1
echo "gloo device tensors + synchronize: rc $?"
grep -c "Timeout" gpurun_out/r3_gloo_sync.log; tail -n 3 gpurun_out/r3_gloo_sync.log | cut -c1-400
exit 0

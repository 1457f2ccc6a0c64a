#!/bin/bash
# round 3, first GPU call: the whole GPU suite, then the two-rank gloo run on device tensors under the watchdog
R=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_t1.log 2>&1
rc=$?
tail -n 15 gpurun_out/r3_t1.log
[ $rc -ne 0 ] && exit $rc
# the stalled configuration (gloo all_reduce on device tensors, two ranks on GPU 0): stack dump after 60 s
DPD_BENCH_REHEARSAL=1 DPD_GLOO_DEVICE_TENSORS=1 DPD_BENCH_WATCHDOG=60 timeout -k 10 150 python -u bench.py --gpus 2 --steps 30 --warmup 2 --rows 1024 --no-extras --no-cpu-baseline > gpurun_out/r3_gloo_stall.log 2>&1
echo "gloo device-tensor run rc $?"
tail -n 60 gpurun_out/r3_gloo_stall.log
exit 0

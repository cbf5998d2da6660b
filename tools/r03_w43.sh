#!/bin/bash
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wino43" > gpurun_out/r3_w43_test.log 2>&1; rc=$?
tail -25 gpurun_out/r3_w43_test.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/wino43_layers.py 2>&1 | grep -v amdgpu.ids

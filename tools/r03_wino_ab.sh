#!/bin/bash
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wino or plan or conv" > gpurun_out/r3_wino_test.log 2>&1 || { tail -30 gpurun_out/r3_wino_test.log; exit 1; }
tail -2 gpurun_out/r3_wino_test.log
STEPS=80 bash tools/lib_ab.sh

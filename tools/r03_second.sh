#!/bin/bash
set -x
cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_all.log 2>&1; rc=$?
tail -15 gpurun_out/r3_gpu_all.log
exit $rc

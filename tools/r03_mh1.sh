#!/bin/bash
cd /root/repo
ST3D_WINO_MH1_MAXK=512 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wino or plan" > gpurun_out/r3_mh1_test.log 2>&1 || { tail -30 gpurun_out/r3_mh1_test.log; exit 1; }
tail -2 gpurun_out/r3_mh1_test.log
for k in 0 512 0 512; do echo "== MAXK=$k"; ST3D_WINO_MH1_MAXK=$k timeout -k 10 300 python tools/wino_layers.py 2>&1 | grep -v amdgpu.ids; done

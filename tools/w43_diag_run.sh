#!/bin/bash
# on the GPU box: F(4x4,3x3) per-layer sums for the shipped library and every diagnostic build given (tools/w43_diag_build.sh)
cd /root/repo
echo "== base"; timeout -k 10 200 python tools/wino43_layers.py 2>&1 | grep -v amdgpu.ids | cut -c1-125 || exit 1
for n in "$@"; do
    echo "== diag $n"
    ST3D_DIAG_LIB=2d-to-3d-style-transfer_amd/lib/diag/libst3d_d$n.so timeout -k 10 200 python tools/wino43_layers.py 2>&1 | grep -v amdgpu.ids | cut -c1-125 || exit 1
done

#!/bin/bash
# on the GPU box: time every lib/diag/libst3d_<name>.so given as argument, twice, interleaved (F(4x4) sum and per layer)
cd /root/repo
for rep in 1 2; do
  for n in "$@"; do
    echo "== $n (run $rep)"
    ST3D_DIAG_LIB=2d-to-3d-style-transfer_amd/lib/diag/libst3d_$n.so timeout -k 10 200 python tools/wino43_layers.py 2>&1 | grep -v amdgpu.ids | cut -c1-125 || exit 1
  done
done

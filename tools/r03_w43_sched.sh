#!/bin/bash
cd /root/repo
L=2d-to-3d-style-transfer_amd/lib
cp $L/libst3d.so $L/keep.so
for v in 0 1; do cp $L/libst3d_s$v.so $L/libst3d.so; echo "== variant $v (0 = U always the same 36 KB, 1 = real)"; timeout -k 10 300 python tools/wino43_layers.py 2>&1 | grep -E "conv3_2|conv4_2|conv4_4|sum"; done
cp $L/keep.so $L/libst3d.so

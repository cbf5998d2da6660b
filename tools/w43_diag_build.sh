#!/bin/bash
# Timing-only diagnostic builds of csrc/wino43.hip (wrong results): lib/diag/libst3d_d<N>.so for ST3D_W43_DIAG=N
#   1 = filter operands from one cache-resident k-step, 2 = no epilogue, 3 = epilogue without global loads / stores
# Run on the GPU box with:  ST3D_DIAG_LIB=2d-to-3d-style-transfer_amd/lib/diag/libst3d_d2.so python tools/wino43_layers.py
set -e
cd "$(dirname "$0")/.."
P=2d-to-3d-style-transfer_amd
python $P/build.py --jobs 8 > /dev/null
mkdir -p $P/lib/diag
for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-variable -Wno-unused-but-set-variable -fno-slp-vectorize \
        -DST3D_W43_DIAG=$n $W43_EXTRA -c $P/csrc/wino43.hip -o $P/lib/diag/wino43_d$n.o &
done
wait
for n in "$@"; do
    objs=$(ls $P/lib/obj/*.o | grep -v wino43.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/diag/libst3d_d$n.so $objs $P/lib/diag/wino43_d$n.o -ldl
    rm $P/lib/diag/wino43_d$n.o
done
ls -la $P/lib/diag

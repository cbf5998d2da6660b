#!/bin/bash
# A/B builds of csrc/wino43.hip variants: tools/w43_ab_build.sh name=path.hip [name=path.hip ...] -> lib/diag/libst3d_<name>.so
# (the variant source is compiled in place of csrc/wino43.hip; everything else comes from lib/obj).  Timing:
#   ST3D_DIAG_LIB=2d-to-3d-style-transfer_amd/lib/diag/libst3d_<name>.so python tools/wino43_layers.py
set -e
cd "$(dirname "$0")/.."
P=2d-to-3d-style-transfer_amd
python $P/build.py --jobs 8 > /dev/null
mkdir -p $P/lib/diag
for kv in "$@"; do
    n=${kv%%=*}; src=${kv#*=}
    cp "$src" $P/csrc/_ab_$n.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-variable -Wno-unused-but-set-variable -fno-slp-vectorize \
        $W43_EXTRA -c $P/csrc/_ab_$n.hip -o $P/lib/diag/wino43_$n.o &
done
wait
for kv in "$@"; do
    n=${kv%%=*}
    rm -f $P/csrc/_ab_$n.hip
    objs=$(ls $P/lib/obj/*.o | grep -v wino43.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/diag/libst3d_$n.so $objs $P/lib/diag/wino43_$n.o -ldl
    rm $P/lib/diag/wino43_$n.o
done
ls $P/lib/diag

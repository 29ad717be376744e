#!/bin/bash
# build_variant.sh <outname> <extra flags...>: builds scripts/bin/<outname>.so with conv.hip recompiled with flags
out=$1; shift
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c vqa_amd/csrc/conv.hip -o /tmp/conv_$out.o || exit 1
objs=$(ls vqa_amd/csrc/*.o | grep -v conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv_$out.o -o scripts/bin/$out.so

#!/bin/bash
# Build libvxm_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# Links libamdhip64 by SONAME only (no RPATH): the process must resolve it to the copy torch has
# already loaded, so that streams / device pointers are shared.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libvxm_hip.so
SRCS="api.hip warp.hip planar.hip conv_fwd.hip conv_bwd_weight.hip conv_bf16.hip conv_s3.hip conv_s3u.hip pool.hip losses.hip diag.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment"
mkdir -p build
objs=""
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ vxm_device.h -nt "$o" ] || [ vxm_common.h -nt "$o" ] || [ conv_common.h -nt "$o" ] || [ s3_pieces.h -nt "$o" ] || [ ../../include/vxm_hip.h -nt "$o" ]; then
    hipcc $FLAGS -c "$s" -o "$o" &
  fi
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs
echo "built $(realpath $OUT)"
# the data-parallel exchange (include/vxm_comm.h): separate library, the only one that needs RCCL
if [ ! -f ../libvxm_comm.so ] || [ comm.cpp -nt ../libvxm_comm.so ] || [ ../../include/vxm_comm.h -nt ../libvxm_comm.so ]; then
  hipcc -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include comm.cpp -o ../libvxm_comm.so -L/opt/rocm/lib -lrccl
fi
echo "built $(realpath ../libvxm_comm.so)"

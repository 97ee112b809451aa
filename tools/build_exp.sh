#!/bin/bash
# Developer build: libvxm_hip_exp.so = the product library with conv_s3.hip compiled with -DVXM_S3_EXP (timing experiments selected by
# VXM_S3_DBG at run time; see S3_DBG in csrc/conv_s3.hip).  Never loaded by the package: tools/s3_bench.py --lib picks it up.
set -euo pipefail
cd "$(dirname "$0")/../voxelmorph_amd/csrc"
bash build.sh > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment"
hipcc $FLAGS -DVXM_S3_EXP -c conv_s3.hip -o build/conv_s3_exp.o &
hipcc $FLAGS -DVXM_S3_EXP -c conv_s3u.hip -o build/conv_s3u_exp.o &
wait
objs=""
for s in api warp planar conv_fwd conv_bwd_weight conv_bf16 pool losses diag; do objs="$objs build/$s.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvxm_hip_exp.so $objs build/conv_s3_exp.o build/conv_s3u_exp.o
echo "built $(realpath ../libvxm_hip_exp.so)"

#!/bin/bash
# Register / LDS / scratch use of every kernel in one .hip file of csrc (cross-compiles the device code to ISA, no GPU needed).
# usage: tools/kernel_regs.sh conv_s3.hip [extra hipcc flags]
set -eu
SRC=$1; shift || true
cd "$(dirname "$0")/../voxelmorph_amd/csrc"
OUT=/tmp/isa/${SRC%.hip}.s
mkdir -p /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment --cuda-device-only -S "$SRC" -o "$OUT" "$@" 2>/dev/null
awk '$1==".amdhsa_kernel" {name=$2} $1==".amdhsa_next_free_vgpr" {v=$2} $1==".amdhsa_accum_offset" {a=$2} $1==".amdhsa_private_segment_fixed_size" {s=$2} $1==".amdhsa_group_segment_fixed_size" {l=$2} $1==".end_amdhsa_kernel" {printf "regs %-4s arch %-4s scratch %-5s lds %-7s %s\n", v, a, s, l, name}' "$OUT" | while read -r a b c d e f g h n; do echo "$a $b $c $d $e $f $g $h $(echo "$n" | c++filt | cut -c1-100)"; done

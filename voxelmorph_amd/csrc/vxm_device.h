// gfx950 device-side helpers shared by every kernel file: MFMA wrapper, wave reductions,
// the bit-exact source-coordinate helper and the dynamic-LDS declaration.
#ifndef VXM_DEVICE_H
#define VXM_DEVICE_H
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VXM_WAVE 64
// dynamic LDS, 16-byte aligned base (cdna guide G17: no static __shared__ in front of it)
#define VXM_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact fp32 (fmaf chain), 32-cycle issue.
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; holds D[row = 4*(l>>4)+r][col = l&15].
__device__ __forceinline__ f32x4 vxm_mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float vxm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double vxm_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Source coordinate of SpatialTransformer + grid_sample(align_corners=True), with the reference's
// fp32 operation order and one IEEE rounding per operation (no FMA contraction):
//   layers.py:32      loc = i + f
//   layers.py:37      c   = 2 * (loc / (S-1) - 0.5)
//   ATen unnormalize  x   = ((c + 1) / 2) * (S-1)
// Reproducing the round trip is what makes mode='nearest' bit-exact.
__device__ __forceinline__ float vxm_src_coord(int i, float f, int S) {
#pragma clang fp contract(off)
    const float sm1 = (float)(S - 1);
    float loc = (float)i + f;
    float q = loc / sm1;
    float c = 2.0f * (q - 0.5f);
    float h = (c + 1.0f) / 2.0f;
    return h * sm1;
}

__device__ __forceinline__ float vxm_lrelu_grad(float y, float slope) { return y > 0.0f ? 1.0f : slope; }

#endif

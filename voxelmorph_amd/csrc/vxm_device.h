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

// One axis of ATen's grid_sampler_3d corner construction: x0 = floor(x), weights (x0 + 1 - x) and (x - x0).
// The two corner indices are clamped into the volume (so that every gather address is valid) and the in-volume
// tests kept separately: padding_mode='zeros' is applied by zeroing the WEIGHT (forward) or the VALUE (backward)
// of an outside corner, which leaves the in-volume terms bit-identical and needs no branches.
struct AxisTaps { int i0, i1; float w0, w1; bool ok0, ok1; };
__device__ __forceinline__ AxisTaps axis_corners(float x, int S) {
    AxisTaps a;
    const float f = floorf(x);
    a.w1 = x - f; a.w0 = (f + 1.0f) - x;
    const int i = (int)fminf(fmaxf(f, -2.0f), (float)S);      // clamp before the int conversion: wild coordinates stay defined
    a.ok0 = (unsigned)i < (unsigned)S; a.ok1 = (unsigned)(i + 1) < (unsigned)S;
    a.i0 = min(max(i, 0), S - 1); a.i1 = min(max(i + 1, 0), S - 1);
    return a;
}
// ATen upsample_trilinear3d(align_corners=True) index/lambda: real = ratio*dst; i0 = (int)real;
// i1 = i0 + (i0 < in-1); l1 = real - i0; l0 = 1 - l1.
// No contraction: `r - i0` must subtract from the ROUNDED product, as ATen does -- fused into fma(ratio, dst, -i0) the weights came out ~1e-6 off
// the reference's at index ~30 (closer to fp64 than ATen's own, but not the reference's: 1.2e-5 on O(10) noise values, round 6).
__device__ __forceinline__ void lin_src(int dst, float ratio, int n_in, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)
    const float r = ratio * (float)dst;
    i0 = min((int)r, n_in - 1);
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = fminf(fmaxf(r - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

// Raw buffer descriptor over `bytes` bytes at p: loads / stores take 32-bit byte offsets (a lane part + a wave-uniform scalar
// part) instead of 64-bit pointer arithmetic per access, and a lane whose offset is beyond the range loads 0.0 / is not stored.
constexpr int VXM_OOB = (int)0x80000000;            // voffset beyond any num_records
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vxm_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float vxm_bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void vxm_bstore(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

__device__ __forceinline__ float vxm_lrelu_grad(float y, float slope) { return y > 0.0f ? 1.0f : slope; }

#endif

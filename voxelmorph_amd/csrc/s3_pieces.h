// Piece schemes of the split-fp32 convolution kernels (conv_s3.hip, conv_s3u.hip): how an fp32 operand becomes 16-bit pieces for the
// matrix pipe, which piece products are accumulated, and the power-of-two scaling of the fp16 scheme.
#ifndef VXM_S3_PIECES_H
#define VXM_S3_PIECES_H
#include <type_traits>
#include <utility>
#include "conv_common.h"

namespace {

typedef __bf16 s3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s3_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 s3_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s3_bf16x8, a), __builtin_bit_cast(s3_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned s3_pack2(float lo, float hi) {            // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, s3_bf16x2));
}
__device__ __forceinline__ float s3_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float s3_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// (x0, x1) -> the three packed bf16 pairs (h, m, l); both remainders are exact fp32 differences
__device__ __forceinline__ void s3_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#pragma clang fp contract(off)
    h = s3_pack2(x0, x1);
    const float r0 = x0 - s3_lo(h), r1 = x1 - s3_hi(h);
    m = s3_pack2(r0, r1);
    const float q0 = r0 - s3_lo(m), q1 = r1 - s3_hi(m);
    l = s3_pack2(q0, q1);
}

// ---- the second piece scheme (round 4): TWO fp16 pieces, three products ("f16x2").
// fp16 carries 11 significand bits against bf16's 8: x s = h + l up to 2^-22 |x s| with h = fp16(x s), l = fp16(x s - h), and a product
// needs only hh + hl + lh (the dropped l l is below 2^-22 |x w|) -- three MFMAs of the same shape and rate instead of six, two LDS
// planes per operand instead of three.  What fp16 lacks is exponent range (5 bits), so every staged tile is scaled by a power of two s
// (exact) that brings ITS largest magnitude into [2^14, 2^15): the high piece cannot overflow, and the low piece of a value v keeps
// full precision down to |v| = 2^-18 max and an absolute error of 2^-40 max below that (fp16 subnormals, which the MFMA and the
// conversion keep -- tools/probe/f16_mfma_probe.hip, measured on the MI355X).  Because the scale is per staged tile / chunk, an MFMA
// chain lives for one chunk and is folded into the fp32 running totals by the vector ALU with the inverse scale (exact power of two,
// round-to-nearest add) -- which also keeps the chains short (the matrix pipe's accumulation truncates, see k_s3_bwd_weight).
// Weights get one scale per packed operator (k_s3_wmax), undone in the epilogue.
typedef _Float16 s3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s3_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 s3_mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s3_f16x8, a), __builtin_bit_cast(s3_f16x8, b), c, 0, 0, 0);
}
// (x0, x1) * s -> packed fp16 pairs (h, l); s is a power of two (exact), the remainder an exact fp32 difference
__device__ __forceinline__ void s3_split2_f16(float x0, float x1, float s, unsigned& h, unsigned& l) {
#pragma clang fp contract(off)
    const float a0 = x0 * s, a1 = x1 * s;
    const s3_f16x2 hh = __builtin_convertvector((f32x2){a0, a1}, s3_f16x2);       // v_cvt_pk_f16_f32, round to nearest even
    const float r0 = a0 - (float)hh[0], r1 = a1 - (float)hh[1];
    const s3_f16x2 ll = __builtin_convertvector((f32x2){r0, r1}, s3_f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
// scale of a tile whose largest magnitude is mx (>= 0): s = 2^k with mx s in [2^14, 2^15), and its inverse.  Exponent field E of mx,
// k = 141 - E.  E is clamped from below at 15 (mx < 2^-112: everything is far below the range anyway); E = 255 (inf / nan) gives 2^-114.
__device__ __forceinline__ void s3_scale_of(float mx, float& s, float& inv) {
    int E = (int)(__float_as_uint(mx) >> 23) & 255;
    E = E < 15 ? 15 : E;
    s = __uint_as_float((unsigned)(268 - E) << 23);
    inv = __uint_as_float((unsigned)(E - 14) << 23);
}
// maximum of a NON-NEGATIVE value over the wave, returned in every lane.  Four DPP steps fold each row of 16 lanes (quad permutes, then
// row_half_mirror / row_mirror), three readlanes fetch the other rows' results: ~10 vector / scalar instructions.  (The first version
// used six __shfl_xor steps = six dependent ds_bpermute round trips, on the critical path between a stage's last MFMA and its barrier.)
__device__ __forceinline__ float s3_wave_max(float v) {
    auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));        // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));        // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));       // row_half_mirror
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));       // row_mirror: every lane of a row holds the row's maximum
    const int b = (int)__float_as_uint(v);
    const float r0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 0)), r1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 16));
    const float r2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 32)), r3 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// Largest FINITE magnitude of what a wave staged for one scale unit, in every lane.  each(f) calls f(value) for every element of the lane.
// Fast path: one v_max per element (NaN operands are ignored by the maximum, an Inf dominates it).  When the wave's maximum comes out as Inf --
// wave-uniform, rare: a diverged step -- the maximum is taken again over the finite elements only, so that the unit's power-of-two scale is that
// of its finite values and a non-finite element stays what it is (h = Inf / NaN, l = NaN): like the reference's convolution, it then poisons
// exactly the outputs whose 3 x 3 x 3 window contains it, instead of pushing the whole unit's scale to 2^-114 (round 6, late; round-5 verdict item 6).
template <class Each>
__device__ __forceinline__ float s3_unit_max(Each&& each) {
    float m = 0.0f;
    each([&](float v) __attribute__((always_inline)) { m = fmaxf(m, __builtin_fabsf(v)); });
    m = s3_wave_max(m);
    if (__builtin_expect((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(m)) >= 0x7f800000u, 0)) {
        m = 0.0f;
        each([&](float v) __attribute__((always_inline)) {
            const float a = __builtin_fabsf(v);
            m = fmaxf(m, a < __builtin_inff() ? a : 0.0f);
        });
        m = s3_wave_max(m);
    }
    return m;
}
// (pack kernels: the same filter per element, their cost does not matter)
__device__ __forceinline__ float s3_finite_mag(float v) {
    const float a = __builtin_fabsf(v);
    return a < __builtin_inff() ? a : 0.0f;
}
// piece scheme NP: 3 = bf16 (h, m, l), six products; 2 = fp16 (h, l), three products.  PA / PB: piece of the A / B operand of product t,
// small terms first.
template <bool FIRST, class T> __device__ __forceinline__ T& s3_sel(T& a, T& b) { if constexpr (FIRST) return a; else return b; }
template <int NP> struct S3P;
template <> struct S3P<3> {
    static constexpr int NPROD = 6;
    static constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};       // (m,m) (l,h) (h,l) (m,h) (h,m) (h,h)
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) { return s3_mfma(a, b, c); }
    static constexpr unsigned ONES = 0x3f803f80u;                                      // bf16 1.0 x 2
};
template <> struct S3P<2> {
    static constexpr int NPROD = 3;
    static constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                         // (l,h) (h,l) (h,h)
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) { return s3_mfma_f16(a, b, c); }
    static constexpr unsigned ONES = 0x3c003c00u;                                      // fp16 1.0 x 2
};

// the transposing LDS read of gfx950 (ds_read_b64_tr_b16): from 16 consecutive 32-byte rows [voxel][16 channels] a lane receives four
// consecutive voxels of one channel -- half of a K = 32 MFMA operand of a contraction over voxels (lane pattern: tools/probe/tr16_probe.hip)
__device__ __forceinline__ u32x2 s3_tr_read(const char* lds_base, int byte_off) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)(lds_base + byte_off));
    return __builtin_bit_cast(u32x2, v);
}

// f(integral_constant<int, 0>), f(integral_constant<int, 1>), ... in order: a fully unrolled loop whose index is a compile-time constant inside f
template <class F, int... I> __device__ __forceinline__ void s3_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

}  // namespace
#endif

// libvxm_hip.so: version / error-string entry points, and the two umbrella names of SURVEY.md section 8b (include/vxm_hip.h).
#include "vxm_common.h"

thread_local char vxm_err_buf[512] = "";

extern "C" {
int vxm_version(void) { return 501; }                       /* 0.5.1: pooling codes + fused pooling backward / first weight gradient, small-volume conv kernel (additive); 0.5.0: round 6 (vecint_bwd_ws, fused upsample + warp, weighted loss finishers); 0.4.0: round 5 (capturable Adam, range probe, bwd_data / workspace_bytes names); 0.3.0: split-fp32 convs */
const char* vxm_last_error_string(void) { return vxm_err_buf; }

/* convolution_backward w.r.t. the input under its SURVEY name: pack the transposed / flipped operator into the caller's scratch and run the
 * forward kernel on it (what voxelmorph_amd/torch/functional.py conv_bwd_data does with the two calls) */
int vxm_conv3d_k3_bwd_data(const float* dz, int Cout, int64_t dz_bstride, const float* w, int Cw_in, int ci_lo, int ci_n, float* wpacked_scratch,
                           float* gx, int64_t gx_bstride, const float* mask, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W,
                           void* stream) {
    VXM_REQUIRE(dz && w && wpacked_scratch && gx, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_bwd_data: null pointer");
    if (int e = vxm_conv3d_k3_pack_weights_range(w, wpacked_scratch, Cw_in, Cout, ci_lo, ci_n, 1, stream)) return e;
    return vxm_conv3d_k3_fwd(dz, Cout, dz_bstride, 0, nullptr, 0, 0, wpacked_scratch, nullptr, gx, gx_bstride, ci_n, 1.0f, mask, mask_bstride,
                             mask_slope, B, D, H, W, stream);
}

size_t vxm_workspace_bytes(int op, int Cin, int Cout, int B, int D, int H, int W) {
    switch (op) {
        case VXM_WS_CONV_BWD_WEIGHT: return vxm_conv3d_k3_bwd_weight_workspace_bytes(Cin, Cout, B, D, H, W);
        case VXM_WS_S3_BWD_WEIGHT: return vxm_conv3d_k3_s3_bwd_weight_workspace_bytes(Cin, Cout, B, D, H, W);
        case VXM_WS_S3U_BWD_WEIGHT: return vxm_conv3d_k3_s3u_bwd_weight_workspace_bytes(Cin, Cout, B, D, H, W);
        case VXM_WS_BF16_BWD_WEIGHT: return vxm_bf16_conv_bwd_weight_workspace_bytes(Cin, Cout, B, D, H, W);
        case VXM_WS_CONV_BWD_DATA: return sizeof(float) * vxm_conv3d_k3_packed_elems(Cout, Cin);            /* the adjoint operator: Cout -> Cin */
        case VXM_WS_VECINT_BWD:         /* two gradient buffers + the per-step statistics + the per-tile (4 x 8 x 32) displacement records of up to 30 steps */
            return sizeof(float) * (2 * (size_t)B * 3 * D * H * W + VXM_VECINT_WORK_EXTRA +
                                    30 * (size_t)B * ((size_t)((D + 3) / 4) * ((H + 7) / 8) * ((W + 31) / 32)));
        default: return 0;
    }
}
}

// Host-side helpers of libvxm_hip.so: status/error convention of include/vxm_hip.h.
#ifndef VXM_COMMON_H
#define VXM_COMMON_H
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/vxm_hip.h"

extern thread_local char vxm_err_buf[512];

static inline int vxm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(vxm_err_buf, sizeof(vxm_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

static inline int vxm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vxm_fail(VXM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return VXM_OK;
}

#define VXM_REQUIRE(cond, code, ...) do { if (!(cond)) return vxm_fail(code, __VA_ARGS__); } while (0)
#define VXM_STREAM(s) reinterpret_cast<hipStream_t>(s)

static inline unsigned vxm_blocks(long long n, int per) { return (unsigned)((n + per - 1) / per); }
#endif

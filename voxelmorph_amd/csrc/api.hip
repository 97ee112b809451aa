// libvxm_hip.so: version / error-string entry points (include/vxm_hip.h).
#include "vxm_common.h"

thread_local char vxm_err_buf[512] = "";

extern "C" {
int vxm_version(void) { return 300; }                       /* 0.3.0: round 3 (bf16 entry points: 0.2.0; split-fp32 convs: 0.3.0) */
const char* vxm_last_error_string(void) { return vxm_err_buf; }
}

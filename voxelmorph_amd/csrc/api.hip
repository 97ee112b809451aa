// libvxm_hip.so: version / error-string entry points (include/vxm_hip.h).
#include "vxm_common.h"

thread_local char vxm_err_buf[512] = "";

extern "C" {
int vxm_version(void) { return 100; }                       /* 0.1.0: round 1 */
const char* vxm_last_error_string(void) { return vxm_err_buf; }
}

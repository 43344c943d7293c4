// libvc_hip.so: version / error plumbing shared by all entry points (include/vc_hip.h).
#include <cstdarg>
#include <cstdio>
#include "vc_common.h"

namespace vc {

static thread_local char g_err[512] = "";

char* last_error_buf() { return g_err; }

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace vc

extern "C" {

int vc_version(void) { return VC_ABI_VERSION; }
const char* vc_last_error(void) { return vc::last_error_buf(); }
const char* vc_target_arch(void) { return "gfx950"; }

}  // extern "C"

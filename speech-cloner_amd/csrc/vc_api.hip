// libvc_hip.so: version / error plumbing shared by all entry points (include/vc_hip.h).
#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <cstring>
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

static std::atomic<int> g_opt[OPT_COUNT];
static const struct { const char* name; Option id; bool ablate; } k_opts[] = {
    {"bank256", OPT_BANK256, false}, {"bank256_xcd", OPT_BANK256_XCD, false}, {"conv256", OPT_CONV256, false},
    {"conv256_min_k", OPT_CONV256_MINK, false}, {"conv256_wm", OPT_CONV256_WM, false}, {"proj256", OPT_PROJ256, false}, {"proj256_split", OPT_PROJ256_SPLIT, false},
    {"wgrad_xcd", OPT_WGRAD_XCD, false}, {"gru_mfma", OPT_GRU_MFMA, false}, {"gru_mfma4", OPT_GRU_MFMA4, false}, {"gru_small_mfma", OPT_GRU_SMALL_MFMA, false}, {"fe_fused", OPT_FE_FUSED, false}, {"fe_fused_spin", OPT_FE_FUSED_SPIN, false}, {"gru_train_resident", OPT_GRU_TRAIN_RESIDENT, false}, {"prenet_lds", OPT_PRENET_LDS, false}, {"cbhg_front_mi", OPT_CBHG_FRONT_MI, false}, {"gemm16_split", OPT_GEMM16_SPLIT, false}, {"f32_f16x3", OPT_F32_F16X3, false}, {"gru_f32_wide", OPT_GRU_F32_WIDE, false},
    {"ablate_bank256", OPT_ABLATE_BANK256, true}, {"ablate_bank256_only", OPT_ABLATE_BANK256_ONLY, true},
    {"ablate_cbhg_front", OPT_ABLATE_CBHG_FRONT, true},
};
static struct OptInit { OptInit() { for (auto& o : g_opt) o.store(-1); } } g_opt_init;

int opt(Option o) { return g_opt[o].load(std::memory_order_relaxed); }

}  // namespace vc

extern "C" {

int vc_set_option(const char* name, int value) {
    VC_REQUIRE(name, "vc_set_option: NULL name");
    for (const auto& o : vc::k_opts)
        if (!std::strcmp(o.name, name)) {
#ifndef VC_ABLATE
            VC_REQUIRE(!o.ablate, "vc_set_option: %s exists only in -DVC_ABLATE builds (tools/build_ablate.sh)", name);
#endif
            vc::g_opt[o.id].store(value);
            return VC_OK;
        }
    return vc::set_error(VC_ERR_INVALID, "vc_set_option: unknown option %s", name);
}

int vc_get_option(const char* name, int* value) {
    VC_REQUIRE(name && value, "vc_get_option: NULL argument");
    for (const auto& o : vc::k_opts)
        if (!std::strcmp(o.name, name)) { *value = vc::g_opt[o.id].load(); return VC_OK; }
    return vc::set_error(VC_ERR_INVALID, "vc_get_option: unknown option %s", name);
}

int vc_ablate_build(void) {
#ifdef VC_ABLATE
    return 1;
#else
    return 0;
#endif
}

int vc_version(void) { return VC_ABI_VERSION; }
const char* vc_last_error(void) { return vc::last_error_buf(); }
const char* vc_target_arch(void) { return "gfx950"; }

}  // extern "C"

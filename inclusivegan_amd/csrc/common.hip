// Error reporting and ABI version for libigan_hip.so.
#include "igan_common.h"
#include <cstring>

namespace igan {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace igan

extern "C" int igan_abi_version(void) { return IGAN_ABI_VERSION; }
extern "C" size_t igan_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(igan_upfirdn2d_params);
        case 1: return sizeof(igan_fused_bias_act_params);
        case 2: return sizeof(igan_conv2d_params);
        case 3: return sizeof(igan_conv2d_wgrad_params);
        case 4: return sizeof(igan_dense_params);
        case 5: return sizeof(igan_dense_wgrad_params);
        case 6: return sizeof(igan_taps_params);
        default: return 0;
    }
}
extern "C" const char* igan_last_error(void) { return igan::error_buffer(); }

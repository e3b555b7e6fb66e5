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
extern "C" const char* igan_last_error(void) { return igan::error_buffer(); }

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


// ---- device-side time stamps --------------------------------------------------------------------------------------
// One-wave kernels that read the constant 100 MHz counter (s_memrealtime).  Stream order puts a stamp after everything
// launched before it and before everything launched after it, so a pair of stamps brackets a kernel launch -- inside a
// captured hipGraph as well, where host-side event timing cannot reach.  The accumulate kernel (also capturable) folds the
// pairs of one replay into running sums, so that per-launch averages over many replays need no host work in between.
namespace {
__global__ void stamp_kernel(unsigned long long* slot) {
    if (threadIdx.x == 0) *slot = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_accumulate_kernel(const unsigned long long* stamps, unsigned long long* acc, int first, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) acc[first + i] += stamps[2 * (first + i) + 1] - stamps[2 * (first + i)];
}
}  // namespace

extern "C" int igan_stamp(igan_stream_t stream, unsigned long long* slot) {
    IGAN_REQUIRE(slot != nullptr, "stamp: null slot");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot);
    IGAN_LAUNCH_CHECK("stamp launch");
    return IGAN_OK;
}

extern "C" int igan_stamp_accumulate(igan_stream_t stream, const unsigned long long* stamps, unsigned long long* acc, int first, int count) {
    IGAN_REQUIRE(stamps && acc, "stamp_accumulate: null buffer");
    IGAN_REQUIRE(first >= 0 && count >= 1, "stamp_accumulate: bad range");
    hipLaunchKernelGGL(stamp_accumulate_kernel, dim3(igan::ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream, stamps, acc, first, count);
    IGAN_LAUNCH_CHECK("stamp_accumulate launch");
    return IGAN_OK;
}

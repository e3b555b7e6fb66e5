// Shared host-side helpers for libigan_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/igan_hip.h"

namespace igan {

// Thread-local message buffer behind igan_last_error(); the reference surfaces the
// same information through tensorflow::Status (upfirdn_2d.cu:20,228-229).
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return IGAN_OK;
    return fail(IGAN_ERR_HIP, "%s: %s", what, hipGetErrorName(e));
}

#define IGAN_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::igan::fail(IGAN_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

#define IGAN_LAUNCH_CHECK(what)                                   \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return ::igan::check_hip(e__, what); \
    } while (0)

#define IGAN_HIP_CHECK(call, what)                                \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) return ::igan::check_hip(e__, what); \
    } while (0)

// dense_small.hip: small-batch (M <= 32 rows) dense layers, dispatched from igan_conv2d / igan_conv2d_wgrad
bool dense_small_ok(int M, int K, int N, const void* x, const void* w, bool wt);
bool dense_small_fits(long long M, long long K, long long N, long long ldx);
int dense_small_rows(int M);
void dense_small_launch(hipStream_t stream, const igan_dense_params& a);
void dense_small(hipStream_t stream, const float* x, const float* w, float* y, int M, int K, int N, bool wt, float alpha);
bool dense_small_wgrad_ok(int M, int N, const void* dy, const void* dw);
void dense_small_wgrad(hipStream_t stream, const float* x, const float* dy, float* dw, int M, int K, int N, float alpha);

// thin_conv.hip: layers with <= 4 channels on one side (ToRGB, FromRGB, VGG conv1_1), dispatched from igan_conv2d / _wgrad
int thin_conv_kind(const igan_conv2d_params* p);
void thin_conv(hipStream_t stream, const igan_conv2d_params* p, int kind);
int thin_wgrad_kind(const igan_conv2d_wgrad_params* p);
size_t thin_wgrad_workspace(const igan_conv2d_wgrad_params* p, int kind);
void thin_wgrad(hipStream_t stream, const igan_conv2d_wgrad_params* p, int kind);

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace igan

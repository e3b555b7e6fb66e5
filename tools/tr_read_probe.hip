#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[16 * 128];
    for (int i = threadIdx.x; i < 16 * 128; i += 64) sm[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    lds_s16x4* ptr = (lds_s16x4*)(sm + (q + 4 * (g >> 1)) * 128 + 16 * (g & 1) + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int e = 0; e < 4; e++) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    std::vector<unsigned short> h(16 * 128), o(256);
    for (int r = 0; r < 16; r++) for (int c = 0; c < 128; c++) h[r * 128 + c] = r * 256 + c;
    unsigned short *d, *e;
    hipMalloc(&d, h.size() * 2); hipMalloc(&e, 512);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o.data(), e, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        const int g = lane >> 4, i = lane & 15;
        for (int el = 0; el < 4; el++) {
            const int want = (4 * (g >> 1) + el) * 256 + 16 * (g & 1) + i;
            if (o[lane * 4 + el] != want) { if (bad < 8) printf("lane %d el %d got r%d c%d want r%d c%d\n", lane, el, o[lane*4+el] >> 8, o[lane*4+el] & 255, want >> 8, want & 255); bad++; }
        }
    }
    printf("tr read: %d mismatches\n", bad);
    return 0;
}

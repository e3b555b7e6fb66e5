// How does v_mfma_f32_32x32x16_bf16 round?  One wave, one instruction, crafted operands: every output element is
// c + sum_k a[k] * b[k] with bf16 a, b and fp32 c; the exact value is known, so the result shows whether the 16 products and the
// addend are summed exactly and rounded once (round to nearest even), or aligned to the largest term and cut.  The exact-f32
// instruction v_mfma_f32_32x32x2_f32 runs the same sums (2 products per instruction, 8 chained) for comparison.
//   build: hipcc -O2 --offload-arch=gfx950 tools/mfma_round_probe.hip -o tools/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Case { float a[16], b[16], c; };

__global__ void probe(const Case* cs, float* out_bf, float* out_f32, int n) {
    const int lane = threadIdx.x, h = lane >> 5;
    for (int t = 0; t < n; t++) {
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; i++) { a[i] = (__bf16)cs[t].a[8 * h + i]; b[i] = (__bf16)cs[t].b[8 * h + i]; }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = cs[t].c;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (lane == 0) out_bf[t] = acc[0];
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; r++) acc2[r] = cs[t].c;
#pragma unroll
        for (int s = 0; s < 8; s++)      // k = 2 s + h
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(cs[t].a[2 * s + h], cs[t].b[2 * s + h], acc2, 0, 0, 0);
        if (lane == 0) out_f32[t] = acc2[0];
    }
}

int main() {
    std::vector<Case> cs;
    std::vector<const char*> names;
    auto blank = [] { Case c; memset(&c, 0, sizeof c); return c; };
    // 1/2/6: one small product x = j * 2^-26 next to a unit term (the addend c = +-1, or a product 1 * 1)
    for (int mode = 0; mode < 3; mode++)
        for (int j = -12; j <= 12; j++) {
            Case c = blank();
            c.a[1] = (float)j; c.b[1] = ldexpf(1.f, -26);
            if (mode == 0) c.c = 1.f;
            if (mode == 1) { c.a[0] = 1.f; c.b[0] = 1.f; }
            if (mode == 2) c.c = -1.f;
            cs.push_back(c);
            names.push_back(mode == 0 ? "c=+1, one product j*2^-26" : mode == 1 ? "c=0, products 1 and j*2^-26" : "c=-1, one product j*2^-26");
        }
    // 1b: the large product negative
    for (int j = -12; j <= 12; j++) {
        Case c = blank();
        c.a[1] = (float)j; c.b[1] = ldexpf(1.f, -26); c.a[0] = -1.f; c.b[0] = 1.f;
        cs.push_back(c); names.push_back("c=0, products -1 and j*2^-26");
    }
    // 1c: the ADDEND is the small term: c = j * 2^-26 (and j * 2^-29: more bits below the window) next to one product +-1
    for (int mode = 0; mode < 4; mode++)
        for (int j = -12; j <= 12; j += (mode < 2 ? 1 : 3)) {
            Case c = blank();
            c.a[0] = (mode & 1) ? -1.f : 1.f; c.b[0] = 1.f;
            c.c = (float)j * ldexpf(1.f, mode < 2 ? -26 : -29);
            cs.push_back(c);
            names.push_back(mode == 0 ? "product +1, c = j*2^-26" : mode == 1 ? "product -1, c = j*2^-26" : mode == 2 ? "product +1, c = j*2^-29" : "product -1, c = j*2^-29");
        }
    // 3/4: many small products whose sum matters only if they are added before the cut
    for (int mode = 0; mode < 4; mode++) {
        Case c = blank();
        for (int k = 0; k < 16; k++) { c.a[k] = 1.f; c.b[k] = ldexpf(1.f, -27); }
        if (mode == 0) c.c = 1.f;                          // exact 1 + 2^-23
        if (mode == 1) { c.a[0] = 1.f; c.b[0] = 1.f; }     // exact 1 + 15 * 2^-27
        if (mode == 2) c.c = -1.f;                         // exact -1 + 2^-23
        if (mode == 3) { c.c = 1.f; for (int k = 0; k < 16; k++) c.b[k] = -ldexpf(1.f, -27); }   // exact 1 - 2^-23
        cs.push_back(c);
        names.push_back(mode == 0 ? "c=+1, 16 products 2^-27" : mode == 1 ? "product 1 + 15 products 2^-27" : mode == 2 ? "c=-1, 16 products 2^-27" : "c=+1, 16 products -2^-27");
    }
    // 5: how far below the largest term does a product still count?  c = 1, one product 2^-s, then a second product -2^-s... simpler:
    // c = 2^e large, products sum to exactly 1 + 2^-10: the result should be RN(2^e + 1 + 2^-10)
    for (int e = 20; e <= 32; e += 4) {
        Case c = blank();
        c.c = ldexpf(1.f, e);
        c.a[0] = 1.f; c.b[0] = 1.f; c.a[9] = 1.f; c.b[9] = ldexpf(1.f, -10);
        cs.push_back(c); names.push_back("c=2^e (e=20,24,28,32), products 1 and 2^-10");
    }
    const int n = (int)cs.size();
    Case* d; float *o1, *o2;
    CK(hipMalloc(&d, n * sizeof(Case))); CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
    CK(hipMemcpy(d, cs.data(), n * sizeof(Case), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o1, o2, n);
    CK(hipDeviceSynchronize());
    std::vector<float> r1(n), r2(n);
    CK(hipMemcpy(r1.data(), o1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, n * 4, hipMemcpyDeviceToHost));
    printf("%-44s %22s %16s %16s %16s   (differences from the exact sum in units of 2^-26)\n", "case", "exact", "RN(exact)-exact", "bf16 mfma", "f32 mfma chain");
    for (int t = 0; t < n; t++) {
        double ex = cs[t].c;
        for (int k = 0; k < 16; k++) ex += (double)cs[t].a[k] * (double)cs[t].b[k];
        const double u = ldexp(1.0, -26) * fmax(1.0, fabs(cs[t].c) > 2 ? fabs(cs[t].c) : 1.0);
        printf("%-44s %22.17g %16.3f %16.3f %16.3f\n", names[t], ex, ((double)(float)ex - ex) / u, ((double)r1[t] - ex) / u, ((double)r2[t] - ex) / u);
    }
    // statistical: random bf16 operands, c = 0 and c ~ N(0, 16): signed mean of (result - exact) over many trials, in units of the rms result
    {
        const int T = 4096;
        std::vector<Case> rc(T);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f; };
        auto nrm = [&]() { float u1 = rnd() + 1e-6f, u2 = rnd(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
        auto tobf = [&](float x) { unsigned u; memcpy(&u, &x, 4); u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; float y; memcpy(&y, &u, 4); return y; };
        for (int mode = 0; mode < 2; mode++) {
            for (int t = 0; t < T; t++) {
                for (int k = 0; k < 16; k++) { rc[t].a[k] = tobf(nrm()); rc[t].b[k] = tobf(nrm()); }
                rc[t].c = mode ? 16.f * nrm() : 0.f;
            }
            Case* dd; float *p1, *p2;
            CK(hipMalloc(&dd, T * sizeof(Case))); CK(hipMalloc(&p1, T * 4)); CK(hipMalloc(&p2, T * 4));
            CK(hipMemcpy(dd, rc.data(), T * sizeof(Case), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dd, p1, p2, T);
            CK(hipDeviceSynchronize());
            std::vector<float> q1(T), q2(T);
            CK(hipMemcpy(q1.data(), p1, T * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(q2.data(), p2, T * 4, hipMemcpyDeviceToHost));
            double m1 = 0, m2 = 0, v1 = 0, v2 = 0, rms = 0, mr = 0;
            for (int t = 0; t < T; t++) {
                double ex = rc[t].c;
                for (int k = 0; k < 16; k++) ex += (double)rc[t].a[k] * (double)rc[t].b[k];
                const double e1 = q1[t] - ex, e2 = q2[t] - ex, er = (double)(float)ex - ex;
                m1 += e1; m2 += e2; v1 += e1 * e1; v2 += e2 * e2; rms += ex * ex; mr += er * er;
            }
            rms = sqrt(rms / T);
            printf("random operands, c %s: bf16 mfma mean %+.3e rms %.3e | f32 chain mean %+.3e rms %.3e | one rounding rms %.3e   (units of the rms result; %d trials)\n",
                   mode ? "~ N(0,16^2)" : "= 0", m1 / T / rms, sqrt(v1 / T) / rms, m2 / T / rms, sqrt(v2 / T) / rms, sqrt(mr / T) / rms, T);
        }
    }
    return 0;
}

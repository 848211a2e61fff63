// Issue cost of vector instructions on one SIMD: cycles per wave-instruction of v_fma_f32, v_pk_fma_f32, v_exp_f32 and
// v_cvt_pk_f16_f32 (independent instructions, 8 accumulators each), with 1, 2, 3 or 4 waves per SIMD on every CU.
// Cycles from s_memtime around the loop (wave 0 of each workgroup), wall clock from s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(2))) float f2;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters, float seed) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float c = seed * 0.999f, d = seed * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) a[i] = __builtin_fmaf(a[i], c, d);
                else if constexpr (KIND == 1) p[i] = p[i] * f2{c, c} + f2{d, d};
                else if constexpr (KIND == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
                else { auto h = __builtin_amdgcn_cvt_pkrtz(a[i], p[i].x); a[i] += (float)h[0]; }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int iters = 20000;
    float* out; unsigned long long* st;
    hipMalloc(&out, 4 * 256 * 1024 * sizeof(float)); hipMalloc(&st, 2 * 4096 * sizeof(unsigned long long));
    unsigned long long* h = (unsigned long long*)malloc(2 * 4096 * sizeof(unsigned long long));
    const char* names[3] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32"};
    for (int kind = 0; kind < 3; ++kind)
        for (int wps = 1; wps <= 4; ++wps) {
            const int blocks = 256 * wps;            // 256-thread workgroups: one wave per SIMD each
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) k<0><<<blocks, 256>>>(out, st, iters, 1.0001f);
                if (kind == 1) k<1><<<blocks, 256>>>(out, st, iters, 1.0001f);
                if (kind == 2) k<2><<<blocks, 256>>>(out, st, iters, -0.5f);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, st, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
            cyc /= blocks; real /= blocks;
            const double n = (double)iters * 32;
            printf("%-14s %d waves/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD slot; clock %.2f GHz\n", names[kind], wps, cyc / n, cyc / n / wps,
                   cyc / (real * 10.0) );
        }
    return 0;
}

// Micro-benchmark: the memory side of the fused row kernel without its arithmetic.
// Persistent workgroups; a wavefront owns 32-row tiles, reads 13 lane-ordered float4 per lane and tile (416 B per row)
// and writes 256 B per row.
//   mode 0: read + write in the row kernel's store shape (per instruction: 32 rows x 32 B)
//   mode 1: read + write with whole-row stores (per instruction: 4 rows x 256 B)
//   mode 2: read only      mode 3: write only (shape of mode 0)      mode 4: write only (shape of mode 1)
// argv: rows (default 4096 * 809), workgroups per CU (default 3)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int PV = 13;

__global__ __launch_bounds__(256, 3) void stream_kernel(const float4* __restrict__ packed, float* __restrict__ out, int64_t n_rows, int mode) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int64_t tile0 = (int64_t)wave_global * 32; tile0 < n_rows; tile0 += (int64_t)n_waves * 32) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mode <= 2) {
            const float4* p = packed + (tile0 >> 5) * (PV * 64) + lane;
            float4 v[PV];
#pragma unroll
            for (int j = 0; j < PV; ++j) v[j] = p[64 * j];
#pragma unroll
            for (int j = 0; j < PV; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
        if (mode == 2) {
            if (acc.x == 12345.678f) out[lane] = acc.y;       // never true: keeps the loads alive
            continue;
        }
        if (mode == 0 || mode == 3) {
            const int64_t row = tile0 + c;
            if (row < n_rows) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(out + row * 64 + 32 * nt + 8 * g + 4 * h) = acc;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = tile0 + 4 * i + (lane >> 4);
                if (row < n_rows) *reinterpret_cast<float4*>(out + row * 64 + 4 * (lane & 15)) = acc;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : (int64_t)4096 * 809;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t tiles = (n_rows + 31) / 32;
    float4* packed;
    float* out;
    hipMalloc(&packed, (size_t)tiles * PV * 64 * sizeof(float4));
    hipMalloc(&out, (size_t)n_rows * 64 * sizeof(float));
    hipMemset(packed, 0, (size_t)tiles * PV * 64 * sizeof(float4));
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const double rd = (double)tiles * PV * 64 * 16, wr = (double)n_rows * 256;
    for (int mode = 0; mode < 5; ++mode) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(stream_kernel, dim3(256 * per_cu), dim3(256), 0, 0, packed, out, n_rows, mode);
        hipEventRecord(s);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(stream_kernel, dim3(256 * per_cu), dim3(256), 0, 0, packed, out, n_rows, mode);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        const double bytes = (mode <= 2 ? rd : 0.0) + (mode != 2 ? wr : 0.0);
        printf("mode %d: %.1f us per launch, %.0f GB/s\n", mode, ms * 100, bytes / (ms * 1e-4) / 1e9);
    }
    hipFree(packed);
    hipFree(out);
    return 0;
}

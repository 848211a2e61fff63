// Micro-benchmark: HBM store bandwidth of a 256x128-float tile epilogue, by store pattern.
//   pattern 0: dword stores, 32 lanes x 4 B per row segment (MFMA accumulator layout)
//   pattern 1: float4 stores, 32 lanes x 16 B = one 512-B tile row per half-wave
//   pattern 2: float4 stores of the 32x32 epilogue patches: 8 lanes x 16 B = a 128-B row segment, 8 rows per instruction
//   pattern 3: pattern 2 walked like the row-stationary GEMM: a workgroup owns 256 rows and visits all N/32 column tiles
// usage: store_pattern [N [M]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512) void store_kernel(float* Y, int M, int N, int tiles_n, int pattern) {
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, fr = lane & 31, fc = lane >> 5;
    if (pattern == 0) {
        for (int i = 0; i < 2; ++i)
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fc;
                for (int j = 0; j < 2; ++j) Y[(int64_t)m * N + n0 + wc * 64 + j * 32 + fr] = (float)(m + j);
            }
    } else if (pattern == 2) {
        for (int t = 0; t < 4; ++t)
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + wr * 64 + (t >> 1) * 32 + rr * 8 + (lane >> 3);
                *reinterpret_cast<float4*>(Y + (int64_t)m * N + n0 + wc * 64 + (t & 1) * 32 + (lane & 7) * 4) = make_float4((float)m, 1.f, 2.f, 3.f);
            }
    } else if (pattern == 3) {
        const int nt = N / 32;
        if (tile >= M / 256) return;
        const int jrot = (tile * 7) % nt;
        for (int j = 0; j < nt; ++j) {
            const int c = (j + jrot) % nt;
            for (int rr = 0; rr < 4; ++rr) {
                const int m = tile * 256 + wave * 32 + rr * 8 + (lane >> 3);
                *reinterpret_cast<float4*>(Y + (int64_t)m * N + c * 32 + (lane & 7) * 4) = make_float4((float)m, 1.f, 2.f, 3.f);
            }
        }
    } else {
        // wave w writes rows w*32 .. w*32+31 of the tile, two rows (2 x 512 B) per instruction
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + 2 * r + fc;
            *reinterpret_cast<float4*>(Y + (int64_t)m * N + n0 + 4 * fr) = make_float4((float)m, 1.f, 2.f, 3.f);
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, M = argc > 2 ? atoi(argv[2]) : 20480;
    float* Y;
    hipMalloc(&Y, (size_t)M * N * 4);
    const int tiles_n = N / 128, tiles = (M / 256) * tiles_n;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int pattern = 0; pattern < 4; ++pattern) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(store_kernel, dim3(tiles), dim3(512), 0, 0, Y, M, N, tiles_n, pattern);
        hipEventRecord(s);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(store_kernel, dim3(tiles), dim3(512), 0, 0, Y, M, N, tiles_n, pattern);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        printf("N=%d pattern %d: %.1f us per launch, %.0f GB/s\n", N, pattern, ms * 100, (double)M * N * 4 / (ms * 1e-4) / 1e9 / 1e0 * 1e-0 / 1.0);
    }
    hipMemset(Y, 0, 16);
    hipFree(Y);
    return 0;
}

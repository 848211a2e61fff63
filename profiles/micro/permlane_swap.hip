#include <hip/hip_runtime.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xor16_add(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__global__ void k(float* p) {
    float x = p[threadIdx.x];
    p[threadIdx.x] = xor32_add(xor16_add(x));
    p[64 + threadIdx.x] = x + __shfl_xor(x, 16, 64);
    p[128 + threadIdx.x] = x + __shfl_xor(x, 32, 64);
    p[192 + threadIdx.x] = xor16_add(x);
    p[256 + threadIdx.x] = xor32_add(x);
}
int main() {
    float* d; hipMalloc(&d, 320 * 4);
    float h[320];
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 97);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 320 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { if (h[64 + i] != h[192 + i]) ++bad; if (h[128 + i] != h[256 + i]) ++bad; }
    printf("mismatches %d\n", bad);
    return bad != 0;
}

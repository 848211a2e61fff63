#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float a, float b, float* out) {
    f16x8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)a; B[j] = (_Float16)b; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)A[0]; out[2] = (float)((_Float16)(a * 0.5f)); }
}
int main() {
    float* d; hipMalloc(&d, 16);
    float h[3];
    const float as[] = {3e-6f, 3e-5f, 1e-4f, 7e-8f};
    for (float a : as) {
        k<<<1, 64>>>(a, 1024.f, d); hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g: mfma sum(32 x a x 1024) = %g (exact %g)  cvt(a) = %g  cvt(a/2) = %g\n", a, h[0], 32.0 * a * 1024, h[1], h[2]);
    }
    return 0;
}

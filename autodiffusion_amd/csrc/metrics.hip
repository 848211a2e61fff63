// Per-scenario evaluation metrics of a batch of predicted trajectories against labels, in one pass on the device
// (the reference copies every trajectory to the host and runs NumPy: transformer4planning/trainer.py:184-305).
// One wavefront per scenario; lane p owns points p and p + 64.  Output row (18 floats):
//   0-3 ade at 1/3/5/8 s | 4-7 fde at 1/3/5/8 s | 8 heading error of point 0 | 9-11 mean heading error 3/5/8 s |
//   12-14 heading error at points 29/49/79 | 15-17 max displacement over 3/5/8 s
// Horizons are 10/30/50/80 points divided by index_rescale = 80 / S when S < 80 (trainer.py:191-194).
#include "adx_ops.h"

#include <math.h>

namespace adx {
namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wrap_angle(float a) { return atan2f(sinf(a), cosf(a)); }

__global__ __launch_bounds__(256) void metrics_kernel(const float* pred, const float* label, int B, int S, int C, float* out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int rs = S < 80 ? 80 / S : 1;
    const int h1 = 10 / rs, h3 = 30 / rs, h5 = 50 / rs, h8 = 80 / rs;
    const float* P = pred + (int64_t)b * S * C;
    const float* Lb = label + (int64_t)b * S * C;
    float s1 = 0.f, s3 = 0.f, s5 = 0.f, s8 = 0.f, m3 = 0.f, m5 = 0.f, m8 = 0.f, a3 = 0.f, a5 = 0.f, a8 = 0.f;
    float f1 = 0.f, f3 = 0.f, f5 = 0.f, f8 = 0.f, e0 = 0.f, g3 = 0.f, g5 = 0.f, g8 = 0.f;
    for (int p = lane; p < S; p += 64) {
        const float dx = P[p * C] - Lb[p * C], dy = P[p * C + 1] - Lb[p * C + 1];
        const float d = sqrtf(dx * dx + dy * dy);
        if (p < h1) s1 += d;
        if (p < h3) { s3 += d; m3 = fmaxf(m3, d); }
        if (p < h5) { s5 += d; m5 = fmaxf(m5, d); }
        if (p < h8) { s8 += d; m8 = fmaxf(m8, d); }
        if (p == 9 / rs) f1 = d;
        if (p == 29 / rs) f3 = d;
        if (p == 49 / rs) f5 = d;
        if (p == S - 1) f8 = d;
        if (C == 4) {
            const float he = fabsf(wrap_angle(wrap_angle(P[p * C + 3]) - wrap_angle(Lb[p * C + 3])));
            if (p < h3) a3 += he;
            if (p < h5) a5 += he;
            if (p < h8) a8 += he;
            if (p == 0) e0 = he;
            if (p == 29 / rs) g3 = he;
            if (p == 49 / rs) g5 = he;
            if (p == 79 / rs) g8 = he;
        }
    }
    float v[18] = {wsum(s1) / h1, wsum(s3) / h3, wsum(s5) / h5, wsum(s8) / h8, wsum(f1), wsum(f3), wsum(f5), wsum(f8), wsum(e0),
                   wsum(a3) / h3, wsum(a5) / h5, wsum(a8) / h8, wsum(g3), wsum(g5), wsum(g8), wmax(m3), wmax(m5), wmax(m8)};
    if (lane < 18) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i) r = lane == i ? v[i] : r;
        out[(int64_t)b * 18 + lane] = r;
    }
}

}  // namespace

hipError_t trajectory_metrics(const float* pred, const float* label, int B, int S, int C, float* out, hipStream_t stream) {
    if (!pred || !label || !out || B <= 0 || S <= 0 || S > 80 || (80 % S) != 0 || (C != 2 && C != 4)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(metrics_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, pred, label, B, S, C, out);
    return hipGetLastError();
}

}  // namespace adx

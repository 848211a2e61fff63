// Multi-modal reduction of sampled trajectories: farthest-point sampling of k modes on the end points, then EM
// iterations of a diagonal Gaussian mixture over whole trajectories (what the reference runs on the CPU after
// multi-sample generation: transformer4planning/models/utils.py:120-132, utils/modify_traj_utils.py:8-30,124-219).
//
// One 256-thread workgroup per scenario; the k modes (mean, and the two per-dimension constants of the Gaussian
// log-density) and the M x k responsibility matrix live in LDS, the M samples are re-read from L2.
//   E step: logn[i][o] = sum_d ( -log(2 pi s_od) / 2 - (x_id - mu_od)^2 / (2 s_od) ), s = sigma + 1e-5; one wavefront per
//           (sample, mode) pair, lanes over d; responsibilities w = q n / sum_o q n with the row maximum subtracted.
//   M step: q_o = sum_i w_io / M; mu_od = sum_i w_io x_id / (M q_o); sigma_od = sum_i w_io (0.1 + (x_id - mu_od)^2) / (M q_o).
#include "adx_ops.h"

#include <math.h>

namespace adx {
namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void select_modes_kernel(const SelectModesArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = a.L * a.D, M = a.M, k = a.k;
    float* mean = sm;                       // [LD]
    float* mu = mean + LD;                  // [k][LD]
    float* c1 = mu + k * LD;                // [k][LD]  -log(2 pi s) / 2
    float* c2 = c1 + k * LD;                // [k][LD]  1 / (2 s)
    float* w = c2 + k * LD;                 // [M][k]   log-densities, then responsibilities / M
    float* q = w + M * k;                   // [k]
    float* dist = q + k;                    // [M]
    int* idx = reinterpret_cast<int*>(dist + M);   // [k]
    __shared__ int far;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = a.traj + (int64_t)blockIdx.x * M * LD;

    for (int d = tid; d < LD; d += 256) {
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += X[(int64_t)i * LD + d];
        mean[d] = s / (float)M;
    }
    for (int i = tid; i < M; i += 256) dist[i] = 1e10f;
    if (tid == 0) far = 0;
    __syncthreads();
    // ---- farthest-point sampling on the end points (the last D values of a trajectory)
    const float* E = X + (LD - a.D);
    for (int t = 0; t < k; ++t) {
        const int f = far;
        if (tid == 0) idx[t] = f;
        for (int i = tid; i < M; i += 256) {
            float dd = 0.f;
            for (int c = 0; c < a.D; ++c) { const float u = E[(int64_t)i * LD + c] - E[(int64_t)f * LD + c]; dd += u * u; }
            dist[i] = fminf(dist[i], dd);
        }
        __syncthreads();
        if (tid == 0) {
            int best = 0;
            float bv = dist[0];
            for (int i = 1; i < M; ++i) if (dist[i] > bv) { bv = dist[i]; best = i; }      // first maximum
            far = best;
        }
        __syncthreads();
    }
    // ---- initial modes
    const float s0 = a.init_sigma + 1e-5f;
    for (int e = tid; e < k * LD; e += 256) {
        const int o = e / LD, d = e % LD;
        mu[e] = X[(int64_t)idx[o] * LD + d] - mean[d];
        c1[e] = -0.5f * logf(2.0f * 3.14159265358979323846f * s0);
        c2[e] = 1.0f / (2.0f * s0);
    }
    if (tid < k) q[tid] = 1.0f / (float)k;
    __syncthreads();
    const float inv_m = 1.0f / (float)M;
    for (int it = 0; it < a.em_iter; ++it) {
        // E step
        for (int p = wave; p < M * k; p += 4) {
            const int i = p / k, o = p % k;
            float s = 0.f;
            for (int d = lane; d < LD; d += 64) {
                const float u = X[(int64_t)i * LD + d] - mean[d] - mu[o * LD + d];
                s += c1[o * LD + d] - u * u * c2[o * LD + d];
            }
            s = wave_sum_f(s);
            if (lane == 0) w[p] = s;
        }
        __syncthreads();
        for (int i = tid; i < M; i += 256) {
            float mx = -INFINITY;
            for (int o = 0; o < k; ++o) mx = fmaxf(mx, w[i * k + o]);
            float tot = 0.f;
            for (int o = 0; o < k; ++o) { const float v = q[o] * expf(w[i * k + o] - mx); w[i * k + o] = v; tot += v; }
            const float r = inv_m / tot;
            for (int o = 0; o < k; ++o) w[i * k + o] *= r;
        }
        __syncthreads();
        // M step
        if (tid < k) {
            float s = 0.f;
            for (int i = 0; i < M; ++i) s += w[i * k + tid];
            q[tid] = s;
        }
        __syncthreads();
        for (int e = tid; e < k * LD; e += 256) {
            const int o = e / LD, d = e % LD;
            const float den = 1.0f / (q[o] + 1e-10f);
            float s = 0.f;
            for (int i = 0; i < M; ++i) s += w[i * k + o] * (X[(int64_t)i * LD + d] - mean[d]);
            const float nm = s * den;
            float v = 0.f;
            for (int i = 0; i < M; ++i) { const float u = X[(int64_t)i * LD + d] - mean[d] - nm; v += w[i * k + o] * (a.org_sigma + u * u); }
            const float sg = v * den + 1e-5f;
            mu[e] = nm;
            c1[e] = -0.5f * logf(2.0f * 3.14159265358979323846f * sg);
            c2[e] = 1.0f / (2.0f * sg);
        }
        __syncthreads();
    }
    // ---- order by weight, descending (ties keep the sampling order)
    if (tid == 0) {
        for (int o = 0; o < k; ++o) idx[o] = o;
        for (int i = 1; i < k; ++i) {
            const int v = idx[i];
            int j = i - 1;
            while (j >= 0 && q[idx[j]] < q[v]) { idx[j + 1] = idx[j]; --j; }
            idx[j + 1] = v;
        }
    }
    __syncthreads();
    float* out = a.modes + (int64_t)blockIdx.x * k * LD;
    for (int e = tid; e < k * LD; e += 256) {
        const int r = e / LD, d = e % LD;
        out[e] = mu[idx[r] * LD + d] + mean[d];
    }
    if (tid < k) a.weights[(int64_t)blockIdx.x * k + tid] = q[idx[tid]];
}

}  // namespace

hipError_t select_modes(const SelectModesArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.M <= 0 || a.L <= 0 || a.D <= 0 || a.k <= 0 || a.k > a.M || a.k > 64 || a.M > 4096 || a.em_iter < 0 || !a.traj ||
        !a.modes || !a.weights)
        return hipErrorInvalidValue;
    const size_t LD = (size_t)a.L * a.D;
    const size_t smem = sizeof(float) * (LD + 3 * (size_t)a.k * LD + (size_t)a.M * a.k + a.k + a.M) + sizeof(int) * a.k;
    if (smem > 150 * 1024) return hipErrorInvalidValue;
    static size_t attr_set = 0;
    if (smem > attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&select_modes_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set = smem;
    }
    hipLaunchKernelGGL(select_modes_kernel, dim3(a.B), dim3(256), smem, stream, a);
    return hipGetLastError();
}

}  // namespace adx

// Elementwise / reduction kernels around the denoiser that are not part of a sampling step:
//   * device-side Gaussian noise for callers that pass none (counter-based Philox4x32-10, keyed by the GLOBAL batch row,
//     so a shard of the batch draws exactly what the same rows draw in a single-GPU run);
//   * the forward-diffusion sample q(x_t | x_0) and the min-SNR weighted squared-error of the evaluation-time loss
//     (TrajRefiner.py:408-452 in the reference).
// All HBM-bound, one pass over their data.
#include "adx_ops.h"

#include <math.h>

namespace adx {
namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// Philox4x32 with 10 rounds (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter c, key k -> 4 words.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Two words -> two independent N(0,1) values (Box-Muller on u1 in (0,1], u2 in [0,1), 24-bit mantissas).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-8f;      // 2^-24
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    z0 = r * c; z1 = r * s;
}

// out[lead][B][row_len]: the value of element e of batch row b under leading index l depends only on
// (seed, row0 + b, l, e / 4): the counter is {row lo, row hi, l, e / 4}, the key is the seed.
__global__ __launch_bounds__(256) void randn_rows_kernel(const uint64_t* ctl, float* out, int n_lead, int B, int row_len) {
    const int quads = (row_len + 3) >> 2;
    const int64_t total = (int64_t)n_lead * B * quads;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % quads);
    const int b = (int)((i / quads) % B);
    const int l = (int)(i / ((int64_t)quads * B));
    const uint64_t seed = ctl[0], grow = ctl[1] + (uint64_t)b;
    uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)l, (uint32_t)q};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z[4];
    box_muller(c[0], c[1], z[0], z[1]);
    box_muller(c[2], c[3], z[2], z[3]);
    float* dst = out + ((int64_t)l * B + b) * row_len + 4 * q;
    if (4 * q + 4 <= row_len && (row_len & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
        for (int e = 0; e < 4 && 4 * q + e < row_len; ++e) dst[e] = z[e];
    }
}

__global__ void set_noise_ctl_kernel(uint64_t* ctl, uint64_t seed, uint64_t row0) {
    if (threadIdx.x == 0) { ctl[0] = seed; ctl[1] = row0; }
}

// x_t = sqrt(acp[t]) x0 + sqrt(1 - acp[t]) noise, v = sqrt(acp[t]) noise - sqrt(1 - acp[t]) x0 (optional); t per batch row.
__global__ __launch_bounds__(256) void q_sample_kernel(const float* x0, const float* noise, const int64_t* t, const float* sa,
                                                        const float* s1, float* xt, float* v, int64_t total, int row_len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t tb = t[i / row_len];
    const float a = sa[tb], b = s1[tb];
    const float x = x0[i], n = noise[i];
    xt[i] = a * x + b * n;
    if (v) v[i] = a * n - b * x;
}

// the same with four elements per lane and access (row_len a multiple of 4, 16-byte aligned buffers)
__global__ __launch_bounds__(256) void q_sample4_kernel(const float4* __restrict__ x0, const float4* __restrict__ noise, const int64_t* __restrict__ t,
                                                         const float* __restrict__ sa, const float* __restrict__ s1, float4* __restrict__ xt,
                                                         float4* __restrict__ v, int64_t total4, int row_len4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int64_t tb = t[i / row_len4];
    const float a = sa[tb], b = s1[tb];
    const float4 x = x0[i], n = noise[i];
    xt[i] = make_float4(a * x.x + b * n.x, a * x.y + b * n.y, a * x.z + b * n.z, a * x.w + b * n.w);
    if (v) v[i] = make_float4(a * n.x - b * x.x, a * n.y - b * x.y, a * n.z - b * x.z, a * n.w - b * x.w);
}

// loss_rows[b] = weight[t[b]] * mean_e (out[b][e] - target[b][e])^2: one wavefront per row, lanes stride the row, the
// lane sums are combined in a fixed butterfly order (the result does not depend on the launch shape).
__global__ __launch_bounds__(256) void weighted_mse_kernel(const float* out, const float* target, const int64_t* t, const float* weight,
                                                            float* loss_rows, int n_rows, int row_len) {
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const float* a = out + (int64_t)row * row_len;
    const float* b = target + (int64_t)row * row_len;
    float s = 0.f;
    for (int e = lane; e < row_len; e += 64) { const float d = a[e] - b[e]; s = fmaf(d, d, s); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) loss_rows[row] = s / (float)row_len * weight[t[row]];
}

// The default branch of TrajectoryDecoder.compute_traj_loss (models/decoder/base.py:9-35, 238-241): positions -- the first two
// channels -- under the squared error or SmoothL1 (beta = 1); heading -- the last channel -- under the squared angular difference,
// every angle and the difference wrapped into (-pi, pi] by atan2(sin, cos).  One wavefront per batch row; part[b] = {sum of the
// position terms, sum of the heading terms}, lanes combined in a fixed butterfly order.
__device__ __forceinline__ float wrap_angle(float a) { return atan2f(sinf(a), cosf(a)); }

__global__ __launch_bounds__(256) void traj_loss_rows_kernel(const float* logits, const float* label, float* part, int B, int T, int C, int Cl,
                                                              int smooth_l1) {
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* a = logits + (int64_t)row * T * C;
    const float* b = label + (int64_t)row * T * Cl;
    float sxy = 0.f, syaw = 0.f;
    for (int t = lane; t < T; t += 64) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float d = a[t * C + c] - b[t * Cl + c], ad = fabsf(d);
            sxy += smooth_l1 ? (ad < 1.f ? 0.5f * d * d : ad - 0.5f) : d * d;
        }
        const float d = wrap_angle(wrap_angle(b[t * Cl + Cl - 1]) - wrap_angle(a[t * C + C - 1]));
        syaw = fmaf(d, d, syaw);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sxy += __shfl_xor(sxy, o, 64); syaw += __shfl_xor(syaw, o, 64); }
    if (lane == 0) { part[2 * row] = sxy; part[2 * row + 1] = syaw; }
}

// out[0] = mean position term, out[1] = mean heading term: one workgroup, fixed summation order
__global__ __launch_bounds__(256) void traj_loss_mean_kernel(const float* part, int B, int T, float* out) {
    __shared__ float px[256], py[256];
    float sx = 0.f, sy = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) { sx += part[2 * i]; sy += part[2 * i + 1]; }
    px[threadIdx.x] = sx; py[threadIdx.x] = sy;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { px[threadIdx.x] += px[threadIdx.x + o]; py[threadIdx.x] += py[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = px[0] / ((float)B * T * 2.f); out[1] = py[0] / ((float)B * T); }
}

// mean of n values, one workgroup, fixed summation order
__global__ __launch_bounds__(256) void mean_kernel(const float* x, int n, float* out) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / (float)n;
}

}  // namespace

hipError_t set_noise_ctl(uint64_t* ctl_dev, uint64_t seed, uint64_t row0, hipStream_t stream) {
    hipLaunchKernelGGL(set_noise_ctl_kernel, dim3(1), dim3(64), 0, stream, ctl_dev, seed, row0);
    return hipGetLastError();
}

hipError_t randn_rows(const uint64_t* ctl_dev, float* out, int n_lead, int B, int row_len, hipStream_t stream) {
    if (n_lead <= 0 || B <= 0 || row_len <= 0) return hipSuccess;
    const int64_t total = (int64_t)n_lead * B * ((row_len + 3) >> 2);
    hipLaunchKernelGGL(randn_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ctl_dev, out, n_lead, B, row_len);
    return hipGetLastError();
}

hipError_t q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, float* xt,
                    float* v, int n_rows, int row_len, hipStream_t stream) {
    const int64_t total = (int64_t)n_rows * row_len;
    if (total <= 0) return hipSuccess;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((row_len & 3) == 0 && al(x0) && al(noise) && al(xt) && al(v))
        hipLaunchKernelGGL(q_sample4_kernel, dim3((unsigned)(((total >> 2) + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const float4*>(x0),
                           reinterpret_cast<const float4*>(noise), t, sqrt_acp, sqrt_1m_acp, reinterpret_cast<float4*>(xt), reinterpret_cast<float4*>(v),
                           total >> 2, row_len >> 2);
    else
        hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x0, noise, t, sqrt_acp, sqrt_1m_acp, xt, v,
                           total, row_len);
    return hipGetLastError();
}

hipError_t weighted_mse(const float* out, const float* target, const int64_t* t, const float* weight, float* loss_rows, float* loss_mean,
                        int n_rows, int row_len, hipStream_t stream) {
    if (n_rows <= 0 || row_len <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(weighted_mse_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, stream, out, target, t, weight, loss_rows, n_rows,
                       row_len);
    if (loss_mean) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, loss_rows, n_rows, loss_mean);
    return hipGetLastError();
}

hipError_t traj_loss(const float* logits, const float* label, float* part, float* out, int B, int T, int C, int Cl, int smooth_l1,
                     hipStream_t stream) {
    if (B <= 0 || T <= 0 || C < 2 || Cl < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(traj_loss_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, logits, label, part, B, T, C, Cl, smooth_l1);
    hipLaunchKernelGGL(traj_loss_mean_kernel, dim3(1), dim3(256), 0, stream, part, B, T, out);
    return hipGetLastError();
}

}  // namespace adx

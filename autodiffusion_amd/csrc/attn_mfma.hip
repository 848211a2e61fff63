// Attention for short sequences (Skv <= 128) and head dimensions that are multiples of 16, on the fp32 matrix cores.
//
// One wavefront per (batch item, head, 16 queries).  Everything is a 16x16x4 fp32 MFMA (exact fp32 products):
//   scores  S[q][key]  = sum_d Q[q][d] K[key][d]     A = Q fragment, B = K fragment, contraction over the head dim
//   output  O[q][c]    = sum_key P[q][key] V[key][c] A = P fragment, B = V fragment, contraction over the keys
// Q and K fragments are read from global memory as float4 along the head dimension: one load feeds four MFMAs, the
// j-th of which contracts dims {16c + 4k + j : k = 0..3} on both operands.  The accumulator of the score MFMAs holds
// (query = 4 (lane >> 4) + reg, key = lane & 15), so a query's softmax is a reduction over registers (key tiles) and
// over the 16 lanes of a row; the probabilities go through a 16 x Skv LDS patch to come back in A-fragment order.
// The softmax scale is folded into exp2; the output is normalised in the accumulator before the store.
#include "adx_ops.h"
#include "fastmath.h"

#include <math.h>

namespace adx {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int MAX_KT = 8;             // key tiles of 16: Skv <= 128
constexpr int P_LD = 132;             // floats per row of the probability patch (16-byte aligned, staggers banks)
constexpr float LOG2E_A = 1.4426950408889634f;

template <int DV_TILES>   // output tiles of 16 columns handled per pass over the keys
__global__ __launch_bounds__(64) void attn_mfma_kernel(const AttentionArgs a) {
    __shared__ __attribute__((aligned(16))) float Ps[16 * P_LD];
    const int lane = threadIdx.x, r16 = lane & 15, k4 = lane >> 4;
    const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int Skv = a.Skv, n_kt = (Skv + 15) / 16;
    const float* Qb = a.Q + (int64_t)b * a.q_batch_stride + (int64_t)h * a.dh;
    const float* Kb = a.K + (int64_t)b * a.k_batch_stride + (int64_t)h * a.dh;
    const float* Vb = a.V + (int64_t)b * a.v_batch_stride + (int64_t)h * a.dv;

    // ---- scores: rows clamp to the last valid row (their results are masked / never stored)
    f32x4 s[MAX_KT];
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* qrow = Qb + (int64_t)min(q0 + r16, a.Sq - 1) * a.ldq + 4 * k4;
    const float* krow[MAX_KT];
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) krow[t] = Kb + (int64_t)min(16 * t + r16, Skv - 1) * a.ldk + 4 * k4;
    for (int c = 0; c < a.dh; c += 16) {
        const float4 qv = *reinterpret_cast<const float4*>(qrow + c);
#pragma unroll
        for (int t = 0; t < MAX_KT; ++t) {
            if (t < n_kt) {
                const float4 kv = *reinterpret_cast<const float4*>(krow[t] + c);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kv.x, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kv.y, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kv.z, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kv.w, s[t], 0, 0, 0);
            }
        }
    }
    // ---- softmax per query (= per register index and 16-lane row): max, exp2, sum
    const float f = a.scale * LOG2E_A;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) {
        if (t < n_kt) {
            const bool ok = 16 * t + r16 < Skv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = ok ? s[t][r] * f : -INFINITY;
                mx[r] = fmaxf(mx[r], s[t][r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) {
        if (t < n_kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2_fast(s[t][r] - mx[r]);
                sum[r] += p;
                Ps[(4 * k4 + r) * P_LD + 16 * t + r16] = p;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
        sum[r] = rcp_fast(sum[r]);
    }
    __syncthreads();
    // ---- output: DV_TILES column tiles per pass over the keys
    const int n_k4 = n_kt * 4;                       // key groups of 4 (probabilities of padded keys are 0)
    for (int c0 = 0; c0 < a.dv; c0 += 16 * DV_TILES) {
        f32x4 o[DV_TILES];
#pragma unroll
        for (int u = 0; u < DV_TILES; ++u) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        // two key tiles (32 keys) per iteration: all their V loads are issued before the first MFMA needs one
#pragma unroll 1
        for (int g = 0; g < n_k4; g += 8) {
            float vv[2][4][DV_TILES];
            float4 pv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ge = min(g + 4 * e, n_k4 - 4);                 // a missing second tile re-reads the first; its p is zeroed
                pv[e] = *reinterpret_cast<const float4*>(&Ps[r16 * P_LD + 4 * ge + 4 * k4]);
                if (g + 4 * e >= n_k4) pv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = min(4 * ge + 4 * k4 + j, Skv - 1);
                    const float* vrow = Vb + (int64_t)key * a.ldv + c0 + r16;
#pragma unroll
                    for (int u = 0; u < DV_TILES; ++u) vv[e][j][u] = (c0 + 16 * u < a.dv) ? vrow[16 * u] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float pj[4] = {pv[e].x, pv[e].y, pv[e].z, pv[e].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < DV_TILES; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(pj[j], vv[e][j][u], o[u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < DV_TILES; ++u) {
            if (c0 + 16 * u >= a.dv) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 4 * k4 + r;
                if (q < a.Sq) a.O[(int64_t)b * a.o_batch_stride + (int64_t)q * a.ldo + (int64_t)h * a.dv + c0 + 16 * u + r16] = o[u][r] * sum[r];
            }
        }
    }
}

}  // namespace

bool attention_mfma_ok(const AttentionArgs& a) {
    return a.mask_off < 0 && !a.state && !a.key0 && a.Skv <= 16 * MAX_KT && a.dh >= 16 && (a.dh % 16) == 0 && (a.dv % 16) == 0 &&
           (a.ldq % 4) == 0 && (a.ldk % 4) == 0 && (a.q_batch_stride % 4) == 0 && (a.k_batch_stride % 4) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Q) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.K) & 15) == 0;
}

hipError_t attention_mfma(const AttentionArgs& a, hipStream_t stream) {
    if (!attention_mfma_ok(a)) return hipErrorInvalidValue;
    dim3 grid((a.Sq + 15) / 16, a.H, a.B);
    if (a.dv >= 64) hipLaunchKernelGGL(attn_mfma_kernel<4>, grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(attn_mfma_kernel<2>, grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace adx

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
#include <stdlib.h>

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

// ============================================================================================================
// Long or masked sequences (the causal passes of the mixture-of-experts decoder: 788 .. 873 keys, and its one-token passes over
// the cached keys): the kernel above with the keys walked in chunks of 64 and an online softmax.  One wavefront per
// (batch item, head, 16 queries); per chunk: four score tiles (fp32 16x16x4 MFMAs), mask, running maximum m and the factor
// alpha = 2^(m_old - m_new) that rescales the running sum and the output accumulators (both indexed by query the way the score
// accumulator is: query = 4 (lane >> 4) + register), probabilities through a 16 x 64 LDS patch back into A-fragment order,
// O += P V.  Chunks whose first key no query of the tile may see are never visited, so a causal pass costs half the square and
// a one-token pass over n cached keys n / 64 chunks of one wavefront.  Key 0 must be visible to every query (mask_off >= 0 or
// no mask): a row of the first chunk is then never empty and m stays finite.
// ============================================================================================================
constexpr int LC = 64;                // keys per chunk
constexpr int LP_LD = LC + 4;         // floats per row of the probability patch

template <int DT>                     // head width / 16 (score contraction steps and output tiles)
__global__ __launch_bounds__(64) void attn_long_kernel(const AttentionArgs a) {
    __shared__ __attribute__((aligned(16))) float Ps[16 * LP_LD];
    const int lane = threadIdx.x, r16 = lane & 15, k4 = lane >> 4;
    const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int Skv = a.Skv;
    const float* Qb = a.Q + (int64_t)b * a.q_batch_stride + (int64_t)h * a.dh;
    const float* Kb = a.K + (int64_t)b * a.k_batch_stride + (int64_t)h * a.dh;
    const float* Vb = a.V + (int64_t)b * a.v_batch_stride + (int64_t)h * a.dv;
    const float* qrow = Qb + (int64_t)min(q0 + r16, a.Sq - 1) * a.ldq + 4 * k4;
    float4 qv[DT];
#pragma unroll
    for (int c = 0; c < DT; ++c) qv[c] = *reinterpret_cast<const float4*>(qrow + 16 * c);
    const float f = a.scale * LOG2E_A;
    // keys the tile may see: all of them, or up to the last query's limit
    const int last_q = min(q0 + 15, a.Sq - 1);
    const int n_keys = a.mask_off < 0 ? Skv : min(Skv, last_q + a.mask_off + 1);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l[4] = {0.f, 0.f, 0.f, 0.f};       // l: this lane's part of the row sum
    f32x4 o[DT];
#pragma unroll
    for (int u = 0; u < DT; ++u) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k0 = 0; k0 < n_keys; k0 += LC) {
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* krow = Kb + (int64_t)min(k0 + 16 * t + r16, Skv - 1) * a.ldk + 4 * k4;
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                const float4 kv = *reinterpret_cast<const float4*>(krow + 16 * c);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[c].x, kv.x, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[c].y, kv.y, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[c].z, kv.z, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[c].w, kv.w, s[t], 0, 0, 0);
            }
        }
        float mx[4] = {m[0], m[1], m[2], m[3]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = k0 + 16 * t + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 4 * k4 + r;
                const bool ok = key < Skv && (a.mask_off < 0 || key <= q + a.mask_off);
                s[t][r] = ok ? s[t][r] * f : -INFINITY;
                mx[r] = fmaxf(mx[r], s[t][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], d, 64));
            const float alpha = exp2_fast(m[r] - mx[r]);            // 0 on the first chunk (m = -inf, mx finite)
            m[r] = mx[r];
            l[r] *= alpha;
#pragma unroll
            for (int u = 0; u < DT; ++u) o[u][r] *= alpha;
        }
        __syncthreads();                                             // the previous chunk's patch has been read
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2_fast(s[t][r] - m[r]);
                l[r] += p;
                Ps[(4 * k4 + r) * LP_LD + 16 * t + r16] = p;
            }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < LC / 16; ++g) {                          // 16 keys per step: this lane's 4 keys are 16 g + 4 k4 + j
            const float4 pv = *reinterpret_cast<const float4*>(&Ps[r16 * LP_LD + 16 * g + 4 * k4]);
            const float pj[4] = {pv.x, pv.y, pv.z, pv.w};
            float vv[4][DT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vrow = Vb + (int64_t)min(k0 + 16 * g + 4 * k4 + j, Skv - 1) * a.ldv + r16;        // masked keys carry p = 0
#pragma unroll
                for (int u = 0; u < DT; ++u) vv[j][u] = vrow[16 * u];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < DT; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(pj[j], vv[j][u], o[u], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) l[r] += __shfl_xor(l[r], d, 64);
        l[r] = rcp_fast(l[r]);
        const int q = q0 + 4 * k4 + r;
        if (q < a.Sq) {
#pragma unroll
            for (int u = 0; u < DT; ++u) a.O[(int64_t)b * a.o_batch_stride + (int64_t)q * a.ldo + (int64_t)h * a.dv + 16 * u + r16] = o[u][r] * l[r];
        }
    }
}

// ============================================================================================================
// The same computation with ONE workgroup per (batch item, head): wave w owns queries 16 w .. 16 w + 15 and all waves share
// K and V through LDS, so every K / V element is read from memory once (the kernel above reads them once per query tile)
// and with 16-byte loads.  For head dimensions that are multiples of 32 and Sq == Skv <= 128.
//   scores: the head dimension advances in chunks of 32: the K chunk (Skv rows x 32 floats) is fetched into registers by all
//           threads while the previous chunk is multiplied, then written to LDS; Q fragments come straight from memory.
//   output: V advances in chunks of 32 columns the same way; the probabilities sit in a wave-private LDS patch as above.
// Image rows are 36 floats: the float4 K-fragment reads (16 rows x one 16-byte column) and the scalar V-fragment reads
// (keys 4 apart x 16 columns) are both bank-conflict free.  Rows beyond Skv are zero (a zero probability times an
// undefined V would be NaN).
// ============================================================================================================
constexpr int KV_LD = 36;

__global__ __launch_bounds__(512) void attn_mfma_shared_kernel(const AttentionArgs a, int p_ld) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const int nthreads = blockDim.x, n_kt = nthreads >> 6;          // one wave per 16 queries = per key tile
    const int h = blockIdx.x, b = blockIdx.y;
    const int S = a.Skv, rows_pad = 16 * n_kt;
    float* kv = sh;                                                 // rows_pad x KV_LD floats
    float* Ps = sh + rows_pad * KV_LD + wave * 16 * p_ld;
    const float* Qb = a.Q + (int64_t)b * a.q_batch_stride + (int64_t)h * a.dh;
    const float* Kb = a.K + (int64_t)b * a.k_batch_stride + (int64_t)h * a.dh;
    const float* Vb = a.V + (int64_t)b * a.v_batch_stride + (int64_t)h * a.dv;
    const int q0 = 16 * wave;

    // chunk staging: float4 i = tid + nthreads j (j = 0, 1) of the rows_pad x 8 grid: row i >> 3, 16-byte column i & 7
    int srow[2], scol[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int i = tid + nthreads * j; srow[j] = i >> 3; scol[j] = (i & 7) * 4; }
    float4 stage[2];
    auto fetch = [&](const float* base, int64_t ld, int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            stage[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (srow[j] < S) stage[j] = *reinterpret_cast<const float4*>(base + (int64_t)srow[j] * ld + c + scol[j]);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(kv + srow[j] * KV_LD + scol[j]) = stage[j];
    };

    // ---- scores
    f32x4 s[MAX_KT];
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* qrow = Qb + (int64_t)min(q0 + r16, a.Sq - 1) * a.ldq + 4 * k4;
    fetch(Kb, a.ldk, 0);
    float4 qn0 = *reinterpret_cast<const float4*>(qrow), qn1 = *reinterpret_cast<const float4*>(qrow + 16);
    for (int c = 0; c < a.dh; c += 32) {
        __syncthreads();                                            // everyone is done with the previous image
        stash();
        const float4 q0v = qn0, q1v = qn1;
        __syncthreads();
        if (c + 32 < a.dh) {
            fetch(Kb, a.ldk, c + 32);
            qn0 = *reinterpret_cast<const float4*>(qrow + c + 32);
            qn1 = *reinterpret_cast<const float4*>(qrow + c + 48);
        }
#pragma unroll
        for (int t = 0; t < MAX_KT; ++t) {
            if (t < n_kt) {
                const float4 k0 = *reinterpret_cast<const float4*>(kv + (16 * t + r16) * KV_LD + 4 * k4);
                const float4 k1 = *reinterpret_cast<const float4*>(kv + (16 * t + r16) * KV_LD + 16 + 4 * k4);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q0v.x, k0.x, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q0v.y, k0.y, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q0v.z, k0.z, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q0v.w, k0.w, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q1v.x, k1.x, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q1v.y, k1.y, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q1v.z, k1.z, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q1v.w, k1.w, s[t], 0, 0, 0);
            }
        }
    }
    fetch(Vb, a.ldv, 0);                                            // the first V chunk flies under the softmax

    // ---- softmax per query (= per register index and 16-lane row)
    const float f = a.scale * LOG2E_A;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) {
        if (t < n_kt) {
            const bool ok = 16 * t + r16 < S;
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
                Ps[(4 * k4 + r) * p_ld + 16 * t + r16] = p;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
        sum[r] = rcp_fast(sum[r]);
    }

    // ---- output, 32 columns per V chunk.  MFMA j of key tile t contracts keys 16 t + 4 k + j (k = 0..3 = lane >> 4).
    for (int c0 = 0; c0 < a.dv; c0 += 32) {
        __syncthreads();                                            // the previous image is free; the P patch is complete
        stash();
        __syncthreads();
        if (c0 + 32 < a.dv) fetch(Vb, a.ldv, c0 + 32);
        f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int t = 0; t < MAX_KT; ++t) {
            if (t >= n_kt) break;
            const float4 pv = *reinterpret_cast<const float4*>(&Ps[r16 * p_ld + 16 * t + 4 * k4]);
            const float* vr = kv + (16 * t + 4 * k4) * KV_LD + r16;
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.x, vr[0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.x, vr[16], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.y, vr[KV_LD], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.y, vr[KV_LD + 16], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.z, vr[2 * KV_LD], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.z, vr[2 * KV_LD + 16], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.w, vr[3 * KV_LD], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv.w, vr[3 * KV_LD + 16], o1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + 4 * k4 + r;
            if (q < a.Sq) {
                float* orow = a.O + (int64_t)b * a.o_batch_stride + (int64_t)q * a.ldo + (int64_t)h * a.dv + c0 + r16;
                orow[0] = o0[r] * sum[r];
                orow[16] = o1[r] * sum[r];
            }
        }
    }
}

// ============================================================================================================
// The same attention on the bf16 matrix cores (engine precisions bf16 / split bf16), Sq == Skv <= 128, head dims multiples of
// 32.  One workgroup per (batch item, head), wave w owns queries 32 w .. 32 w + 31.
//   scores   S^T[key][q] = K Q^T: A = K tile (lane <-> key), B = Q (lane <-> query).  The accumulator then holds "query on the
//            lane, keys on the registers": the softmax of a query is a reduction over registers plus one exchange between the
//            two lane halves, and the probabilities are the B operand of the next product as they stand -- no LDS patch.
//   output   O^T[c][q] = V^T P^T: A = V^T tile (lane <-> output column, k = keys), B = P^T from the registers; the result has the
//            query on the lane again, so 1 / sum is applied per lane and a lane stores 16-byte pieces of its query's row.
// K chunks (all keys x 32 dims) and V chunks (all keys x 32 columns) go through one LDS image as bf16 (high and low halves with
// split operands), converted once per workgroup; the V image is stored transposed ([column][key], a 4 x 4 register block per
// thread) with the keys of each group of 16 in the order the accumulator registers imply (index bits 2 and 3 swapped).  The next
// chunk is fetched into registers while the current one is multiplied.  Rows beyond Skv are zero.
// ============================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BK_ROW_B = 80;          // bytes per key row of the K image: 32 bf16 + 16 (conflict-free 16-byte fragment reads)

template <bool X3, int NT>
__global__ __launch_bounds__(64 * NT) void attn_bf_kernel(const AttentionArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char shb[];
    constexpr int ROWS = 32 * NT, NTHREADS = 64 * NT;
    constexpr int BV_ROW_B = 64 * NT + 16;                                  // bytes per column row of the V image
    constexpr int IMG_B = ROWS * BK_ROW_B > 32 * BV_ROW_B ? ROWS * BK_ROW_B : 32 * BV_ROW_B;     // one plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fc = lane >> 5;
    const int h = blockIdx.x, b = blockIdx.y, S = a.Skv;
    const float* Qb = a.Q + (int64_t)b * a.q_batch_stride + (int64_t)h * a.dh;
    const float* Kb = a.K + (int64_t)b * a.k_batch_stride + (int64_t)h * a.dh;
    const float* Vb = a.V + (int64_t)b * a.v_batch_stride + (int64_t)h * a.dv;

    float4 stage[4];
    // K chunk: float4 i = tid + NTHREADS j of the ROWS x 8 grid (row i >> 3, 16-byte column i & 7)
    auto fetch_k = [&](int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + NTHREADS * j, row = i >> 3;
            stage[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < S) stage[j] = *reinterpret_cast<const float4*>(Kb + (int64_t)row * a.ldk + c + (i & 7) * 4);
        }
    };
    auto split4 = [](const float (&v)[4], bf16x4_t& hi, bf16x4_t& lo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
    };
    auto stash_k = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + NTHREADS * j;
            const float v[4] = {stage[j].x, stage[j].y, stage[j].z, stage[j].w};
            bf16x4_t hi, lo;
            split4(v, hi, lo);
            unsigned char* dst = shb + (i >> 3) * BK_ROW_B + (i & 7) * 8;
            *reinterpret_cast<bf16x4_t*>(dst) = hi;
            if constexpr (X3) *reinterpret_cast<bf16x4_t*>(dst + IMG_B) = lo;
        }
    };
    // V chunk: thread = (key group kg = tid >> 3 of four keys, column group cg = tid & 7 of four columns)
    const int kg = tid >> 3, cg = tid & 7;
    auto fetch_v = [&](int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = 4 * kg + j;
            stage[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (key < S) stage[j] = *reinterpret_cast<const float4*>(Vb + (int64_t)key * a.ldv + c + 4 * cg);
        }
    };
    auto stash_v = [&]() {
        const int u = kg & 3, pos = ((4 * kg) & ~15) + 8 * (u & 1) + 4 * (u >> 1);      // keys 4 kg .. + 3 stay together
        const float col[4][4] = {{stage[0].x, stage[1].x, stage[2].x, stage[3].x}, {stage[0].y, stage[1].y, stage[2].y, stage[3].y},
                                 {stage[0].z, stage[1].z, stage[2].z, stage[3].z}, {stage[0].w, stage[1].w, stage[2].w, stage[3].w}};
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            bf16x4_t hi, lo;
            split4(col[ci], hi, lo);
            unsigned char* dst = shb + (4 * cg + ci) * BV_ROW_B + pos * 2;
            *reinterpret_cast<bf16x4_t*>(dst) = hi;
            if constexpr (X3) *reinterpret_cast<bf16x4_t*>(dst + IMG_B) = lo;
        }
    };
    auto split8 = [](const float4& u0, const float4& u1, bf16x8_t& hi, bf16x8_t& lo) {
        const float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
    };

    // ---- scores
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int q = 32 * wave + fr;
    const float* qrow = Qb + (int64_t)min(q, a.Sq - 1) * a.ldq + 8 * fc;          // this lane's k elements of step s: 16 s + 8 fc .. + 7
    fetch_k(0);
    float4 qn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qn[j] = *reinterpret_cast<const float4*>(qrow + 16 * (j >> 1) + 4 * (j & 1));
    for (int c = 0; c < a.dh; c += 32) {
        __syncthreads();                                            // everyone is done with the previous image
        stash_k();
        bf16x8_t qh[2], ql[2];
        split8(qn[0], qn[1], qh[0], ql[0]);
        split8(qn[2], qn[3], qh[1], ql[1]);
        __syncthreads();
        if (c + 32 < a.dh) {
            fetch_k(c + 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) qn[j] = *reinterpret_cast<const float4*>(qrow + c + 32 + 16 * (j >> 1) + 4 * (j & 1));
        } else {
            fetch_v(0);                                             // the first V chunk flies under the last products and the softmax
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const unsigned char* src = shb + (32 * t + fr) * BK_ROW_B + (16 * s + 8 * fc) * 2;
                const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(src);
                if constexpr (X3) {
                    const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(src + IMG_B);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], acc[t], 0, 0, 0);
            }
    }

    // ---- softmax of this lane's query: register r of tile t is key 32 t + 8 (r >> 2) + 4 fc + (r & 3)
    const float f = a.scale * LOG2E_A;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * t + 8 * (r >> 2) + 4 * fc + (r & 3);
            acc[t][r] = key < S ? acc[t][r] * f : -INFINITY;
            mx = fmaxf(mx, acc[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    bf16x8_t ph[2 * NT], pl[X3 ? 2 * NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float p = exp2_fast(acc[t][8 * u + i] - mx);
                sum += p;
                const __bf16 hi = (__bf16)p;
                ph[2 * t + u][i] = hi;
                if constexpr (X3) pl[2 * t + u][i] = (__bf16)(p - (float)hi);
            }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = rcp_fast(sum);

    // ---- output, 32 columns per V chunk
    for (int c0 = 0; c0 < a.dv; c0 += 32) {
        __syncthreads();
        stash_v();
        __syncthreads();
        if (c0 + 32 < a.dv) fetch_v(c0 + 32);
        f32x16_t o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int kq = 0; kq < 2 * NT; ++kq) {
            const unsigned char* src = shb + fr * BV_ROW_B + (16 * kq + 8 * fc) * 2;
            const bf16x8_t vh = *reinterpret_cast<const bf16x8_t*>(src);
            if constexpr (X3) {
                const bf16x8_t vl = *reinterpret_cast<const bf16x8_t*>(src + IMG_B);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[kq], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[kq], o, 0, 0, 0);
            }
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[kq], o, 0, 0, 0);
        }
        if (q < a.Sq) {
            float* orow = a.O + (int64_t)b * a.o_batch_stride + (int64_t)q * a.ldo + (int64_t)h * a.dv + c0 + 4 * fc;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(orow + 8 * g) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        }
    }
}

template <bool X3>
static hipError_t launch_attn_bf(const AttentionArgs& a, hipStream_t stream) {
    const int nt = (a.Skv + 31) / 32;
    // (A form that walked several heads per workgroup, requesting the next head's first chunk early, needed 169 registers instead
    // of 128 and was slower inside the decoders: -0.7 % at d=256, -10 % for MM-DiT.)
    const dim3 grid(a.H, a.B);
#define ADX_ATTN_BF(NT)                                                                                                            \
    {                                                                                                                              \
        constexpr int img = (32 * NT * BK_ROW_B > 32 * (64 * NT + 16)) ? 32 * NT * BK_ROW_B : 32 * (64 * NT + 16);                 \
        hipLaunchKernelGGL((attn_bf_kernel<X3, NT>), grid, dim3(64 * NT), (size_t)img * (X3 ? 2 : 1), stream, a);                  \
        return hipGetLastError();                                                                                                  \
    }
    switch (nt) {
        case 1: ADX_ATTN_BF(1)
        case 2: ADX_ATTN_BF(2)
        case 3: ADX_ATTN_BF(3)
        case 4: ADX_ATTN_BF(4)
        default: return hipErrorInvalidValue;
    }
#undef ADX_ATTN_BF
}

}  // namespace

static bool attention_bf_ok(const AttentionArgs& a) {
    return (a.prec == PREC_BF16 || a.prec == PREC_BF16X3) && a.Sq == a.Skv && a.Skv <= 128 && (a.dh % 32) == 0 && (a.dv % 32) == 0 &&
           (a.ldv % 4) == 0 && (a.v_batch_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(a.V) & 15) == 0 &&
           (a.ldo % 4) == 0 && (a.o_batch_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(a.O) & 15) == 0 && a.H <= 65535 && a.B <= 65535;
}

static bool attention_shared_ok(const AttentionArgs& a) {
    return a.Sq == a.Skv && (a.dh % 32) == 0 && (a.dv % 32) == 0 && (a.ldv % 4) == 0 && (a.v_batch_stride % 4) == 0 &&
           (reinterpret_cast<uintptr_t>(a.V) & 15) == 0 && a.H <= 65535 && a.B <= 65535;
}

bool attention_mfma_ok(const AttentionArgs& a) {
    return a.mask_off < 0 && !a.state && !a.key0 && a.Skv <= 16 * MAX_KT && a.dh >= 16 && (a.dh % 16) == 0 && (a.dv % 16) == 0 &&
           (a.ldq % 4) == 0 && (a.ldk % 4) == 0 && (a.q_batch_stride % 4) == 0 && (a.k_batch_stride % 4) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Q) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.K) & 15) == 0;
}

hipError_t attention_mfma(const AttentionArgs& a, hipStream_t stream) {
    if (!attention_mfma_ok(a)) return hipErrorInvalidValue;
    const char* f32_env = getenv("ADX_ATTN_F32");                // measurement switch, read per call: 1 = the fp32 kernels in every precision
    if (attention_bf_ok(a) && !(f32_env && f32_env[0] == '1'))
        return a.prec == PREC_BF16X3 ? launch_attn_bf<true>(a, stream) : launch_attn_bf<false>(a, stream);
    static const bool no_shared = [] { const char* e = getenv("ADX_ATTN_NO_SHARED"); return e && e[0] == '1'; }();   // measurement switch
    if (!no_shared && attention_shared_ok(a)) {
        const int n_kt = (a.Skv + 15) / 16, p_ld = 16 * n_kt + 4;
        const size_t smem = (size_t)(16 * n_kt * KV_LD + n_kt * 16 * p_ld) * sizeof(float);      // <= 52 KB
        hipLaunchKernelGGL(attn_mfma_shared_kernel, dim3(a.H, a.B), dim3(64 * n_kt), smem, stream, a, p_ld);
        return hipGetLastError();
    }
    dim3 grid((a.Sq + 15) / 16, a.H, a.B);
    if (a.dv >= 64) hipLaunchKernelGGL(attn_mfma_kernel<4>, grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(attn_mfma_kernel<2>, grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

bool attention_long_ok(const AttentionArgs& a) {
    return !a.state && !a.key0 && a.dh == a.dv && a.dh >= 16 && (a.dh % 16) == 0 && a.dh <= 128 && (a.ldq % 4) == 0 && (a.ldk % 4) == 0 &&
           (a.q_batch_stride % 4) == 0 && (a.k_batch_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(a.Q) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.K) & 15) == 0;
}

hipError_t attention_long(const AttentionArgs& a, hipStream_t stream) {
    if (!attention_long_ok(a)) return hipErrorInvalidValue;
    dim3 grid((a.Sq + 15) / 16, a.H, a.B);
    switch (a.dh / 16) {
        case 1: hipLaunchKernelGGL(attn_long_kernel<1>, grid, dim3(64), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(attn_long_kernel<2>, grid, dim3(64), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(attn_long_kernel<3>, grid, dim3(64), 0, stream, a); break;
        case 4: hipLaunchKernelGGL(attn_long_kernel<4>, grid, dim3(64), 0, stream, a); break;
        case 5: hipLaunchKernelGGL(attn_long_kernel<5>, grid, dim3(64), 0, stream, a); break;
        case 6: hipLaunchKernelGGL(attn_long_kernel<6>, grid, dim3(64), 0, stream, a); break;
        case 7: hipLaunchKernelGGL(attn_long_kernel<7>, grid, dim3(64), 0, stream, a); break;
        default: hipLaunchKernelGGL(attn_long_kernel<8>, grid, dim3(64), 0, stream, a); break;
    }
    return hipGetLastError();
}

}  // namespace adx

// Fused "tail" of one variant-A denoiser call plus the scheduler update (d_model 32, 8 heads, FFN 128, T <= 11).
//
// After the encoder's last attention everything left concerns only T+1 memory rows and T trajectory tokens of
// each batch item: the remainder of the last encoder layer (out_proj + residual, LayerNorm, FFN + residual), the
// whole decoder (4 x [causal self-attention, masked cross-attention to the T+1 memory rows, FFN], ln_f, head) and
// the scheduler update x_t -> x_{t-1}.  As separate launches that is ~75 kernels of a few microseconds each;
// here it is ONE launch: one 128-thread workgroup per batch item, activations in LDS, fp32 VALU arithmetic
// (this is the trajectory path, which stays fp32 in every precision mode), weights streamed from L2 in a
// k-major packing ([K/4][N][4]) so that a wavefront's weight load is one contiguous 1 KiB read.
#include "adx_ops.h"
#include "fastmath.h"

#include <math.h>

namespace adx {
namespace {

constexpr int TRM = 12;       // max rows (memory rows T+1 <= 12)
constexpr int D = 32, LD = 36, NH = 8, DH = 4;
constexpr float LOG2E_T = 1.4426950408889634f;

__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = rcp_fast(fmaf(0.3275911f, ax, 1.0f));
    const float p = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float r = 1.0f - p * exp2_fast(-ax * ax * LOG2E_T);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752f)); }

// Y[t][n] = [R[t][n] +] act(sum_k X[t][k] W[n][k] + b[n]) for t < TRM (rows >= the live count are zero rows of X and
// are stored too: they stay finite and are never read as results).  G row groups share the 128 threads when N is small.
template <int G>
__device__ __forceinline__ void gemm_g(const float* X, int ldx, const float* w, const TailMat m, float* Y, int ldy, bool gelu,
                                       const float* R, int ldr, int rows) {
    constexpr int RPT = (TRM + G - 1) / G;          // rows per thread
    const int tid = threadIdx.x;
    const int n = tid % (128 / G), rg = tid / (128 / G);
    if (n < m.N) {
        float acc[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc[i] = 0.f;
        const float4* wp = reinterpret_cast<const float4*>(w + m.w_off) + n;
        // weights come from L2: fetch eight k-steps (32 k) at a time so their latencies overlap
        for (int k0 = 0; k0 < m.K / 4; k0 += 8) {
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = (k0 + u < m.K / 4) ? wp[(int64_t)(k0 + u) * m.N] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int t = rg + i * G;
                    if (t < TRM) {
                        const float4 xv = *reinterpret_cast<const float4*>(X + t * ldx + 4 * min(k0 + u, m.K / 4 - 1));
                        acc[i] = fmaf(wv[u].x, xv.x, fmaf(wv[u].y, xv.y, fmaf(wv[u].z, xv.z, fmaf(wv[u].w, xv.w, acc[i]))));
                    }
                }
            }
        }
        const float bias = w[m.b_off + n];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int t = rg + i * G;
            if (t < rows) {
                float v = acc[i] + bias;
                if (gelu) v = gelu_erf(v);
                if (R) v += R[t * ldr + n];
                Y[t * ldy + n] = v;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void gemm(const float* X, int ldx, const float* w, const TailMat m, float* Y, int ldy, bool gelu,
                                     const float* R, int ldr, int rows) {
    if (m.N <= 32) gemm_g<4>(X, ldx, w, m, Y, ldy, gelu, R, ldr, rows);
    else if (m.N <= 64) gemm_g<2>(X, ldx, w, m, Y, ldy, gelu, R, ldr, rows);
    else gemm_g<1>(X, ldx, w, m, Y, ldy, gelu, R, ldr, rows);
}

// LayerNorm of `rows` rows of 32 features: 32 lanes per row, 4 rows per pass
__device__ __forceinline__ void layer_norm(const float* X, float* Y, const float* w, int g_off, int b_off, int rows) {
    const int f = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    for (int t = r0; t < rows; t += 4) {
        const float v = X[t * LD + f];
        float s = v;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
        const float mean = s * (1.0f / 32.0f);
        const float dlt = v - mean;
        float q = dlt * dlt;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 32);
        const float rstd = rsqrt_fast(fmaf(q, 1.0f / 32.0f, 1e-5f));
        Y[t * LD + f] = fmaf(dlt * rstd, w[g_off + f], w[b_off + f]);
    }
    __syncthreads();
}

// softmax attention, one thread per (query, head); key j visible iff j <= q + mask_off; at most TRM keys
__device__ __forceinline__ void attend(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int nq, int nk,
                                       int mask_off) {
    const int tid = threadIdx.x;
    if (tid < nq * NH) {
        const int q = tid >> 3, h = tid & 7;
        const float4 qv = *reinterpret_cast<const float4*>(Q + q * ldq + h * DH);
        const float sc = 0.5f * LOG2E_T;               // 1 / sqrt(head_dim 4), in log2 units
        float s[TRM];
        float mx = -INFINITY;
        const int last = min(nk - 1, q + mask_off);
#pragma unroll
        for (int j = 0; j < TRM; ++j) {
            if (j <= last) {
                const float4 kv = *reinterpret_cast<const float4*>(K + j * ldkv + h * DH);
                s[j] = (qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w) * sc;
                mx = fmaxf(mx, s[j]);
            } else {
                s[j] = -INFINITY;
            }
        }
        float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
        for (int j = 0; j < TRM; ++j) {
            if (j <= last) {
                const float p = exp2_fast(s[j] - mx);
                const float4 vv = *reinterpret_cast<const float4*>(V + j * ldkv + h * DH);
                l += p;
                o0 = fmaf(p, vv.x, o0); o1 = fmaf(p, vv.y, o1); o2 = fmaf(p, vv.z, o2); o3 = fmaf(p, vv.w, o3);
            }
        }
        const float inv = rcp_fast(l);
        *reinterpret_cast<float4*>(O + q * LD + h * DH) = make_float4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
    }
    __syncthreads();
}

__global__ __launch_bounds__(128) void tail_a_kernel(const TailArgs a) {
    __shared__ __attribute__((aligned(16))) float XM[TRM * LD];      // encoder rows -> memory
    __shared__ __attribute__((aligned(16))) float Y[TRM * LD];       // decoder stream
    __shared__ __attribute__((aligned(16))) float HB[TRM * LD];      // LayerNorm output
    __shared__ __attribute__((aligned(16))) float AT[TRM * LD];      // attention output / cross-attention queries
    __shared__ __attribute__((aligned(16))) float QB[TRM * LD];
    __shared__ __attribute__((aligned(16))) float QKV[TRM * 100];
    __shared__ __attribute__((aligned(16))) float MKV[TRM * 68];
    __shared__ __attribute__((aligned(16))) float FB[TRM * 132];
    __shared__ __attribute__((aligned(16))) float XI[TRM * 68];      // x_t rows (Cin <= 64), later the model output
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = a.T, TR = a.T + 1, C = a.C;
    const float* w = a.w;
    for (int i = tid; i < TRM * LD; i += 128) { XM[i] = 0.f; Y[i] = 0.f; HB[i] = 0.f; AT[i] = 0.f; QB[i] = 0.f; }
    for (int i = tid; i < TRM * 100; i += 128) QKV[i] = 0.f;
    for (int i = tid; i < TRM * 68; i += 128) { MKV[i] = 0.f; XI[i] = 0.f; }
    for (int i = tid; i < TRM * 132; i += 128) FB[i] = 0.f;
    __syncthreads();
    // ---- loads: layer-1 attention output and residual rows 0..T, current sample x_t
    for (int i = tid; i < TR * D; i += 128) {
        const int t = i >> 5, f = i & 31;
        AT[t * LD + f] = a.att1[((int64_t)b * a.L + t) * D + f];
        XM[t * LD + f] = a.x1[((int64_t)b * a.L + t) * D + f];
    }
    for (int i = tid; i < T * C; i += 128) XI[(i / C) * 68 + (i % C)] = a.x[(int64_t)b * T * C + i];
    __syncthreads();
    // ---- remainder of the last encoder layer on the T+1 visible rows -> memory
    gemm(AT, LD, w, a.e_wo, XM, LD, false, XM, LD, TR);
    layer_norm(XM, HB, w, a.e_ln2_g, a.e_ln2_b, TR);
    gemm(HB, LD, w, a.e_w1, FB, 132, true, nullptr, 0, TR);
    gemm(FB, 132, w, a.e_w2, XM, LD, false, XM, LD, TR);
    // ---- decoder
    gemm(XI, 68, w, a.d_in, Y, LD, false, w + a.pos_off, D, T);          // input_emb + pos_emb (pos table row stride 32)
    for (int l = 0; l < a.n_dec; ++l) {
        const TailLayer& p = a.dec[l];
        layer_norm(Y, HB, w, p.ln1_g, p.ln1_b, T);
        gemm(HB, LD, w, p.wqkv, QKV, 100, false, nullptr, 0, T);
        attend(QKV, 100, QKV + D, QKV + 2 * D, 100, AT, T, T, 0);
        gemm(AT, LD, w, p.wo, Y, LD, false, Y, LD, T);
        layer_norm(Y, HB, w, p.ln2_g, p.ln2_b, T);
        gemm(HB, LD, w, p.wq, QB, LD, false, nullptr, 0, T);
        gemm(XM, LD, w, p.wkv, MKV, 68, false, nullptr, 0, TR);
        attend(QB, LD, MKV, MKV + D, 68, AT, T, TR, 1);
        gemm(AT, LD, w, p.wo2, Y, LD, false, Y, LD, T);
        layer_norm(Y, HB, w, p.ln3_g, p.ln3_b, T);
        gemm(HB, LD, w, p.w1, FB, 132, true, nullptr, 0, T);
        gemm(FB, 132, w, p.w2, Y, LD, false, Y, LD, T);
    }
    layer_norm(Y, HB, w, a.lnf_g, a.lnf_b, T);
    gemm(HB, LD, w, a.d_head, FB, 132, false, nullptr, 0, T);             // model output (T x C) in FB rows
    // ---- scheduler update
    const StepCoef k = *a.coef;
    for (int i = tid; i < T * C; i += 128) {
        const int t = i / C, c = i % C;
        const int64_t gi = (int64_t)b * T * C + i;
        const float xv = XI[t * 68 + c], ov = FB[t * 132 + c];
        float x0 = fminf(fmaxf(k.a0 * xv + k.a1 * ov, -k.clip), k.clip);
        const float eps = k.eps_mode != 0.f ? (k.b0 * xv - x0) * k.b1 : ov;
        float r = k.c0 * x0 + k.c1 * xv + k.c2 * eps;
        if (k.c3 != 0.f && a.z) r += k.c3 * a.z[gi];
        a.x_next[gi] = r;
        if (a.model_out) a.model_out[gi] = ov;
    }
}

__global__ void pack_kernel(const float* W, int N, int K, float* out) {
    const int total = N * K;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / K, k = i % K;
        out[((k >> 2) * N + n) * 4 + (k & 3)] = W[i];
    }
}

}  // namespace

hipError_t tail_pack_matrix(const float* W, int N, int K, float* out, hipStream_t stream) {
    if (N <= 0 || K <= 0 || (K & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_kernel, dim3((N * K + 255) / 256), dim3(256), 0, stream, W, N, K, out);
    return hipGetLastError();
}

hipError_t tail_a(const TailArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.T <= 0 || a.T + 1 > TRM || a.C <= 0 || a.C > 64 || a.n_dec < 0 || a.n_dec > 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tail_a_kernel, dim3(a.B), dim3(128), 0, stream, a);
    return hipGetLastError();
}

}  // namespace adx

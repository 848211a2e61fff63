// Backward kernels of the refiner's training path (the reference trains its denoiser under torch autograd:
// TrajRefiner.py:417-452 builds the loss, loss.backward() differentiates transformer.py:189-273).  Everything here is fp32:
//   * weight / bias gradients of a Linear, dW = dY^T X and db = column sums of dY, on the fp32 matrix cores
//     (v_mfma_f32_32x32x2_f32: exact f32 products, one accumulation chain per output element), partial sums per row chunk
//     combined in a fixed order, so the result does not depend on the launch timing (no atomics);
//   * LayerNorm backward (input gradient + the per-element products whose column sums are the gain gradient);
//   * softmax attention backward for small head widths (the refiner's heads are 4 wide: vector arithmetic, not matrix
//     cores), recomputing the probabilities from q, k and the saved log-sum-exp instead of storing an Sq x Skv matrix;
//   * exact-erf GELU forward / backward, the gradient of the weighted squared error, transpose, add, column sums.
// The input gradient of a Linear, dX = dY W, is adx_op_linear on the transposed weight.
#include "engine.h"

#include <math.h>

using namespace adx;

#define ADX_API extern "C" __attribute__((visibility("default")))

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

int fail(int code, const char* msg) { set_error(msg); return code; }
int done(hipError_t e, const char* what) { return e == hipSuccess ? ADX_OK : hip_fail(e, what); }

constexpr float kLog2e = 1.4426950408889634f;

// 2^x as the single hardware instruction: the arguments here are differences to a running or final maximum (<= 0 up to
// rounding), so the library function's range handling (compare, two selects, ldexp per call) buys nothing
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// four elements per lane and access, for n a multiple of 4 and 16-byte aligned buffers (the scalar forms move 4 bytes per lane
// and reach 0.64 of the HBM peak); OP 0: a + b, 1: a * b
template <int OP>
__global__ __launch_bounds__(256) void binary4_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 u = a[i], v = b[i];
    out[i] = OP == 0 ? make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w) : make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
}
inline bool vec4_ok(int64_t n, const void* a, const void* b, const void* c) {
    return (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// y = x Phi(x), Phi the standard normal distribution function (torch's default "gelu": the erf form)
__global__ __launch_bounds__(256) void gelu_kernel(const float* x, float* y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = 0.5f * v * erfcf(-v * 0.70710678118654752f); }
}

// dx = dy (Phi(x) + x phi(x)), phi the standard normal density
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* x, const float* dy, float* dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float cdf = 0.5f * erfcf(-v * 0.70710678118654752f);
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    dx[i] = dy[i] * fmaf(v, pdf, cdf);
}

__global__ __launch_bounds__(256) void mul_kernel(const float* a, const float* b, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

// The activations of the decoder twin (ADX_ACT_* codes): value and derivative with respect to the pre-activation.
__device__ __forceinline__ float act_value(float v, int act) {
    switch (act) {
        case ADX_ACT_RELU: return fmaxf(v, 0.f);
        case ADX_ACT_GELU_ERF: return 0.5f * v * erfcf(-v * 0.70710678118654752f);
        case ADX_ACT_GELU_TANH: { const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v); return 0.5f * v * (1.0f + tanhf(u)); }
        case ADX_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        case ADX_ACT_SILU: return v / (1.0f + expf(-v));
        case ADX_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
__device__ __forceinline__ float act_slope(float v, int act) {
    switch (act) {
        case ADX_ACT_RELU: return v > 0.f ? 1.0f : 0.f;
        case ADX_ACT_GELU_ERF: return fmaf(v, 0.3989422804014327f * expf(-0.5f * v * v), 0.5f * erfcf(-v * 0.70710678118654752f));
        case ADX_ACT_GELU_TANH: {
            const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v), th = tanhf(u);
            return 0.5f * (1.0f + th) + 0.5f * v * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * v * v);
        }
        case ADX_ACT_LEAKY: return v > 0.f ? 1.0f : 0.01f;
        case ADX_ACT_SILU: { const float sg = 1.0f / (1.0f + expf(-v)); return sg * (1.0f + v * (1.0f - sg)); }
        case ADX_ACT_TANH: { const float th = tanhf(v); return 1.0f - th * th; }
        default: return 1.0f;
    }
}
__global__ __launch_bounds__(256) void act_kernel(const float* x, float* y, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = act_value(x[i], act);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* x, const float* dy, float* dx, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = dy[i] * act_slope(x[i], act);
}

// d loss / d out for loss = mean_b( w[t_b] * mean_e (out - target)^2 ), times the upstream scalar gradient
__global__ __launch_bounds__(256) void weighted_mse_bwd_kernel(const float* out, const float* target, const int64_t* t, const float* weight,
                                                                const float* upstream, float* dout, int64_t total, int row_len, float inv_count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float g = (upstream ? upstream[0] : 1.0f) * 2.0f * inv_count * weight[t[i / row_len]];
    dout[i] = g * (out[i] - target[i]);
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* x, float* y, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = x[(int64_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) y[(int64_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

// ------------------------------------------------------------------------------------------------ column sums
// partial[chunk][c] = sum of x[r][c] over the rows of the chunk.  A workgroup covers a tile of up to 256 columns; when the
// tile is narrower, 256 / width row phases share it and are combined through LDS in phase order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* x, int64_t M, int N, int64_t rows_per_chunk, float* partial) {
    __shared__ float red[256];
    const int c0 = blockIdx.x * 256;
    const int width = min(256, N - c0);
    const int P = 256 / width;
    const int c = threadIdx.x % width, p = threadIdx.x / width;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    if (p < P) {
        int64_t r = r0 + p;
        for (; r + 3 * P < r1; r += 4 * P) {                        // four loads in flight, added in row order
            const float v0 = x[r * N + c0 + c], v1 = x[(r + P) * N + c0 + c], v2 = x[(r + 2 * P) * N + c0 + c], v3 = x[(r + 3 * P) * N + c0 + c];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; r < r1; r += P) s += x[r * N + c0 + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (p == 0) {
        for (int q = 1; q < P; ++q) s += red[q * width + c];
        partial[(int64_t)blockIdx.y * N + c0 + c] = s;
    }
}

// out[i] (+)= sum over chunks of partial[chunk * stride + i].  A workgroup owns 16 consecutive outputs; 16 groups of threads
// each add every 16th chunk (ascending), and the group sums are added in group order: a fixed order for a given shape.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partial, int n_chunks, int64_t stride, int64_t n, float* out, int accumulate) {
    __shared__ float red[16][17];
    const int il = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + il;
    float s = 0.f;
    if (i < n)
        for (int ch = cg; ch < n_chunks; ch += 16) s += partial[(int64_t)ch * stride + i];
    red[cg][il] = s;
    __syncthreads();
    if (cg == 0 && i < n) {
        float t = red[0][il];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][il];
        out[i] = accumulate ? out[i] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------ dW = dY^T X, db = colsum(dY)
// One workgroup per (32 x 32 tile of dW, row chunk).  v_mfma_f32_32x32x2_f32 contracts two rows per instruction: lane l
// supplies dY[row + l / 32][n0 + l % 32] as the A operand and X[row + l / 32][k0 + l % 32] as the B operand -- both are plain
// coalesced loads, nothing is staged.  Each of the four waves walks a quarter of the chunk with four independent
// accumulation chains; the waves' tiles are added through LDS in wave order.
__global__ __launch_bounds__(256) void wgrad_partial_kernel(const float* dy, const float* x, int64_t M, int N, int K, int64_t rows_per_chunk,
                                                             float* partial, int64_t partial_stride, int want_bias) {
    __shared__ float red[4][17][64];
    const int kt_n = (K + 31) >> 5;
    const int nt = blockIdx.x / kt_n, kt = blockIdx.x % kt_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n = nt * 32 + col, k = kt * 32 + col;
    const bool n_ok = n < N, k_ok = k < K;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const int64_t quarter = rows_per_chunk >> 2;                    // rows_per_chunk is a multiple of 32
    const int64_t w0 = r0 + wave * quarter, w1 = min(r1, w0 + quarter);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;
    float a[4], b[4];
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t rr = base + 2 * j + half;
            const bool ok = rr < w1;
            a[j] = (ok && n_ok) ? dy[rr * N + n] : 0.f;
            b[j] = (ok && k_ok) ? x[rr * K + k] : 0.f;
        }
    };
    fetch(w0);
    for (int64_t base = w0; base < w1; base += 8) {
        float ca[4], cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ca[j] = a[j]; cb[j] = b[j]; }
        fetch(base + 8);                                            // the next eight rows are in flight under these products
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[j], cb[j], acc[j], 0, 0, 0);
            bsum += ca[j];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
    red[wave][16][lane] = bsum;
    __syncthreads();
    if (wave != 0) return;
    float* out = partial + (int64_t)blockIdx.y * partial_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
        const int m = nt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);           // accumulator register r of this lane: row m, column col
        if (m < N && k_ok) out[(int64_t)m * K + k] = s;
    }
    if (want_bias && kt == 0 && half == 0 && n_ok) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += red[w][16][lane] + red[w][16][lane + 32];
        out[(int64_t)N * K + n] = s;
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// G lanes per row.  y = xhat * gamma + beta with xhat = (x - mean) * rstd:
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;   dgamma = colsum(dy * xhat),  dbeta = colsum(dy).
template <int G>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* x, const float* gamma, const float* dy, int64_t rows, int d, float eps,
                                                             float* dx, float* dy_xhat) {
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int sub = threadIdx.x % G;
    if (row >= rows) return;                                        // uniform over the G lanes of a row
    const float* xr = x + row * d;
    const float* gr = dy + row * d;
    float s = 0.f;
    for (int e = sub; e < d; e += G) s += xr[e];
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, G);
    const float mean = s / (float)d;
    float q = 0.f;
    for (int e = sub; e < d; e += G) { const float t = xr[e] - mean; q = fmaf(t, t, q); }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, G);
    const float rstd = 1.0f / sqrtf(q / (float)d + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int e = sub; e < d; e += G) {
        const float xh = (xr[e] - mean) * rstd;
        const float g = gr[e] * (gamma ? gamma[e] : 1.0f);
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, G); s2 += __shfl_xor(s2, o, G); }
    const float m1 = s1 / (float)d, m2 = s2 / (float)d;
    for (int e = sub; e < d; e += G) {
        const float xh = (xr[e] - mean) * rstd;
        const float g = gr[e] * (gamma ? gamma[e] : 1.0f);
        dx[row * d + e] = rstd * (g - m1 - xh * m2);
        if (dy_xhat) dy_xhat[row * d + e] = gr[e] * xh;
    }
}

// ------------------------------------------------------------------------------------------------ attention backward
// Workgroup -> (item, head, block of queries / keys) with every workgroup of an item on ONE XCD.  Heads are interleaved in
// memory (a 128-byte line of a row holds all eight 4-wide heads), so the workgroups of an item's heads read the same lines;
// the hardware deals consecutive workgroup ids round-robin over the eight XCDs, each with its own L2, and with the natural
// (block, head, item) order an item's lines were fetched from HBM by up to eight L2s (counters: 17 x the algorithmic bytes).
// Ids congruent modulo 8 share an XCD, so item = 8 * (slot / per_item) + id % 8: the grid is padded to a multiple of 8 items.
struct AttnBlock { int b, h, x; bool ok; };
// ------------------------------------------------------------------------------------------------ dropout
// nn.Dropout(p) of the reference's training mode (transformer.py:52,70-98; torch's TransformerEncoder / DecoderLayer): every
// element (and, inside the attention, every (query, key) probability) is kept with probability 1 - p and scaled by 1 / (1 - p).
// The decision is a pure function of (seed, site, position): a 32-bit integer hash (two multiply / xor-shift rounds) compared with
// a threshold, so the backward pass -- and the key pass of the attention backward, which walks the pairs in another order --
// regenerates the mask instead of storing it, and results do not depend on the launch shape.  `site` numbers the dropout layers
// of one forward pass; seed and site are mixed on the host into `base`.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool drop_keep_elem(uint32_t base, int64_t i, uint32_t thr) {
    return mix32(mix32(base ^ ((uint32_t)i * 0x9E3779B1u)) ^ ((uint32_t)((uint64_t)i >> 32) * 0x85EBCA77u)) < thr;
}
// attention: the hash of (item, head, query) is shared by the whole row of keys
struct DropArgs {
    const float* mask = nullptr;     // host-supplied multipliers [B][H][Sq][Skv] (0 or 1 / (1 - p)); otherwise the hash below
    uint32_t base = 0, thr = 0;
    float scale = 1.0f;
};
__device__ __forceinline__ uint32_t drop_row_hash(uint32_t base, int bh, int i) {
    return mix32(base ^ ((uint32_t)bh * 0x9E3779B1u) ^ ((uint32_t)i * 0x85EBCA77u));
}
__device__ __forceinline__ float drop_mult(const DropArgs& d, uint32_t row_hash, int64_t row_off, int j) {
    if (d.mask) return d.mask[row_off + j];
    return mix32(row_hash ^ ((uint32_t)j * 0xC2B2AE3Du)) < d.thr ? d.scale : 0.f;
}

// y = keep ? x * scale : 0 (x == nullptr: the multipliers themselves, i.e. the mask a caller can inspect or replay)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint32_t base, uint32_t thr, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float m = drop_keep_elem(base, i, thr) ? scale : 0.f;
    y[i] = x ? x[i] * m : m;
}
// the attention multipliers as a tensor [B][H][Sq][Skv]
__global__ __launch_bounds__(256) void attention_dropout_mask_kernel(float* __restrict__ out, int BH, int Sq, int Skv, DropArgs d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)BH * Sq * Skv) return;
    const int j = (int)(i % Skv), q = (int)((i / Skv) % Sq), bh = (int)(i / ((int64_t)Skv * Sq));
    out[i] = drop_mult(d, drop_row_hash(d.base, bh, q), 0, j);
}

__device__ __forceinline__ AttnBlock attn_block(int B, int H, int nx) {
    const int per_item = H * nx, id = blockIdx.x, slot = id >> 3, w = slot % per_item;
    AttnBlock r;
    r.b = (slot / per_item) * 8 + (id & 7);
    r.h = w / nx;
    r.x = w % nx;
    r.ok = r.b < B;
    return r;
}
unsigned attn_grid(int B, int H, int nx) { return (unsigned)(((B + 7) / 8) * 8 * H * nx); }


// q[B][Sq][H*DH], k / v[B][Skv][H*DH], o / dout[B][Sq][H*DH]; key j is visible to query i iff mask_off < 0 or j <= i + mask_off.
// Head widths this small fit no matrix-core shape (the f32 matrix rate equals the vector rate on this part, and bf16 products
// on split operands would waste 7/8 of a tile on the zeros of a block-diagonal head layout), so both passes are vector code.
// What bounds them is the LDS broadcast of the other side's rows (one LDS pipe serves the four SIMDs of a CU), so a thread
// owns PT queries (keys) 128 apart and every row read from LDS is used PT times.
//
// Pass over queries: the log-sum-exp of a query's scores (base 2) and D = dout . o are kept for the key pass;
//   dq_i = scale * sum_j dS_ij k_j,   dS_ij = P_ij (dout_i . v_j - D_i).
// First the largest score only (no exponential), then one exponential per pair with weights 2^(s - m), divided by their sum
// at the end:  dq = scale / l * sum_j 2^(s_j - m) (dout . v_j - D) k_j,  log-sum-exp = m + log2 l.
// HAVE_LSE: stats already holds the log-sum-exp of every query (written by attn_fwd_stats_kernel): the maximum pass is skipped
// and the weights are the probabilities themselves.
template <int DH, int PT, bool MASKED, bool HAVE_LSE, bool DROP = false>
__global__ __launch_bounds__(128) void attn_bwd_query_kernel(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                                              int B, int nx, int H, int Sq, int Skv, float scale, int mask_off, float* dq, float* stats,
                                                              DropArgs dr = DropArgs()) {
    __shared__ float ks[128][DH], vs[128][DH];
    const AttnBlock blk = attn_block(B, H, nx);
    if (!blk.ok) return;
    const int b = blk.b, h = blk.h, bx = blk.x;
    const int ld = H * DH;
    int iq[PT], last[PT];
    bool valid[PT];
    int64_t qoff[PT];
    float qs[PT][DH], go[PT][DH], dqa[PT][DH], D[PT], m[PT], l[PT];
    uint32_t rh[PT];
    int64_t roff[PT];
    int last_all = 0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        iq[u] = bx * (128 * PT) + 128 * u + threadIdx.x;
        valid[u] = iq[u] < Sq;
        const int ic = valid[u] ? iq[u] : Sq - 1;
        qoff[u] = ((int64_t)b * Sq + ic) * ld + h * DH;
        rh[u] = DROP ? drop_row_hash(dr.base, b * H + h, ic) : 0u;
        roff[u] = (((int64_t)b * H + h) * Sq + ic) * Skv;
        D[u] = 0.f; m[u] = -INFINITY; l[u] = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) {
            qs[u][e] = q[qoff[u] + e] * (scale * kLog2e);
            go[u][e] = dout[qoff[u] + e];
            D[u] = fmaf(go[u][e], o[qoff[u] + e], D[u]);
            dqa[u][e] = 0.f;
        }
        last[u] = MASKED ? min(Skv - 1, ic + mask_off) : Skv - 1;               // last visible key of this query
        last_all = max(last_all, last[u]);
        if (HAVE_LSE) { m[u] = stats[(((int64_t)b * H + h) * Sq + ic) * 2]; l[u] = 1.0f; }
    }
    const float* kb = k + (int64_t)b * Skv * ld + h * DH;
    const float* vb = v + (int64_t)b * Skv * ld + h * DH;
    for (int j0 = 0; j0 < (HAVE_LSE ? 0 : Skv); j0 += 128) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 128 * DH; idx += 128) {
            const int jj = idx / DH, e = idx % DH;
            ks[jj][e] = j0 + jj < Skv ? kb[(int64_t)(j0 + jj) * ld + e] : 0.f;
        }
        __syncthreads();
        const int jn = min(128, last_all - j0 + 1);
        for (int jj = 0; jj < jn; ++jj) {
            float kr[DH];
#pragma unroll
            for (int e = 0; e < DH; ++e) kr[e] = ks[jj][e];
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < DH; ++e) s = fmaf(qs[u][e], kr[e], s);
                if (!MASKED || j0 + jj <= last[u]) m[u] = fmaxf(m[u], s);
            }
        }
    }
    for (int j0 = 0; j0 < Skv; j0 += 128) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 128 * DH; idx += 128) {
            const int jj = idx / DH, e = idx % DH;
            const bool ok = j0 + jj < Skv;
            ks[jj][e] = ok ? kb[(int64_t)(j0 + jj) * ld + e] : 0.f;
            vs[jj][e] = ok ? vb[(int64_t)(j0 + jj) * ld + e] : 0.f;
        }
        __syncthreads();
        const int jn = min(128, last_all - j0 + 1);
        for (int jj = 0; jj < jn; ++jj) {
            float kr[DH], vr[DH];
#pragma unroll
            for (int e = 0; e < DH; ++e) { kr[e] = ks[jj][e]; vr[e] = vs[jj][e]; }
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int e = 0; e < DH; ++e) { s = fmaf(qs[u][e], kr[e], s); dp = fmaf(go[u][e], vr[e], dp); }
                float p = exp2_raw(s - m[u]);
                if (MASKED && j0 + jj > last[u]) p = 0.f;
                if (!HAVE_LSE) l[u] += p;
                if constexpr (DROP) dp *= drop_mult(dr, rh[u], roff[u], min(j0 + jj, Skv - 1));      // d P = d P_hat * multiplier
                const float ds = p * (dp - D[u]);
#pragma unroll
                for (int e = 0; e < DH; ++e) dqa[u][e] = fmaf(ds, kr[e], dqa[u][e]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        if (!valid[u]) continue;
        const float f = scale / l[u];
#pragma unroll
        for (int e = 0; e < DH; ++e) dq[qoff[u] + e] = dqa[u][e] * f;
        float* st = stats + (((int64_t)b * H + h) * Sq + iq[u]) * 2;
        if (!HAVE_LSE) st[0] = m[u] + log2f(l[u]);
        st[1] = D[u];
    }
}

// Forward attention of the training path: o, and per (query, head) the log-sum-exp of the scores (base 2) in stats[..][0], so
// that the backward needs no pass of its own to find it.  Thread = PT queries, online softmax over blocks of 8 keys.
template <int DH, int PT, bool MASKED, bool DROP = false>
__global__ __launch_bounds__(128) void attn_fwd_stats_kernel(const float* q, const float* k, const float* v, int B, int nx, int H, int Sq, int Skv,
                                                              float scale, int mask_off, float* o, float* stats, DropArgs dr = DropArgs()) {
    __shared__ float ks[128][DH], vs[128][DH];
    const AttnBlock blk = attn_block(B, H, nx);
    if (!blk.ok) return;
    const int b = blk.b, h = blk.h, bx = blk.x;
    const int ld = H * DH;
    int iq[PT], last[PT];
    bool valid[PT];
    int64_t qoff[PT];
    float qs[PT][DH], acc[PT][DH], m[PT], l[PT];
    uint32_t rh[PT];
    int64_t roff[PT];
    int last_all = 0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        iq[u] = bx * (128 * PT) + 128 * u + threadIdx.x;
        valid[u] = iq[u] < Sq;
        const int ic = valid[u] ? iq[u] : Sq - 1;
        qoff[u] = ((int64_t)b * Sq + ic) * ld + h * DH;
        rh[u] = DROP ? drop_row_hash(dr.base, b * H + h, ic) : 0u;
        roff[u] = (((int64_t)b * H + h) * Sq + ic) * Skv;
        m[u] = -INFINITY; l[u] = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) { qs[u][e] = q[qoff[u] + e] * (scale * kLog2e); acc[u][e] = 0.f; }
        last[u] = MASKED ? min(Skv - 1, ic + mask_off) : Skv - 1;
        last_all = max(last_all, last[u]);
    }
    const float* kb = k + (int64_t)b * Skv * ld + h * DH;
    const float* vb = v + (int64_t)b * Skv * ld + h * DH;
    for (int j0 = 0; j0 < Skv; j0 += 128) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 128 * DH; idx += 128) {
            const int jj = idx / DH, e = idx % DH;
            const bool ok = j0 + jj < Skv;
            ks[jj][e] = ok ? kb[(int64_t)(j0 + jj) * ld + e] : 0.f;
            vs[jj][e] = ok ? vb[(int64_t)(j0 + jj) * ld + e] : 0.f;
        }
        __syncthreads();
        const int jn = min(128, last_all - j0 + 1);
        for (int jb = 0; jb < jn; jb += 8) {
            float sc[PT][8], mx[PT];
#pragma unroll
            for (int u = 0; u < PT; ++u) mx[u] = m[u];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int jj = min(jb + w, 127);
                float kr[DH];
#pragma unroll
                for (int e = 0; e < DH; ++e) kr[e] = ks[jj][e];
#pragma unroll
                for (int u = 0; u < PT; ++u) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < DH; ++e) d = fmaf(qs[u][e], kr[e], d);
                    if (jb + w >= jn || (MASKED && j0 + jb + w > last[u])) d = -INFINITY;
                    sc[u][w] = d;
                    mx[u] = fmaxf(mx[u], d);
                }
            }
            float alpha[PT];
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                alpha[u] = mx[u] == -INFINITY ? 1.0f : exp2_raw(m[u] - mx[u]);      // nothing visible yet: keep the empty state
                if (mx[u] == -INFINITY) mx[u] = 0.f;                                // (every score of the block is -inf: all weights 0)
                l[u] *= alpha[u];
#pragma unroll
                for (int e = 0; e < DH; ++e) acc[u][e] *= alpha[u];
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int jj = min(jb + w, 127);
                float vr[DH];
#pragma unroll
                for (int e = 0; e < DH; ++e) vr[e] = vs[jj][e];
#pragma unroll
                for (int u = 0; u < PT; ++u) {
                    float p = exp2_raw(sc[u][w] - mx[u]);
                    l[u] += p;                              // the softmax is normalised over ALL visible keys; dropout acts on the weights
                    if constexpr (DROP) p *= drop_mult(dr, rh[u], roff[u], min(j0 + jb + w, Skv - 1));
#pragma unroll
                    for (int e = 0; e < DH; ++e) acc[u][e] = fmaf(p, vr[e], acc[u][e]);
                }
            }
#pragma unroll
            for (int u = 0; u < PT; ++u)
                if (l[u] > 0.f) m[u] = mx[u];
        }
    }
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        if (!valid[u]) continue;
        const float inv = 1.0f / l[u];
#pragma unroll
        for (int e = 0; e < DH; ++e) o[qoff[u] + e] = acc[u][e] * inv;
        stats[(((int64_t)b * H + h) * Sq + iq[u]) * 2] = m[u] + log2f(l[u]);
    }
}

// Pass over keys: thread = PT keys;  dv_j = sum_i P_ij dout_i,   dk_j = scale * sum_i dS_ij q_i.
template <int DH, int PT, bool MASKED, bool DROP = false>
__global__ __launch_bounds__(128) void attn_bwd_key_kernel(const float* q, const float* k, const float* v, const float* dout, const float* stats,
                                                            int B, int nx, int H, int Sq, int Skv, float scale, int mask_off, float* dk, float* dv,
                                                            DropArgs dr = DropArgs()) {
    __shared__ float qsh[128][DH], gsh[128][DH], ssh[128][2];
    __shared__ uint32_t hsh[DROP ? 128 : 1];                   // row hashes of the staged queries
    const AttnBlock blk = attn_block(B, H, nx);
    if (!blk.ok) return;
    const int b = blk.b, h = blk.h, bx = blk.x;
    const int ld = H * DH;
    int jk[PT], first[PT];
    bool valid[PT];
    int64_t koff[PT];
    float kk[PT][DH], vv[PT][DH], dka[PT][DH], dva[PT][DH];
    int first_all = Sq;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        jk[u] = bx * (128 * PT) + 128 * u + threadIdx.x;
        valid[u] = jk[u] < Skv;
        const int jc = valid[u] ? jk[u] : Skv - 1;
        koff[u] = ((int64_t)b * Skv + jc) * ld + h * DH;
#pragma unroll
        for (int e = 0; e < DH; ++e) { kk[u][e] = k[koff[u] + e]; vv[u][e] = v[koff[u] + e]; dka[u][e] = 0.f; dva[u][e] = 0.f; }
        first[u] = MASKED ? max(0, jc - mask_off) : 0;                          // first query that sees this key
        first_all = min(first_all, first[u]);
    }
    const float* qb = q + (int64_t)b * Sq * ld + h * DH;
    const float* gb = dout + (int64_t)b * Sq * ld + h * DH;
    const float* sb = stats + ((int64_t)b * H + h) * Sq * 2;
    for (int i0 = 0; i0 < Sq; i0 += 128) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 128 * DH; idx += 128) {
            const int ii = idx / DH, e = idx % DH;
            const bool ok = i0 + ii < Sq;
            qsh[ii][e] = ok ? qb[(int64_t)(i0 + ii) * ld + e] * (scale * kLog2e) : 0.f;
            gsh[ii][e] = ok ? gb[(int64_t)(i0 + ii) * ld + e] : 0.f;
        }
        for (int idx = threadIdx.x; idx < 256; idx += 128) {
            const int ii = idx >> 1;
            ssh[ii][idx & 1] = i0 + ii < Sq ? sb[(int64_t)(i0 + ii) * 2 + (idx & 1)] : 0.f;
        }
        if constexpr (DROP) hsh[threadIdx.x] = drop_row_hash(dr.base, b * H + h, min(i0 + (int)threadIdx.x, Sq - 1));
        __syncthreads();
        const int in = min(128, Sq - i0);
        for (int ii = max(0, first_all - i0); ii < in; ++ii) {
            float qr[DH], gr[DH];
#pragma unroll
            for (int e = 0; e < DH; ++e) { qr[e] = qsh[ii][e]; gr[e] = gsh[ii][e]; }
            const float lse = ssh[ii][0], Di = ssh[ii][1];
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int e = 0; e < DH; ++e) { s = fmaf(qr[e], kk[u][e], s); dp = fmaf(gr[e], vv[u][e], dp); }
                float p = exp2_raw(s - lse);
                if (MASKED && i0 + ii < first[u]) p = 0.f;
                float pv = p;                                     // weight of dout_i in dv_j: the dropped-out probability
                if constexpr (DROP) {
                    const float mk = drop_mult(dr, hsh[ii], (((int64_t)b * H + h) * Sq + i0 + ii) * Skv, valid[u] ? jk[u] : Skv - 1);
                    pv *= mk; dp *= mk;
                }
                const float ds = p * (dp - Di);
#pragma unroll
                for (int e = 0; e < DH; ++e) { dva[u][e] = fmaf(pv, gr[e], dva[u][e]); dka[u][e] = fmaf(ds, qr[e], dka[u][e]); }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        if (!valid[u]) continue;
#pragma unroll
        for (int e = 0; e < DH; ++e) { dk[koff[u] + e] = dka[u][e] * (1.0f / kLog2e); dv[koff[u] + e] = dva[u][e]; }
    }
}

// ------------------------------------------------------------------------------------------------ short sequences, wide heads
// The joint attention of the MM-DiT blocks (80 tokens, head width 2 d = 128 ... 512; libs/diffusionTransformer.py:99-117): one
// workgroup per (item, head), the Sq x Skv probability matrix lives in LDS (and in the backward a second matrix for dS), the
// head dimension is walked with 16-byte loads.  Scores in base 2: the statistics are those of attn_fwd_stats_kernel.
// out[i][j] = sum_e A[i][e] B[j][e] for i < Sa, j < Sb (rows of stride lda / ldb in global memory, out in LDS with row stride ldo).
// A wave owns a 4 x 4 block of pairs and its lanes split the head dimension, so every load is a coalesced 256-byte piece of a
// row and is used four times; the sixteen lane-partial sums are combined with butterfly exchanges.
constexpr int kWideThreads = 1024;           // 16 waves per workgroup, four per SIMD: the products below are chains of dependent
constexpr int kWideWaves = kWideThreads / 64;   // global loads, and one workgroup per (item, head) leaves a CU nothing else to run

__device__ __forceinline__ void pair_dots(const float* A, int lda, const float* B, int ldb, int Sa, int Sb, int DH, float* out, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb = (Sb + 3) >> 2, n_tiles = ((Sa + 3) >> 2) * tb;
    for (int tile = wave; tile < n_tiles; tile += kWideWaves) {
        const int i0 = (tile / tb) * 4, j0 = (tile % tb) * 4;
        float acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) acc[u][w] = 0.f;
        for (int e = lane; e < DH; e += 64) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = A[(int64_t)min(i0 + u, Sa - 1) * lda + e];
                b[u] = B[(int64_t)min(j0 + u, Sb - 1) * ldb + e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[u][w] = fmaf(a[u], b[w], acc[u][w]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float v = acc[u][w];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0 && i0 + u < Sa && j0 + w < Sb) out[(i0 + u) * ldo + j0 + w] = v;
            }
    }
}

// out[i][e] = scale * sum_j M(i, j) V[j][e] for i < Sa, e < DH, with M(i, j) = TRANS ? Mat[j][i] : Mat[i][j] read from LDS
// (broadcast) and V rows in global memory.  A thread owns one column e and eight rows i: a value of V is loaded once per
// eight products, adjacent threads read adjacent columns.
template <bool TRANS>
__device__ __forceinline__ void matrix_times_rows(const float* Mat, int ldm, int Sa, int Sb, const float* V, int ldv, int DH, float scale,
                                                  float* out, int ldo) {
    const int n_groups = (Sa + 7) >> 3;
    for (int idx = threadIdx.x; idx < n_groups * DH; idx += kWideThreads) {
        const int e = idx % DH, i0 = (idx / DH) * 8;
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
#pragma unroll 4
        for (int j = 0; j < Sb; ++j) {
            const float vv = V[(int64_t)j * ldv + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, Sa - 1);
                acc[u] = fmaf(TRANS ? Mat[j * ldm + i] : Mat[i * ldm + j], vv, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u < Sa) out[(int64_t)(i0 + u) * ldo + e] = acc[u] * scale;
    }
}

__global__ __launch_bounds__(kWideThreads) void attn_wide_fwd_kernel(const float* q, const float* k, const float* v, int H, int Sq, int Skv, int DH,
                                                             float scale, int mask_off, float* o, float* stats) {
    extern __shared__ float wide_lds[];
    float* P = wide_lds;                                            // [Sq][Skv + 1]
    const int b = blockIdx.y, h = blockIdx.x, ld = H * DH, ldp = Skv + 1;
    const float* qb = q + (int64_t)b * Sq * ld + h * DH;
    const float* kb = k + (int64_t)b * Skv * ld + h * DH;
    const float* vb = v + (int64_t)b * Skv * ld + h * DH;
    pair_dots(qb, ld, kb, ld, Sq, Skv, DH, P, ldp);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < Sq; i += kWideWaves) {
        const int last = mask_off < 0 ? Skv - 1 : min(Skv - 1, i + mask_off);
        float m = -INFINITY;
        for (int j = lane; j <= last; j += 64) m = fmaxf(m, P[i * ldp + j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        m *= scale * kLog2e;
        float l = 0.f;
        for (int j = lane; j < Skv; j += 64) {
            const float p = j <= last ? exp2_raw(fmaf(P[i * ldp + j], scale * kLog2e, -m)) : 0.f;
            P[i * ldp + j] = p;
            l += p;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
        const float inv = 1.0f / l;
        for (int j = lane; j < Skv; j += 64) P[i * ldp + j] *= inv;
        if (lane == 0) stats[(((int64_t)b * H + h) * Sq + i) * 2] = m + log2f(l);
    }
    __syncthreads();
    matrix_times_rows<false>(P, ldp, Sq, Skv, vb, ld, DH, 1.0f, o + (int64_t)b * Sq * ld + h * DH, ld);
}

__global__ __launch_bounds__(kWideThreads) void attn_wide_bwd_kernel(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                                             float* stats, int have_lse, int H, int Sq, int Skv, int DH, float scale, int mask_off,
                                                             float* dq, float* dk, float* dv) {
    extern __shared__ float wide_lds[];
    const int b = blockIdx.y, h = blockIdx.x, ld = H * DH, ldp = Skv + 1;
    float* P = wide_lds;                                            // [Sq][Skv + 1]
    float* dS = P + Sq * ldp;                                       // [Sq][Skv + 1]
    const float* qb = q + (int64_t)b * Sq * ld + h * DH;
    const float* kb = k + (int64_t)b * Skv * ld + h * DH;
    const float* vb = v + (int64_t)b * Skv * ld + h * DH;
    const float* ob = o + (int64_t)b * Sq * ld + h * DH;
    const float* gb = dout + (int64_t)b * Sq * ld + h * DH;
    float* st = stats + ((int64_t)b * H + h) * Sq * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pair_dots(qb, ld, kb, ld, Sq, Skv, DH, P, ldp);                 // raw scores
    pair_dots(gb, ld, vb, ld, Sq, Skv, DH, dS, ldp);                // dP = dout . v
    __syncthreads();
    for (int i = wave; i < Sq; i += kWideWaves) {
        const int last = mask_off < 0 ? Skv - 1 : min(Skv - 1, i + mask_off);
        float lse;
        if (have_lse) {
            lse = st[i * 2];
        } else {
            float m = -INFINITY;
            for (int j = lane; j <= last; j += 64) m = fmaxf(m, P[i * ldp + j]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            m *= scale * kLog2e;
            float l = 0.f;
            for (int j = lane; j <= last; j += 64) l += exp2_raw(fmaf(P[i * ldp + j], scale * kLog2e, -m));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
            lse = m + log2f(l);
        }
        float d = 0.f;
        for (int e = lane; e < DH; e += 64) d = fmaf(gb[(int64_t)i * ld + e], ob[(int64_t)i * ld + e], d);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
        for (int j = lane; j < Skv; j += 64) {
            const float p = j <= last ? exp2_raw(fmaf(P[i * ldp + j], scale * kLog2e, -lse)) : 0.f;
            P[i * ldp + j] = p;
            dS[i * ldp + j] = p * (dS[i * ldp + j] - d);
        }
        if (lane == 0) { st[i * 2] = lse; st[i * 2 + 1] = d; }
    }
    __syncthreads();
    matrix_times_rows<false>(dS, ldp, Sq, Skv, kb, ld, DH, scale, dq + (int64_t)b * Sq * ld + h * DH, ld);      // dq = scale dS k
    matrix_times_rows<true>(dS, ldp, Skv, Sq, qb, ld, DH, scale, dk + (int64_t)b * Skv * ld + h * DH, ld);      // dk = scale dS^T q
    matrix_times_rows<true>(P, ldp, Skv, Sq, gb, ld, DH, 1.0f, dv + (int64_t)b * Skv * ld + h * DH, ld);        // dv = P^T dout
}

// bytes of LDS the wide-head kernels need (forward: one matrix; backward: two matrices and two vectors); 0 = shape not served
size_t wide_lds_bytes(int Sq, int Skv, int dh, bool bwd) {
    if (dh % 4 != 0 || dh <= 32) return 0;
    const size_t n = (size_t)Sq * (Skv + 1) * (bwd ? 2 : 1);
    return n * sizeof(float) <= 64 * 1024 ? n * sizeof(float) : 0;
}

unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }
unsigned reduce_blocks(int64_t n) { return (unsigned)((n + 15) / 16); }

// rows per chunk (a multiple of 32) and number of chunks for a reduction over M rows
void chunking(int64_t M, int64_t* rows_per_chunk, int* n_chunks) {
    int64_t rpc = (M + 511) / 512;                                  // at most 512 chunks; short chunks keep small problems parallel
    rpc = ((rpc + 31) / 32) * 32;
    if (rpc < 32) rpc = 32;
    *rows_per_chunk = rpc;
    *n_chunks = (int)((M + rpc - 1) / rpc);
}

}  // namespace

// ================================================================================================ C ABI
ADX_API int adx_op_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return fail(ADX_EINVAL, "add: null pointer or empty shape");
    if (vec4_ok(n, a, b, out))
        hipLaunchKernelGGL(binary4_kernel<0>, dim3(blocks_for(n >> 2)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                           reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n >> 2);
    else
        hipLaunchKernelGGL(add_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return done(hipGetLastError(), "add");
}

ADX_API int adx_op_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return fail(ADX_EINVAL, "mul: null pointer or empty shape");
    if (vec4_ok(n, a, b, out))
        hipLaunchKernelGGL(binary4_kernel<1>, dim3(blocks_for(n >> 2)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                           reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n >> 2);
    else
        hipLaunchKernelGGL(mul_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return done(hipGetLastError(), "mul");
}

ADX_API int adx_op_act(const float* x, float* y, int64_t n, int act, void* stream) {
    if (!x || !y || n <= 0) return fail(ADX_EINVAL, "act: null pointer or empty shape");
    if (act < ADX_ACT_NONE || act > ADX_ACT_TANH) return fail(ADX_EINVAL, "act: unknown activation code");
    hipLaunchKernelGGL(act_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
    return done(hipGetLastError(), "act");
}

ADX_API int adx_op_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int act, void* stream) {
    if (!x || !dy || !dx || n <= 0) return fail(ADX_EINVAL, "act_bwd: null pointer or empty shape");
    if (act < ADX_ACT_NONE || act > ADX_ACT_TANH) return fail(ADX_EINVAL, "act_bwd: unknown activation code");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, act);
    return done(hipGetLastError(), "act_bwd");
}

ADX_API int adx_op_group_colsum(const float* x, int G, int64_t R, int N, float* out, void* stream) {
    if (!x || !out) return fail(ADX_EINVAL, "group_colsum: null pointer");
    if (G <= 0 || G > 65535 || R <= 0 || N <= 0) return fail(ADX_EINVAL, "group_colsum: empty shape or more than 65535 groups");
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, G), dim3(256), 0, (hipStream_t)stream, x, (int64_t)G * R, N, R, out);
    return done(hipGetLastError(), "group_colsum");
}

ADX_API int adx_op_gelu(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return fail(ADX_EINVAL, "gelu: null pointer or empty shape");
    hipLaunchKernelGGL(gelu_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return done(hipGetLastError(), "gelu");
}

ADX_API int adx_op_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    if (!x || !dy || !dx || n <= 0) return fail(ADX_EINVAL, "gelu_bwd: null pointer or empty shape");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
    return done(hipGetLastError(), "gelu_bwd");
}

ADX_API int adx_op_weighted_mse_bwd(const float* out, const float* target, const int64_t* t, const float* weight, const float* upstream,
                                    float* dout, int n_rows, int row_len, void* stream) {
    if (!out || !target || !t || !weight || !dout) return fail(ADX_EINVAL, "weighted_mse_bwd: null pointer");
    if (n_rows <= 0 || row_len <= 0) return fail(ADX_EINVAL, "weighted_mse_bwd: empty shape");
    const int64_t total = (int64_t)n_rows * row_len;
    hipLaunchKernelGGL(weighted_mse_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, out, target, t, weight, upstream,
                       dout, total, row_len, 1.0f / (float)total);
    return done(hipGetLastError(), "weighted_mse_bwd");
}

ADX_API int adx_op_transpose(const float* x, float* y, int rows, int cols, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return fail(ADX_EINVAL, "transpose: null pointer or empty shape");
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols);
    return done(hipGetLastError(), "transpose");
}

ADX_API size_t adx_reduce_ws_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 0) return 0;
    int64_t rpc; int nc;
    chunking(M, &rpc, &nc);
    return (size_t)nc * ((size_t)N * (K > 0 ? K : 1) + (K > 0 ? N : 0)) * sizeof(float);
}

ADX_API int adx_op_colsum(const float* x, int64_t M, int N, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace) return fail(ADX_EINVAL, "colsum: null pointer");
    if (M <= 0 || N <= 0) return fail(ADX_EINVAL, "colsum: empty shape");
    if (workspace_bytes < adx_reduce_ws_bytes(M, N, 0)) return fail(ADX_EINVAL, "colsum: workspace smaller than adx_reduce_ws_bytes(M, N, 0)");
    int64_t rpc; int nc;
    chunking(M, &rpc, &nc);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, nc), dim3(256), 0, (hipStream_t)stream, x, M, N, rpc, partial);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_blocks(N)), dim3(256), 0, (hipStream_t)stream, partial, nc, (int64_t)N, (int64_t)N, out,
                       accumulate);
    return done(hipGetLastError(), "colsum");
}

ADX_API int adx_op_linear_wgrad(const float* dy, const float* x, int64_t M, int N, int K, float* dw, float* db, int accumulate, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!dy || !x || !dw || !workspace) return fail(ADX_EINVAL, "linear_wgrad: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) return fail(ADX_EINVAL, "linear_wgrad: empty shape");
    if (workspace_bytes < adx_reduce_ws_bytes(M, N, K)) return fail(ADX_EINVAL, "linear_wgrad: workspace smaller than adx_reduce_ws_bytes(M, N, K)");
    int64_t rpc; int nc;
    chunking(M, &rpc, &nc);
    float* partial = static_cast<float*>(workspace);
    const int64_t stride = (int64_t)N * K + N;
    const int tiles = ((N + 31) / 32) * ((K + 31) / 32);
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(tiles, nc), dim3(256), 0, (hipStream_t)stream, dy, x, M, N, K, rpc, partial, stride, db ? 1 : 0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_blocks((int64_t)N * K)), dim3(256), 0, (hipStream_t)stream, partial, nc, stride,
                       (int64_t)N * K, dw, accumulate);
    if (db)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_blocks(N)), dim3(256), 0, (hipStream_t)stream, partial + (int64_t)N * K, nc, stride,
                           (int64_t)N, db, accumulate);
    return done(hipGetLastError(), "linear_wgrad");
}

ADX_API int adx_op_layernorm_bwd(const float* x, const float* gamma, const float* dy, int64_t rows, int d, float eps, float* dx, float* dy_xhat,
                                 void* stream) {
    if (!x || !dy || !dx) return fail(ADX_EINVAL, "layernorm_bwd: null pointer");
    if (rows <= 0 || d <= 0) return fail(ADX_EINVAL, "layernorm_bwd: empty shape");
    if (d <= 32)
        hipLaunchKernelGGL(layernorm_bwd_kernel<32>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, rows, d, eps,
                           dx, dy_xhat);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, rows, d, eps,
                           dx, dy_xhat);
    return done(hipGetLastError(), "layernorm_bwd");
}

// seed and site mixed into the 32-bit base of the mask hash (splitmix64 finaliser), the keep threshold and the scale of a rate p
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static bool drop_args(float p, uint64_t seed, int site, const float* host_mask, DropArgs* d, const char* who) {
    if (!(p >= 0.f) || p >= 1.f) { set_error(std::string(who) + ": dropout rate must lie in [0, 1)"); return false; }
    d->mask = host_mask;
    d->base = (uint32_t)splitmix64(seed ^ splitmix64((uint64_t)(uint32_t)site + 0x5851F42D4C957F2Dull));
    const double keep = 1.0 - (double)p;
    d->thr = keep >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(keep * 4294967296.0);
    d->scale = (float)(1.0 / keep);
    return true;
}

static int attention_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* dout, int B, int H, int Sq, int Skv,
                              int dh, float scale, int mask_off, float* dq, float* dk, float* dv, float* stats, int stats_have_lse, const DropArgs* drp,
                              void* stream) {
    if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !stats) return fail(ADX_EINVAL, "attention_bwd: null pointer");
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return fail(ADX_EINVAL, "attention_bwd: empty shape");
    if (B > 65535 || H > 65535) return fail(ADX_EINVAL, "attention_bwd: batch or head count above 65535");
    hipStream_t s = (hipStream_t)stream;
    const DropArgs dr = drp ? *drp : DropArgs();
    if (drp && wide_lds_bytes(Sq, Skv, dh, true)) return fail(ADX_EINVAL, "attention_bwd: attention dropout is built for head widths 4, 8, 16 and 32");
    if (const size_t lds = wide_lds_bytes(Sq, Skv, dh, true)) {
        hipLaunchKernelGGL(attn_wide_bwd_kernel, dim3(H, B), dim3(kWideThreads), lds, s, q, k, v, o, dout, stats, stats_have_lse, H, Sq, Skv, dh, scale,
                           mask_off, dq, dk, dv);
        return done(hipGetLastError(), "attention_bwd");
    }
    const int PTr = dh <= 16 ? 2 : 1;                               // queries (keys) per thread: 2 while the registers allow it
    const int nq = (Sq + 128 * PTr - 1) / (128 * PTr), nk = (Skv + 128 * PTr - 1) / (128 * PTr);
    const dim3 gq(attn_grid(B, H, nq)), gk(attn_grid(B, H, nk));
#define PT (DH_ <= 16 ? 2 : 1)
#define ADX_ATTN_BWD3(DH, MASKED, LSE)                                                                                                                  \
    if (drp) {                                                                                                                                           \
        hipLaunchKernelGGL((attn_bwd_query_kernel<DH, PT, MASKED, LSE, true>), gq, dim3(128), 0, s, q, k, v, o, dout, B, nq, H, Sq, Skv, scale, mask_off, dq, stats, dr);  \
        hipLaunchKernelGGL((attn_bwd_key_kernel<DH, PT, MASKED, true>), gk, dim3(128), 0, s, q, k, v, dout, stats, B, nk, H, Sq, Skv, scale, mask_off, dk, dv, dr);        \
    } else {                                                                                                                                             \
        hipLaunchKernelGGL((attn_bwd_query_kernel<DH, PT, MASKED, LSE, false>), gq, dim3(128), 0, s, q, k, v, o, dout, B, nq, H, Sq, Skv, scale, mask_off, dq, stats, dr); \
        hipLaunchKernelGGL((attn_bwd_key_kernel<DH, PT, MASKED, false>), gk, dim3(128), 0, s, q, k, v, dout, stats, B, nk, H, Sq, Skv, scale, mask_off, dk, dv, dr);       \
    }
#define ADX_ATTN_BWD2(DH, MASKED) if (stats_have_lse) { ADX_ATTN_BWD3(DH, MASKED, true) } else { ADX_ATTN_BWD3(DH, MASKED, false) }
#define ADX_ATTN_BWD(DH)                                              \
    case DH: {                                                        \
        constexpr int DH_ = DH;                                       \
        if (mask_off < 0) { ADX_ATTN_BWD2(DH, false) } else { ADX_ATTN_BWD2(DH, true) } \
        break; }
    switch (dh) {
        ADX_ATTN_BWD(4)
        ADX_ATTN_BWD(8)
        ADX_ATTN_BWD(16)
        ADX_ATTN_BWD(32)
        default: return fail(ADX_EINVAL, "attention_bwd: head width must be 4, 8, 16 or 32, or a multiple of 4 with at most ~88 tokens on either side");
    }
#undef ADX_ATTN_BWD
#undef ADX_ATTN_BWD2
#undef ADX_ATTN_BWD3
#undef PT
    return done(hipGetLastError(), "attention_bwd");
}

ADX_API int adx_op_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout, int B, int H, int Sq, int Skv,
                                 int dh, float scale, int mask_off, float* dq, float* dk, float* dv, float* stats, int stats_have_lse, void* stream) {
    return attention_bwd_impl(q, k, v, o, dout, B, H, Sq, Skv, dh, scale, mask_off, dq, dk, dv, stats, stats_have_lse, nullptr, stream);
}

ADX_API int adx_op_attention_bwd_dropout(const float* q, const float* k, const float* v, const float* o, const float* dout, int B, int H, int Sq,
                                         int Skv, int dh, float scale, int mask_off, float* dq, float* dk, float* dv, float* stats, int stats_have_lse,
                                         float p, uint64_t seed, int site, const float* host_mask, void* stream) {
    DropArgs d;
    if (!drop_args(p, seed, site, host_mask, &d, "attention_bwd_dropout")) return ADX_EINVAL;
    return attention_bwd_impl(q, k, v, o, dout, B, H, Sq, Skv, dh, scale, mask_off, dq, dk, dv, stats, stats_have_lse, &d, stream);
}

static int attention_stats_impl(const float* q, const float* k, const float* v, float* o, float* stats, int B, int H, int Sq, int Skv, int dh,
                                float scale, int mask_off, const DropArgs* drp, void* stream) {
    if (!q || !k || !v || !o || !stats) return fail(ADX_EINVAL, "attention_stats: null pointer");
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return fail(ADX_EINVAL, "attention_stats: empty shape");
    if (B > 65535 || H > 65535) return fail(ADX_EINVAL, "attention_stats: batch or head count above 65535");
    hipStream_t s = (hipStream_t)stream;
    const DropArgs dr = drp ? *drp : DropArgs();
    if (drp && wide_lds_bytes(Sq, Skv, dh, false)) return fail(ADX_EINVAL, "attention_stats: attention dropout is built for head widths 4, 8, 16 and 32");
    if (const size_t lds = wide_lds_bytes(Sq, Skv, dh, false)) {
        hipLaunchKernelGGL(attn_wide_fwd_kernel, dim3(H, B), dim3(kWideThreads), lds, s, q, k, v, H, Sq, Skv, dh, scale, mask_off, o, stats);
        return done(hipGetLastError(), "attention_stats");
    }
    const int PTr = dh <= 16 ? 2 : 1;
    const int nq = (Sq + 128 * PTr - 1) / (128 * PTr);
    const dim3 gq(attn_grid(B, H, nq));
#define ADX_ATTN_FWD(DH)                                                                                                                         \
    case DH: {                                                                                                                                   \
        constexpr int PT = DH <= 16 ? 2 : 1;                                                                                                     \
        if (drp) {                                                                                                                               \
            if (mask_off < 0) hipLaunchKernelGGL((attn_fwd_stats_kernel<DH, PT, false, true>), gq, dim3(128), 0, s, q, k, v, B, nq, H, Sq, Skv, scale, mask_off, o, stats, dr); \
            else hipLaunchKernelGGL((attn_fwd_stats_kernel<DH, PT, true, true>), gq, dim3(128), 0, s, q, k, v, B, nq, H, Sq, Skv, scale, mask_off, o, stats, dr);     \
        } else if (mask_off < 0) hipLaunchKernelGGL((attn_fwd_stats_kernel<DH, PT, false, false>), gq, dim3(128), 0, s, q, k, v, B, nq, H, Sq, Skv, scale, mask_off, o, stats, dr); \
        else hipLaunchKernelGGL((attn_fwd_stats_kernel<DH, PT, true, false>), gq, dim3(128), 0, s, q, k, v, B, nq, H, Sq, Skv, scale, mask_off, o, stats, dr);     \
        break; }
    switch (dh) {
        ADX_ATTN_FWD(4)
        ADX_ATTN_FWD(8)
        ADX_ATTN_FWD(16)
        ADX_ATTN_FWD(32)
        default: return fail(ADX_EINVAL, "attention_stats: head width must be 4, 8, 16 or 32, or a multiple of 4 with at most ~88 tokens on either side");
    }
#undef ADX_ATTN_FWD
    return done(hipGetLastError(), "attention_stats");
}

ADX_API int adx_op_attention_stats(const float* q, const float* k, const float* v, float* o, float* stats, int B, int H, int Sq, int Skv, int dh,
                                   float scale, int mask_off, void* stream) {
    return attention_stats_impl(q, k, v, o, stats, B, H, Sq, Skv, dh, scale, mask_off, nullptr, stream);
}

ADX_API int adx_op_attention_stats_dropout(const float* q, const float* k, const float* v, float* o, float* stats, int B, int H, int Sq, int Skv,
                                           int dh, float scale, int mask_off, float p, uint64_t seed, int site, const float* host_mask, void* stream) {
    DropArgs d;
    if (!drop_args(p, seed, site, host_mask, &d, "attention_stats_dropout")) return ADX_EINVAL;
    return attention_stats_impl(q, k, v, o, stats, B, H, Sq, Skv, dh, scale, mask_off, &d, stream);
}

ADX_API int adx_op_attention_dropout_mask(float* out, int B, int H, int Sq, int Skv, float p, uint64_t seed, int site, void* stream) {
    if (!out || B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return fail(ADX_EINVAL, "attention_dropout_mask: null pointer or empty shape");
    DropArgs d;
    if (!drop_args(p, seed, site, nullptr, &d, "attention_dropout_mask")) return ADX_EINVAL;
    const int64_t n = (int64_t)B * H * Sq * Skv;
    hipLaunchKernelGGL(attention_dropout_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, B * H, Sq, Skv, d);
    return done(hipGetLastError(), "attention_dropout_mask");
}

ADX_API int adx_op_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, int site, void* stream) {
    if (!y || n <= 0) return fail(ADX_EINVAL, "dropout: null pointer or empty shape");
    DropArgs d;
    if (!drop_args(p, seed, site, nullptr, &d, "dropout")) return ADX_EINVAL;
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, d.base, d.thr, d.scale);
    return done(hipGetLastError(), "dropout");
}

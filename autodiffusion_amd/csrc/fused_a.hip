// Fused per-row pipeline of the variant-A conditioning encoder (d_model 32, 8 heads, FFN 128), gfx950.
//
// One launch per diffusion step replaces, for all B*L conditioning rows:
//     layer 0:  fold the time-token key into the cached softmax state  ->  out_proj + residual  ->  LayerNorm
//               ->  FFN (GELU)  + residual
//     layer 1:  LayerNorm  ->  K/V projection
// Rows are independent, so every wavefront owns a tile of 32 rows and keeps the whole chain in registers.
//
// Layout.  All products are computed TRANSPOSED, Y^T = W X^T, with v_mfma 32x32: the weight tile is the A operand
// (32 output features x k), the activations are the B operand (k x 32 rows).  The accumulator of a 32x32 MFMA holds,
// in lane l (c = l & 31, h = l >> 5) and register reg, feature  f(reg, h) = (reg & 3) + 8 (reg >> 2) + 4 h  of row c.
// An accumulator is therefore already "k on registers, row on the lane" and feeds the next product as B operand with
// no data movement:
//   * PREC_F32   : v_mfma_f32_32x32x2_f32, one instruction per register (k pair = features f(reg,0), f(reg,1));
//   * PREC_BF16X3: v_mfma_f32_32x32x16_bf16 on registers 8s..8s+7 converted to bf16 (k order inside the step is
//                  16s + 8 (j >> 2) + 4h + (j & 3)); operands split in high + low halves, three products.
// The weight fragments are pre-arranged in exactly that k order by build_image_kernel and staged in LDS once per
// workgroup.  LayerNorm reduces 16 registers in-lane plus one exchange with lane l ^ 32.
//
// Rows r < n_keep of every batch item (the ones the decoder can see) also get the layer-1 query projection.
//
// HBM traffic per row and step (f32): read x0 128 B, q 128 B, softmax state 160 B; write k/v 256 B  -> 672 B.
#include "adx_ops.h"
#include "fastmath.h"

#include <math.h>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

namespace adx {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

namespace {

constexpr int D = 32, NH = 8;
constexpr float LOG2E_F = 1.4426950408889634f;

// ---- fragment image -------------------------------------------------------------------------------------------
// matrices in image order: Wo (32x32), W1 (128x32), W2 (32x128), Wkv (64x32), Wq (32x32, layer 1)
// F32   : fragment (mat, nt, it, reg)  = 64 floats: lane l -> W[32 nt + (l & 31)][32 it + f(reg, l >> 5)]
// BF16X3: fragment (mat, nt, it, s)    = 64 lanes x 8 bf16 (hi image, then lo image)
constexpr int NFRAG_F32 = 16 * (1 + 4 + 4 + 2 + 1);      // 192 fragments of 256 B
constexpr int NFRAG_BF = 2 * (1 + 4 + 4 + 2 + 1);        // 24 fragments of 1 KiB per half
constexpr int IMG_F32_FLOATS = NFRAG_F32 * 64;           // 12288 floats = 49152 B
constexpr int ROWS_LDS_IMG_H = (NFRAG_BF + 2) * 64 * 8 * 2;  // 26624 B: the LDS copy of the half-width row kernel (all high halves + one tile of low halves)
constexpr int IMG_BF_SHORTS = 2 * NFRAG_BF * 64 * 8;     // hi + lo: 24576 shorts = 49152 B
constexpr int VEC_FLOATS = 32 * 6 + 128 + 64 + 32;       // bo g2 be2 b2 g1 be1 (6 x 32) + b1 (128) + bkv (64) + bq (32) = 416

// base fragment index of each matrix (in units of "nt, it" tiles)
__host__ __device__ constexpr int tile_base(int mat) { return mat == 0 ? 0 : (mat == 1 ? 1 : (mat == 2 ? 5 : (mat == 3 ? 9 : 11))); }
// number of input tiles (K / 32) of each matrix
__host__ __device__ constexpr int in_tiles(int mat) { return mat == 2 ? 4 : 1; }

__device__ __forceinline__ int feat(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ unsigned short f2bf_(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f_(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct ImageSrc {
    const float* W[5];   // Wo [32][32], W1 [128][32], W2 [32][128], Wkv [64][32], Wq [32][32]
};

__global__ void build_image_kernel(ImageSrc src, float* img_f32, unsigned short* img_bf, int f16_mats) {
    const int nt_count[5] = {1, 4, 1, 2, 1};
    const int Kdim[5] = {32, 32, 128, 32, 32};
    int tile = 0;
    for (int mat = 0; mat < 5; ++mat)
        for (int nt = 0; nt < nt_count[mat]; ++nt)
            for (int it = 0; it < in_tiles(mat); ++it, ++tile) {
                const float wscale = mat == 2 ? 0.5f : 1.0f;      // the row kernel's activation returns twice the GELU
                for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 16 * 64; e += gridDim.x * blockDim.x) {
                    const int reg = e >> 6, l = e & 63;
                    img_f32[(tile * 16 + reg) * 64 + l] = wscale * src.W[mat][(32 * nt + (l & 31)) * Kdim[mat] + 32 * it + feat(reg, l >> 5)];
                }
                for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * 64 * 8; e += gridDim.x * blockDim.x) {
                    const int s = e >> 9, l = (e >> 3) & 63, j = e & 7;
                    const int k = 32 * it + 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                    const float w = wscale * src.W[mat][(32 * nt + (l & 31)) * Kdim[mat] + k];
                    const int idx = ((tile * 2 + s) * 64 + l) * 8 + j;
                    if ((f16_mats >> mat) & 1) {                // this matrix as fp16 halves (the mixed modes read the high ones)
                        const _Float16 hh = (_Float16)w, hl = (_Float16)(w - (float)hh);
                        img_bf[idx] = __builtin_bit_cast(unsigned short, hh);
                        img_bf[NFRAG_BF * 64 * 8 + idx] = __builtin_bit_cast(unsigned short, hl);
                        continue;
                    }
                    const unsigned short hi = f2bf_(w);
                    img_bf[idx] = hi;
                    img_bf[NFRAG_BF * 64 * 8 + idx] = f2bf_(w - bf2f_(hi));
                }
            }
}

// ---- Y^T tile += W[mat][nt][it] X^T, input in accumulator layout ----------------------------------------------
// lo_off: where the low halves of this tile's fragments live, in shorts from its high halves (the full image keeps all low halves
// behind all high halves; the compact LDS copy of the half-width row kernel keeps only the one tile of low halves it needs)
template <int PREC>
__device__ __forceinline__ f32x16 mm(const unsigned char* lds_img, int tile, const f32x16& in, f32x16 acc, int lane, int lo_off = NFRAG_BF * 64 * 8) {
    if constexpr (PREC == PREC_F32) {
        const float* w = reinterpret_cast<const float*>(lds_img) + tile * 16 * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[reg * 64], in[reg], acc, 0, 0, 0);
    } else {
        const unsigned short* whi = reinterpret_cast<const unsigned short*>(lds_img) + (tile * 2 * 64 + lane) * 8;
        const unsigned short* wlo = whi + lo_off;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 bhv, blv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = in[8 * s + j];
                const __bf16 hi = (__bf16)v;              // v_cvt_pk_bf16_f32, round to nearest even
                bhv[j] = hi;
                blv[j] = (__bf16)(v - (float)hi);
            }
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(whi + s * 64 * 8);
            if constexpr (PREC == PREC_BF16X3) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wlo + s * 64 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blv, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhv, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhv, acc, 0, 0, 0);
        }
    }
    return acc;
}

// The split-bf16 product with the weight fragments fetched separately, so that the LDS reads can be issued well before
// the MFMAs that consume them (ahead of a block of vector work) instead of right in front of them.
struct WFrag { bf16x8 ah[2], al[2]; };

__device__ __forceinline__ WFrag load_w(const unsigned char* lds_img, int tile, int lane) {
    const unsigned short* whi = reinterpret_cast<const unsigned short*>(lds_img) + (tile * 2 * 64 + lane) * 8;
    const unsigned short* wlo = whi + NFRAG_BF * 64 * 8;
    WFrag w;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        w.ah[s] = *reinterpret_cast<const bf16x8*>(whi + s * 64 * 8);
        w.al[s] = *reinterpret_cast<const bf16x8*>(wlo + s * 64 * 8);
    }
    return w;
}

__device__ __forceinline__ f32x16 mm_w(const WFrag& w, const f32x16& in, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8 bhv, blv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = in[8 * s + j];
            const __bf16 hi = (__bf16)v;
            bhv[j] = hi;
            blv[j] = (__bf16)(v - (float)hi);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.ah[s], blv, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.al[s], bhv, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.ah[s], bhv, acc, 0, 0, 0);
    }
    return acc;
}

// The same product with BOTH operands rounded once to fp16 (one matrix instruction per 16 k, one conversion per pair of values
// instead of six vector instructions): w.ah holds fp16 bit patterns here (build_image_kernel with w2_f16).  fp16 keeps 11
// significand bits against bf16's 8; the values it is used on are bounded (GELU outputs, softmax weights <= 1, LayerNorm rows).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ f16x8 to_f16x8(const f32x16& in, int s) {
    f16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)in[8 * s + j];
    return b;
}

__device__ __forceinline__ f32x16 mm_h(const unsigned char* lds_img, int tile, const f32x16& in, f32x16 acc, int lane) {
    const unsigned short* whi = reinterpret_cast<const unsigned short*>(lds_img) + (tile * 2 * 64 + lane) * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(whi + s * 64 * 8), to_f16x8(in, s), acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x16 mm_w_h(const WFrag& w, const f32x16& in, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.ah[s]), to_f16x8(in, s), acc, 0, 0, 0);
    return acc;
}

// GELU(v) = v Phi(v) = max(v, 0) - |v| Phi(-|v|), with Phi(-t) = 2^q(t) and q a degree-5 polynomial fitted to
// log2 Phi(-t) on [0, 5.6] (minimax for the error of the product; beyond 5.6 the term is below 1e-7 and t is
// clamped).  |error| <= 9e-7 over the whole line evaluated in fp32 (4.6e-7 in exact arithmetic; a sixth-degree fit reaches
// 5e-7 in fp32, so the last coefficient bought nothing), with one transcendental and no division: the exponential and the
// reciprocal of the erfc formulas are quarter-rate instructions and this kernel is bound by vector issue.  (A cubic q, within 5.5e-5 of
// the exact function, was tried for the fp16 form, whose product rounds the GELU's output to fp16 anyway: 2 % faster, and the ADE against
// fp32 went from 1.1e-3 to 2.5e-3 m median, 3.8e-3 to 7.0e-3 m worst -- a systematic error of the function is not rounding noise.)
typedef __attribute__((ext_vector_type(2))) float f32x2;

// No clamp is needed past the fitted range: the polynomial keeps falling (its leading coefficient is negative), 2^q underflows
// to zero and the term stays below 6e-8 for every t >= 5.6.
// Both functions return TWICE the GELU, v + |v| (1 - 2 Phi(-|v|)) (max(v, 0) = (v + |v|) / 2): two packed instructions per pair
// after the exponential instead of three; the weight image holds W2 / 2 (build_image_kernel; exact, a power of two).  For v < 0 the
// two terms cancel to -2 |v| Phi(-|v|) with an absolute error of at most half an ulp of |v|.
// Plain FMAs with |v| as a source modifier of each: measured on this chip (profiles/micro/valu_issue.hip, four waves per SIMD)
// a packed fp32 FMA issues in 1.9 x the time of a plain one, so packing buys 5 % and costs the instruction that forms |v|.
__device__ __forceinline__ float gelu2_(float v) {
    float q = fmaf(fabsf(v), -4.72791765e-04f, 7.08039424e-03f);
    q = fmaf(fabsf(v), q, -5.18154655e-02f);
    q = fmaf(fabsf(v), q, -4.60007281e-01f);
    q = fmaf(fabsf(v), q, -1.15078008e+00f);
    q = fmaf(fabsf(v), q, -1.00003892e+00f);
    float r = fmaf(fabsf(v), fmaf(-2.0f, exp2_fast(q), 1.0f), v);
    asm("" : "+v"(r));          // opaque to the vectoriser, which would pair the last two steps into packed FMAs and form |v| for them
    return r;
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) { return f32x2{gelu2_(v.x), gelu2_(v.y)}; }

// the same function on one value with plain (unpacked) FMAs: measurement aid for the packed form
__device__ __forceinline__ float gelu_erf1(float v) {
    const float av = fabsf(v), t = av;
    float q = fmaf(t, -4.72791765e-04f, 7.08039424e-03f);
    q = fmaf(t, q, -5.18154655e-02f);
    q = fmaf(t, q, -4.60007281e-01f);
    q = fmaf(t, q, -1.15078008e+00f);
    q = fmaf(t, q, -1.00003892e+00f);
    return fmaf(t, fmaf(-2.0f, exp2_fast(q), 1.0f), v);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// LayerNorm over the 32 features of each row: 16 in this lane, 16 in lane ^ 32
__device__ __forceinline__ f32x16 layer_norm32(const f32x16& y, const float* gamma, const float* beta, int h) {
    // written on pairs: the sums, the centring and the affine map as packed instructions (41 instead of ~60 per call)
    f32x2 s2 = {y[0], y[1]};
#pragma unroll
    for (int i = 2; i < 16; i += 2) s2 += f32x2{y[i], y[i + 1]};
    const float mean = xor32_add(s2.x + s2.y) * (1.0f / 32.0f);
    f32x2 d[8], q2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = f32x2{y[2 * i], y[2 * i + 1]} - f32x2{mean, mean}; q2 += d[i] * d[i]; }
    const float rstd = rsqrt_fast(fmaf(xor32_add(q2.x + q2.y), 1.0f / 32.0f, 1e-5f));
    f32x16 o;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 8 * g + 4 * h);
        const float4 be = *reinterpret_cast<const float4*>(beta + 8 * g + 4 * h);
        const f32x2 a = d[2 * g] * f32x2{rstd, rstd} * f32x2{ga.x, ga.y} + f32x2{be.x, be.y};
        const f32x2 b = d[2 * g + 1] * f32x2{rstd, rstd} * f32x2{ga.z, ga.w} + f32x2{be.z, be.w};
        o[4 * g + 0] = a.x; o[4 * g + 1] = a.y; o[4 * g + 2] = b.x; o[4 * g + 3] = b.y;
    }
    return o;
}

// Inputs of one row, as the lane that owns (row, half h) needs them.
struct RowIn {
    float4 x[4];      // x0 features 8g + 4h .. +3
    float4 q[4];      // layer-0 query of heads 2g + h
    float4 st[5];     // normalised softmax states of heads 2g + h: 4 x {c, o0..o3}, contiguous (see pack_rows_kernel)
};
constexpr int PACK_VECS = 13;   // float4 per lane per 32-row tile: exactly the 416 B a row contributes

// The step-independent inputs of a tile are re-read on every diffusion step, so they are stored once per chunk in the
// order the lanes consume them: packed[(tile * 13 + j) * 64 + lane].  Every load instruction of the row kernel then
// covers 1 KB of contiguous memory (eight full 128-B lines) instead of 32 B out of each of 32 lines.
// The cached softmax state {m, l, o[4]} of a head (scores in the log2 domain) is stored normalised, as
// {c = m + log2 l, o / l}: folding one more key (score s, value v) into it is then
//     out = o/l + g (v - o/l),   g = 1 / (1 + 2^(c - s)),
// which needs five floats instead of six and one exponential instead of two, and cannot overflow to NaN
// (2^(c - s) = inf gives g = 0).
// x0 is packed with the bias of the attention out-projection already added (vecs[0 .. 31]): the row kernel starts the
// out-projection's accumulators from it, and the residual and bias additions cost it no instruction.
__device__ __forceinline__ float4 x0_plus_bias(const EncRowsArgs& a, int64_t row, int f) {
    const float4 x = *reinterpret_cast<const float4*>(a.x0 + row * D + f), b = *reinterpret_cast<const float4*>(a.vecs + f);
    return make_float4(x.x + b.x, x.y + b.y, x.z + b.z, x.w + b.w);
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const EncRowsArgs a, float4* __restrict__ packed, int64_t n_tiles) {
    // one thread per (tile, lane): the 13 vectors that lane will load
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tiles * 64) return;
    const int lane = (int)(i & 63);
    const int64_t tile = i >> 6;
    const int64_t n_rows = (int64_t)a.B * a.L;
    const int64_t row = min(tile * 32 + (lane & 31), n_rows - 1);
    const int h = lane >> 5;
    float4* dst = packed + tile * (PACK_VECS * 64) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dst[64 * g] = x0_plus_bias(a, row, 4 * h + 8 * g);
        dst[64 * (4 + g)] = *reinterpret_cast<const float4*>(a.q + row * a.ldq + 4 * h + 8 * g);
    }
    const float4* sp = reinterpret_cast<const float4*>(a.state + row * (NH * 6) + h * 24);      // 96 bytes: 4 x {m, l, o[4]}
    float st[24], e[20];
#pragma unroll
    for (int v = 0; v < 6; ++v) { const float4 t = sp[v]; st[4 * v] = t.x; st[4 * v + 1] = t.y; st[4 * v + 2] = t.z; st[4 * v + 3] = t.w; }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float inv = rcp_fast(st[6 * g + 1]);
        e[5 * g] = st[6 * g] + __builtin_amdgcn_logf(st[6 * g + 1]);      // v_log_f32 is log2, 1 ulp
#pragma unroll
        for (int k = 0; k < 4; ++k) e[5 * g + 1 + k] = st[6 * g + 2 + k] * inv;
    }
#pragma unroll
    for (int v = 0; v < 5; ++v) dst[64 * (8 + v)] = make_float4(e[4 * v], e[4 * v + 1], e[4 * v + 2], e[4 * v + 3]);
}

// Gate-table form: the only step-dependent part of the fold, the gate g of each head, is tabulated per chunk for every
// step (pack_gates_kernel), so a step reads x0 (128 B), o / l (128 B) and four gates (32 B per row: 16 B per lane) --
// 288 B per row instead of 416 -- and evaluates no exponential or reciprocal for the fold.
struct RowInG {
    float4 x[4];      // x0 features 8g + 4h .. +3
    float4 o[4];      // o / l of heads 2g + h
    float4 gate;      // gates of heads h, 2 + h, 4 + h, 6 + h for this step
};
constexpr int PACK_VECS_G = 8;

__device__ __forceinline__ float4 ld_stream(const float4* p) {       // read once per step, 240 MB per launch: do not keep it in the caches
    typedef __attribute__((ext_vector_type(4))) float v4;
    const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ void load_row(RowInG& in, const float4* __restrict__ p, const float4* __restrict__ gp) {
#pragma unroll
    for (int g = 0; g < 4; ++g) in.x[g] = ld_stream(p + 64 * g);
#pragma unroll
    for (int g = 0; g < 4; ++g) in.o[g] = ld_stream(p + 64 * (4 + g));
    in.gate = ld_stream(gp);
}

__device__ __forceinline__ void land_row(RowInG& in) {
    typedef __attribute__((ext_vector_type(4))) float v4;
    v4* p = reinterpret_cast<v4*>(&in);
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]));
}

// Half-width form of the same inputs and of the output (the fp16 mode only: every one of these values is rounded to fp16 by the
// product that consumes it anyway): o / l as sixteen fp16 per lane (two 16-byte vectors), the four gates as fp16 (8 bytes), the
// LayerNorm rows stored as fp16 -- x0 (128) + o / l (64) + gates (16) read and 64 written = 272 B per row instead of 416.
struct RowInH {
    float4 x[4];      // x0 features 8g + 4h .. +3 (the residual stream stays fp32)
    uint4 o[2];       // o / l of heads 2g + h as fp16: values 4g + i in order
    uint2 gate;       // gates of heads h, 2 + h, 4 + h, 6 + h as fp16
};
constexpr int PACK_VECS_H = 6;        // 16-byte vectors per lane per tile

__device__ __forceinline__ void load_row(RowInH& in, const float4* __restrict__ p, const uint2* __restrict__ gp) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
#pragma unroll
    for (int g = 0; g < 4; ++g) in.x[g] = ld_stream(p + 64 * g);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + 64 * (4 + v)));
        in.o[v] = make_uint4(t.x, t.y, t.z, t.w);
    }
    const u2 t = __builtin_nontemporal_load(reinterpret_cast<const u2*>(gp));
    in.gate = make_uint2(t.x, t.y);
}

__device__ __forceinline__ void land_row(RowInH& in) {
    typedef __attribute__((ext_vector_type(4))) float v4;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    v4* p = reinterpret_cast<v4*>(in.x);
    u4* o = reinterpret_cast<u4*>(in.o);
    u2* g = reinterpret_cast<u2*>(&in.gate);
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(o[0]), "+v"(o[1]), "+v"(g[0]));
}

// One thread per (tile, lane): the step-independent inputs in lane order (x0, o / l) and, for every step of the schedule,
// the gates of the lane's four heads: g = 1 / (1 + 2^(c - s)), c = m + log2 l of the cached state, s the scaled score of the
// step's time-token key against the row's layer-0 query -- the very expressions of the in-kernel fold, so both forms of the
// row kernel return the same bits.  tq: the per-step q | k | v rows of the time token ([n_steps][96], keys at +32).
template <bool HALF>     // HALF: the layout of RowInH (o / l and the gates as fp16)
__global__ __launch_bounds__(256) void pack_gates_kernel(const EncRowsArgs a, float4* __restrict__ packed, float4* __restrict__ gates, int64_t n_tiles,
                                                         const float* __restrict__ tq, int n_steps, float sc) {
    constexpr int CH = 128;                                   // steps staged in LDS at a time
    __shared__ float ks[CH * D];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_tiles * 64;
    const int lane = (int)(i & 63);
    const int64_t tile = live ? i >> 6 : 0;
    const int64_t n_rows = (int64_t)a.B * a.L;
    const int64_t row = min(tile * 32 + (lane & 31), n_rows - 1);
    const int h = lane >> 5;
    float4 q[4];
    float c[4];
    {
        float4* dst = packed + tile * ((HALF ? PACK_VECS_H : PACK_VECS_G) * 64) + lane;
        const float4* sp = reinterpret_cast<const float4*>(a.state + row * (NH * 6) + h * 24);      // 96 bytes: 4 x {m, l, o[4]}
        float st[24];
#pragma unroll
        for (int v = 0; v < 6; ++v) { const float4 t = sp[v]; st[4 * v] = t.x; st[4 * v + 1] = t.y; st[4 * v + 2] = t.z; st[4 * v + 3] = t.w; }
        _Float16 oh[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            q[g] = *reinterpret_cast<const float4*>(a.q + row * a.ldq + 4 * h + 8 * g);
            const float inv = rcp_fast(st[6 * g + 1]);
            c[g] = st[6 * g] + __builtin_amdgcn_logf(st[6 * g + 1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) oh[4 * g + k] = f16_sat(st[6 * g + 2 + k] * inv);
            if (live) {
                dst[64 * g] = x0_plus_bias(a, row, 4 * h + 8 * g);
                if constexpr (!HALF) dst[64 * (4 + g)] = make_float4(st[6 * g + 2] * inv, st[6 * g + 3] * inv, st[6 * g + 4] * inv, st[6 * g + 5] * inv);
            }
        }
        if constexpr (HALF) {
            if (live) {
                f16x8 lo, hi;
#pragma unroll
                for (int k = 0; k < 8; ++k) { lo[k] = oh[k]; hi[k] = oh[8 + k]; }
                reinterpret_cast<f16x8*>(dst)[64 * 4] = lo;
                reinterpret_cast<f16x8*>(dst)[64 * 5] = hi;
            }
        }
    }
    for (int s0 = 0; s0 < n_steps; s0 += CH) {
        const int ns = min(CH, n_steps - s0);
        __syncthreads();
        for (int e = threadIdx.x; e < ns * D; e += 256) ks[e] = tq[(int64_t)(s0 + e / D) * (3 * D) + D + (e % D)] * sc;
        __syncthreads();
        if (live) {
            for (int k = 0; k < ns; ++k) {
                float gt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 kk = *reinterpret_cast<const float4*>(ks + k * D + 8 * g + 4 * h);
                    const float sv = fmaf(q[g].x, kk.x, fmaf(q[g].y, kk.y, fmaf(q[g].z, kk.z, q[g].w * kk.w)));
                    gt[g] = rcp_fast(1.0f + exp2_fast(c[g] - sv));
                }
                // streamed: written once here, read once by the step that uses it, 26 MB apart from its neighbours in time
                typedef __attribute__((ext_vector_type(4))) float v4;
                if constexpr (HALF) {
                    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
                    typedef __attribute__((ext_vector_type(2))) unsigned u2;
                    const h4 gh = {(_Float16)gt[0], (_Float16)gt[1], (_Float16)gt[2], (_Float16)gt[3]};
                    __builtin_nontemporal_store(__builtin_bit_cast(u2, gh), reinterpret_cast<u2*>(gates) + ((int64_t)(s0 + k) * n_tiles + tile) * 64 + lane);
                } else
                __builtin_nontemporal_store(v4{gt[0], gt[1], gt[2], gt[3]}, reinterpret_cast<v4*>(gates + ((int64_t)(s0 + k) * n_tiles + tile) * 64 + lane));
            }
        }
    }
}

__device__ __forceinline__ void load_row(RowIn& in, const float4* __restrict__ p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) in.x[g] = p[64 * g];
#pragma unroll
    for (int g = 0; g < 4; ++g) in.q[g] = p[64 * (4 + g)];
#pragma unroll
    for (int i = 0; i < 5; ++i) in.st[i] = p[64 * (8 + i)];
}

// Makes the loaded row "used" at this point of the program, so the wait for its loads is placed HERE, in straight-line code,
// and is counted exactly (everything issued after the loads -- the stores of the tile -- may still be in flight).  Left to
// the first real use at the top of the tile loop, the wait is merged with the loop-entry state and also covers those stores.
__device__ __forceinline__ void land_row(RowIn& in) {
    typedef __attribute__((ext_vector_type(4))) float v4;
    v4* p = reinterpret_cast<v4*>(&in);
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]),
                      "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]));
}

// WAVES = resident wavefronts per SIMD the register budget is set for; a workgroup has 4 waves (WAVES <= 3, three
// workgroups per CU) or 8 waves (WAVES = 4, two workgroups per CU), so that the LDS copies of the weight image fit.
// GOUT: the kernel ends at the layer-1 LayerNorm and stores those rows (lane order, one float4 per lane and register group:
// four stores of 1 KB of contiguous memory per tile) for attention_g instead of projecting and storing K | V.
// ABL: measurement aid (results are wrong except for 16): 1 the FFN activation is the identity, 2 no FFN products, 4 no stores of
// the row output, 8 every tile re-reads the wave's first tile (L2 hits instead of HBM), 16 GELU with unpacked FMAs (same bits)
// MIX (split-bf16 mode only), bits: products with both operands rounded once to fp16 -- 1 the second FFN product, 2 the attention
// out-projection, 4 the first FFN product (the image must have been built with the same matrices as fp16: bit m <-> matrix m of the image);
// 8 (with GOUT and GATES) the half-width inputs and output of RowInH
template <int PREC, int WAVES, bool GOUT, bool GATES, int ABL = 0, int MIX = 0>
__global__ __launch_bounds__(WAVES == 4 ? 512 : 256, WAVES) void enc_rows_kernel(const EncRowsArgs a) {
    constexpr int NT = WAVES == 4 ? 512 : 256;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool HIO = (MIX & 8) != 0;
    // LDS copy of the weight image: 49152 B, or ROWS_LDS_IMG_H for the half-width form, which reads the high halves (fp16 Wo, W1, W2 and
    // bf16 Wq) and the low halves of Wq only -- that leaves room for the tail's workgroups on the same CU (three workgroups of this
    // kernel with full images fill 147 of the 160 KB, and the tail, which runs beside it on the second stream, found no CU to start on)
    constexpr int IMG_LDS = HIO ? ROWS_LDS_IMG_H : IMG_F32_FLOATS * 4;
    constexpr int Q_LO_OFF = HIO ? 2 * 64 * 8 : NFRAG_BF * 64 * 8;              // low halves of Wq: the tile behind it / the full image's place
    unsigned char* img = smem;
    float* vec = reinterpret_cast<float*>(smem + IMG_LDS);                      // VEC_FLOATS floats, then k0 (scaled) and v0
    const float sc = a.scale * LOG2E_F;
    {
        const uint4* src = reinterpret_cast<const uint4*>(PREC == PREC_F32 ? (const void*)a.img_f32 : (const void*)a.img_bf);
        uint4* dst = reinterpret_cast<uint4*>(img);
        if constexpr (HIO) {
            constexpr int HI16 = NFRAG_BF * 64 * 8 * 2 / 16, TILE16 = 2 * 64 * 8 * 2 / 16;     // uint4 per region / per tile
            static_assert(ROWS_LDS_IMG_H == (HI16 + TILE16) * 16, "compact image = all high halves + one tile of low halves");
            for (int i = threadIdx.x; i < HI16; i += NT) dst[i] = src[i];
            for (int i = threadIdx.x; i < TILE16; i += NT) dst[HI16 + i] = src[HI16 + tile_base(4) * TILE16 + i];
        } else
        for (int i = threadIdx.x; i < IMG_F32_FLOATS / 4; i += NT) dst[i] = src[i];
        for (int i = threadIdx.x; i < VEC_FLOATS; i += NT) vec[i] = a.vecs[i];
        if (threadIdx.x < 64 && (!HIO || threadIdx.x >= 32)) vec[VEC_FLOATS + threadIdx.x] = a.key0[threadIdx.x] * (threadIdx.x < 32 ? sc : 1.0f);
        if constexpr (HIO) {
            // half-width form: the gates come from the table, so the key's place holds the time token's value as fp16 in the order a lane's
            // operand registers want it: [half h][k-step s][j] <-> feature 8 (2 s + (j >> 2)) + 4 h + (j & 3)
            if (threadIdx.x < 32) {
                const int hh = threadIdx.x >> 4, ss = (threadIdx.x >> 3) & 1, j = threadIdx.x & 7;
                reinterpret_cast<_Float16*>(vec + VEC_FLOATS)[threadIdx.x] = (_Float16)a.key0[32 + 8 * (2 * ss + (j >> 2)) + 4 * hh + (j & 3)];
            }
        }
    }
    __syncthreads();
    const float *bo = vec, *g2 = vec + 32, *be2 = vec + 64, *b2 = vec + 96, *g1 = vec + 128, *be1 = vec + 160, *b1 = vec + 192,
                *bkv = vec + 320, *bq = vec + 384, *k0s = vec + VEC_FLOATS, *v0s = vec + VEC_FLOATS + 32;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave_global = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), n_waves = gridDim.x * (NT / 64);
    const int64_t n_rows = (int64_t)a.B * a.L;

    // The inputs of a wave's next tile are requested while the current tile is in its last phase (layer-1 LayerNorm and
    // projections), where few registers are live: the loads are in flight during a quarter of the tile time instead of
    // only while the wave waits for them, which is what keeps enough bytes in flight to use the HBM.
    const int64_t last_tile = (n_rows - 1) >> 5;
    static_assert(!HIO || (GATES && GOUT), "the half-width form exists for the gate-table, LayerNorm-row kernel only");
    typename std::conditional<HIO, RowInH, typename std::conditional<GATES, RowInG, RowIn>::type>::type in;
    auto fetch_row = [&](int64_t tile) {
        if constexpr (HIO) load_row(in, a.packed + tile * (PACK_VECS_H * 64) + lane, reinterpret_cast<const uint2*>(a.gates) + tile * 64 + lane);
        else if constexpr (GATES) load_row(in, a.packed + tile * (PACK_VECS_G * 64) + lane, a.gates + tile * 64 + lane);
        else load_row(in, a.packed + tile * (PACK_VECS * 64) + lane);
    };
    fetch_row(min((int64_t)wave_global, last_tile));
    land_row(in);
    for (int64_t tile0 = (int64_t)wave_global * 32; tile0 < n_rows; tile0 += (int64_t)n_waves * 32) {
        asm volatile("" ::: "memory");      // keep the per-lane bias / LayerNorm vectors in LDS: hoisting them costs ~100 registers
        const int64_t row = min(tile0 + c, n_rows - 1);          // clamp: out-of-range lanes recompute the last row, never store
        const bool live = tile0 + c < n_rows;
        const int r = (int)((unsigned)row % (unsigned)a.L);      // B * L < 2^31 is checked by the launcher
        // ---- attention output of layer 0 in accumulator layout: register 4g+i <-> feature 8g + 4h + i
        f32x16 att;
        f16x8 ah[2] = {};                                       // half-width form: the same values as the fp16 operand of the out-projection
        if (r == 0) {                                           // the time token: inputs change every step
            const float* ap = a.att0 + (int64_t)((unsigned)row / (unsigned)a.L) * D;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(ap + 8 * g + 4 * h);
                att[4 * g] = t.x; att[4 * g + 1] = t.y; att[4 * g + 2] = t.z; att[4 * g + 3] = t.w;
                const float4 xt = *reinterpret_cast<const float4*>(a.x0_time + 8 * g + 4 * h), bt = *reinterpret_cast<const float4*>(bo + 8 * g + 4 * h);
                in.x[g] = make_float4(xt.x + bt.x, xt.y + bt.y, xt.z + bt.z, xt.w + bt.w);
            }
            // These loads run in 1 tile of 25.  Their wait must sit INSIDE this branch: placed at the first use after the
            // branch it would run on every tile and, with nothing loaded, wait for the previous tile's stores instead.
            typedef __attribute__((ext_vector_type(4))) float v4;
            v4* xp = reinterpret_cast<v4*>(in.x);
            asm volatile("" : "+v"(att), "+v"(xp[0]), "+v"(xp[1]), "+v"(xp[2]), "+v"(xp[3]));
            if constexpr (HIO) { ah[0] = to_f16x8(att, 0); ah[1] = to_f16x8(att, 1); }
        } else if constexpr (HIO) {
            // o (1 - gate) + gate v0 in packed fp16 arithmetic, straight into the operand registers (o arrives as fp16 and the product rounds
            // its operand to fp16 anyway; the gates are ~1 / 800, so rounding v0 - o moves the result by far less than that rounding):
            // 16 packed instructions instead of 56 conversions, subtractions and multiply-adds
            typedef __attribute__((ext_vector_type(4))) _Float16 h4;
            const h4 gt = __builtin_bit_cast(h4, in.gate);
            const f16x8* v0h = reinterpret_cast<const f16x8*>(k0s) + 2 * h;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f16x8 ov = __builtin_bit_cast(f16x8, in.o[s2]);
                const _Float16 ga = gt[2 * s2], gb = gt[2 * s2 + 1], na = (_Float16)1 - ga, nb = (_Float16)1 - gb;
                const f16x8 g8 = {ga, ga, ga, ga, gb, gb, gb, gb}, n8 = {na, na, na, na, nb, nb, nb, nb};
                ah[s2] = ov * n8 + g8 * v0h[s2];
            }
        } else if constexpr (GATES) {
            const float gt[4] = {in.gate.x, in.gate.y, in.gate.z, in.gate.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 vv = *reinterpret_cast<const float4*>(v0s + 8 * g + 4 * h);
                const float4 o = in.o[g];
                att[4 * g + 0] = fmaf(gt[g], vv.x - o.x, o.x);
                att[4 * g + 1] = fmaf(gt[g], vv.y - o.y, o.y);
                att[4 * g + 2] = fmaf(gt[g], vv.z - o.z, o.z);
                att[4 * g + 3] = fmaf(gt[g], vv.w - o.w, o.w);
            }
        } else {
            const float* stf = reinterpret_cast<const float*>(in.st);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = in.q[g];
                const float4 kk = *reinterpret_cast<const float4*>(k0s + 8 * g + 4 * h);
                const float4 vv = *reinterpret_cast<const float4*>(v0s + 8 * g + 4 * h);
                const float s = fmaf(q.x, kk.x, fmaf(q.y, kk.y, fmaf(q.z, kk.z, q.w * kk.w)));
                const float gate = rcp_fast(1.0f + exp2_fast(stf[5 * g] - s));
                const float o0 = stf[5 * g + 1], o1 = stf[5 * g + 2], o2 = stf[5 * g + 3], o3 = stf[5 * g + 4];
                att[4 * g + 0] = fmaf(gate, vv.x - o0, o0);
                att[4 * g + 1] = fmaf(gate, vv.y - o1, o1);
                att[4 * g + 2] = fmaf(gate, vv.z - o2, o2);
                att[4 * g + 3] = fmaf(gate, vv.w - o3, o3);
            }
        }
        // ---- y1 = (x0 + bo) + Wo att: the packed row is the accumulators' initial value
        f32x16 y1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 x = in.x[g];
            y1[4 * g + 0] = x.x; y1[4 * g + 1] = x.y; y1[4 * g + 2] = x.z; y1[4 * g + 3] = x.w;
        }
        if constexpr (HIO) {
            const unsigned short* wo = reinterpret_cast<const unsigned short*>(img) + (tile_base(0) * 2 * 64 + lane) * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(wo + s2 * 64 * 8), ah[s2], y1, 0, 0, 0);
        } else
        y1 = (MIX & 2) ? mm_h(img, tile_base(0), att, y1, lane) : mm<PREC>(img, tile_base(0), att, y1, lane);
        // ---- y2 = y1 + W2 gelu(W1 LN(y1) + b1) + b2, one 32-wide slice of the hidden layer at a time
        const f32x16 hn = layer_norm32(y1, g2, be2, h);
        f32x16 acc2;                                    // starts from the bias of the second FFN layer
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(b2 + 8 * g + 4 * h);
            acc2[4 * g + 0] = b.x; acc2[4 * g + 1] = b.y; acc2[4 * g + 2] = b.z; acc2[4 * g + 3] = b.w;
        }
        if constexpr (PREC == PREC_BF16X3) {
            WFrag w1 = load_w(img, tile_base(1), lane);
#pragma unroll 1
            for (int nt = 0; nt < 4; ++nt) {
                // the bias is the accumulator's initial value: the matrix pipe adds it
                f32x16 f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(b1 + 32 * nt + 8 * g + 4 * h);
                    f[4 * g + 0] = b.x; f[4 * g + 1] = b.y; f[4 * g + 2] = b.z; f[4 * g + 3] = b.w;
                }
                if constexpr ((ABL & 2) != 0) { for (int q = 0; q < 16; ++q) f[q] += hn[q]; } else if constexpr ((MIX & 4) != 0) f = mm_w_h(w1, hn, f); else f = mm_w(w1, hn, f);
                const WFrag w2 = load_w(img, tile_base(2) + nt, lane);          // lands under the GELU
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2 lo = (ABL & 1) ? f32x2{f[4 * g + 0], f[4 * g + 1]}
                                     : (ABL & 16) ? f32x2{gelu_erf1(f[4 * g + 0]), gelu_erf1(f[4 * g + 1])} : gelu_erf2(f32x2{f[4 * g + 0], f[4 * g + 1]});
                    const f32x2 hi = (ABL & 1) ? f32x2{f[4 * g + 2], f[4 * g + 3]}
                                     : (ABL & 16) ? f32x2{gelu_erf1(f[4 * g + 2]), gelu_erf1(f[4 * g + 3])} : gelu_erf2(f32x2{f[4 * g + 2], f[4 * g + 3]});
                    f[4 * g + 0] = lo.x; f[4 * g + 1] = lo.y; f[4 * g + 2] = hi.x; f[4 * g + 3] = hi.y;
                }
                w1 = load_w(img, tile_base(1) + min(nt + 1, 3), lane);          // lands under the MFMAs below
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((ABL & 2) != 0) { for (int q = 0; q < 16; ++q) acc2[q] += f[q]; } else if constexpr ((MIX & 1) != 0) acc2 = mm_w_h(w2, f, acc2); else acc2 = mm_w(w2, f, acc2);
            }
        } else {
#pragma unroll 1
            for (int nt = 0; nt < 4; ++nt) {
                f32x16 f = mm<PREC>(img, tile_base(1) + nt, hn, zero16(), lane);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(b1 + 32 * nt + 8 * g + 4 * h);
                    const f32x2 lo = gelu_erf2(f32x2{f[4 * g + 0] + b.x, f[4 * g + 1] + b.y});
                    const f32x2 hi = gelu_erf2(f32x2{f[4 * g + 2] + b.z, f[4 * g + 3] + b.w});
                    f[4 * g + 0] = lo.x; f[4 * g + 1] = lo.y; f[4 * g + 2] = hi.x; f[4 * g + 3] = hi.y;
                }
                acc2 = mm<PREC>(img, tile_base(2) + nt, f, acc2, lane);
            }
        }
        f32x16 y2;
#pragma unroll
        for (int q = 0; q < 16; ++q) y2[q] = y1[q] + acc2[q];
        // ---- next tile's inputs (clamped: a wave's last iteration re-reads the last tile and drops it)
        asm volatile("" ::: "memory");
        fetch_row((ABL & 8) ? min((int64_t)wave_global, last_tile) : min((tile0 >> 5) + n_waves, last_tile));
        // ---- layer 1: LayerNorm, K/V projection
        const f32x16 gn = layer_norm32(y2, g1, be1, h);
        const bool keep = live && r < a.n_keep;     // rows the decoder can see: they also need the layer-1 residual and query
        if (__any(keep)) {                          // wave-uniform: ~1 tile in 25 holds the head of a batch item
            const f32x16 q1 = mm<PREC>(img, tile_base(4), gn, zero16(), lane, Q_LO_OFF);
            if (keep) {
                float* xo = a.x1 + row * D;
                float* qo = a.q1 + (int64_t)((unsigned)row / (unsigned)a.L) * a.q1_item_stride + (int64_t)r * a.ldq1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(bq + 8 * g + 4 * h);
                    *reinterpret_cast<float4*>(xo + 8 * g + 4 * h) = make_float4(y2[4 * g], y2[4 * g + 1], y2[4 * g + 2], y2[4 * g + 3]);
                    *reinterpret_cast<float4*>(qo + 8 * g + 4 * h) =
                        make_float4(q1[4 * g] + b.x, q1[4 * g + 1] + b.y, q1[4 * g + 2] + b.z, q1[4 * g + 3] + b.w);
                }
            }
        }
        // Lanes past the last row hold a copy of it (the packed tile repeats it) and store the same values to the same
        // place: the stores are unconditional, so exactly eight vector-memory operations follow the loads of the next tile.
        if constexpr (HIO) {
            f16x8* gp = reinterpret_cast<f16x8*>(a.gpk) + (tile0 >> 5) * (2 * 64) + lane;
            gp[0] = to_f16x8(gn, 0);
            gp[64] = to_f16x8(gn, 1);
        } else if constexpr (GOUT) {
            float4* gp = a.gpk + (tile0 >> 5) * (4 * 64) + lane;
            if constexpr ((ABL & 4) != 0) { if (gn[0] == 123.456f) gp[0] = make_float4(gn[0], gn[1], gn[2], gn[3]); } else
#pragma unroll
            for (int g = 0; g < 4; ++g) gp[64 * g] = make_float4(gn[4 * g], gn[4 * g + 1], gn[4 * g + 2], gn[4 * g + 3]);
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x16 kv = mm<PREC>(img, tile_base(3) + nt, gn, zero16(), lane);
                float* op = a.kv + row * a.ldkv + 32 * nt;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(bkv + 32 * nt + 8 * g + 4 * h);
                    *reinterpret_cast<float4*>(op + 8 * g + 4 * h) =
                        make_float4(kv[4 * g] + b.x, kv[4 * g + 1] + b.y, kv[4 * g + 2] + b.z, kv[4 * g + 3] + b.w);
                }
            }
        }
        land_row(in);
    }
}

// ---- last-layer attention of the few visible rows, straight from the LayerNorm rows -----------------------------------
// An MFMA operand with "outer index on the lane, k on the registers" in the k order of feat(): exactly what an accumulator
// holds, so every product below feeds the next one without moving data between lanes.  mma(A, B, acc) returns acc +
// (lane <-> B's outer index, register <-> A's outer index).
template <int PREC> struct Opnd;
template <> struct Opnd<PREC_F32> { f32x16 v; };
template <> struct Opnd<PREC_BF16> { bf16x8 h[2]; };
template <> struct Opnd<PREC_BF16X3> { bf16x8 h[2], l[2]; };

template <int PREC>
__device__ __forceinline__ Opnd<PREC> to_opnd(const f32x16& v) {
    Opnd<PREC> o;
    if constexpr (PREC == PREC_F32) {
        o.v = v;
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = v[8 * s + j];
                const __bf16 hi = (__bf16)x;
                o.h[s][j] = hi;
                if constexpr (PREC == PREC_BF16X3) o.l[s][j] = (__bf16)(x - (float)hi);
            }
    }
    return o;
}

template <int PREC>
__device__ __forceinline__ Opnd<PREC> weight_opnd(const float* img_f32, const unsigned short* img_bf, int tile, int lane) {
    Opnd<PREC> o;
    if constexpr (PREC == PREC_F32) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) o.v[reg] = img_f32[(tile * 16 + reg) * 64 + lane];
    } else {
        const unsigned short* whi = img_bf + (tile * 2 * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            o.h[s] = *reinterpret_cast<const bf16x8*>(whi + s * 64 * 8);
            if constexpr (PREC == PREC_BF16X3) o.l[s] = *reinterpret_cast<const bf16x8*>(whi + NFRAG_BF * 64 * 8 + s * 64 * 8);
        }
    }
    return o;
}

template <int PREC>
__device__ __forceinline__ f32x16 mma(const Opnd<PREC>& A, const Opnd<PREC>& B, f32x16 acc) {
    if constexpr (PREC == PREC_F32) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A.v[reg], B.v[reg], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (PREC == PREC_BF16X3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[s], B.l[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l[s], B.h[s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[s], B.h[s], acc, 0, 0, 0);
        }
    }
    return acc;
}

// N products issued with their accumulation chains interleaved: consecutive matrix instructions never depend on each other
template <int N>
__device__ __forceinline__ void mma_x3_interleaved(const Opnd<PREC_BF16X3>* const (&A)[N], const Opnd<PREC_BF16X3>* const (&B)[N], f32x16 (&acc)[N]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[n]->h[s], B[n]->l[s], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[n]->l[s], B[n]->h[s], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[n]->h[s], B[n]->h[s], acc[n], 0, 0, 0);
    }
}

constexpr int AG_COLS = 96;            // (visible row, head) pairs: three 32-wide column blocks
constexpr float AG_NEG = -1.0e30f;

// One workgroup (two wavefronts) per item; the wavefronts take alternate 32-row groups of the item's rows and merge at the end.
// Per group, with g the stored LayerNorm rows (lane <-> row):
//     V   = g Wv^T        (lane <-> feature, registers <-> key)
//     S^T = g Qt^T        (lane <-> column (row i, head), registers <-> key), Qt = Qb Wk: Qb holds the query of head hd on that
//                          head's four features and zero elsewhere, so S is the per-head dot product of the query with Wk g;
//                          the K bias shifts all scores of a query alike and is dropped
//     O^T += V^T P^T      (lane <-> column, registers <-> feature), P = 2^(S - running max): the online softmax keeps its
//                          maximum, sum and rescale factor per LANE, because a column lives on a lane in S^T and in O^T alike.
// Only the four features of a column's own head are meaningful in O^T; they are picked at the end.
// MIX (split-bf16 mode only), bits: 1 the weighted sums O^T += V^T P^T with V and P rounded once to fp16; 2 also the scores and V
// from g, the folded queries and Wv rounded once to fp16; 4 (with 2) the LayerNorm rows arrive as fp16, as the half-width row kernel stores them
template <int PREC, int ABL = 0, int MIX = 0>      // ABL: measurement aid (results are wrong except for 16): 1 no exponentials, 2 no weighted-sum products, 4 no score products, 8 no operand splits of the weights, 16 hand-placed instruction order
__device__ __forceinline__ void attn_g_body(const AttnGArgs& a, const float* img_f32, const unsigned short* img_bf) {
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    __shared__ float mrg[(6 + 48) * 64];
    __shared__ Opnd<PREC> qsh[3][64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h2 = lane >> 5;
    const Opnd<PREC> wv = weight_opnd<PREC>(img_f32, img_bf, tile_base(3) + 1, lane);
    f16x8 wvh[2] = {};                                      // Wv rounded once to fp16 (MIX & 2)
    if constexpr ((MIX & 2) != 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) wvh[s][j] = (_Float16)((float)wv.h[s][j] + (float)wv.l[s][j]);
    }
    const float qs = a.scale * LOG2E_F;
    const int64_t r0 = (int64_t)b * a.L, r1 = r0 + a.L;
    // The query operand is the same for both wavefronts and every group: built once, kept in LDS (48 registers otherwise).
    // The K projection is folded into it: q . (Wk g) = (Wk^T q) . g, so Qt^T = Wk^T Qb^T (lane <-> column, registers <->
    // feature of g) is computed here and no group projects its keys.
    if (wave == 0) {
        f32x16 wt;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) wt[reg] = a.wk[feat(reg, h2) * D + n];       // Wk^T: lane <-> input feature, registers <-> output feature
        const Opnd<PREC> wkt = to_opnd<PREC>(wt);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int col = 32 * nb + n, i = col >> 3, hd = col & 7;
            f32x16 f = zero16();
            if (i < a.n_keep && (hd & 1) == h2) {
                const float4 q = *reinterpret_cast<const float4*>(a.q1 + (int64_t)b * a.q1_item_stride + (int64_t)i * a.ldq1 + 4 * hd);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (g == (hd >> 1)) { f[4 * g] = q.x * qs; f[4 * g + 1] = q.y * qs; f[4 * g + 2] = q.z * qs; f[4 * g + 3] = q.w * qs; }
            }
            const f32x16 qt = mma<PREC>(wkt, to_opnd<PREC>(f), zero16());
            if constexpr ((MIX & 2) != 0) {
                Opnd<PREC> qh;                                  // fp16 bit patterns in the high-half slots
                qh.h[0] = __builtin_bit_cast(bf16x8, to_f16x8(qt, 0)); qh.h[1] = __builtin_bit_cast(bf16x8, to_f16x8(qt, 1));
                qh.l[0] = qh.h[0]; qh.l[1] = qh.h[1];
                qsh[nb][lane] = qh;
            } else qsh[nb][lane] = to_opnd<PREC>(qt);
        }
    }
    __syncthreads();
    float m[3] = {AG_NEG, AG_NEG, AG_NEG}, l[3] = {0.f, 0.f, 0.f};
    f32x16 o[3] = {zero16(), zero16(), zero16()};
    // Keys are walked in groups of 32 rows counted from the ITEM's first row (not in the global tiles enc_rows stored them
    // in), so that the order of the online softmax -- and with it every bit of the result -- does not depend on where in
    // the batch the item sits.  Lane (c, h2) of group t reads row r0 + 32 t + c: two contiguous runs of the lane-ordered store.
    const int n_groups = (a.L + 31) >> 5;
    int t = wave;
    float4 gv[4];
    f16x8 gq[2];                                                 // the same rows as fp16 (MIX & 4)
    static_assert((MIX & 4) == 0 || (MIX & 2) != 0, "fp16 rows feed the fp16 products only");
    auto fetch = [&](int grp) {
        const int64_t row = min(r0 + 32 * (int64_t)min(grp, n_groups - 1) + n, r1 - 1);
        if constexpr ((MIX & 4) != 0) {
            const f16x8* gp = reinterpret_cast<const f16x8*>(a.gpk) + (row >> 5) * (2 * 64) + h2 * 32 + (row & 31);
            gq[0] = gp[0]; gq[1] = gp[64];
        } else {
            const float4* gp = a.gpk + (row >> 5) * (4 * 64) + h2 * 32 + (row & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) gv[g] = gp[64 * g];      // written by the row kernel just before: kept cacheable (streamed loads measured 1 % slower)
        }
    };
    fetch(t);
    for (; t < n_groups; t += 2) {
        f32x16 g16 = zero16();
        f16x8 gh[2] = {};
        if constexpr ((MIX & 4) != 0) { gh[0] = gq[0]; gh[1] = gq[1]; }
        else {
#pragma unroll
            for (int g = 0; g < 4; ++g) { g16[4 * g] = gv[g].x; g16[4 * g + 1] = gv[g].y; g16[4 * g + 2] = gv[g].z; g16[4 * g + 3] = gv[g].w; }
            if constexpr ((MIX & 2) != 0) { gh[0] = to_f16x8(g16, 0); gh[1] = to_f16x8(g16, 1); }
        }
        fetch(t + 2);                                                   // next group (clamped: the last iteration re-reads and drops it)
        Opnd<PREC> gop;
        if constexpr ((MIX & 2) == 0) gop = to_opnd<PREC>(g16);
        // One group, as straight-line code (the masked form of the item's last group is a second copy): a wavefront issues
        // in order, so vector work hides under matrix work only where the two alternate in the instruction stream.
        auto group = [&](auto partial_c) {
            constexpr bool PARTIAL = decltype(partial_c)::value;
            // online softmax of one column block; returns the weights as an operand
            auto soft = [&](int nb, f32x16 sc) {
                if constexpr (PARTIAL) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (32 * t + feat(reg, h2) >= a.L) sc[reg] = AG_NEG;
                }
                float mx = sc[0];
#pragma unroll
                for (int reg = 1; reg < 16; ++reg) mx = fmaxf(mx, sc[reg]);
                mx = xor32_max(mx);
                // The reference value of a column moves only when a score exceeds it by more than a.lazy (log2 units): the weights
                // then stay below 2^lazy, well inside fp16 and bf16, and the partial sums are rescaled in the first group and
                // hardly ever after it (wave-uniform branch) instead of in every group.  Depends on the item's own rows only.
                const bool up = mx > m[nb] + a.lazy;
                if (__any(up)) {
                    const float mn = up ? mx : m[nb];
                    const float fac = exp2_fast(m[nb] - mn);
                    m[nb] = mn;
                    l[nb] *= fac;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) o[nb][reg] *= fac;
                }
                const float mn = m[nb];
                float sum = 0.f;
                if constexpr (MIX != 0) {                       // pairs: the subtraction and the sum as packed instructions
                    f32x2 sum2 = {0.f, 0.f};                    // (the split-bf16 instance has no registers to spare for this form)
#pragma unroll
                    for (int reg = 0; reg < 16; reg += 2) {
                        f32x2 dd = f32x2{sc[reg], sc[reg + 1]} - f32x2{mn, mn};
                        dd = f32x2{exp2_fast(dd.x), exp2_fast(dd.y)};
                        sc[reg] = dd.x; sc[reg + 1] = dd.y;
                        sum2 += dd;
                    }
                    sum = sum2.x + sum2.y;
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) { sc[reg] = (ABL & 1) ? sc[reg] - mn : exp2_fast(sc[reg] - mn); sum += sc[reg]; }
                }
                l[nb] += sum;
                if constexpr (MIX != 0) return sc;
                else if constexpr ((ABL & 8) != 0) { Opnd<PREC> r = gop; r.h[0][0] = (__bf16)sum; return r; }
                else return to_opnd<PREC>(sc);
            };
            if constexpr (MIX != 0) {                                  // one fp16 instruction per weighted sum and 16 keys
                f32x16 acc4[4] = {zero16(), zero16(), zero16(), zero16()};
                if constexpr ((MIX & 2) != 0) {                        // scores and V from fp16 operands as well
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
                            acc4[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], __builtin_bit_cast(f16x8, qsh[nb][lane].h[s]), acc4[nb], 0, 0, 0);
                        acc4[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], wvh[s], acc4[3], 0, 0, 0);
                    }
                } else {
                    const Opnd<PREC> q0 = qsh[0][lane], q1 = qsh[1][lane], q2 = qsh[2][lane];
                    const Opnd<PREC>* const A4[4] = {&gop, &gop, &gop, &gop};
                    const Opnd<PREC>* const B4[4] = {&q0, &q1, &q2, &wv};
                    mma_x3_interleaved<4>(A4, B4, acc4);
                }
                const f16x8 v0 = to_f16x8(acc4[3], 0), v1 = to_f16x8(acc4[3], 1);
                const f32x16 p0 = soft(0, acc4[0]), p1 = soft(1, acc4[1]), p2 = soft(2, acc4[2]);
                const f16x8 pa[3][2] = {{to_f16x8(p0, 0), to_f16x8(p0, 1)}, {to_f16x8(p1, 0), to_f16x8(p1, 1)}, {to_f16x8(p2, 0), to_f16x8(p2, 1)}};
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pa[nb][0], o[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pa[nb][1], o[nb], 0, 0, 0);
                return;
            }
            else if constexpr (ABL == 0 && PREC == PREC_BF16X3) {     // (the measurement variants remove parts of the hand-placed form below)
                // split-bf16 operands: the four products of these rows, then the three weighted sums, each set issued with its
                // accumulation chains interleaved (no matrix instruction depends on its predecessor): 45.5 us against 48.3 us for
                // the hand-placed order below (ADX_ATTN_G_ABLATE=16), same bits
                const Opnd<PREC> q0 = qsh[0][lane], q1 = qsh[1][lane], q2 = qsh[2][lane];
                const Opnd<PREC>* const A4[4] = {&gop, &gop, &gop, &gop};
                const Opnd<PREC>* const B4[4] = {&q0, &q1, &q2, &wv};
                f32x16 acc4[4] = {zero16(), zero16(), zero16(), zero16()};
                mma_x3_interleaved<4>(A4, B4, acc4);
                const Opnd<PREC> vop = to_opnd<PREC>(acc4[3]);
                const Opnd<PREC> p0 = soft(0, acc4[0]), p1 = soft(1, acc4[1]), p2 = soft(2, acc4[2]);
                const Opnd<PREC>* const A3[3] = {&vop, &vop, &vop};
                const Opnd<PREC>* const B3[3] = {&p0, &p1, &p2};
                mma_x3_interleaved<3>(A3, B3, o);
                return;
            }
            if constexpr (MIX == 0) {
            const f32x16 s0 = (ABL & 4) ? g16 : mma<PREC>(gop, qsh[0][lane], zero16());
            __builtin_amdgcn_sched_barrier(0);
            // the other products of these rows (18 matrix instructions), with block 0's vector work between them
            const f32x16 s1 = (ABL & 4) ? g16 : mma<PREC>(gop, qsh[1][lane], zero16());
            const f32x16 vv = mma<PREC>(gop, wv, zero16());
            const f32x16 s2 = (ABL & 4) ? g16 : mma<PREC>(gop, qsh[2][lane], zero16());
            const Opnd<PREC> p0 = soft(0, s0);
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // eight vector instructions
            }
            __builtin_amdgcn_sched_barrier(0);
            const Opnd<PREC> vop = to_opnd<PREC>(vv);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 2) != 0) o[0][0] += (float)p0.h[0][0] + (float)vop.h[0][1]; else o[0] = mma<PREC>(vop, p0, o[0]);
            const Opnd<PREC> p1 = soft(1, s1);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 2) != 0) o[1][0] += (float)p1.h[0][0] + (float)vop.h[0][1]; else o[1] = mma<PREC>(vop, p1, o[1]);
            const Opnd<PREC> p2 = soft(2, s2);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 2) != 0) o[2][0] += (float)p2.h[0][0] + (float)vop.h[0][1]; else o[2] = mma<PREC>(vop, p2, o[2]);
            }
        };
        if (32 * t + 32 > a.L) group(std::true_type{});          // wave-uniform: the item's last group
        else group(std::false_type{});
    }
    // ---- merge the two wavefronts, normalise, add the V bias, store the four features of each column's head
    if (wave == 1) {
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            mrg[nb * 64 + lane] = m[nb];
            mrg[(3 + nb) * 64 + lane] = l[nb];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) mrg[(6 + nb * 16 + reg) * 64 + lane] = o[nb][reg];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const float* bv = a.vecs + 320 + 32;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const float m1 = mrg[nb * 64 + lane], l1 = mrg[(3 + nb) * 64 + lane];
            const float mn = fmaxf(m[nb], m1);
            const float f0 = exp2_fast(m[nb] - mn), f1 = exp2_fast(m1 - mn);
            const float inv = rcp_fast(xor32_add(fmaf(l[nb], f0, l1 * f1)));
            const int col = 32 * nb + n, i = col >> 3, hd = col & 7;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g == (hd >> 1)) {
                    r.x = fmaf(o[nb][4 * g], f0, mrg[(6 + nb * 16 + 4 * g) * 64 + lane] * f1);
                    r.y = fmaf(o[nb][4 * g + 1], f0, mrg[(6 + nb * 16 + 4 * g + 1) * 64 + lane] * f1);
                    r.z = fmaf(o[nb][4 * g + 2], f0, mrg[(6 + nb * 16 + 4 * g + 2) * 64 + lane] * f1);
                    r.w = fmaf(o[nb][4 * g + 3], f0, mrg[(6 + nb * 16 + 4 * g + 3) * 64 + lane] * f1);
                }
            if (i < a.n_keep && (hd & 1) == h2) {
                const float4 bb = *reinterpret_cast<const float4*>(bv + 4 * hd);
                *reinterpret_cast<float4*>(a.O + (int64_t)b * a.o_batch_stride + (int64_t)i * a.ldo + 4 * hd) =
                    make_float4(fmaf(r.x, inv, bb.x), fmaf(r.y, inv, bb.y), fmaf(r.z, inv, bb.z), fmaf(r.w, inv, bb.w));
            }
        }
    }
}

template <int PREC, int ABL = 0, int MIX = 0>
__global__ __launch_bounds__(128, 2) void attn_g_kernel(const AttnGArgs a, const float* img_f32, const unsigned short* img_bf) {
    attn_g_body<PREC, ABL, MIX>(a, img_f32, img_bf);
}

// The fp16 forms with a budget of 192 registers per lane (the form the fp16 engine runs, 7, keeps it: 191 without spilling; the compiler
// exceeds it for the partial forms 1 and 3): two of its waves and one wave of the tail (120 registers), which runs beside them on the second
// stream, then fit the 512 registers of a SIMD lane together (at 200 a SIMD holding a tail wave takes only one of these): +0.4 %, same bits.
template <int MIX>
__global__ __launch_bounds__(128, 2) __attribute__((amdgpu_num_vgpr(192))) void attn_g_kernel_h(const AttnGArgs a, const float* img_f32, const unsigned short* img_bf) {
    attn_g_body<PREC_BF16X3, 0, MIX>(a, img_f32, img_bf);
}

}  // namespace

size_t enc_rows_image_bytes() { return sizeof(float) * IMG_F32_FLOATS + sizeof(unsigned short) * IMG_BF_SHORTS; }

hipError_t enc_rows_build_image(const float* Wo, const float* W1, const float* W2, const float* Wkv, const float* Wq, void* image,
                                hipStream_t stream, int f16_mats) {
    ImageSrc s;
    s.W[0] = Wo; s.W[1] = W1; s.W[2] = W2; s.W[3] = Wkv; s.W[4] = Wq;
    float* f = reinterpret_cast<float*>(image);
    unsigned short* b = reinterpret_cast<unsigned short*>(f + IMG_F32_FLOATS);
    hipLaunchKernelGGL(build_image_kernel, dim3(8), dim3(256), 0, stream, s, f, b, f16_mats);
    return hipGetLastError();
}

size_t enc_rows_packed_bytes(int B, int L) { return (size_t)(((int64_t)B * L + 31) / 32) * PACK_VECS * 64 * sizeof(float4); }

hipError_t enc_rows_pack(const EncRowsArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.L <= 0 || !a.packed || !a.x0 || !a.q || !a.state) return hipErrorInvalidValue;
    const int64_t tiles = ((int64_t)a.B * a.L + 31) / 32, n = tiles * 64;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, const_cast<float4*>(a.packed), tiles);
    return hipGetLastError();
}

hipError_t enc_rows(const EncRowsArgs& a0, int prec, hipStream_t stream, int mix) {
    if (a0.B <= 0 || a0.L <= 0 || !a0.image || !a0.packed || !a0.att0 || !a0.x0_time || (int64_t)a0.B * a0.L >= (int64_t)1 << 31) return hipErrorInvalidValue;
    EncRowsArgs a = a0;
    if (a.q1_item_stride == 0) a.q1_item_stride = (int64_t)a.L * a.ldq1;
    a.img_f32 = reinterpret_cast<const float*>(a.image);
    a.img_bf = reinterpret_cast<const unsigned short*>(a.img_f32 + IMG_F32_FLOATS);
    const size_t smem = IMG_F32_FLOATS * 4 + (VEC_FLOATS + 64) * 4;
    const int64_t tiles = ((int64_t)a.B * a.L + 31) / 32;
    static const int waves = [] { const char* e = getenv("ADX_ENC_ROWS_WAVES"); return e ? atoi(e) : 3; }();   // tuning knob (2, 3 or 4 resident waves per SIMD)
    const int wpb = waves == 4 ? 8 : 4;                 // waves per workgroup
    int blocks = (int)((tiles + wpb - 1) / wpb);
    const int cap = 256 * (waves == 3 ? 3 : 2);        // resident workgroups; waves stride over the remaining tiles
    if (blocks > cap) blocks = cap;
#define ADX_LAUNCH_ROWS2(P, G, T)                                                                                          \
    do {                                                                                                                  \
        if (waves == 4) hipLaunchKernelGGL((enc_rows_kernel<P, 4, G, T>), dim3(blocks), dim3(512), smem, stream, a);      \
        else if (waves == 3) hipLaunchKernelGGL((enc_rows_kernel<P, 3, G, T>), dim3(blocks), dim3(256), smem, stream, a); \
        else hipLaunchKernelGGL((enc_rows_kernel<P, 2, G, T>), dim3(blocks), dim3(256), smem, stream, a);                 \
    } while (0)
#ifdef ADX_MEASURE          // measurement library only: the ablation instances give wrong results
#define ADX_LAUNCH_ROWS(P)                                       \
    do {                                                         \
        if (a.gpk && a.gates && abl && waves == 3) {             \
            switch (abl) {                                       \
                case 1: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 1>), dim3(blocks), dim3(256), smem, stream, a); break;  \
                case 2: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 2>), dim3(blocks), dim3(256), smem, stream, a); break;  \
                case 3: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 3>), dim3(blocks), dim3(256), smem, stream, a); break;  \
                case 4: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 4>), dim3(blocks), dim3(256), smem, stream, a); break;  \
                case 8: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 8>), dim3(blocks), dim3(256), smem, stream, a); break;  \
                case 12: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 12>), dim3(blocks), dim3(256), smem, stream, a); break; \
                case 15: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 15>), dim3(blocks), dim3(256), smem, stream, a); break; \
                case 16: hipLaunchKernelGGL((enc_rows_kernel<P, 3, true, true, 16>), dim3(blocks), dim3(256), smem, stream, a); break; \
                default: ADX_LAUNCH_ROWS2(P, true, true);        \
            }                                                    \
        } else if (a.gpk && a.gates) ADX_LAUNCH_ROWS2(P, true, true);   \
        else if (a.gpk) ADX_LAUNCH_ROWS2(P, true, false);        \
        else ADX_LAUNCH_ROWS2(P, false, false);                  \
    } while (0)
    const char* abl_env = getenv("ADX_ENC_ROWS_ABLATE");         // measurement aid, read per call (split-bf16 mode only)
    const int abl = (abl_env && prec == PREC_BF16X3) ? atoi(abl_env) : 0;
#else
#define ADX_LAUNCH_ROWS(P)                                       \
    do {                                                         \
        if (a.gpk && a.gates) ADX_LAUNCH_ROWS2(P, true, true);   \
        else if (a.gpk) ADX_LAUNCH_ROWS2(P, true, false);        \
        else ADX_LAUNCH_ROWS2(P, false, false);                  \
    } while (0)
#endif
    if (a.gates && !a.gpk) return hipErrorInvalidValue;          // the gate table exists for the LayerNorm-row form only
    if (!a.gpk && !a.kv) return hipErrorInvalidValue;
    if (mix) {                                                   // the image must hold Wo, W1 and W2 as fp16; always three resident waves per SIMD
        if (prec != PREC_BF16X3 || (mix != 7 && mix != 15) || (a.gates && !a.gpk) || (!a.gpk && !a.kv)) return hipErrorInvalidValue;
        const int blocks3 = (int)std::min<int64_t>((tiles + 3) / 4, 256 * 3);
        if (mix == 15) {                                         // half-width inputs (enc_rows_pack_gates with half = true) and output
            if (!a.gpk || !a.gates) return hipErrorInvalidValue;
            // tuning knob: resident workgroups per CU of this form (its compact LDS image and 124 registers allow 4; the default 3
            // leaves registers for a wave of the tail on every SIMD)
            static const int wgs = [] { const char* e = getenv("ADX_ENC_ROWS_WGS"); const int v = e ? atoi(e) : 3; return v >= 1 && v <= 4 ? v : 3; }();
            const int blocks_h = (int)std::min<int64_t>((tiles + 3) / 4, 256 * wgs);
            hipLaunchKernelGGL((enc_rows_kernel<PREC_BF16X3, 3, true, true, 0, 15>), dim3(blocks_h), dim3(256), ROWS_LDS_IMG_H + (VEC_FLOATS + 64) * 4, stream, a);
        } else
        if (a.gpk && a.gates) hipLaunchKernelGGL((enc_rows_kernel<PREC_BF16X3, 3, true, true, 0, 7>), dim3(blocks3), dim3(256), smem, stream, a);
        else if (a.gpk) hipLaunchKernelGGL((enc_rows_kernel<PREC_BF16X3, 3, true, false, 0, 7>), dim3(blocks3), dim3(256), smem, stream, a);
        else hipLaunchKernelGGL((enc_rows_kernel<PREC_BF16X3, 3, false, false, 0, 7>), dim3(blocks3), dim3(256), smem, stream, a);
        return hipGetLastError();
    }
    if (prec == PREC_F32) ADX_LAUNCH_ROWS(PREC_F32);
    else if (prec == PREC_BF16) ADX_LAUNCH_ROWS(PREC_BF16);
    else ADX_LAUNCH_ROWS(PREC_BF16X3);
#undef ADX_LAUNCH_ROWS
#undef ADX_LAUNCH_ROWS2
    return hipGetLastError();
}

size_t enc_rows_gates_bytes(int B, int L, int n_steps) { return (size_t)(((int64_t)B * L + 31) / 32) * 64 * sizeof(float4) * (size_t)n_steps; }

hipError_t enc_rows_pack_gates(const EncRowsArgs& a, float4* gates, const float* tq, int n_steps, hipStream_t stream, bool half) {
    if (a.B <= 0 || a.L <= 0 || !a.packed || !a.x0 || !a.q || !a.state || !gates || !tq || n_steps <= 0) return hipErrorInvalidValue;
    const int64_t tiles = ((int64_t)a.B * a.L + 31) / 32, n = tiles * 64;
    if (half) hipLaunchKernelGGL(pack_gates_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, const_cast<float4*>(a.packed), gates, tiles, tq,
                                 n_steps, a.scale * LOG2E_F);
    else hipLaunchKernelGGL(pack_gates_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, const_cast<float4*>(a.packed), gates, tiles, tq,
                            n_steps, a.scale * LOG2E_F);
    return hipGetLastError();
}

size_t enc_rows_g_bytes(int B, int L) { return (size_t)(((int64_t)B * L + 31) / 32) * 4 * 64 * sizeof(float4); }

bool attention_g_ok(int n_keep) { return n_keep >= 1 && n_keep * NH <= AG_COLS; }

hipError_t attention_g(const AttnGArgs& a0, int prec, hipStream_t stream, int mix) {
    AttnGArgs a = a0;
    if (a.q1_item_stride == 0) a.q1_item_stride = (int64_t)a.L * a.ldq1;
    if (a.B <= 0 || a.L <= 0 || !attention_g_ok(a.n_keep) || a.n_keep > a.L || !a.gpk || !a.image || !a.vecs || !a.wk || !a.q1 || !a.O ||
        (a.ldq1 & 3) != 0 || (a.q1_item_stride & 3) != 0 || (a.ldo & 3) != 0 || (a.o_batch_stride & 3) != 0 || (int64_t)a.B * a.L >= (int64_t)1 << 31)
        return hipErrorInvalidValue;
    const float* img_f32 = reinterpret_cast<const float*>(a.image);
    const unsigned short* img_bf = reinterpret_cast<const unsigned short*>(img_f32 + IMG_F32_FLOATS);
    if (mix) {
        if (prec != PREC_BF16X3) return hipErrorInvalidValue;
        if ((mix & 6) == 6) hipLaunchKernelGGL((attn_g_kernel_h<7>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
        else if (mix & 4) return hipErrorInvalidValue;
        else if (mix & 2) hipLaunchKernelGGL((attn_g_kernel_h<3>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
        else hipLaunchKernelGGL((attn_g_kernel_h<1>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
        return hipGetLastError();
    }
    if (prec == PREC_F32) hipLaunchKernelGGL(attn_g_kernel<PREC_F32>, dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
    else if (prec == PREC_BF16) hipLaunchKernelGGL(attn_g_kernel<PREC_BF16>, dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
    else {
#ifdef ADX_MEASURE
        const char* abl_env = getenv("ADX_ATTN_G_ABLATE");      // measurement aid, read per call (measurement library only)
        switch (abl_env ? atoi(abl_env) : 0) {
            case 1: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 1>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 2: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 2>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 4: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 4>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 6: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 6>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 8: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 8>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 9: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 9>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 15: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 15>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            case 16: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 16>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf); break;
            default: hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 0>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
        }
#else
        hipLaunchKernelGGL((attn_g_kernel<PREC_BF16X3, 0>), dim3(a.B), dim3(128), 0, stream, a, img_f32, img_bf);
#endif
    }
    return hipGetLastError();
}

}  // namespace adx

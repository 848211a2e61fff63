// MFMA form of the fused variant-A tail (see tail_a.hip for what the tail covers): one WAVEFRONT per batch item (up to 16
// rows per item), or one wavefront per 16-row tile of an item and one workgroup per item (up to 96 rows).
//
// The tail is a chain of 32-wide linear maps on at most 16 rows (T+1 memory rows, T trajectory tokens) of each item.
// Every map runs transposed on the fp32 matrix cores, Y^T = W X^T, as 16x16x4 MFMA tiles: M = 16 output features,
// N = 16 rows (lane & 15), K = 4 input features (lane >> 4).  The accumulator of one map (lane (n, q) holds features
// 16t + 4q + r of row n, t in {0,1}, r in 0..3) is fed back as the B operand of the next one, one register per MFMA;
// the weight image stores A fragments with the matching permuted k order, so an activation never leaves the
// registers between two maps.  In that layout a lane owns two complete attention heads (q and q + 4) of its row, a
// LayerNorm is 8 registers + two cross-lane adds, and keys/values reach the other rows through a 4 KB LDS tile.
// Weights are not staged in LDS: the image is laid out in consumption order and streamed from L2 with a two-block
// register look-ahead (a block = one 32x32 map = 16 MFMAs), so the loads of block i+2 fly under the MFMAs of block i;
// a deeper look-ahead costs the fourth resident wave per SIMD and measured slower.
// Two arithmetic forms of the 32x32 maps: exact fp32 (16x16x4 MFMAs, the fp32 engine mode) and split-bf16 (both
// operands as bf16 high + low halves, three 16x16x32 bf16 MFMAs per output tile: 6 MFMAs of 16 cycles per map instead of
// 16 of 32 cycles, products good to ~2^-17).  In the split form the accumulator layout is unchanged -- the 8 features a lane
// holds are its 8 k-slots of the K = 32 MFMA, the weight image carries the matching permutation -- so LayerNorm, attention,
// GELU and the scheduler update are the same fp32 code in both.  The split form applies to the maps on MEMORY rows only
// (the encoder remainder and the cross-attention key / value projections, blocks packed with split = 1); the maps on
// trajectory tokens stay fp32 in the fp32 and split-bf16 modes.
// A third form, split-fp16 (fp16 high + low halves, the same three MFMAs per output tile; 22 significand bits, so the products
// are good to ~2^-21), is close enough to fp32 for the maps on trajectory tokens as well, which split-bf16 is not: it is the form of
// EVERY map in the bf16 and fp16 engine modes (measured at 1024 scenarios of the bench configuration against memory rows split-bf16
// + trajectory tokens fp32: 79.6 -> 76.7 ms per generate and ADE against the fp32 engine 5.0e-4 -> 4.7e-4 m median in the split-bf16
// engine, 60.0 -> 56.0 ms and 1.09e-3 -> 1.06e-3 m in the fp16 engine).  Memory rows rounded ONCE to fp16 -- what the row kernel does
// to every row of the first encoder layer in the fp16 engine -- fail here: 1.8e-2 m median.
#include "adx_ops.h"
#include "fastmath.h"

#include <math.h>

namespace adx {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int SLOT = 64;                 // float4 per slot: one per lane
constexpr int BLOCK_SLOTS = 4;           // a 32x32 block: (mt, t) = (0,0) (0,1) (1,0) (1,1)
constexpr int KV_LD = 68;                // LDS row stride of the key/value tile (floats)
constexpr float LOG2E_M = 1.4426950408889634f;

// ---- offsets into the vector table (floats); mirrored by tail_mfma_vec_layout below
constexpr int V_E_BO = 0, V_E_LN2G = 32, V_E_LN2B = 64, V_E_B1 = 96, V_E_B2 = 224, V_DIN_B = 256, V_POS = 288, V_LAYER0 = 800;
constexpr int VL_LN1G = 0, VL_LN1B = 32, VL_LN2G = 64, VL_LN2B = 96, VL_LN3G = 128, VL_LN3B = 160, VL_BQKV = 192, VL_BO = 288,
              VL_BQ = 320, VL_BKV = 352, VL_BO2 = 416, VL_B1 = 448, VL_B2 = 576, VL_SIZE = 608;
constexpr int V_LNFG = V_LAYER0 + 4 * VL_SIZE, V_LNFB = V_LNFG + 32, V_HEAD_B = V_LNFB + 32, V_TOTAL = V_HEAD_B + 16;

struct A8 { f32x4 t[2]; };               // a [16 rows x 32 features] activation in accumulator layout

// LDS accesses of one wavefront execute in program order; this only stops the compiler from moving them across.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// DEPTH-block look-ahead over the weight image (registers; the shifts are moves that issue under the MFMAs).
constexpr int DEPTH = 2;
struct WStream {
    const float4* p;                     // next block to fetch (already offset by the lane)
    float4 a[DEPTH][BLOCK_SLOTS];
    __device__ __forceinline__ void fetch(float4* dst) {
#pragma unroll
        for (int i = 0; i < BLOCK_SLOTS; ++i) dst[i] = p[i * SLOT];
        p += BLOCK_SLOTS * SLOT;
    }
    __device__ __forceinline__ void init(const float4* first) {
        p = first;
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) fetch(a[k]);
    }
    // hand out the oldest block, shift, and start fetching the block DEPTH ahead
    __device__ __forceinline__ void pop(float4* w) {
#pragma unroll
        for (int i = 0; i < BLOCK_SLOTS; ++i) w[i] = a[0][i];
#pragma unroll
        for (int k = 0; k + 1 < DEPTH; ++k)
#pragma unroll
            for (int i = 0; i < BLOCK_SLOTS; ++i) a[k][i] = a[k + 1][i];
        fetch(a[DEPTH - 1]);
    }
};

__device__ __forceinline__ f32x4 mfma4(const float4& w, const f32x4& x, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x[3], acc, 0, 0, 0);
    return acc;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// Arithmetic form of a map (the value of `split` its block was packed with)
constexpr int F_F32 = 0, F_BF16X3 = 1, F_F16X3 = 2;
// split-fp16: the low halves of both operands are stored times 2^11, which keeps them normal fp16 numbers whenever the value itself is
// (unscaled, the low half of anything below 0.125 -- every weight -- is an fp16 subnormal, good to 3e-8 absolute only: measured
// on the bench configuration, that alone moved the final trajectories by 6e-4 m)
constexpr float LO_SCALE = 2048.f;

// An activation prepared as the B operand of a map: itself (fp32 form) or its bf16 or fp16 high / low halves (split forms).
template <int F> struct Opnd { A8 x; };
template <> struct Opnd<F_BF16X3> { bf16x8_t hi, lo; };
template <> struct Opnd<F_F16X3> { f16x8_t hi, lo; };

template <int F>
__device__ __forceinline__ Opnd<F> prep(const A8& in) {
    Opnd<F> o;
    if constexpr (F == F_BF16X3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = in.t[j >> 2][j & 3];
            const __bf16 h = (__bf16)v;
            o.hi[j] = h;
            o.lo[j] = (__bf16)(v - (float)h);
        }
    } else if constexpr (F == F_F16X3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = in.t[j >> 2][j & 3];
            const _Float16 h = (_Float16)v;
            o.hi[j] = h;
            o.lo[j] = (_Float16)((v - (float)h) * LO_SCALE);
        }
    } else {
        o.x = in;
    }
    return o;
}

// out = acc + W_block in   (32 inputs -> 32 outputs), next block of the stream.
// Split forms: slot 2 mt = high halves, slot 2 mt + 1 = low halves of the A fragment of output tile mt (8 bf16 / fp16 per lane).
template <int F>
__device__ __forceinline__ A8 mm32(WStream& s, const Opnd<F>& in, A8 acc) {
    float4 w[BLOCK_SLOTS];
    s.pop(w);
    if constexpr (F == F_BF16X3) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bf16x8_t wh = __builtin_bit_cast(bf16x8_t, w[2 * mt]), wl = __builtin_bit_cast(bf16x8_t, w[2 * mt + 1]);
            acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, in.lo, acc.t[mt], 0, 0, 0);
            acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, in.hi, acc.t[mt], 0, 0, 0);
            acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, in.hi, acc.t[mt], 0, 0, 0);
        }
    } else if constexpr (F == F_F16X3) {
        // the cross terms (both low halves carry the factor LO_SCALE) collect in accumulators of their own: four independent chains
        const f16x8_t wh0 = __builtin_bit_cast(f16x8_t, w[0]), wl0 = __builtin_bit_cast(f16x8_t, w[1]);
        const f16x8_t wh1 = __builtin_bit_cast(f16x8_t, w[2]), wl1 = __builtin_bit_cast(f16x8_t, w[3]);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 lo0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, in.lo, z, 0, 0, 0);
        f32x4 lo1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, in.lo, z, 0, 0, 0);
        acc.t[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, in.hi, acc.t[0], 0, 0, 0);
        acc.t[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, in.hi, acc.t[1], 0, 0, 0);
        lo0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, in.hi, lo0, 0, 0, 0);
        lo1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, in.hi, lo1, 0, 0, 0);
        acc.t[0] = acc.t[0] + lo0 * (1.0f / LO_SCALE);
        acc.t[1] = acc.t[1] + lo1 * (1.0f / LO_SCALE);
    } else {
        acc.t[0] = mfma4(w[0], in.x.t[0], acc.t[0]);
        acc.t[1] = mfma4(w[2], in.x.t[0], acc.t[1]);
        acc.t[0] = mfma4(w[1], in.x.t[1], acc.t[0]);
        acc.t[1] = mfma4(w[3], in.x.t[1], acc.t[1]);
    }
    return acc;
}

// bias of 32 features as an accumulator initial value
__device__ __forceinline__ A8 bias8(const float* b, int q) {
    A8 o;
    o.t[0] = ld4(b + 4 * q);
    o.t[1] = ld4(b + 16 + 4 * q);
    return o;
}

__device__ __forceinline__ A8 add8(const A8& x, const A8& y) {
    A8 o;
    o.t[0] = x.t[0] + y.t[0];
    o.t[1] = x.t[1] + y.t[1];
    return o;
}

// GELU(v) = max(v, 0) - |v| Phi(-|v|) with Phi(-t) = 2^q(t), q a degree-6 fit of log2 Phi(-t) on [0, 5.6] (|error| <= 3e-7
// over the whole line): one transcendental and no division
__device__ __forceinline__ float gelu_erf(float v) {
    const float t = fminf(fabsf(v), 5.6f);
    float q = fmaf(t, 3.309289604658261e-05f, -0.0007692204089835286f);
    q = fmaf(t, q, 0.008080719038844109f);
    q = fmaf(t, q, -0.05341210588812828f);
    q = fmaf(t, q, -0.4587709605693817f);
    q = fmaf(t, q, -1.1512017250061035f);
    q = fmaf(t, q, -0.999993085861206f);
    return fmaf(-fabsf(v), exp2_fast(q), fmaxf(v, 0.f));
}

__device__ __forceinline__ A8 gelu8(A8 x) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) x.t[t][r] = gelu_erf(x.t[t][r]);
    return x;
}

// LayerNorm over the 32 features of each row: 8 in this lane, the rest in lanes ^16, ^32, ^48
__device__ __forceinline__ A8 ln8(const A8& y, const float* g, const float* b, int q) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += y.t[t][r];
    s = xor32_add(xor16_add(s));
    const float mean = s * (1.0f / 32.0f);
    A8 d;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { d.t[t][r] = y.t[t][r] - mean; v = fmaf(d.t[t][r], d.t[t][r], v); }
    v = xor32_add(xor16_add(v));
    const float rstd = rsqrt_fast(fmaf(v, 1.0f / 32.0f, 1e-5f));
    A8 o;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4 ga = ld4(g + 16 * t + 4 * q), be = ld4(b + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o.t[t][r] = fmaf(d.t[t][r] * rstd, ga[r], be[r]);
    }
    return o;
}

// FFN: y + W2 gelu(W1 h + b1) + b2, the hidden layer one 32-wide slice at a time (8 blocks of the stream)
template <int F>
__device__ __forceinline__ A8 ffn(WStream& s, const A8& y, const A8& h, const float* b1, const float* b2, int q) {
    A8 acc = add8(y, bias8(b2, q));
    const Opnd<F> ho = prep<F>(h);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const A8 f = gelu8(mm32<F>(s, ho, bias8(b1 + 32 * blk, q)));
        acc = mm32<F>(s, prep<F>(f), acc);
    }
    return acc;
}

// keys/values of this wave's rows -> LDS tile [row][K(32) | V(32)], natural feature order
__device__ __forceinline__ void store_kv(float* kv, const A8& k, const A8& v, int n, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        *reinterpret_cast<f32x4*>(kv + n * KV_LD + 16 * t + 4 * q) = k.t[t];
        *reinterpret_cast<f32x4*>(kv + n * KV_LD + 32 + 16 * t + 4 * q) = v.t[t];
    }
}

// softmax attention of this lane's two heads (q and q + 4) over keys j < nk with j <= row + mask_off (row = the lane's row
// index among the item's rows)
__device__ __forceinline__ A8 attend(const float* kv, const A8& qv, int row, int q, int nk, int mask_off) {
    const float sc = 0.5f * LOG2E_M;                    // 1 / sqrt(head_dim 4), in log2 units
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    A8 o;
    o.t[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    o.t[1] = o.t[0];
    const int last = row + mask_off;
#pragma unroll 2
    for (int j = 0; j < nk; ++j) {
        const float* kr = kv + j * KV_LD + 4 * q;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 kk = ld4(kr + 16 * t), vv = ld4(kr + 32 + 16 * t);
            float s = (qv.t[t][0] * kk[0] + qv.t[t][1] * kk[1] + qv.t[t][2] * kk[2] + qv.t[t][3] * kk[3]) * sc;
            s = j <= last ? s : -INFINITY;
            const float mn = fmaxf(m[t], s);            // key 0 is visible to every row, so mn is finite from j = 0 on
            const float al = exp2_fast(m[t] - mn), p = exp2_fast(s - mn);
            l[t] = fmaf(l[t], al, p);
            o.t[t] = o.t[t] * al + vv * p;
            m[t] = mn;
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) o.t[t] = o.t[t] * rcp_fast(l[t]);
    return o;
}

// The same attention when an item's rows fit one tile (at most 16 keys): the scores stay in registers, so the maximum is known
// before the first exponential -- one exponential per (key, head) instead of the two of the running-maximum form and no rescaling
// of the partial sums (13 vector instructions per key and head instead of 21, and the exponentials are quarter rate).
__device__ __forceinline__ A8 attend16(const float* kv, const A8& qv, int row, int q, int nk, int mask_off) {
    const float sc = 0.5f * LOG2E_M;
    const f32x4 qs[2] = {qv.t[0] * sc, qv.t[1] * sc};
    float s[2][16], m[2] = {-INFINITY, -INFINITY};
    const int last = row + mask_off;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < nk) {                                   // wave-uniform
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 kk = ld4(kv + j * KV_LD + 4 * q + 16 * t);
                const float v = fmaf(qs[t][0], kk[0], fmaf(qs[t][1], kk[1], fmaf(qs[t][2], kk[2], qs[t][3] * kk[3])));
                s[t][j] = j <= last ? v : -INFINITY;
                m[t] = fmaxf(m[t], s[t][j]);            // key 0 is visible to every row: m is finite
            }
        }
    }
    float l[2] = {0.f, 0.f};
    A8 o;
    o.t[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    o.t[1] = o.t[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < nk) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 vv = ld4(kv + j * KV_LD + 4 * q + 32 + 16 * t);
                const float p = exp2_fast(s[t][j] - m[t]);
                l[t] += p;
                o.t[t] = o.t[t] + vv * p;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) o.t[t] = o.t[t] * rcp_fast(l[t]);
    return o;
}

// NW = 1: a workgroup of four wavefronts, one batch item each (at most 16 rows per item).
// NW > 1: one batch item per workgroup of NW wavefronts; wavefront w owns rows 16 w .. 16 w + 15 of the item (memory rows
//         and trajectory tokens alike), keys / values of all rows meet in one LDS tile and the workgroup barriers around
//         each attention replace the wave-private ordering.  Rows past the end are clamped copies that nobody can see.
// MEM / TRJ: arithmetic form of the maps on memory rows / on trajectory tokens.
template <int MEM, int TRJ, int NW>
__global__ __launch_bounds__(NW == 1 ? 256 : 64 * NW, NW == 1 ? 4 : 2) void tail_mfma_kernel(const TailMfmaArgs a) {
    constexpr int NT = NW == 1 ? 256 : 64 * NW;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) float vec[V_TOTAL];
    __shared__ __attribute__((aligned(16))) float kv_all[NW == 1 ? 4 : 1][(NW == 1 ? 16 : 16 * NW) * KV_LD];
    for (int i = threadIdx.x; i < V_TOTAL; i += NT) vec[i] = a.vecs[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    // waves past the batch recompute the last item and store nothing
    const int item = NW == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    const int rb = NW == 1 ? 0 : 16 * wave;              // first row of this wavefront's tile
    const bool live = item < a.B;
    const int b = min(item, a.B - 1);
    float* kv = kv_all[NW == 1 ? wave : 0];
    const int T = a.T, TR = a.T + 1;
    const float4* img = reinterpret_cast<const float4*>(a.image) + lane;
    auto tile_sync = [&]() {
        if constexpr (NW == 1) wave_sync();              // the K/V tile is private to the wave: order its LDS traffic only
        else __syncthreads();
    };

    WStream ws;
    ws.init(img + 3 * SLOT);                             // slots 0..2: input embedding and head, read directly
    // ---- remainder of the last encoder layer on the T+1 visible rows -> memory rows XM
    A8 att, xm;
    {
        const int64_t row = (int64_t)b * a.L + min(rb + n, TR - 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            att.t[t] = ld4(a.att1 + row * 32 + 16 * t + 4 * q);
            xm.t[t] = ld4(a.x1 + row * 32 + 16 * t + 4 * q);
        }
    }
    xm = mm32<MEM>(ws, prep<MEM>(att), add8(xm, bias8(vec + V_E_BO, q)));
    xm = ffn<MEM>(ws, xm, ln8(xm, vec + V_E_LN2G, vec + V_E_LN2B, q), vec + V_E_B1, vec + V_E_B2, q);
    const Opnd<MEM> xmo = prep<MEM>(xm);               // the memory rows feed the K/V projections of every decoder layer
    // ---- decoder input: input_emb(x_t) + pos_emb
    const int nt = min(rb + n, T - 1);
    const float* xrow = a.x + ((int64_t)b * T + nt) * 4;
    A8 y;
    {
        const float4 w = img[0];
        const float xin = xrow[q];
        const float* pos = NW == 1 ? vec + V_POS + nt * 32 : a.pos + nt * 32;      // the table in LDS holds 16 rows
        y.t[0] = ld4(vec + V_DIN_B + 4 * q) + ld4(pos + 4 * q);
        y.t[1] = ld4(vec + V_DIN_B + 16 + 4 * q) + ld4(pos + 16 + 4 * q);
        y.t[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, xin, y.t[0], 0, 0, 0);
        y.t[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, xin, y.t[1], 0, 0, 0);
    }
    const int nk_self = min(T, rb + 16), nk_cross = min(TR, rb + 17);      // keys this tile's rows can see
#pragma unroll 1
    for (int l = 0; l < a.n_dec; ++l) {
        const float* lv = vec + V_LAYER0 + l * VL_SIZE;
        // causal self-attention.  The maps on trajectory tokens are never split-bf16: the sampler multiplies an
        // error of the noise estimate by up to 1 / sqrt(alpha_bar) = 160 per step, and split-bf16 products there move the
        // final trajectory by millimetres (measured: ADE against the fp32 engine 3e-3 m median, 8e-3 m worst of 1024 scenarios,
        // against 4e-4 / 8e-4 m with fp32 maps).  They are fp32, or split-fp16 (TRJ) in the fp16 engine.
        Opnd<TRJ> h = prep<TRJ>(ln8(y, lv + VL_LN1G, lv + VL_LN1B, q));
        const A8 qs = mm32<TRJ>(ws, h, bias8(lv + VL_BQKV, q));
        const A8 ks = mm32<TRJ>(ws, h, bias8(lv + VL_BQKV + 32, q));
        const A8 vs = mm32<TRJ>(ws, h, bias8(lv + VL_BQKV + 64, q));
        tile_sync();
        store_kv(kv + rb * KV_LD, ks, vs, n, q);
        tile_sync();
        A8 at = NW == 1 ? attend16(kv, qs, n, q, nk_self, 0) : attend(kv, qs, rb + n, q, nk_self, 0);
        y = mm32<TRJ>(ws, prep<TRJ>(at), add8(y, bias8(lv + VL_BO, q)));
        // cross-attention: query i sees memory rows 0..i+1; keys / values come from the memory rows (split form allowed)
        h = prep<TRJ>(ln8(y, lv + VL_LN2G, lv + VL_LN2B, q));
        const A8 qc = mm32<TRJ>(ws, h, bias8(lv + VL_BQ, q));
        const A8 km = mm32<MEM>(ws, xmo, bias8(lv + VL_BKV, q));
        const A8 vm = mm32<MEM>(ws, xmo, bias8(lv + VL_BKV + 32, q));
        tile_sync();
        store_kv(kv + rb * KV_LD, km, vm, n, q);
        tile_sync();
        at = NW == 1 ? attend16(kv, qc, n, q, nk_cross, 1) : attend(kv, qc, rb + n, q, nk_cross, 1);
        y = mm32<TRJ>(ws, prep<TRJ>(at), add8(y, bias8(lv + VL_BO2, q)));
        // feed-forward
        y = ffn<TRJ>(ws, y, ln8(y, lv + VL_LN3G, lv + VL_LN3B, q), lv + VL_B1, lv + VL_B2, q);
    }
    // ---- ln_f, head (4 outputs: only the q == 0 lanes hold real ones), scheduler update
    const A8 hf = ln8(y, vec + V_LNFG, vec + V_LNFB, q);
    f32x4 out = ld4(vec + V_HEAD_B + 4 * q);
    out = mfma4(img[1 * SLOT], hf.t[0], out);
    out = mfma4(img[2 * SLOT], hf.t[1], out);
    if (live && q == 0 && rb + n < T) {
        const StepCoef k = *a.coef;
        const int64_t gi = ((int64_t)b * T + rb + n) * 4;
        const f32x4 xv = ld4(xrow);
        f32x4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x0 = fminf(fmaxf(k.a0 * xv[c] + k.a1 * out[c], -k.clip), k.clip);
            const float eps = k.eps_mode != 0.f ? (k.b0 * xv[c] - x0) * k.b1 : out[c];
            r[c] = k.c0 * x0 + k.c1 * xv[c] + k.c2 * eps;
        }
        if (k.c3 != 0.f && a.z) r = r + ld4(a.z + gi) * k.c3;
        *reinterpret_cast<f32x4*>(a.x_next + gi) = r;
        if (a.model_out) *reinterpret_cast<f32x4*>(a.model_out + gi) = out;
    }
}

// One 32x32 block of W (row-major, leading dimension ldw) at (row0, col0) -> 4 slots of A fragments.
// Rows >= n_rows and columns >= n_cols of the block are zero.
//   fp32 form : slot mt * 2 + t = W[16 mt + m][16 t + 4 k + i], i = 0..3, for lane (m = lane & 15, k = lane >> 4)
//   split forms (split = 1 bf16, 2 fp16): slot mt * 2 = high halves, slot mt * 2 + 1 = low halves of
//               W[16 mt + m][16 (j >> 2) + 4 k + (j & 3)], j = 0..7: the k-slot order in which a lane of the accumulator layout
//               holds its 8 features
__global__ void pack_block_kernel(const float* W, int ldw, int row0, int col0, int n_rows, int n_cols, float4* dst, int split) {
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;      // 256 threads
    const int mt = slot >> 1, t = slot & 1;
    const int r = 16 * mt + (lane & 15);
    if (!split) {
        float v[4];
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * t + 4 * (lane >> 4) + i;
            v[i] = (r < n_rows && c < n_cols) ? W[(int64_t)(row0 + r) * ldw + col0 + c] : 0.f;
        }
        dst[slot * SLOT + lane] = make_float4(v[0], v[1], v[2], v[3]);
    } else if (split == F_BF16X3) {
        bf16x8_t f;
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
            const float w = (r < n_rows && c < n_cols) ? W[(int64_t)(row0 + r) * ldw + col0 + c] : 0.f;
            const __bf16 h = (__bf16)w;
            f[j] = t == 0 ? h : (__bf16)(w - (float)h);
        }
        dst[slot * SLOT + lane] = __builtin_bit_cast(float4, f);
    } else {                    // fp16 high halves / low halves times LO_SCALE
        f16x8_t f;
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
            const float w = (r < n_rows && c < n_cols) ? W[(int64_t)(row0 + r) * ldw + col0 + c] : 0.f;
            const _Float16 h = (_Float16)w;
            f[j] = t == 0 ? h : (_Float16)((w - (float)h) * LO_SCALE);
        }
        dst[slot * SLOT + lane] = __builtin_bit_cast(float4, f);
    }
}

// Slots 0..2: input embedding (K = 4: lane (m, k) holds {W[m][k], W[16 + m][k], 0, 0}) and the head (16-row tile, 2 slots).
__global__ void pack_ends_kernel(const float* Win, const float* Whead, float4* dst) {
    const int lane = threadIdx.x;
    const int m = lane & 15, k = lane >> 4;
    dst[lane] = make_float4(Win[m * 4 + k], Win[(16 + m) * 4 + k], 0.f, 0.f);
    for (int t = 0; t < 2; ++t) {
        float v[4];
        for (int i = 0; i < 4; ++i) v[i] = m < 4 ? Whead[m * 32 + 16 * t + 4 * k + i] : 0.f;
        dst[(1 + t) * SLOT + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

int tail_mfma_vec_floats() { return V_TOTAL; }
int tail_mfma_blocks(int n_dec) { return 9 + 16 * n_dec; }
size_t tail_mfma_image_bytes(int n_dec) {
    return sizeof(float4) * SLOT * (3 + (size_t)BLOCK_SLOTS * (tail_mfma_blocks(n_dec) + DEPTH));    // + look-ahead padding
}

TailMfmaVecLayout tail_mfma_vec_layout() {
    TailMfmaVecLayout v;
    v.e_bo = V_E_BO; v.e_ln2_g = V_E_LN2G; v.e_ln2_b = V_E_LN2B; v.e_b1 = V_E_B1; v.e_b2 = V_E_B2;
    v.din_b = V_DIN_B; v.pos = V_POS; v.layer0 = V_LAYER0; v.layer_size = VL_SIZE;
    v.ln1_g = VL_LN1G; v.ln1_b = VL_LN1B; v.ln2_g = VL_LN2G; v.ln2_b = VL_LN2B; v.ln3_g = VL_LN3G; v.ln3_b = VL_LN3B;
    v.bqkv = VL_BQKV; v.bo = VL_BO; v.bq = VL_BQ; v.bkv = VL_BKV; v.bo2 = VL_BO2; v.b1 = VL_B1; v.b2 = VL_B2;
    v.lnf_g = V_LNFG; v.lnf_b = V_LNFB; v.head_b = V_HEAD_B;
    return v;
}

hipError_t tail_mfma_pack_block(const float* W, int ldw, int row0, int col0, void* image, int block, int split, hipStream_t stream) {
    float4* dst = reinterpret_cast<float4*>(image) + (size_t)SLOT * (3 + (size_t)BLOCK_SLOTS * block);
    hipLaunchKernelGGL(pack_block_kernel, dim3(1), dim3(256), 0, stream, W, ldw, row0, col0, 32, 32, dst, split);
    return hipGetLastError();
}

hipError_t tail_mfma_pack_ends(const float* Win, const float* Whead, void* image, hipStream_t stream) {
    hipLaunchKernelGGL(pack_ends_kernel, dim3(1), dim3(64), 0, stream, Win, Whead, reinterpret_cast<float4*>(image));
    return hipGetLastError();
}

template <int MEM, int TRJ>
static hipError_t launch_tail_mfma(const TailMfmaArgs& a, hipStream_t stream) {
    if (a.T + 1 <= 16) {
        hipLaunchKernelGGL((tail_mfma_kernel<MEM, TRJ, 1>), dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
    } else {
        if (!a.pos) return hipErrorInvalidValue;
        hipLaunchKernelGGL((tail_mfma_kernel<MEM, TRJ, 6>), dim3(a.B), dim3(384), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t tail_mfma(const TailMfmaArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.T <= 0 || a.T + 1 > 96 || a.n_dec < 0 || a.n_dec > 4 || !a.image || !a.vecs) return hipErrorInvalidValue;
    // (form of the memory-row maps, form of the trajectory-token maps) the image was packed for
    if (a.split == F_F32 && a.traj == F_F32) return launch_tail_mfma<F_F32, F_F32>(a, stream);
    if (a.split == F_BF16X3 && a.traj == F_F32) return launch_tail_mfma<F_BF16X3, F_F32>(a, stream);
    if (a.split == F_F16X3 && a.traj == F_F16X3) return launch_tail_mfma<F_F16X3, F_F16X3>(a, stream);
    if (a.split == F_F16X3 && a.traj == F_F32) return launch_tail_mfma<F_F16X3, F_F32>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace adx

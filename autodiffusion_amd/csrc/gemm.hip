// Large-M Linear for the DiT decoders: Y = res + gate * act(X W^T + bias + pos) with bf16 MFMA operands.
//
// 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave a 64x64 sub-tile held as 2x2 accumulators of
// v_mfma_f32_32x32x16_bf16 (64 VGPRs).  K advances in steps of 32 through a double-buffered LDS image:
//   * X is fp32 in memory.  The next K-step's X tile (128x32 floats, 4 float4 per thread) is fetched into registers
//     BEFORE the MFMAs of the current step are issued and converted / written to the other LDS buffer AFTER them, so the
//     global latency hides under the matrix work and there is one barrier per K-step.
//   * W is bf16 (high halves, plus low halves in the split-operand mode), fetched the same way.
//   * LDS rows are 64 bytes (32 bf16); the 16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3), which makes
//     both the staging writes and the ds_read_b128 fragment reads (32 rows x one chunk per half-wave) bank-conflict free.
// Split-operand mode (PREC_BF16X3): x = x_hi + x_lo, w = w_hi + w_lo, three MFMAs per product (x_lo w_hi, x_hi w_lo,
// x_hi w_hi).  A row's result depends only on that row of X, so outputs do not depend on how the batch is sharded.
#include "adx_ops.h"

#include <mutex>
#include "fastmath.h"

#include <math.h>
#include <stdlib.h>

#include <type_traits>

namespace adx {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TN = 128, TK = 32;                  // tile rows: 64 * WM (WM = 2 or 4 wave rows)
constexpr int ROW_B = 64;                         // bytes per LDS row
constexpr int BTILE_B = TN * ROW_B;               // 8 KB: W image (high or low halves)

// Activations in a few instructions each: the epilogue applies this to 64 values per lane, fully unrolled, and the
// library erff / tanhf expansions would push the kernel past the instruction cache.
__device__ __forceinline__ float act_fn(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_GELU_ERF: {        // v Phi(v) through erfc (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7)
            const float ax = fabsf(v) * 0.70710678118654752f;
            const float t = rcp_fast(fmaf(0.3275911f, ax, 1.0f));
            const float p = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            const float hr = 0.5f * v * p * exp2_fast(-1.4426950408889634f * ax * ax);
            return v > 0.f ? v - hr : hr;
        }
        case ACT_GELU_TANH: {       // 0.5 v (1 + tanh(u)) = v - v / (1 + e^(2u))
            const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
            return v - v * rcp_fast(1.f + exp2_fast(2.f * 1.4426950408889634f * u));
        }
        case ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        case ACT_SILU: return v * rcp_fast(1.f + exp2_fast(-1.4426950408889634f * v));
        default: return v;
    }
}

__device__ __forceinline__ int64_t row_of(const RowMap& r, int m) {
    if (r.per_batch == 0) return m;
    return (int64_t)(m / r.per_batch) * r.batch_stride + (m % r.per_batch) + r.off;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_B + ((chunk ^ ((row >> 2) & 3)) << 4); }

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// Registers that carry the next K-step from global memory to LDS, and the per-thread addresses they are filled from.
// X: 4 rows per thread (xr + 16 WM i), floats xk..xk+3.  W: rows br + 64 i (2 per thread with 256 threads, 1 with 512).
template <bool X3, int WM> struct Stage { float4 x0, x1, x2, x3; uint4 w0, w1, l0, l1; };
template <> struct Stage<true, 4> { float4 x0, x1, x2, x3; uint4 w0, l0; };
template <> struct Stage<false, 4> { float4 x0, x1, x2, x3; uint4 w0; };
template <> struct Stage<false, 2> { float4 x0, x1, x2, x3; uint4 w0, w1; };
template <bool X3, int WM> struct StagePlan {
    const float *xp0, *xp1, *xp2, *xp3;
    const unsigned short *wp0, *wp1, *lp0, *lp1;
    int xr, xk, br, bc, in_act;
};

template <bool X3, int WM>
__device__ __forceinline__ void fetch(Stage<X3, WM>& s, const StagePlan<X3, WM>& p, int k0) {
    s.x0 = *reinterpret_cast<const float4*>(p.xp0 + k0);
    s.x1 = *reinterpret_cast<const float4*>(p.xp1 + k0);
    s.x2 = *reinterpret_cast<const float4*>(p.xp2 + k0);
    s.x3 = *reinterpret_cast<const float4*>(p.xp3 + k0);
    s.w0 = *reinterpret_cast<const uint4*>(p.wp0 + k0);
    if constexpr (WM == 2) s.w1 = *reinterpret_cast<const uint4*>(p.wp1 + k0);
    if constexpr (X3) {
        s.l0 = *reinterpret_cast<const uint4*>(p.lp0 + k0);
        if constexpr (WM == 2) s.l1 = *reinterpret_cast<const uint4*>(p.lp1 + k0);
    }
}

template <bool X3, bool SILU>
__device__ __forceinline__ void stash_x(float4 v, int r, int xk, unsigned char* Ah, unsigned char* Al) {
    if constexpr (SILU) {                         // the only input activation on the path (adaLN modulation)
        v.x = act_fn(v.x, ACT_SILU); v.y = act_fn(v.y, ACT_SILU); v.z = act_fn(v.z, ACT_SILU); v.w = act_fn(v.w, ACT_SILU);
    }
    const int o = swz(r, xk >> 3) + ((xk & 4) << 1);
    bf16x4 hi;
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(Ah + o) = hi;
    if (X3) {
        bf16x4 lo;
        lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
        lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
        *reinterpret_cast<bf16x4*>(Al + o) = lo;
    }
}

// stage layout: [A_hi (64 WM rows)] [B_hi (128 rows)] [A_lo] [B_lo]
template <int WM> __device__ __forceinline__ constexpr int atile_b() { return 64 * WM * ROW_B; }

template <bool X3, int WM, bool SILU>
__device__ __forceinline__ void stash(const Stage<X3, WM>& s, const StagePlan<X3, WM>& p, unsigned char* st) {
    unsigned char* Ah = st;
    unsigned char* Bh = st + atile_b<WM>();
    unsigned char* Al = Bh + BTILE_B;
    unsigned char* Bl = Al + atile_b<WM>();
    constexpr int XS = 16 * WM;
    stash_x<X3, SILU>(s.x0, p.xr, p.xk, Ah, Al);
    stash_x<X3, SILU>(s.x1, p.xr + XS, p.xk, Ah, Al);
    stash_x<X3, SILU>(s.x2, p.xr + 2 * XS, p.xk, Ah, Al);
    stash_x<X3, SILU>(s.x3, p.xr + 3 * XS, p.xk, Ah, Al);
    *reinterpret_cast<uint4*>(Bh + swz(p.br, p.bc)) = s.w0;
    if constexpr (WM == 2) *reinterpret_cast<uint4*>(Bh + swz(p.br + 64, p.bc)) = s.w1;
    if constexpr (X3) {
        *reinterpret_cast<uint4*>(Bl + swz(p.br, p.bc)) = s.l0;
        if constexpr (WM == 2) *reinterpret_cast<uint4*>(Bl + swz(p.br + 64, p.bc)) = s.l1;
    }
}

constexpr int EP_LD = 36;     // floats per row of a wave's 32x32 epilogue patch (144 B: 16-byte aligned, staggers banks)

// The activation of a lane's epilogue values with the switch OUTSIDE the element loop (one uniform branch per patch).
template <int RP>
__device__ __forceinline__ void act_patch(float (&v)[RP][4], int act) {
#define ADX_ACT_ALL(A)                                      \
    _Pragma("unroll") for (int rr = 0; rr < RP; ++rr)       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[rr][e] = act_fn(v[rr][e], A);
    switch (act) {
        case ACT_RELU: ADX_ACT_ALL(ACT_RELU) break;
        case ACT_GELU_ERF: ADX_ACT_ALL(ACT_GELU_ERF) break;
        case ACT_GELU_TANH: ADX_ACT_ALL(ACT_GELU_TANH) break;
        case ACT_LEAKY: ADX_ACT_ALL(ACT_LEAKY) break;
        case ACT_SILU: ADX_ACT_ALL(ACT_SILU) break;
        default: break;
    }
#undef ADX_ACT_ALL
}

// Vector form of the epilogue of one 32x32 patch: a lane owns four columns of rows (lane >> 3) + 8 rr, handled RP rows at a
// time.  Every load of a pass (position table, gate, residual) is issued before its first store and the stores go out back
// to back: vector-memory operations retire in order, so a load issued after a store waits for that store to complete.
// EXTRA = false is the instance for calls without position table, gate and residual: it contains no vector-memory load at
// all, so no wait at all (where the optional loads sit under branches the compiler emits one unconditional wait for
// everything in flight after them -- the previous patch's stores included -- whether or not a load was issued).
// F16OUT: element type of the output planes -- 0 bf16, 1 fp16, -1 whatever a.f16 says at run time.  Callers whose loop over
// patches is large (eight patches per wave) pass a constant, chosen once outside that loop: both conversion paths in one loop
// body took the body past the size at which the compiler still unrolls it, and the accumulators, then indexed by the loop
// counter, went to scratch memory.
template <int RP, bool EXTRA, int F16OUT = -1>
__device__ __forceinline__ void epilogue_rows(const LinearArgs& a, const float* ep, int m_base, int n, int nc, int lane, float4 bvec) {
    const int c4 = (lane & 7) * 4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += RP) {
        float v[RP][4];
        float4 g[RP], r[RP];
        float* yp[RP];
        int pm[RP];
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int row = (r0 + i) * 8 + (lane >> 3), m = m_base + row;
            const int mc = m < a.M ? m : a.M - 1;                 // clamped rows are loaded and computed but not stored
            const int64_t yrow = row_of(a.ymap, mc);
            yp[i] = (m < a.M && a.Y) ? a.Y + yrow * a.ldy + nc : nullptr;
            pm[i] = m < a.M ? m : -1;
            const float4 raw = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
            float4 p = zero;
            g[i] = zero; r[i] = zero;
            if constexpr (EXTRA) {
                if (a.pos) p = *reinterpret_cast<const float4*>(a.pos + (int64_t)((mc % a.pos_period) + a.pos_off) * a.ldp + n);
                if (a.gate) g[i] = *reinterpret_cast<const float4*>(a.gate + (a.gate_group ? (yrow / a.gate_group) * a.gate_period + yrow % a.gate_period : yrow) * a.ldg + n);
                if (a.res) r[i] = *reinterpret_cast<const float4*>(a.res + yrow * a.ldr + n);
            }
            v[i][0] = raw.x + bvec.x + p.x; v[i][1] = raw.y + bvec.y + p.y; v[i][2] = raw.z + bvec.z + p.z; v[i][3] = raw.w + bvec.w + p.w;
        }
        act_patch<RP>(v, a.act);
        if constexpr (EXTRA) {
            if (a.gate) {
#pragma unroll
                for (int i = 0; i < RP; ++i) { v[i][0] *= g[i].x; v[i][1] *= g[i].y; v[i][2] *= g[i].z; v[i][3] *= g[i].w; }
            }
            if (a.res) {
#pragma unroll
                for (int i = 0; i < RP; ++i) { v[i][0] += r[i].x; v[i][1] += r[i].y; v[i][2] += r[i].z; v[i][3] += r[i].w; }
            }
        }
#pragma unroll
        for (int i = 0; i < RP; ++i)
            if (yp[i]) *reinterpret_cast<float4*>(yp[i]) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        if (a.Yh) {             // the result as operand planes of the next product: [N/16][M][16], this lane's 4 columns = 8 bytes
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                if (pm[i] < 0) continue;
                const int64_t o = ((int64_t)(n >> 4) * a.M + pm[i]) * 16 + (n & 15);
                if (F16OUT < 0 ? a.f16 : F16OUT == 1) {
                    f16x4 hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hi[e] = f16_sat(v[i][e]);
                    *reinterpret_cast<f16x4*>(a.Yh + o) = hi;
                    if (a.Yl) {
                        f16x4 lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) lo[e] = f16_sat(v[i][e] - (float)hi[e]);
                        *reinterpret_cast<f16x4*>(a.Yl + o) = lo;
                    }
                } else {
                    bf16x4 hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hi[e] = (__bf16)v[i][e];
                    *reinterpret_cast<bf16x4*>(a.Yh + o) = hi;
                    if (a.Yl) {
                        bf16x4 lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) lo[e] = (__bf16)(v[i][e] - (float)hi[e]);
                        *reinterpret_cast<bf16x4*>(a.Yl + o) = lo;
                    }
                }
            }
        }
    }
}

// Y rows m_base .. m_base+31, columns n_base .. n_base+31 from a 32x32 patch of raw accumulators in LDS.
// lds_bias: the bias vector (zeros without one) in LDS; bias_reg: this lane's four bias values, fetched by the caller before
// its first patch -- either keeps the bias load (which would wait for the stores of the previous patch) out of the patch.
// With RP = 4 nothing in a patch waits for a store of the same patch; RP = 1 needs the fewest registers.
// EXTRA: the call has a position table, gate or residual (the caller tests that once, outside its loop over patches, so that
// each loop body holds one instance of this code and still unrolls: the accumulators are indexed by the loop counter).
template <int RP, bool EXTRA, int F16OUT = -1>
__device__ __forceinline__ void epilogue_patch(const LinearArgs& a, const float* ep, int m_base, int n_base, int lane, bool vec,
                                               const float* lds_bias = nullptr, const float4* bias_reg = nullptr) {
    const int c4 = (lane & 7) * 4, n = n_base + c4;
    if (vec) {
        if (n >= a.N) return;
        // column map of Y (groups are multiples of four columns here, so a lane's four columns stay together)
        const int nc = a.col_group ? (n / a.col_group) * a.col_stride + (n % a.col_group) + a.col_off : n;
        float4 bvec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias_reg) bvec = *bias_reg;
        else if (lds_bias) bvec = *reinterpret_cast<const float4*>(lds_bias + n);
        else if (a.bias) bvec = *reinterpret_cast<const float4*>(a.bias + n);
        epilogue_rows<RP, EXTRA, F16OUT>(a, ep, m_base, n, nc, lane, bvec);
        return;
    }
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
        const int row = rr * 8 + (lane >> 3), m = m_base + row;
        if (m >= a.M || n >= a.N) continue;
        const float4 raw = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
        const int64_t yrow = row_of(a.ymap, m);
        const float* posrow = a.pos ? a.pos + (int64_t)((m % a.pos_period) + a.pos_off) * a.ldp : nullptr;
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
            const int ne = n + e;
            if (ne >= a.N) break;
            float x = e == 0 ? raw.x : e == 1 ? raw.y : e == 2 ? raw.z : raw.w;
            if (a.bias) x += a.bias[ne];
            if (posrow) x += posrow[ne];
            x = act_fn(x, a.act);
            if (a.gate) x *= a.gate[(a.gate_group ? (yrow / a.gate_group) * a.gate_period + yrow % a.gate_period : yrow) * a.ldg + ne];
            if (a.res) x += a.res[yrow * a.ldr + ne];
            const int nc = a.col_group ? (ne / a.col_group) * a.col_stride + (ne % a.col_group) + a.col_off : ne;
            a.Y[yrow * a.ldy + nc] = x;
        }
    }
}

// The epilogue of this call can use 16-byte accesses for every operand.
__device__ __forceinline__ bool linear_vec_ok_dev(const LinearArgs& a) {
    return (a.col_group == 0 || ((a.col_group | a.col_stride | a.col_off) & 3) == 0) && (a.N & 3) == 0 && (a.ldy & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 && (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
           (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
           (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
           (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
}

// The bias of the two 32-column patches of a wave's 64-column sub-tile, as this lane will use them in the vector epilogue.
__device__ __forceinline__ void load_patch_bias(const LinearArgs& a, int n_base, int lane, bool vec, float4 (&bv)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_base + j * 32 + (lane & 7) * 4;
        bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec && a.bias && n < a.N) bv[j] = *reinterpret_cast<const float4*>(a.bias + n);
    }
}

// MFMAs of one staged K-step (two k16 sub-steps) for this wave's 64x64 sub-tile.
template <bool X3, int WM>
__device__ __forceinline__ void compute(f32x16 (&acc)[2][2], const unsigned char* st, int wr, int wc, int fr, int fc) {
    const unsigned char* Bh = st + atile_b<WM>();
    const unsigned char* Al = Bh + BTILE_B;
    const unsigned char* Bl = Al + atile_b<WM>();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[2], bh[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(st + swz(wr * 64 + i * 32 + fr, 2 * ks + fc));
            bh[i] = *reinterpret_cast<const bf16x8*>(Bh + swz(wc * 64 + i * 32 + fr, 2 * ks + fc));
        }
        if (X3) {
            bf16x8 al[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                al[i] = *reinterpret_cast<const bf16x8*>(Al + swz(wr * 64 + i * 32 + fr, 2 * ks + fc));
                bl[i] = *reinterpret_cast<const bf16x8*>(Bl + swz(wc * 64 + i * 32 + fr, 2 * ks + fc));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
}

// Scheduling hint for one half-iteration: spread the conversion / LDS-write instructions of the stash between the MFMAs
// instead of after them (the MFMA pipe runs 8 passes per instruction; the vector ALU is free meanwhile).
template <bool X3>
__device__ __forceinline__ void interleave() {
    constexpr int N_MFMA = X3 ? 24 : 8;
    constexpr int VALU_PER = X3 ? 4 : 6;
#pragma unroll
    for (int g = 0; g < N_MFMA; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);     // a few VALU
        if (g % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // an LDS write
    }
}

template <bool X3, int WM, bool SILU>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void linear_big_kernel(const LinearArgs a, int tiles_n, int n_tiles) {
    constexpr int TM = 64 * WM, NT = 128 * WM;
    constexpr int STAGE_B = (X3 ? 2 : 1) * (atile_b<WM>() + BTILE_B);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // 2 * STAGE_B

    // workgroup -> tile: each XCD (blockIdx % 8) walks a contiguous range of tiles, n fastest, so the workgroups resident
    // on one XCD share X row-tiles and W column-tiles through that XCD's L2
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int M = a.M, N = a.N, K = a.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging assignments
    const int xr = tid >> 3, xk = (tid & 7) * 4;
    const float* xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xrow[i] = a.X + row_of(a.xmap, min(m0 + xr + 16 * WM * i, M - 1)) * a.ldx + xk;   // clamped rows are never stored
    const int br = tid >> 2, bc = tid & 3;
    const unsigned short* wrow[2];
    const unsigned short* wlrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t off = (int64_t)min(n0 + br + 64 * i, N - 1) * K + bc * 8;
        wrow[i] = reinterpret_cast<const unsigned short*>(a.W) + off;
        wlrow[i] = X3 ? reinterpret_cast<const unsigned short*>(a.Wlo) + off : nullptr;
    }
    const int in_act = a.in_act;
    (void)NT;

    const StagePlan<X3, WM> plan{xrow[0], xrow[1], xrow[2], xrow[3], wrow[0], wrow[1], wlrow[0], wlrow[1], xr, xk, br, bc, in_act};
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two K-steps in flight: while step kt is multiplied, step kt+1 sits in one register set (fetched one iteration ago,
    // written to the other LDS buffer after this iteration's MFMAs) and step kt+2 is being fetched into the other set.
    // K is a multiple of 64, so the steps come in pairs.
    const int nk = K / TK;
    const int fr = lane & 31, fc = lane >> 5;
    unsigned char* buf0 = smem;
    unsigned char* buf1 = smem + STAGE_B;
    Stage<X3, WM> sa, sb;
    // Fetches and stashes are unconditional (past the end they re-read the last K-step, which nobody multiplies): with a
    // fixed number of loads in flight the compiler can wait for exactly the older register set and leave the newer one flying.
    const int klast = (nk - 1) * TK;
    fetch<X3, WM>(sa, plan, 0);
    fetch<X3, WM>(sb, plan, TK);
    stash<X3, WM, SILU>(sa, plan, buf0);
    fetch<X3, WM>(sa, plan, min(2 * TK, klast));
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        compute<X3, WM>(acc, buf0, wr, wc, fr, fc);
        stash<X3, WM, SILU>(sb, plan, buf1);
        interleave<X3>();
        fetch<X3, WM>(sb, plan, min((kt + 3) * TK, klast));
        __syncthreads();
        compute<X3, WM>(acc, buf1, wr, wc, fr, fc);
        stash<X3, WM, SILU>(sa, plan, buf0);
        interleave<X3>();
        fetch<X3, WM>(sa, plan, min((kt + 4) * TK, klast));
        __syncthreads();
    }

    // ---- epilogue.  Each wave passes its four 32x32 accumulators through a private LDS patch (the staging buffers are
    // free now), which turns the MFMA layout (lane = column, register = row) into rows of 32 consecutive columns: the
    // bias / activation / gate / residual code then runs in a ROLLED loop on float4s (small code, 16-byte stores).
    __syncthreads();                                        // every wave is done reading the last K-step
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const bool vec = (a.col_group == 0 || ((a.col_group | a.col_stride | a.col_off) & 3) == 0) && (N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 &&
                     (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                     (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
                     (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
                     (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
    float4 bv[2];
    load_patch_bias(a, n0 + wc * 64, lane, vec, bv);
    auto patches = [&](auto extra, auto f16out) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = t >> 1, j = t & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * fc) * EP_LD + fr] = acc[i][j][r];
            epilogue_patch<4, decltype(extra)::value, decltype(f16out)::value>(a, ep, m0 + wr * 64 + i * 32, n0 + wc * 64 + j * 32, lane, vec, nullptr, &bv[j]);
        }
    };
    typedef std::integral_constant<int, 0> out_bf16;
    typedef std::integral_constant<int, 1> out_f16;
    if (a.f16 && a.Yh) { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_f16{}); else patches(std::false_type{}, out_f16{}); }
    else { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_bf16{}); else patches(std::false_type{}, out_bf16{}); }
}

// ============================================================================================================
// Wave-specialised form for large problems: a 256x128 tile per 512-thread workgroup.  Waves 0-3 only multiply (each a
// 128x64 sub-tile, 4x2 accumulators = 128 VGPRs, no vector-ALU work in the loop); waves 4-7 only stage: fetch the fp32
// X rows and bf16 W rows two K-steps ahead, convert / split them and write the LDS image of the next K-step.  A SIMD
// hosts one wave of each kind (waves are placed on SIMDs cyclically), so the conversion work and the LDS writes run
// under the other wave's MFMAs instead of between them.  One barrier per K-step.
// ============================================================================================================
constexpr int WS_TM = 256;
constexpr int WS_ATILE = WS_TM * ROW_B;           // 16 KB

template <bool X3> struct WsStage { float4 x[8]; uint4 w[2]; uint4 l[2]; };

template <bool X3, bool SILU>
__global__ __launch_bounds__(512, 1) void linear_ws_kernel(const LinearArgs a, int tiles_n, int n_tiles) {
    constexpr int STAGE_B = (X3 ? 2 : 1) * (WS_ATILE + BTILE_B);          // [A_hi][B_hi][A_lo][B_lo]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE_B
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int m0 = (tile / tiles_n) * WS_TM, n0 = (tile % tiles_n) * TN;
    const int M = a.M, N = a.N, K = a.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = K / TK;
    unsigned char* buf0 = smem;
    unsigned char* buf1 = smem + STAGE_B;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers (256 threads)
        const int pt = tid - 256;
        const int xr = pt >> 3, xk = (pt & 7) * 4;                // X rows xr + 32 i (i < 8), floats xk..xk+3
        const int br = pt >> 2, bc = pt & 3;                      // W rows br + 64 i (i < 2), chunk bc
        const float* xp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xp[i] = a.X + row_of(a.xmap, min(m0 + xr + 32 * i, M - 1)) * a.ldx + xk;
        const unsigned short* wp[2];
        const unsigned short* lp[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t off = (int64_t)min(n0 + br + 64 * i, N - 1) * K + bc * 8;
            wp[i] = reinterpret_cast<const unsigned short*>(a.W) + off;
            lp[i] = X3 ? reinterpret_cast<const unsigned short*>(a.Wlo) + off : nullptr;
        }
        auto fetch_ws = [&](WsStage<X3>& g, int k0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g.x[i] = *reinterpret_cast<const float4*>(xp[i] + k0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                g.w[i] = *reinterpret_cast<const uint4*>(wp[i] + k0);
                if constexpr (X3) g.l[i] = *reinterpret_cast<const uint4*>(lp[i] + k0);
            }
        };
        auto stash_ws = [&](const WsStage<X3>& g, unsigned char* st) {
            unsigned char* Ah = st;
            unsigned char* Bh = st + WS_ATILE;
            unsigned char* Al = Bh + BTILE_B;
            unsigned char* Bl = Al + WS_ATILE;
#pragma unroll
            for (int i = 0; i < 8; ++i) stash_x<X3, SILU>(g.x[i], xr + 32 * i, xk, Ah, Al);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<uint4*>(Bh + swz(br + 64 * i, bc)) = g.w[i];
                if constexpr (X3) *reinterpret_cast<uint4*>(Bl + swz(br + 64 * i, bc)) = g.l[i];
            }
        };
        const int klast = (nk - 1) * TK;
        WsStage<X3> ga, gb;
        fetch_ws(ga, 0);
        fetch_ws(gb, TK);
        stash_ws(ga, buf0);
        fetch_ws(ga, min(2 * TK, klast));
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            stash_ws(gb, buf1);
            fetch_ws(gb, min((kt + 3) * TK, klast));
            __syncthreads();
            stash_ws(ga, buf0);
            fetch_ws(ga, min((kt + 4) * TK, klast));
            __syncthreads();
        }
        __syncthreads();                                          // pairs with the consumers' barrier before the epilogue
        return;
    }

    // ---------------------------------------------------------------------- consumers (waves 0-3 as 2x2, 128x64 each)
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fc = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute_ws = [&](const unsigned char* st) {
        const unsigned char* Bh = st + WS_ATILE;
        const unsigned char* Al = Bh + BTILE_B;
        const unsigned char* Bl = Al + WS_ATILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(Bh + swz(wc * 64 + j * 32 + fr, 2 * ks + fc));
                if constexpr (X3) bl[j] = *reinterpret_cast<const bf16x8*>(Bl + swz(wc * 64 + j * 32 + fr, 2 * ks + fc));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + swz(wr * 128 + i * 32 + fr, 2 * ks + fc));
                if constexpr (X3) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Al + swz(wr * 128 + i * 32 + fr, 2 * ks + fc));
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    __syncthreads();                                              // K-step 0 is staged
    for (int kt = 0; kt < nk; kt += 2) {
        compute_ws(buf0);
        __syncthreads();
        compute_ws(buf1);
        __syncthreads();
    }
    __syncthreads();                                              // all consumers are done reading the staging buffers
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const bool vec = (a.col_group == 0 || ((a.col_group | a.col_stride | a.col_off) & 3) == 0) && (N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 &&
                     (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                     (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
                     (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
                     (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
    float4 bv[2];
    load_patch_bias(a, n0 + wc * 64, lane, vec, bv);
    auto patches = [&](auto extra, auto f16out) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = t >> 1, j = t & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * fc) * EP_LD + fr] = acc[i][j][r];
            epilogue_patch<4, decltype(extra)::value, decltype(f16out)::value>(a, ep, m0 + wr * 128 + i * 32, n0 + wc * 64 + j * 32, lane, vec, nullptr, &bv[j]);
        }
    };
    typedef std::integral_constant<int, 0> out_bf16;
    typedef std::integral_constant<int, 1> out_f16;
    if (a.f16 && a.Yh) { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_f16{}); else patches(std::false_type{}, out_f16{}); }
    else { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_bf16{}); else patches(std::false_type{}, out_bf16{}); }
}

template <bool X3, bool SILU>
static hipError_t launch_ws2(const LinearArgs& a, hipStream_t stream) {
    constexpr size_t smem = 2 * (X3 ? 2 : 1) * (size_t)(WS_ATILE + BTILE_B);
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_ws_kernel<X3, SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (attr != hipSuccess) return attr;
    const int tiles_m = (a.M + WS_TM - 1) / WS_TM, tiles_n = (a.N + TN - 1) / TN;
    const int64_t n_tiles = (int64_t)tiles_m * tiles_n;
    if (n_tiles > (1 << 30)) return hipErrorInvalidValue;
    const int blocks = (int)((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL((linear_ws_kernel<X3, SILU>), dim3(blocks), dim3(512), smem, stream, a, tiles_n, (int)n_tiles);
    return hipGetLastError();
}

template <bool X3>
static hipError_t launch_ws(const LinearArgs& a, hipStream_t stream) {
    return a.in_act == ACT_SILU ? launch_ws2<X3, true>(a, stream) : launch_ws2<X3, false>(a, stream);
}


// ============================================================================================================
// LDS-DMA form: 256x256 tile, 8 waves as 2x4, each a 128x64 sub-tile (4x2 accumulators of 32x32x16 MFMAs).  Both
// operands arrive as K-BLOCKED bf16 planes -- plane[kb][row][16]: the 16 elements of K-step kb of every row, rows
// contiguous -- written once per call by stage_x_kernel / stage_w_kernel (high halves, plus low halves for split
// operands).  A K-step (k = 16) of a tile is then one contiguous run of 32-byte rows and is staged by 16-byte
// global_load_lds copies of whole cache lines, with no vector-ALU work and no registers.
//   * A ring of R LDS buffers (R = 4 x 32 KB with split operands, 8 x 16 KB without): the copies of R - 1 K-steps are in
//     flight while one is multiplied -- the LDS-DMA path needs ~100 KB in flight per CU to run at its rate.
//   * The fragments of K-step kt+1 are read from LDS while the MFMAs of step kt run (two register sets); one counted
//     wait and one barrier per K-step.
//   * LDS image: rows of 32 bytes; the 16-byte half c of row r sits at half c ^ ((r >> 3) & 1), which makes the
//     ds_read_b128 fragment reads (32 rows x one half per half-wave) bank-conflict free.  LDS-DMA writes 64 lanes x 16 B
//     to consecutive LDS bytes, so the swizzle is applied to the SOURCE address of each lane.
// ============================================================================================================
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const unsigned short* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// The same copy issued from inline assembly, for kernels that overlap copies with LDS reads of OTHER buffers: the compiler
// makes every LDS read wait for all vector-memory operations while a copy it knows of is in flight (the copy may alias
// the read), which would serialise them.  The kernel orders the copies itself with counted waits and barriers.
__device__ __forceinline__ void dma16_unordered(const unsigned short* g, unsigned char* l) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)l);
    unsigned keep;          // m0 is restored: the compiler does not expect inline assembly to change it
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(g) : "memory");
}

// The same copy with a wave-uniform base (SGPR pair), a 32-bit byte offset per lane and a wave-uniform LDS address: no
// vector arithmetic per copy.  m0 is restored: the compiler does not expect inline assembly to change it.
__device__ __forceinline__ void dma16_sbase(const void* base_uniform, unsigned lane_off_bytes, unsigned lds_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_uniform), "v"(lane_off_bytes), "s"(base_uniform) : "memory");
}

struct StagedOperands {
    const unsigned short *xh, *xl, *wh, *wl;      // K-blocked planes: [K/16][M][16] and [K/16][N][16]
};

constexpr int GT = 256;                           // tile rows and columns
constexpr int GK = 16;                            // K-step
constexpr int GROW_B = 32;                        // bytes per image row
constexpr int GIMG_B = GT * GROW_B;               // 8 KB: one operand image (256 rows x 16 bf16)

__device__ __forceinline__ int gswz(int row, int half) { return row * GROW_B + ((half ^ ((row >> 3) & 1)) << 4); }

template <bool X3> struct Frags { bf16x8 ah[4], bh[2], al[4], bl[2]; };
template <> struct Frags<false> { bf16x8 ah[4], bh[2]; };

template <bool X3>
__device__ __forceinline__ void load_frags(Frags<X3>& f, const unsigned char* st, int wr, int wc, int fr, int fc) {
#pragma unroll
    for (int j = 0; j < 2; ++j) f.bh[j] = *reinterpret_cast<const bf16x8*>(st + GIMG_B + gswz(wc * 64 + j * 32 + fr, fc));
#pragma unroll
    for (int i = 0; i < 4; ++i) f.ah[i] = *reinterpret_cast<const bf16x8*>(st + gswz(wr * 128 + i * 32 + fr, fc));
    if constexpr (X3) {
#pragma unroll
        for (int j = 0; j < 2; ++j) f.bl[j] = *reinterpret_cast<const bf16x8*>(st + 3 * GIMG_B + gswz(wc * 64 + j * 32 + fr, fc));
#pragma unroll
        for (int i = 0; i < 4; ++i) f.al[i] = *reinterpret_cast<const bf16x8*>(st + 2 * GIMG_B + gswz(wr * 128 + i * 32 + fr, fc));
    }
}

template <bool X3>
__device__ __forceinline__ void mfma_frags(f32x16 (&acc)[4][2], const Frags<X3>& f) {
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
}

template <bool X3, int ABL>
__global__ __launch_bounds__(512, 1) void linear_glds_kernel(const LinearArgs a, const StagedOperands so, int tiles_n, int n_tiles) {
    constexpr int STAGE_B = (X3 ? 4 : 2) * GIMG_B;                             // plain: [A][B]; split: [A_hi][B_hi][A_lo][B_lo]
    constexpr int R = X3 ? 4 : 8;                                              // ring buffers: 128 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // R * STAGE_B
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int m0 = (tile / tiles_n) * GT, n0 = (tile % tiles_n) * GT;
    const int M = a.M, N = a.N, K = a.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;

    // ---- copy plan: a slab = 32 image rows = 1 KB = one wave-instruction; wave w copies slab w of every image.
    // Offsets are in elements of the planes, K-step 0.
    const int r = wave * 32 + (lane >> 1);
    const unsigned half = 8u * (unsigned)((lane & 1) ^ ((r >> 3) & 1));
    const unsigned xoff = (unsigned)min(m0 + r, M - 1) * 16u + half, woff = (unsigned)min(n0 + r, N - 1) * 16u + half;
    const unsigned xstep = (unsigned)M * 16u, wstep = (unsigned)N * 16u;      // elements per K-step of a plane
    const int slab = wave * 1024;
    const int nk = K / GK;
    auto issue = [&](int kt) {                       // K-step kt (clamped: steps past the end repeat the last one) -> ring slot kt % R
        unsigned char* st = smem + (kt % R) * STAGE_B;
        const int kc = min(kt, nk - 1);
        const unsigned xo = (unsigned)kc * xstep + xoff, wo = (unsigned)kc * wstep + woff;
        dma16(so.xh + xo, st + slab);
        dma16(so.wh + wo, st + GIMG_B + slab);
        if constexpr (X3) {
            dma16(so.xl + xo, st + 2 * GIMG_B + slab);
            dma16(so.wl + wo, st + 3 * GIMG_B + slab);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int fr = lane & 31, fc = lane >> 5;

    // wait until at most R - 2 K-steps of this wave's copies are outstanding
    auto wait_copies = [&]() {
        if constexpr (X3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // (R - 2) * 4 copies per K-step
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // (R - 2) * 2
    };
#pragma unroll
    for (int t = 0; t < R; ++t) issue(t);
    wait_copies();                                   // K-steps 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    Frags<X3> f0, f1;
    load_frags<X3>(f0, smem, wr, wc, fr, fc);
    __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();                    // everyone holds step 0 in registers: its slot can be refilled
    // Each K-step has a LOAD phase and a MULTIPLY phase, each closed by a barrier.  The two wave groups (wr = 0 / 1: the
    // two waves that share a SIMD belong to different groups) run one phase apart -- group 1 takes one extra barrier here,
    // group 0 one after the loop -- so on every SIMD one wave multiplies while the other reads fragments and starts copies,
    // and a barrier is the hand-over of the matrix pipe instead of a point where it drains.
    //   LOAD(kt):      start the copies of step kt+R into the slot of step kt (every wave read that slot in LOAD(kt-1), which
    //                  ended at least one barrier ago for both groups); wait for this wave's copies of step kt+2; read the
    //                  fragments of step kt+1 (every wave waited for its copies of kt+1 in LOAD(kt-1)).
    //   MULTIPLY(kt):  MFMAs on the fragments of step kt.
    // Copies and reads past the end repeat the last K-step.
    if (wr == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; kt += 2) {
        if (!(ABL & 2)) { issue(kt + R); wait_copies(); }
        if (!(ABL & 1)) load_frags<X3>(f1, smem + ((kt + 1) % R) * STAGE_B, wr, wc, fr, fc);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 4)) mfma_frags<X3>(acc, f0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 2)) { issue(kt + 1 + R); wait_copies(); }
        if (!(ABL & 1)) load_frags<X3>(f0, smem + ((kt + 2) % R) * STAGE_B, wr, wc, fr, fc);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 4)) mfma_frags<X3>(acc, f1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // every wave is done reading, no copy is in flight
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const bool vec = (a.col_group == 0 || ((a.col_group | a.col_stride | a.col_off) & 3) == 0) && (N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 &&
                     (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                     (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
                     (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
                     (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
    if (ABL & 8) {          // measurement aid: keep the accumulators alive with one store per lane instead of the epilogue
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += acc[t >> 1][t & 1][q];
        a.Y[(int64_t)min(m0 + wr * 128 + fr, M - 1) * a.ldy + min(n0 + wc * 64 + fc, N - 1)] = sum;
    }
    float4 bv[2];
    load_patch_bias(a, n0 + wc * 64, lane, vec, bv);
    auto patches = [&](auto extra, auto f16out) {
#pragma unroll
        for (int t = 0; t < ((ABL & 8) ? 0 : 8); ++t) {
            const int i = t >> 1, j = t & 1;
#pragma unroll
            for (int q = 0; q < 16; ++q) ep[((q & 3) + 8 * (q >> 2) + 4 * fc) * EP_LD + fr] = acc[i][j][q];
            epilogue_patch<4, decltype(extra)::value, decltype(f16out)::value>(a, ep, m0 + wr * 128 + i * 32, n0 + wc * 64 + j * 32, lane, vec, nullptr, &bv[j]);
        }
    };
    typedef std::integral_constant<int, 0> out_bf16;
    typedef std::integral_constant<int, 1> out_f16;
    if (a.f16 && a.Yh) { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_f16{}); else patches(std::false_type{}, out_f16{}); }
    else { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_bf16{}); else patches(std::false_type{}, out_bf16{}); }
}

// ============================================================================================================
// Two-workgroups-per-CU form of the LDS-DMA kernel: a 256x128 tile per 256-thread workgroup, 4 waves as 2x2, each a 128x64
// sub-tile (4x2 accumulators, the same per-wave work as above).  A CU holds TWO such workgroups (72 KB of LDS and
// <= 256 registers each), one wave of each per SIMD, and nothing synchronises them: while one workgroup writes its
// 128 KB of fp32 results (no MFMAs for thousands of cycles) or waits at its barrier, the other one's main loop has the matrix
// pipe to itself, so the epilogue, the copy waits and the fragment-read latency of one tile hide under the products of
// another -- with one workgroup per CU they were serial (the 256x256 kernel spends 20 % of its time in the epilogue).
// Per K-step (k = 16) and workgroup: ring slot = [A_hi 8 KB][B_hi 4 KB][A_lo 8 KB][B_lo 4 KB] (split) or [A][B] (plain);
// R = 3 slots of 24 KB or 6 of 12 KB: the copies of R - 1 steps are in flight while one step is multiplied; ONE barrier
// per step: it publishes step kt + 1 (every wave waited for its own copies) and frees the slot of step kt (every wave
// holds those fragments in registers).  Copies are issued from inline assembly and ordered by counted waits (see
// dma16_unordered): the fragment reads of step kt + 1 and the copies of step kt + R run under the MFMAs of step kt.
// ============================================================================================================
constexpr int G2_TM = 256, G2_TN = 128;
constexpr int G2_A_B = G2_TM * GROW_B;            // 8 KB
constexpr int G2_B_B = G2_TN * GROW_B;            // 4 KB

template <bool X3>
__device__ __forceinline__ void load_frags2(Frags<X3>& f, const unsigned char* st, int wr, int wc, int fr, int fc) {
#pragma unroll
    for (int j = 0; j < 2; ++j) f.bh[j] = *reinterpret_cast<const bf16x8*>(st + G2_A_B + gswz(wc * 64 + j * 32 + fr, fc));
#pragma unroll
    for (int i = 0; i < 4; ++i) f.ah[i] = *reinterpret_cast<const bf16x8*>(st + gswz(wr * 128 + i * 32 + fr, fc));
    if constexpr (X3) {
        constexpr int LO = G2_A_B + G2_B_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) f.bl[j] = *reinterpret_cast<const bf16x8*>(st + LO + G2_A_B + gswz(wc * 64 + j * 32 + fr, fc));
#pragma unroll
        for (int i = 0; i < 4; ++i) f.al[i] = *reinterpret_cast<const bf16x8*>(st + LO + gswz(wr * 128 + i * 32 + fr, fc));
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// F16: the planes hold fp16 values.  With X3, X arrives as high and low halves and W rounded once to fp16: a product is TWO matrix
// instructions, x_lo w + x_hi w (the slot of the W low halves in the LDS image stays empty).  Without X3 both operands are rounded
// once to fp16: ONE matrix instruction, the single-pass kernel on another element type.  fp16 carries 11 significant bits against
// bf16's 8: one rounding costs 2^-12 relative instead of 2^-9.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <bool X3, int ABL, int S = 1, bool F16 = false>
__global__ __launch_bounds__(256, 2) void linear_glds2_kernel(const LinearArgs a, const StagedOperands so, int tiles_n, int n_tiles, int pw) {
    constexpr int STAGE_B = (X3 ? 2 : 1) * (G2_A_B + G2_B_B);                  // 24 KB / 12 KB
    constexpr int R = X3 ? 3 : 6;                                              // 72 KB
    constexpr int P = X3 ? (F16 ? 5 : 6) : 3;                                  // copy instructions per wave per K-step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // R * STAGE_B
    // workgroup -> tile: XCD x (blockIdx % 8) walks a contiguous range of a PANEL order -- column panels of `pw` tiles, inside a
    // panel row by row -- so the ~64 workgroups resident on an XCD at one time form a block of about 8 x 8 tiles: per K-step
    // they fetch 8 row-slabs of X and 8 of W into that XCD's L2 (a 2 x 32 block of the row-major order fetches 2 + 32)
    const int per_xcd = (n_tiles + 7) / 8;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (lin >= n_tiles) return;
    const int tiles_m = n_tiles / tiles_n;
    int tm, tn;
    if (pw > 0) {
        const int full = (tiles_n / pw) * pw;                    // columns covered by whole panels
        const int in_full = tiles_m * full;
        if (lin < in_full) { const int p = lin / (tiles_m * pw), q = lin % (tiles_m * pw); tm = q / pw; tn = p * pw + q % pw; }
        else { const int w = tiles_n - full, q = lin - in_full; tm = q / w; tn = full + q % w; }
    } else if (pw < 0) {                                         // experiment: bands of -pw tile rows, column by column inside a band
        const int bh = -pw, full = (tiles_m / bh) * bh, in_full = full * tiles_n;
        if (lin < in_full) { const int p = lin / (bh * tiles_n), q = lin % (bh * tiles_n); tn = q / bh; tm = p * bh + q % bh; }
        else { const int h = tiles_m - full, q = lin - in_full; tn = q / h; tm = full + q % h; }
    } else { tm = lin / tiles_n; tn = lin % tiles_n; }
    const int m0 = tm * G2_TM, n0 = tn * G2_TN;
    const int M = a.M, N = a.N, K = a.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: everything derived from it lives in scalar registers
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fc = lane >> 5;

    // ---- copy plan: a slab = 32 image rows = 1 KB = one wave-instruction.  Wave w copies slabs w and w + 4 of the A images
    // and slab w of the B images; the swizzle of the image is applied to the SOURCE element of each lane.
    const int r = wave * 32 + (lane >> 1);
    const unsigned half = 8u * (unsigned)((lane & 1) ^ ((r >> 3) & 1));
    const unsigned xoff0 = (unsigned)min(m0 + r, M - 1) * 16u + half, xoff1 = (unsigned)min(m0 + r + 128, M - 1) * 16u + half;
    const unsigned woff = (unsigned)min(n0 + r, N - 1) * 16u + half;
    const unsigned xstep = (unsigned)M * 16u, wstep = (unsigned)N * 16u;
    const int nk = K / GK;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem + (unsigned)wave * 1024u;       // LDS address of this wave's slab 0 in slot 0
    auto issue = [&](int kt, int slot) {             // K-step kt (clamped: steps past the end repeat the last one) -> ring slot
        unsigned char* st = smem + slot * STAGE_B + wave * 1024;
        const int kc = min(kt, nk - 1);
        const unsigned xo = (unsigned)kc * xstep, wo = (unsigned)kc * wstep + woff;
        dma16_unordered(so.xh + xo + xoff0, st);
        dma16_unordered(so.xh + xo + xoff1, st + 4096);
        dma16_unordered(so.wh + wo, st + G2_A_B);
        if constexpr (X3) {
            constexpr int LO = G2_A_B + G2_B_B;
            dma16_unordered(so.xl + xo + xoff0, st + LO);
            dma16_unordered(so.xl + xo + xoff1, st + LO + 4096);
            if constexpr (!F16) dma16_unordered(so.wl + wo, st + LO + G2_A_B);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // Ring discipline with S K-steps per barrier (R >= 2 S slots, step k lives in slot k % R).  At the barrier that opens the
    // iteration of steps kt .. kt + S - 1, step kt is in registers, everybody's copies of steps kt + 1 .. kt + S have landed
    // and nobody reads the slots of steps kt - S + 1 .. kt any more: those are refilled with steps kt + R - S + 1 .. kt + R
    // while the iteration multiplies.  A wave may keep its copies of the youngest R - 2 S steps in flight across the barrier.
    static_assert(R >= 2 * S, "ring too short for this many K-steps per barrier");
#pragma unroll
    for (int t = 0; t <= R - S; ++t) issue(t, t);
    wait_vmcnt<(R - S) * P>();                       // step 0 has landed
    __builtin_amdgcn_s_barrier();
    Frags<X3> f0, f1;
    load_frags2<X3>(f0, smem, wr, wc, fr, fc);
    __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0)
    int slot = 0;                                    // ring slot of step kt
    constexpr int LO = G2_A_B + G2_B_B;
    // one K-step: its MFMA groups (one accumulator = one group) with the fragment reads of the next step (slot rs) and the copies
    // of K-step kc into slot ds spread between them, so that no burst of LDS reads or copy issues sits in front of an MFMA
    auto sub = [&](bool live, Frags<X3>& cur, Frags<X3>& nxt, int rs, int ds, int kc) {
        const unsigned char* ns = smem + rs * STAGE_B;
        const unsigned lst = lds0 + (unsigned)(ds * STAGE_B);                    // this wave's first slab of the slot (LDS address)
        const unsigned short* sxh = so.xh + (size_t)kc * xstep;
        const unsigned short* swh = so.wh + (size_t)kc * wstep;
        const unsigned short* sxl = X3 ? so.xl + (size_t)kc * xstep : nullptr;
        const unsigned short* swl = (X3 && !F16) ? so.wl + (size_t)kc * wstep : nullptr;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int i = g >> 1, j = g & 1;
            // reads: g 0: bh0, ah0 | 1: bh1, ah1 | ...   (order: what the next step's first groups need comes first)
            if constexpr (!(ABL & 1)) {
            if (g == 0) { nxt.bh[0] = *reinterpret_cast<const bf16x8*>(ns + G2_A_B + gswz(wc * 64 + fr, fc));
                          nxt.ah[0] = *reinterpret_cast<const bf16x8*>(ns + gswz(wr * 128 + fr, fc)); }
            if (g == 1) { nxt.bh[1] = *reinterpret_cast<const bf16x8*>(ns + G2_A_B + gswz(wc * 64 + 32 + fr, fc));
                          nxt.ah[1] = *reinterpret_cast<const bf16x8*>(ns + gswz(wr * 128 + 32 + fr, fc)); }
            if (g == 2) { nxt.ah[2] = *reinterpret_cast<const bf16x8*>(ns + gswz(wr * 128 + 64 + fr, fc));
                          nxt.ah[3] = *reinterpret_cast<const bf16x8*>(ns + gswz(wr * 128 + 96 + fr, fc)); }
            if constexpr (X3) {
                if (g == 3) { if constexpr (!F16) nxt.bl[0] = *reinterpret_cast<const bf16x8*>(ns + LO + G2_A_B + gswz(wc * 64 + fr, fc));
                              nxt.al[0] = *reinterpret_cast<const bf16x8*>(ns + LO + gswz(wr * 128 + fr, fc)); }
                if (g == 4) { if constexpr (!F16) nxt.bl[1] = *reinterpret_cast<const bf16x8*>(ns + LO + G2_A_B + gswz(wc * 64 + 32 + fr, fc));
                              nxt.al[1] = *reinterpret_cast<const bf16x8*>(ns + LO + gswz(wr * 128 + 32 + fr, fc)); }
                if (g == 5) nxt.al[2] = *reinterpret_cast<const bf16x8*>(ns + LO + gswz(wr * 128 + 64 + fr, fc));
                if (g == 6) nxt.al[3] = *reinterpret_cast<const bf16x8*>(ns + LO + gswz(wr * 128 + 96 + fr, fc));
            }
            }
            if constexpr (!(ABL & 2)) {
            if (g == 1) dma16_sbase(sxh, 2u * xoff0, lst);
            if (g == 2) dma16_sbase(sxh, 2u * xoff1, lst + 4096);
            if (g == 3) dma16_sbase(swh, 2u * woff, lst + G2_A_B);
            if constexpr (X3) {
                if (g == 4) dma16_sbase(sxl, 2u * xoff0, lst + LO);
                if (g == 5) dma16_sbase(sxl, 2u * xoff1, lst + LO + 4096);
                if constexpr (!F16) if (g == 6) dma16_sbase(swl, 2u * woff, lst + LO + G2_A_B);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 4)) {
            if (live) {
                if constexpr (F16) {
                    if constexpr (X3) acc[i][j] = mfma_f16(cur.al[i], cur.bh[j], acc[i][j]);
                    acc[i][j] = mfma_f16(cur.ah[i], cur.bh[j], acc[i][j]);
                } else {
                if constexpr (X3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.al[i], cur.bh[j], acc[i][j], 0, 0, 0);
                    // measurement aid (128): two-term product, the low halves of W dropped (W effectively rounded to bf16)
                    if constexpr (!(ABL & 128)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.ah[i], cur.bl[j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.ah[i], cur.bh[j], acc[i][j], 0, 0, 0);
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);          // the next step's fragments are in registers
    };
    // the iteration of steps kt .. kt + S - 1: fa holds step kt on entry; step kt + S ends up in fa (S even) or fb (S odd)
    auto iter = [&](int kt, Frags<X3>& fa, Frags<X3>& fb) {
        wait_vmcnt<(R - 2 * S) * P>();
        __builtin_amdgcn_s_barrier();
        if constexpr ((ABL & 16) != 0) {
#pragma unroll
            for (int t = 0; t < S; ++t) {
                int rs = slot + t + 1; if (rs >= R) rs -= R;
                int ds = slot + t + 1 - S; if (ds < 0) ds += R;
                const int kc = (ABL & 64) ? 0 : min(kt + R - S + 1 + t, nk - 1);     // measurement aid (64): every copy re-reads K-step 0 (L1 / L2 hits only)
                const bool live = (S == 1) || (kt + t < nk);                          // K-steps past the end (nk not a multiple of S) multiply nothing
                if (t & 1) sub(live, fb, fa, rs, ds, kc); else sub(live, fa, fb, rs, ds, kc);
            }
        } else {
            static_assert((ABL & 16) || S == 1, "the un-interleaved step exists for one K-step per barrier only");
            static_assert(!F16, "the fp16 form exists for the interleaved step only");
            const int nslot = slot + 1 == R ? 0 : slot + 1;
            if (!(ABL & 2)) issue(kt + R, slot);
            if (!(ABL & 1)) load_frags2<X3>(fb, smem + nslot * STAGE_B, wr, wc, fr, fc);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 4)) mfma_frags<X3>(acc, fa);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        slot += S; if (slot >= R) slot -= R;
    };
    for (int kt = 0; kt < nk; kt += 2 * S) {
        iter(kt, f0, f1);
        if (kt + S < nk) { if constexpr ((S & 1) != 0) iter(kt + S, f1, f0); else iter(kt + S, f0, f1); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // every wave is done reading, no copy is in flight
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const bool vec = linear_vec_ok_dev(a);
    if (ABL & 8) {          // measurement aid: keep the accumulators alive with one store per lane instead of the epilogue
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += acc[t >> 1][t & 1][q];
        a.Y[(int64_t)min(m0 + wr * 128 + fr, M - 1) * a.ldy + min(n0 + wc * 64 + fc, N - 1)] = sum;
    }
    float4 bv[2];
    load_patch_bias(a, n0 + wc * 64, lane, vec, bv);
    auto patches = [&](auto extra, auto f16out) {
#pragma unroll
        for (int t = 0; t < ((ABL & 8) ? 0 : 8); ++t) {
            const int i = t >> 1, j = t & 1;
#pragma unroll
            for (int q = 0; q < 16; ++q) ep[((q & 3) + 8 * (q >> 2) + 4 * fc) * EP_LD + fr] = acc[i][j][q];
            epilogue_patch<4, decltype(extra)::value, decltype(f16out)::value>(a, ep, m0 + wr * 128 + i * 32, n0 + wc * 64 + j * 32, lane, vec, nullptr, &bv[j]);
        }
    };
    typedef std::integral_constant<int, 0> out_bf16;
    typedef std::integral_constant<int, 1> out_f16;
    if (a.f16 && a.Yh) { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_f16{}); else patches(std::false_type{}, out_f16{}); }
    else { if (a.pos || a.gate || a.res) patches(std::true_type{}, out_bf16{}); else patches(std::false_type{}, out_bf16{}); }
}

// X (fp32, row map, optional SiLU) -> K-blocked bf16 planes [K/16][M][16]: high halves and, when lo != nullptr,
// bf16(x - high).  A thread converts 8 consecutive elements; threads run over the rows of one K-step first, so the
// writes are contiguous.
__global__ __launch_bounds__(256) void stage_x_kernel(const LinearArgs a, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_kb = (int64_t)a.M * 2;
    if (i >= per_kb * (a.K >> 4)) return;
    const int kb = (int)(i / per_kb), rem = (int)(i % per_kb), m = rem >> 1, c = rem & 1;
    const float* src = a.X + row_of(a.xmap, m) * a.ldx + kb * 16 + c * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if (a.f16) {
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (a.in_act == ACT_SILU) v[e] = act_fn(v[e], ACT_SILU);
            h[e] = f16_sat(v[e]);
            l[e] = f16_sat(v[e] - (float)h[e]);
        }
        *reinterpret_cast<f16x8*>(hi + i * 8) = h;
        if (lo) *reinterpret_cast<f16x8*>(lo + i * 8) = l;
        return;
    }
    bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (a.in_act == ACT_SILU) v[e] = act_fn(v[e], ACT_SILU);
        h[e] = (__bf16)v[e];
        l[e] = (__bf16)(v[e] - (float)h[e]);
    }
    *reinterpret_cast<bf16x8*>(hi + i * 8) = h;
    if (lo) *reinterpret_cast<bf16x8*>(lo + i * 8) = l;
}

// W planes [N][K] (bf16 bits) -> K-blocked [K/16][N][16].
__global__ __launch_bounds__(256) void stage_w_kernel(const unsigned short* __restrict__ w, int N, int K, unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_kb = (int64_t)N * 2;
    if (i >= per_kb * (K >> 4)) return;
    const int kb = (int)(i / per_kb), rem = (int)(i % per_kb), n = rem >> 1, c = rem & 1;
    *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(w + (int64_t)n * K + kb * 16 + c * 8);
}

// hipFuncSetAttribute once per kernel instance and process (it is not a stream operation: a launch recorded during graph capture
// must not depend on it, see linear_staged_prepare).
static hipError_t ensure_dynamic_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static const void* done[64];
    static int n_done = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_done; ++i)
        if (done[i] == fn) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && n_done < 64) done[n_done++] = fn;
    return e;
}

template <bool X3>
static hipError_t launch_glds(const LinearArgs& a, hipStream_t stream) {
    constexpr size_t smem = (X3 ? 4 : 8) * (X3 ? 4 : 2) * (size_t)GIMG_B;      // ring of K-step buffers: 128 KB
    auto kern = &linear_glds_kernel<X3, 0>;
#ifdef ADX_MEASURE          // measurement library only (python -m autodiffusion_amd.build --measure): instances whose results are wrong
    const char* abl_env = getenv("ADX_GLDS_ABLATE");        // read per call
    const int abl = abl_env ? atoi(abl_env) : 0;
    switch (abl) {          // bit 0 no LDS reads, bit 1 no copies, bit 2 no MFMAs, bit 3 no epilogue
        case 1: kern = &linear_glds_kernel<X3, 1>; break;
        case 2: kern = &linear_glds_kernel<X3, 2>; break;
        case 3: kern = &linear_glds_kernel<X3, 3>; break;
        case 4: kern = &linear_glds_kernel<X3, 4>; break;
        case 8: kern = &linear_glds_kernel<X3, 8>; break;
        case 11: kern = &linear_glds_kernel<X3, 11>; break;
        default: break;
    }
#else
    constexpr int abl = 0;
#endif
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem) != hipSuccess) return hipErrorInvalidValue;
    const size_t xe = a.Xh ? 0 : (size_t)a.M * a.K, we = (size_t)a.N * a.K;          // scratch layout: [X planes unless given][W planes unless given]
    unsigned short* base = reinterpret_cast<unsigned short*>(a.xstage);
    StagedOperands so;
    if (a.Xh) {
        so.xh = a.Xh; so.xl = a.Xl;
    } else {
        unsigned short* xh = base;
        unsigned short* xl = X3 ? base + xe : nullptr;
        so.xh = xh; so.xl = xl;
        const int64_t nx = (int64_t)a.M * (a.K >> 3);
        hipLaunchKernelGGL(stage_x_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream, a, xh, xl);
    }
    if (a.f16) {                 // fp16 forms: W rounded to fp16, pre-blocked by the caller; no low halves
        if (!a.Wblk_h) return hipErrorInvalidValue;
        so.wh = reinterpret_cast<const unsigned short*>(a.Wblk_h);
        so.wl = nullptr;
    } else if (a.Wblk) {
        so.wh = reinterpret_cast<const unsigned short*>(a.Wblk);
        so.wl = reinterpret_cast<const unsigned short*>(a.Wblk_lo);
    } else {
        unsigned short* wh = base + (X3 ? 2 : 1) * xe;
        unsigned short* wl = X3 ? wh + we : nullptr;
        const int64_t nw = (int64_t)a.N * (a.K >> 3);
        hipLaunchKernelGGL(stage_w_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(a.W), a.N, a.K, wh);
        if (X3) hipLaunchKernelGGL(stage_w_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(a.Wlo), a.N, a.K, wl);
        so.wh = wh; so.wl = wl;
    }
    const char* form_env = getenv("ADX_GLDS_FORM");        // read per call: 1 = one 8-wave workgroup per CU (256x256 tiles), 2 = two 4-wave ones (256x128)
    const int form = form_env ? atoi(form_env) : 2;
    if (form == 2) {
        constexpr size_t smem2 = (X3 ? 3 : 6) * (X3 ? 2 : 1) * (size_t)(G2_A_B + G2_B_B);      // 72 KB: two workgroups per CU
        auto kern2 = &linear_glds2_kernel<X3, 16>;
#ifdef ADX_MEASURE
        switch (abl) {      // measurement aid, as above (32: the un-interleaved step)
            case 32: kern2 = &linear_glds2_kernel<X3, 0>; break;
            case 1: kern2 = &linear_glds2_kernel<X3, 1>; break;
            case 2: kern2 = &linear_glds2_kernel<X3, 2>; break;
            case 3: kern2 = &linear_glds2_kernel<X3, 3>; break;
            case 4: kern2 = &linear_glds2_kernel<X3, 4>; break;
            case 8: kern2 = &linear_glds2_kernel<X3, 8>; break;
            case 11: kern2 = &linear_glds2_kernel<X3, 11>; break;
            case 16: kern2 = &linear_glds2_kernel<X3, 16>; break;
            case 24: kern2 = &linear_glds2_kernel<X3, 24>; break;
            case 17: kern2 = &linear_glds2_kernel<X3, 17>; break;
            case 18: kern2 = &linear_glds2_kernel<X3, 18>; break;
            case 19: kern2 = &linear_glds2_kernel<X3, 19>; break;
            case 20: kern2 = &linear_glds2_kernel<X3, 20>; break;
            case 27: kern2 = &linear_glds2_kernel<X3, 27>; break;
            case 80: kern2 = &linear_glds2_kernel<X3, 80>; break;
            case 144: kern2 = &linear_glds2_kernel<X3, 144>; break;
            default: break;
        }
#endif
        if constexpr (!X3) {            // single-pass bf16 can walk its 6-slot ring 2 or 3 K-steps per barrier.  Measured (M=20480 N=4096 K=1024,
            // product kernel alone): 1 step 226.6 us, 2 steps 232.8, 3 steps 247.0 -- the barrier is not what limits this kernel
            const char* ks_env = getenv("ADX_GLDS_KSTEPS");      // tuning knob, read per call
            const int ks = ks_env ? atoi(ks_env) : 1;
            if (abl == 0 && ks == 2) kern2 = &linear_glds2_kernel<false, 16, 2>;
            if (abl == 0 && ks == 3) kern2 = &linear_glds2_kernel<false, 16, 3>;
        }
        if (a.f16) kern2 = &linear_glds2_kernel<X3, 16, 1, true>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern2), (int)smem2) != hipSuccess) return hipErrorInvalidValue;
        const int tiles_m = (a.M + G2_TM - 1) / G2_TM, tiles_n = (a.N + G2_TN - 1) / G2_TN;
        const int64_t n_tiles = (int64_t)tiles_m * tiles_n;
        const int blocks = (int)((n_tiles + 7) / 8) * 8;
        const char* pw_env = getenv("ADX_GLDS_PANEL");       // tuning knob, read per call: panel width in tiles (0 = row-major order)
        const int pw = pw_env ? atoi(pw_env) : 8;
        hipLaunchKernelGGL(kern2, dim3(blocks), dim3(256), smem2, stream, a, so, tiles_n, (int)n_tiles, (pw < tiles_n) ? pw : 0);
        return hipGetLastError();
    }
    if (a.f16) return hipErrorInvalidValue;         // the fp16 form exists in the two-workgroup kernel only
    const int tiles_m = (a.M + GT - 1) / GT, tiles_n = (a.N + GT - 1) / GT;
    const int64_t n_tiles = (int64_t)tiles_m * tiles_n;
    const int blocks = (int)((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, stream, a, so, tiles_n, (int)n_tiles);
    return hipGetLastError();
}

// ============================================================================================================
// Row-stationary form for short K (K = 16 KS <= 256) and very many rows: a workgroup of 8 waves owns 256 rows of X for ALL
// N output columns, so X is read, converted and split exactly once (the tiled kernels above do that once per column of
// tiles) and no X bytes pass through LDS at all.
//   * Each wave turns its 32 rows of X into MFMA fragments that stay in registers for the whole kernel: KS x 4 VGPRs of
//     high halves, as many again of low halves with split operands (128 VGPRs at K = 256).
//   * W is walked in column tiles of 32: the bf16 rows of a tile (all of K: KS KB per operand image) are copied global -> LDS
//     by 16-byte LDS-DMA into one of two buffers, laid out [k16 step][k half][column] so that the fragment read of a
//     step is one contiguous KB.  A tile costs 3 KS (split) or KS dependent 32x32x16 MFMAs per wave and one 32x32 epilogue.
//   * The two waves of a SIMD (wave w and w + 4) belong to different groups that run half a tile apart: while one
//     multiplies tile j the other runs the epilogue (bias, activation, gate, residual, 16-byte stores) of the tile it
//     finished before, so the matrix pipe always has a wave issuing and the epilogue costs no matrix time.  Two
//     barriers per column tile hand the roles over; group 1 also starts the copies of the next tile.
// ============================================================================================================
constexpr int ROWS_MAX_N = 8192;                  // the bias vector sits in LDS

template <bool X3, int KS>
__device__ __forceinline__ void rows_mul_tile(f32x16& acc, const bf16x8 (&ah)[KS], const bf16x8 (&al)[X3 ? KS : 1], const unsigned char* wb, int lane) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // W fragments run two k16 steps ahead of the MFMAs (three register sets): an LDS read takes longer than the three
    // MFMAs of a step, and the scheduler left to itself reads each step right before it is used.
    constexpr int AHEAD = 4;
    bf16x8 bh[AHEAD + 1], bl[X3 ? AHEAD + 1 : 1];
    auto read = [&](int s) {
        bh[s % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(wb + (s * 64 + lane) * 16);
        if constexpr (X3) bl[s % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(wb + KS * 1024 + (s * 64 + lane) * 16);
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) read(s);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + AHEAD < KS) read(s + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        const int c = s % (AHEAD + 1);
        if constexpr (X3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl[c], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh[c], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// EXTRA: the call has a position table, gate or residual.  The launcher only takes calls whose epilogue can use 16-byte
// accesses (linear_vec_ok), so the kernel holds exactly one epilogue variant.
template <bool X3, bool SILU, int KS, bool EXTRA>
__global__ __launch_bounds__(512, 1) void linear_rows_kernel(const LinearArgs a) {
    constexpr int IMG_B = KS * 1024;                          // one operand image of a 32-column tile
    constexpr int WBUF_B = (X3 ? 2 : 1) * IMG_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // 2 * WBUF_B + 8 epilogue patches
    const int M = a.M, N = a.N, K = a.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fc = lane >> 5;
    const int grp = wave >> 2, w4 = wave & 3;
    const int m0 = (int)blockIdx.x * 256 + wave * 32;
    float* ep = reinterpret_cast<float*>(smem + 2 * WBUF_B) + wave * (32 * EP_LD);
    float* lbias = reinterpret_cast<float*>(smem + 2 * WBUF_B) + 8 * 32 * EP_LD;          // N floats, rounded up to 4
    for (int i = tid; i < ((N + 3) & ~3); i += 512) lbias[i] = (a.bias && i < N) ? a.bias[i] : 0.f;        // visible after the first barrier
    // Workgroups start at different column tiles and wrap around: they all advance at the same rate, and with a common
    // order every store in flight on the chip would land in the same 128 bytes of each 4 N-byte row (one memory channel).
    const int nt = (N + 31) / 32;
    const int jrot = (int)((blockIdx.x * 7u) % (unsigned)nt);
    auto col_tile = [&](int j) { const int c = j + jrot; return c >= nt ? c - nt : c; };

    // ---- this wave's 32 rows of X -> register fragments (lane: row fr, elements 16 s + 8 fc .. + 7 of every k16 step s)
    bf16x8 ah[KS], al[X3 ? KS : 1];
    {
        const float* xp = a.X + row_of(a.xmap, min(m0 + fr, M - 1)) * a.ldx + 8 * fc;          // clamped rows are never stored
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += 4) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(xp + 16 * (s0 + (i >> 1)) + 4 * (i & 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float e[8] = {v[2 * i].x, v[2 * i].y, v[2 * i].z, v[2 * i].w, v[2 * i + 1].x, v[2 * i + 1].y, v[2 * i + 1].z, v[2 * i + 1].w};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if constexpr (SILU) e[c] = act_fn(e[c], ACT_SILU);
                    ah[s0 + i][c] = (__bf16)e[c];
                    if constexpr (X3) al[s0 + i][c] = (__bf16)(e[c] - (float)ah[s0 + i][c]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                // at most eight loads in flight: the fragments need the registers
        }
    }

    // ---- copies of column tile j into buffer j & 1, by the four waves of group 1: wave w4 copies k16 steps 4 r + w4
    const unsigned short* Wh = reinterpret_cast<const unsigned short*>(a.W);
    const unsigned short* Wl = reinterpret_cast<const unsigned short*>(a.Wlo);
    auto copy_tile = [&](int j) {
        unsigned char* dst = smem + (j & 1) * WBUF_B + w4 * 1024;
        const int64_t src = (int64_t)min(col_tile(j) * 32 + fr, N - 1) * K + 16 * w4 + 8 * fc;
#pragma unroll
        for (int r = 0; r < KS / 4; ++r) {
            dma16_unordered(Wh + src + 64 * r, dst + r * 4096);
            if constexpr (X3) dma16_unordered(Wl + src + 64 * r, dst + IMG_B + r * 4096);
        }
    };
    constexpr bool vec = true;
    f32x16 acc;
    auto finish_tile = [&](int j) {
#pragma unroll
        for (int q = 0; q < 16; ++q) ep[((q & 3) + 8 * (q >> 2) + 4 * fc) * EP_LD + fr] = acc[q];
        epilogue_patch<EXTRA ? 1 : (X3 ? 2 : 4), EXTRA>(a, ep, m0, col_tile(j) * 32, lane, vec, lbias);
    };

    if (grp == 1) { copy_tile(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    const bool rows_full = vec && m0 + 32 <= M;
    for (int j = 0; j < nt; ++j) {
        const unsigned char* wb = smem + (j & 1) * WBUF_B;
        // first half: group 0 multiplies tile j; group 1 starts the copies of tile j + 1 (its buffer was last read for tile
        // j - 1, two barriers ago) and then finishes tile j - 1, so that its stores are YOUNGER than the copies
        bool counted = false;
        if (grp == 0) {
            rows_mul_tile<X3, KS>(acc, ah, al, wb, lane);
        } else {
            if (j + 1 < nt) copy_tile(j + 1);
            if (j > 0) {
                finish_tile(j - 1);
                counted = rows_full && (col_tile(j - 1) + 1) * 32 <= N;      // exactly four 16-byte stores per lane
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        // second half: group 1 multiplies tile j, group 0 finishes tile j
        if (grp == 1) {
            rows_mul_tile<X3, KS>(acc, ah, al, wb, lane);
            // The copies must have landed before anyone reads them; the stores issued after them need not have (vector
            // memory operations complete in order, so "at most four outstanding" leaves only those stores in flight).
            if (counted) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            finish_tile(j);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 1) finish_tile(nt - 1);
}

template <bool X3, bool SILU, int KS, bool EXTRA>
static hipError_t launch_rows4(const LinearArgs& a, hipStream_t stream) {
    constexpr size_t smem_fixed = 2 * (X3 ? 2 : 1) * (size_t)KS * 1024 + 8 * 32 * EP_LD * sizeof(float);
    constexpr size_t smem_max = smem_fixed + ROWS_MAX_N * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_rows_kernel<X3, SILU, KS, EXTRA>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max);
    if (attr != hipSuccess) return attr;
    const size_t smem = smem_fixed + (size_t)((a.N + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL((linear_rows_kernel<X3, SILU, KS, EXTRA>), dim3((unsigned)((a.M + 255) / 256)), dim3(512), smem, stream, a);
    return hipGetLastError();
}

template <bool X3, bool SILU, int KS>
static hipError_t launch_rows3(const LinearArgs& a, hipStream_t stream) {
    return (a.pos || a.gate || a.res) ? launch_rows4<X3, SILU, KS, true>(a, stream) : launch_rows4<X3, SILU, KS, false>(a, stream);
}

template <bool X3>
static hipError_t launch_rows(const LinearArgs& a, hipStream_t stream) {
    const bool silu = a.in_act == ACT_SILU;
    switch (a.K) {
        case 256: return silu ? launch_rows3<X3, true, 16>(a, stream) : launch_rows3<X3, false, 16>(a, stream);
        case 128: return silu ? launch_rows3<X3, true, 8>(a, stream) : launch_rows3<X3, false, 8>(a, stream);
        case 64: return silu ? launch_rows3<X3, true, 4>(a, stream) : launch_rows3<X3, false, 4>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

// ============================================================================================================
// Two-layer MLP of width 256 in one kernel: Y = epilogue2(act1(X W1^T + b1) W2^T + b2), X [M][256], W1 [N1][256], W2 [256][N1].
// The hidden layer (4 floats per element and token when it goes through memory as operand planes, written once and read
// once) never leaves the CU.  A workgroup of four waves -- one per SIMD, up to 512 registers each -- owns 128 rows:
//   * a wave keeps its 32 rows of X as MFMA fragments (128 VGPRs with split operands) and the 32 x 256 result as eight
//     accumulators (128 more);
//   * the hidden layer is walked 64 units at a time.  Phase 1: H^T = W1_tile X^T, two 32-unit tiles with interleaved
//     accumulation chains (A = W1 rows, B = the X fragments: the accumulator then holds "token on the lane, hidden unit on
//     the registers", which is the B operand of the next product as it stands).  The rows of a W1 tile are fetched in the
//     order that makes registers 8u .. 8u+7 of a lane half the k elements 8 half .. 8 half + 7 of k16 step u.
//     Phase 2: Y^T += W2_chunk H^T over the eight output tiles, independent chains.
//   * W1 tiles and W2 chunks come from the K-blocked weight copies by LDS-DMA into one 64 KB buffer each; the copies of a
//     phase's operands are issued during the phase before it, two barriers per 64 hidden units.
//   * epilogue: the accumulators go through a wave-private LDS patch (in the W1 buffer, free by then) to epilogue_patch, so
//     bias, activation, gate (with row groups), residual and plane outputs are those of every other product.
// ============================================================================================================
constexpr int MLP_D = 256;

__device__ __forceinline__ int mlp_row_perm(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }     // swaps bits 2 and 3

template <bool X3, int ACT1, int ABL = 0>
__global__ __launch_bounds__(256, 1) void mlp256_kernel(const LinearArgs a1, const LinearArgs a2) {
    constexpr int KS = MLP_D / 16;                            // k16 steps of the first product
    constexpr int PL = X3 ? 2 : 1;
    constexpr int W1_TILE_B = KS * 1024;                      // one plane of a 32-unit tile: [k16 step][k half][unit] 16-byte pieces
    constexpr int W1_BUF_B = 2 * PL * W1_TILE_B;              // two tiles
    constexpr int W2_PLANE_B = 8 * 4 * 1024;                  // one plane of a 64-unit chunk: [out tile][k16 step][k half][row]
    constexpr int W2_BUF_B = PL * W2_PLANE_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* w1buf = smem;
    unsigned char* w2buf = smem + W1_BUF_B;
    const int M = a1.M, N1 = a1.N;
    constexpr int EP_B = 4 * 4 * 32 * EP_LD * (int)sizeof(float);       // the epilogue's patches reuse the weight buffers
    constexpr int BIAS_OFF = W1_BUF_B + W2_BUF_B > EP_B ? W1_BUF_B + W2_BUF_B : EP_B;
    float* lb1 = reinterpret_cast<float*>(smem + BIAS_OFF);
    float* lb2 = lb1 + N1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fc = lane >> 5;
    const int m0 = (int)blockIdx.x * 128 + wave * 32;
    for (int i = tid; i < N1; i += 256) lb1[i] = a1.bias ? a1.bias[i] : 0.f;
    for (int i = tid; i < MLP_D; i += 256) lb2[i] = a2.bias ? a2.bias[i] : 0.f;

    const unsigned short* W1h = reinterpret_cast<const unsigned short*>(a1.Wblk);
    const unsigned short* W1l = reinterpret_cast<const unsigned short*>(a1.Wblk_lo);
    const unsigned short* W2h = reinterpret_cast<const unsigned short*>(a2.Wblk);
    const unsigned short* W2l = reinterpret_cast<const unsigned short*>(a2.Wblk_lo);
    // copy c (0 .. 8 PL - 1 per wave) of pair p's W1 tiles: tile jj, plane pl, k16 step 4 r + wave
    const int prow = mlp_row_perm(fr);
    auto copy_w1 = [&](int p, int c) {
        const int jj = c / (4 * PL), pl = (c / 4) % PL, sidx = 4 * (c & 3) + wave;
        const int64_t src = ((int64_t)sidx * N1 + 64 * p + 32 * jj + prow) * 16 + 8 * fc;
        dma16_unordered((pl ? W1l : W1h) + src, w1buf + (jj * PL + pl) * W1_TILE_B + sidx * 1024);
    };
    // copy c (0 .. 8 PL - 1 per wave) of pair p's W2 chunk: plane pl, out tile t = 2 (c & 3) + (wave >> 1), k16 steps 2 (wave & 1) + {0, 1}
    auto copy_w2 = [&](int p, int c) {
        const int pl = c / 8, t = (c >> 1) & 3, q = 2 * (wave & 1) + (c & 1), tt = 2 * t + (wave >> 1);
        const int64_t src = ((int64_t)(4 * p + q) * MLP_D + 32 * tt + fr) * 16 + 8 * fc;
        dma16_unordered((pl ? W2l : W2h) + src, w2buf + pl * W2_PLANE_B + (tt * 4 + q) * 1024);
    };
#pragma unroll
    for (int c = 0; c < 8 * PL; ++c) copy_w1(0, c);

    // ---- this wave's 32 rows of X -> register fragments (lane: row fr, elements 16 s + 8 fc .. + 7 of every k16 step s)
    bf16x8 xh[KS], xl[X3 ? KS : 1];
    {
        const float* xp = a1.X + row_of(a1.xmap, min(m0 + fr, M - 1)) * a1.ldx + 8 * fc;
        // all 32 loads of the row are in flight together (the accumulators' registers are still free): one wave per SIMD
        // means nobody else covers a round trip to memory
        float4 v[2 * KS];
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) v[i] = (ABL & 16) ? make_float4(0.5f, 0.25f, 1.f, 2.f) : *reinterpret_cast<const float4*>(xp + 16 * (i >> 1) + 4 * (i & 1));
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const float e[8] = {v[2 * i].x, v[2 * i].y, v[2 * i].z, v[2 * i].w, v[2 * i + 1].x, v[2 * i + 1].y, v[2 * i + 1].z, v[2 * i + 1].w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                xh[i][c] = (__bf16)e[c];
                if constexpr (X3) xl[i][c] = (__bf16)(e[c] - (float)xh[i][c]);
            }
        }
    }
    f32x16 y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) y[t][q] = 0.f;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int n_pairs = N1 / 64;
    for (int p = 0; p < n_pairs; ++p) {
        // ---- phase 1: two tiles of H^T, one after the other; the W2 chunk of this pair is copied meanwhile.  Bias, activation
        // and split of tile 0 run in slices between the products of tile 1 (and those of tile 1 between the first products of
        // phase 2, which only read tile 0's half of H): with one wave per SIMD a separate activation section would idle the
        // matrix pipe.  Register 8 u + 4 e + m of tile jj holds hidden unit 64 p + 32 jj + 16 u + 8 fc + 4 e + m.
        f32x16 h[2];
        bf16x8 hh[4], hl[X3 ? 4 : 1];
        auto act_slice = [&](int jj, int r) {
            const int u = r >> 3, i = r & 7;
            const float v = act_fn(h[jj][r] + lb1[64 * p + 32 * jj + 16 * u + 8 * fc + i], ACT1);
            const __bf16 hi = (__bf16)v;
            hh[2 * jj + u][i] = hi;
            if constexpr (X3) hl[2 * jj + u][i] = (__bf16)(v - (float)hi);
        };
        {
            constexpr int AHEAD = 3;
            bf16x8 wh[AHEAD + 1], wl[X3 ? AHEAD + 1 : 1];
            auto read = [&](int k) {            // k = 16 jj + step
                const int jj = k >> 4, sidx = k & 15;
                wh[k % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(w1buf + (jj * PL) * W1_TILE_B + (sidx * 64 + lane) * 16);
                if constexpr (X3) wl[k % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(w1buf + (jj * PL + 1) * W1_TILE_B + (sidx * 64 + lane) * 16);
            };
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) read(k);
            // every copy of the phase goes out at its start: a 64 KB burst per CU takes about a third of the phase to land when
            // all CUs pull from L2 at once, and a copy issued late would be waited for at the barrier
            if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int c = 0; c < 8 * PL; ++c) copy_w2(p, c);
            }
#pragma unroll
            for (int k = 0; k < 2 * KS; ++k) {
                const int jj = k >> 4, sidx = k & 15;
                if (k + AHEAD < 2 * KS) read(k + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                const int c = k % (AHEAD + 1);
                if (sidx == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) h[jj][q] = 0.f;
                }
                if constexpr (ABL & 4) {
                    h[jj][sidx] += (float)wh[c][0] + (X3 ? (float)wl[c][2] : 0.f);
                } else {
                    if constexpr (X3) {
                        h[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[c], xh[sidx], h[jj], 0, 0, 0);
                        h[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[c], xl[sidx], h[jj], 0, 0, 0);
                    }
                    h[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[c], xh[sidx], h[jj], 0, 0, 0);
                }
                if (jj == 1) act_slice(0, sidx);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: Y^T += W2_chunk H^T, four output tiles at a time; the W1 tiles of the next pair are copied meanwhile
        const bool more = p + 1 < n_pairs;
        {
            bf16x8 vh[2][4], vl[X3 ? 2 : 1][4];
            auto read2 = [&](int g) {             // group g: k16 step g >> 1 of the chunk, output tiles 4 (g & 1) .. + 3
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int off = (((4 * (g & 1) + t) * 4 + (g >> 1)) * 64 + lane) * 16;
                    vh[g & 1][t] = *reinterpret_cast<const bf16x8*>(w2buf + off);
                    if constexpr (X3) vl[g & 1][t] = *reinterpret_cast<const bf16x8*>(w2buf + W2_PLANE_B + off);
                }
            };
            read2(0);
            if (more && !(ABL & 1)) {
#pragma unroll
                for (int c = 0; c < 8 * PL; ++c) copy_w1(p + 1, c);
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) read2(g + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int q = g >> 1, t0 = 4 * (g & 1), b = g & 1;
                if constexpr (ABL & 2) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t0 + t][q] += (float)vh[b][t][q] + (X3 ? (float)vl[b][t][q] : 0.f) + (float)hh[q][t];
                } else {
                    if constexpr (X3) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) y[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[b][t], hh[q], y[t0 + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) y[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[b][t], hl[q], y[t0 + t], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[b][t], hh[q], y[t0 + t], 0, 0, 0);
                }
                if (g < 4) {                      // tile 1's half of H is first read by group 4
#pragma unroll
                    for (int r = 4 * g; r < 4 * g + 4; ++r) act_slice(1, r);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // ---- epilogue: lane = token fr, register q of tile t = output column 32 t + 8 (q >> 2) + 4 fc + (q & 3).  Four tiles at a
    // time go through wave-private LDS patches (in the weight buffers, free by now) into "a lane owns four columns of rows
    // (lane >> 3) + 8 rr"; ALL gate and residual loads of the four tiles are issued before the first store: with one wave per
    // SIMD every round trip to memory is exposed, and a patch-by-patch epilogue made 32 of them per wave.
    float* ep = reinterpret_cast<float*>(smem) + wave * (4 * 32 * EP_LD);
    const int c4 = (lane & 7) * 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& yt = y[4 * half + t];
                *reinterpret_cast<float4*>(ep + t * (32 * EP_LD) + fr * EP_LD + 8 * g + 4 * fc) = make_float4(yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]);
            }
        float4 gv[4][4], rv[4][4];
        int mrow[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = m0 + rr * 8 + (lane >> 3);
            mrow[rr] = m < M ? m : -1;
            const int64_t mc = m < M ? m : M - 1;
            const int64_t grow = a2.gate_group ? (mc / a2.gate_group) * a2.gate_period + mc % a2.gate_period : mc;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = 32 * (4 * half + t) + c4;
                gv[t][rr] = (a2.gate && !(ABL & 8)) ? *reinterpret_cast<const float4*>(a2.gate + grow * a2.ldg + n) : make_float4(1.f, 1.f, 1.f, 1.f);
                rv[t][rr] = (a2.res && !(ABL & 8)) ? *reinterpret_cast<const float4*>(a2.res + mc * a2.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 32 * (4 * half + t) + c4;
            const float4 bv = *reinterpret_cast<const float4*>(lb2 + n);
            float v[4][4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4 raw = *reinterpret_cast<const float4*>(ep + t * (32 * EP_LD) + (rr * 8 + (lane >> 3)) * EP_LD + c4);
                v[rr][0] = raw.x + bv.x; v[rr][1] = raw.y + bv.y; v[rr][2] = raw.z + bv.z; v[rr][3] = raw.w + bv.w;
            }
            act_patch<4>(v, a2.act);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (mrow[rr] < 0) continue;
                if ((ABL & 8) && v[rr][0] != 12345.f) continue;
                const float4 g = gv[t][rr], r = rv[t][rr];
                *reinterpret_cast<float4*>(a2.Y + (int64_t)mrow[rr] * a2.ldy + n) =
                    make_float4(fmaf(v[rr][0], g.x, r.x), fmaf(v[rr][1], g.y, r.y), fmaf(v[rr][2], g.z, r.z), fmaf(v[rr][3], g.w, r.w));
            }
        }
    }
}

template <bool X3>
static hipError_t launch_mlp256(const LinearArgs& a1, const LinearArgs& a2, hipStream_t stream) {
    const size_t wbytes = (size_t)(X3 ? 2 : 1) * (2 * 16 * 1024 + 8 * 4 * 1024), epbytes = 4 * 4 * 32 * EP_LD * sizeof(float);
    const size_t smem = (wbytes > epbytes ? wbytes : epbytes) + (size_t)(a1.N + MLP_D) * sizeof(float);
    const dim3 grid((unsigned)((a1.M + 127) / 128)), block(256);
#define ADX_MLP_LAUNCH(ACT)                                                                                                        \
    {                                                                                                                              \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp256_kernel<X3, ACT>),                 \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
        if (attr != hipSuccess) return attr;                                                                                       \
        hipLaunchKernelGGL((mlp256_kernel<X3, ACT>), grid, block, smem, stream, a1, a2);                                           \
        return hipGetLastError();                                                                                                  \
    }
#ifdef ADX_MEASURE
    if (X3 && a1.act == ACT_RELU) {                 // measurement hooks (results are wrong): 1 no weight copies after the first, 2 / 4 no products in phase 2 / 1, 8 no epilogue loads and stores, 16 no X loads
        const char* e = getenv("ADX_MLP_ABLATE");
        const int abl = e ? atoi(e) : 0;
#define ADX_MLP_ABL(A)                                                                                                             \
        if (abl == A) {                                                                                                            \
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp256_kernel<X3, ACT_RELU, A>),     \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
            if (attr != hipSuccess) return attr;                                                                                   \
            hipLaunchKernelGGL((mlp256_kernel<X3, ACT_RELU, A>), grid, block, smem, stream, a1, a2);                               \
            return hipGetLastError();                                                                                              \
        }
        ADX_MLP_ABL(1) ADX_MLP_ABL(2) ADX_MLP_ABL(4) ADX_MLP_ABL(6) ADX_MLP_ABL(7) ADX_MLP_ABL(15) ADX_MLP_ABL(23) ADX_MLP_ABL(31) ADX_MLP_ABL(24)
#undef ADX_MLP_ABL
    }
#endif
    switch (a1.act) {
        case ACT_RELU: ADX_MLP_LAUNCH(ACT_RELU)
        case ACT_GELU_TANH: ADX_MLP_LAUNCH(ACT_GELU_TANH)
        case ACT_NONE: ADX_MLP_LAUNCH(ACT_NONE)
        default: return hipErrorInvalidValue;
    }
#undef ADX_MLP_LAUNCH
}

// The epilogue of this call can use 16-byte accesses for every operand (the same test the kernels make for themselves).
static bool linear_vec_ok(const LinearArgs& a) {
    return (a.col_group == 0 || ((a.col_group | a.col_stride | a.col_off) & 3) == 0) && (a.N & 3) == 0 && (a.ldy & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 && (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
           (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
           (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
           (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
}

}  // namespace

// Shapes the large-tile kernel takes: bf16 operand modes, K a multiple of 64, 16-byte aligned rows, at least two row tiles.
bool linear_big_ok(const LinearArgs& a) {
    return (a.prec == PREC_BF16 || a.prec == PREC_BF16X3) && a.M >= 256 && a.N >= 64 && (a.K % (2 * TK)) == 0 && (a.in_act == ACT_NONE || a.in_act == ACT_SILU) && (a.ldx % 4) == 0 &&
           (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 &&
           (a.prec != PREC_BF16X3 || (a.Wlo && (reinterpret_cast<uintptr_t>(a.Wlo) & 15) == 0));
}

template <bool X3, int WM, bool SILU>
static hipError_t launch_big2(const LinearArgs& a, hipStream_t stream) {
    constexpr int TM = 64 * WM;
    constexpr size_t smem = 2 * (X3 ? 2 : 1) * (size_t)(TM * ROW_B + BTILE_B);
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_big_kernel<X3, WM, SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (attr != hipSuccess) return attr;
    const int tiles_m = (a.M + TM - 1) / TM, tiles_n = (a.N + TN - 1) / TN;
    const int64_t n_tiles = (int64_t)tiles_m * tiles_n;
    if (n_tiles > (1 << 30)) return hipErrorInvalidValue;
    const int blocks = (int)((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL((linear_big_kernel<X3, WM, SILU>), dim3(blocks), dim3(128 * WM), smem, stream, a, tiles_n, (int)n_tiles);
    return hipGetLastError();
}

template <bool X3, int WM>
static hipError_t launch_big(const LinearArgs& a, hipStream_t stream) {
    return a.in_act == ACT_SILU ? launch_big2<X3, WM, true>(a, stream) : launch_big2<X3, WM, false>(a, stream);
}

// Everything the staged product needs that is not a stream operation (the dynamic-LDS attribute of its kernels), so that a later
// call can run while its stream is being captured into a graph.
hipError_t linear_staged_prepare() {
    constexpr int smem_x3 = 3 * 2 * (G2_A_B + G2_B_B), smem_1 = 6 * (G2_A_B + G2_B_B);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&linear_glds2_kernel<true, 16>), smem_x3);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void*>(&linear_glds2_kernel<false, 16>), smem_1);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void*>(&linear_glds2_kernel<true, 16, 1, true>), smem_x3);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void*>(&linear_glds2_kernel<false, 16, 1, true>), smem_1);
    return e;
}

size_t linear_stage_bytes(int M, int N, int K, int prec) { return ((size_t)M + N) * K * 2 * (prec == PREC_BF16X3 ? 2 : 1); }

bool linear_planes_ok(int M, int N, int K, int prec) {
    if (prec != PREC_BF16 && prec != PREC_BF16X3) return false;
    const int64_t tiles = (int64_t)((M + G2_TM - 1) / G2_TM) * ((N + G2_TN - 1) / G2_TN);
    const char* mk = getenv("ADX_PLANES_MIN_K");                  // measurement switch, read per call (default 512)
    const int min_k = mk ? atoi(mk) : 512;
    return M >= 256 && N >= 64 && K >= min_k && (K % 64) == 0 && (N & 3) == 0 && tiles >= 320 && (int64_t)M * K < ((int64_t)1 << 31) &&
           (int64_t)N * K < ((int64_t)1 << 31);
}

hipError_t stage_planes(const float* x, int64_t ldx, int M, int K, int in_act, unsigned short* hi, unsigned short* lo, hipStream_t stream, bool f16) {
    if (!x || !hi || M <= 0 || K <= 0 || (K % 16) != 0 || (ldx & 3) != 0) return hipErrorInvalidValue;
    LinearArgs a;
    a.X = x; a.ldx = ldx; a.M = M; a.K = K; a.in_act = in_act; a.f16 = f16;
    const int64_t nx = (int64_t)M * (K >> 3);
    hipLaunchKernelGGL(stage_x_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream, a, hi, lo);
    return hipGetLastError();
}

hipError_t linear_block_weight(const void* w, int N, int K, void* out, hipStream_t stream) {
    if (!w || !out || N <= 0 || K <= 0 || (K % 32) != 0) return hipErrorInvalidValue;
    const int64_t nw = (int64_t)N * (K >> 3);
    hipLaunchKernelGGL(stage_w_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(w), N, K,
                       reinterpret_cast<unsigned short*>(out));
    return hipGetLastError();
}

// The pair l1 (X -> hidden, bias + activation only) and l2 (hidden -> 256 columns, any epilogue) as one kernel.
bool mlp256_ok(const LinearArgs& l1, const LinearArgs& l2) {
    const bool bf = l1.prec == PREC_BF16 || l1.prec == PREC_BF16X3;
    return bf && l2.prec == l1.prec && l1.M == l2.M && l1.M >= 128 && l1.K == MLP_D && l2.N == MLP_D && l2.K == l1.N && (l1.N % 64) == 0 &&
           l1.N <= 4096 && l1.X && !l1.Xh && (l1.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(l1.X) & 15) == 0 && l1.in_act == ACT_NONE &&
           l1.Wblk && l2.Wblk && (l1.prec != PREC_BF16X3 || (l1.Wblk_lo && l2.Wblk_lo)) && !l1.gate && !l1.res && !l1.pos &&
           l1.col_group == 0 && (l1.act == ACT_RELU || l1.act == ACT_GELU_TANH || l1.act == ACT_NONE) && l2.in_act == ACT_NONE &&
           l2.col_group == 0 && l2.ymap.per_batch == 0 && linear_vec_ok(l2) && l2.Y && !l2.Yh && !l2.pos;
}

hipError_t mlp256(const LinearArgs& l1, const LinearArgs& l2, hipStream_t stream) {
    if (!mlp256_ok(l1, l2)) return hipErrorInvalidValue;
    return l1.prec == PREC_BF16X3 ? launch_mlp256<true>(l1, l2, stream) : launch_mlp256<false>(l1, l2, stream);
}

hipError_t linear_big(const LinearArgs& a, hipStream_t stream) {
    if (!linear_big_ok(a)) return hipErrorInvalidValue;
    if (a.Yh && (!linear_vec_ok(a) || a.col_group != 0 || a.ymap.per_batch != 0 || (a.N & 15) != 0)) return hipErrorInvalidValue;
    static const int force_wm = [] { const char* e = getenv("ADX_GEMM_WM"); return e ? atoi(e) : 0; }();     // tuning knob
    // 256-row tiles halve the X traffic per flop; they need enough tiles to cover the chip (256 CUs, one workgroup each)
    const int64_t tiles256 = (int64_t)((a.M + 255) / 256) * ((a.N + TN - 1) / TN);
    const bool tall = force_wm ? force_wm >= 4 : tiles256 >= 512;
    // With a staging buffer from the caller both operands are re-laid as K-blocked bf16 planes once and the tiles are filled by
    // LDS-DMA (256x256 tiles).  Measured against the register-staged kernels with the two alternating in one process
    // (M=20480 N=4096 K=1024): 0.70x their time for single-pass bf16, 0.96x for split operands; M=163840 N=2048 K=256 split:
    // 0.91x; equal at N <= 1536 with K=256; slower with few tiles (M=5120: 1.27x) or a single column of 256-wide tiles.
    // Short K and enough rows for 2.5 or more rounds of 256-row workgroups: X stays in registers, W streams through LDS.
    // Measured against the tiled kernels with the two alternating on one box (K=256, split operands): M=327680 N=1024 632
    // against 820 us, N=1536 915 against 1154, N=768 501 against 593, N=512 387 against 423; M=163840 N=1024 376 against 429,
    // N=1536 523 against 591; single-pass bf16 M=327680 N=1024 404 against 755.  Slower at N=256 (241 against 228).
    const char* rows_env = getenv("ADX_GEMM_ROWS");             // read per call (tests switch it): 0 never, 1 whenever legal
    const int rows_mode = rows_env ? atoi(rows_env) : -1;
    if (force_wm == 0 && rows_mode != 0 && !a.force_staged && !a.Xh && (a.K == 256 || a.K == 128 || a.K == 64) && a.N <= ROWS_MAX_N && linear_vec_ok(a) &&
        ((a.M >= 160 * 1024 && a.N >= 512) || rows_mode == 1))
        return a.prec == PREC_BF16X3 ? launch_rows<true>(a, stream) : launch_rows<false>(a, stream);
    static const bool no_glds = [] { const char* e = getenv("ADX_GEMM_NO_DMA"); return e && e[0] == '1'; }();
    const int64_t tiles_sq = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const char* any_env = getenv("ADX_GEMM_DMA_ANY");             // read per call (tests switch it): the staged kernel for any shape
    const bool glds_any = any_env && any_env[0] == '1';
    const size_t stage_need = linear_stage_bytes(a.Xh ? 0 : a.M, a.Wblk ? 0 : a.N, a.K, a.prec);
    const bool stage_fits = stage_need == 0 || (a.xstage && a.xstage_bytes >= stage_need && (reinterpret_cast<uintptr_t>(a.xstage) & 15) == 0);
    if (a.Xh) {                 // operand planes from the producing kernel: there may be no fp32 X at all, so only the staged kernel will do
        if (!stage_fits || !linear_planes_ok(a.M, a.N, a.K, a.prec) || (a.prec == PREC_BF16X3 && !a.Xl && !a.f16) || a.xmap.per_batch != 0) return hipErrorInvalidValue;
        if (a.f16) {            // fp16 planes: low halves given = the two-term form, otherwise one term
            if (!a.Wblk_h) return hipErrorInvalidValue;
            return a.Xl ? launch_glds<true>(a, stream) : launch_glds<false>(a, stream);
        }
        return a.prec == PREC_BF16X3 ? launch_glds<true>(a, stream) : launch_glds<false>(a, stream);
    }
    // (a.f16 with an fp32 X: the kernels below multiply split-bf16 operands as always and a.f16 only selects the format of the output
    // planes; the staged form re-lays X as fp16 halves and needs Wblk_h like the planes form)
    const bool f16_ok = !a.f16 || (a.prec == PREC_BF16X3 && a.Wblk_h);
    // force_staged (the allocation-free entry point) overrides the tuning switches: that entry point promises this kernel
    if (f16_ok && (a.force_staged || (!no_glds && force_wm == 0 && ((tiles_sq >= 900 && ((a.N >= 1024 && a.K >= 512) || (a.N >= 2048 && a.K >= 256 && a.prec == PREC_BF16X3))) || glds_any))) && stage_fits &&
        (a.xstage || stage_need == 0) &&
        (int64_t)a.M * a.K < ((int64_t)1 << 31) && (int64_t)a.N * a.K < ((int64_t)1 << 31))
        return a.prec == PREC_BF16X3 ? launch_glds<true>(a, stream) : launch_glds<false>(a, stream);
    // measured: the wave-specialised form wins for single-pass bf16 (its producers keep up with 16 MFMAs per K-step per
    // consumer), the symmetric 256-row form for split operands (L2 -> CU staging bandwidth is the limit there either way)
    if (tall && (force_wm == 8 || (force_wm == 0 && a.prec == PREC_BF16)))
        return a.prec == PREC_BF16X3 ? launch_ws<true>(a, stream) : launch_ws<false>(a, stream);
    if (a.prec == PREC_BF16X3) return tall ? launch_big<true, 4>(a, stream) : launch_big<true, 2>(a, stream);
    return tall ? launch_big<false, 4>(a, stream) : launch_big<false, 2>(a, stream);
}

}  // namespace adx

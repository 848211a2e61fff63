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
#include "fastmath.h"

#include <math.h>
#include <stdlib.h>

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

// Y rows m_base .. m_base+31, columns n_base .. n_base+31 from a 32x32 patch of raw accumulators in LDS.
__device__ __forceinline__ void epilogue_patch(const LinearArgs& a, const float* ep, int m_base, int n_base, int lane, bool vec) {
    const int c4 = (lane & 7) * 4, n = n_base + c4;
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
        const int row = rr * 8 + (lane >> 3), m = m_base + row;
        if (m >= a.M || n >= a.N) continue;
        const float4 raw = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
        float v[4] = {raw.x, raw.y, raw.z, raw.w};
        const int64_t yrow = row_of(a.ymap, m);
        const float* posrow = a.pos ? a.pos + (int64_t)((m % a.pos_period) + a.pos_off) * a.ldp : nullptr;
        if (vec) {
            if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
            if (posrow) { const float4 p = *reinterpret_cast<const float4*>(posrow + n); v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn(v[e], a.act);
            if (a.gate) { const float4 g = *reinterpret_cast<const float4*>(a.gate + yrow * a.ldg + n); v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w; }
            if (a.res) { const float4 r = *reinterpret_cast<const float4*>(a.res + yrow * a.ldr + n); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
            *reinterpret_cast<float4*>(a.Y + yrow * a.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {
                const int ne = n + e;
                if (ne >= a.N) break;
                float x = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3];
                if (a.bias) x += a.bias[ne];
                if (posrow) x += posrow[ne];
                x = act_fn(x, a.act);
                if (a.gate) x *= a.gate[yrow * a.ldg + ne];
                if (a.res) x += a.res[yrow * a.ldr + ne];
                const int nc = a.col_group ? (ne / a.col_group) * a.col_stride + (ne % a.col_group) + a.col_off : ne;
                a.Y[yrow * a.ldy + nc] = x;
            }
        }
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
    const bool vec = a.col_group == 0 && (N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 &&
                     (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                     (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
                     (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
                     (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = t >> 1, j = t & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * fc) * EP_LD + fr] = acc[i][j][r];
        epilogue_patch(a, ep, m0 + wr * 64 + i * 32, n0 + wc * 64 + j * 32, lane, vec);
    }
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
    const bool vec = a.col_group == 0 && (N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 &&
                     (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                     (!a.pos || ((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0)) &&
                     (!a.gate || ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0)) &&
                     (!a.res || ((a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int i = t >> 1, j = t & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * fc) * EP_LD + fr] = acc[i][j][r];
        epilogue_patch(a, ep, m0 + wr * 128 + i * 32, n0 + wc * 64 + j * 32, lane, vec);
    }
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

hipError_t linear_big(const LinearArgs& a, hipStream_t stream) {
    if (!linear_big_ok(a)) return hipErrorInvalidValue;
    static const int force_wm = [] { const char* e = getenv("ADX_GEMM_WM"); return e ? atoi(e) : 0; }();     // tuning knob
    // 256-row tiles halve the X traffic per flop; they need enough tiles to cover the chip (256 CUs, one workgroup each)
    const int64_t tiles256 = (int64_t)((a.M + 255) / 256) * ((a.N + TN - 1) / TN);
    const bool tall = force_wm ? force_wm >= 4 : tiles256 >= 512;
    // measured: the wave-specialised form wins for single-pass bf16 (its producers keep up with 16 MFMAs per K-step per
    // consumer), the symmetric 256-row form for split operands (L2 -> CU staging bandwidth is the limit there either way)
    if (tall && (force_wm == 8 || (force_wm == 0 && a.prec == PREC_BF16)))
        return a.prec == PREC_BF16X3 ? launch_ws<true>(a, stream) : launch_ws<false>(a, stream);
    if (a.prec == PREC_BF16X3) return tall ? launch_big<true, 4>(a, stream) : launch_big<true, 2>(a, stream);
    return tall ? launch_big<false, 4>(a, stream) : launch_big<false, 2>(a, stream);
}

}  // namespace adx

// The sparse mixture-of-experts decoder layer of the conditioning producer (phase 1 of generate(): STR.embedding_to_hidden,
// transformer4planning/models/backbone/str_base.py:379-387, on the MixtralModel that models/backbone/mixtral.py:17-19 builds; the
// decoder itself is third-party `transformers`, pinned 4.42.3 in the reference's requirements.txt:212): RMSNorm, the rotate-half
// rotary embedding, and the expert block  y = res + sum_j w_j E_{e_j}(x),  E_e(x) = W2_e (silu(W1_e x) * W3_e x),  with
// (e_j, w_j) the top_k router probabilities renormalised to sum one.
//
// The expert block runs without a host round trip and without atomics:
//   1. moe_route_kernel     one wavefront per token: E router logits (fp32 dot products), softmax, top_k, renormalisation;
//   2. moe_rank_kernel      one workgroup per expert: the rank of each of its (token, slot) entries, in token order, and its count;
//   3. moe_layout_kernel    entries -> rows of an expert-sorted, 64-row-padded table (src_row, pos) and, per 64-row tile, its expert
//                           and number of live rows;
//   4. moe_gemm_kernel      grouped products over that table: a workgroup owns one 64-row tile (one expert) and 64 columns;
//                           the first product gathers its X rows through src_row and applies silu(g) * u in its epilogue (the
//                           gate and up tiles are accumulated side by side), the second reads the packed hidden rows;
//                           the grid is sized for the worst case (ceil(T k / 64) + E tiles), tiles beyond the table exit at once;
//   5. moe_combine_kernel   y[token] = res[token] + sum_slot w[token][slot] * Y2[pos[token][slot]]  (slot order: deterministic).
// A token's result depends on that token alone, so outputs do not depend on how a batch is split.
#include "adx_ops.h"
#include "engine.h"

namespace adx {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------------------- RMSNorm
// y = w * (x * rsqrt(mean(x^2) + eps)); one wavefront per row, the row read twice (the second read hits the cache).
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps, int64_t rows, int d,
                                                      float* __restrict__ y) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * d;
    float* yr = y + row * d;
    float ss = 0.f;
    if ((d & 3) == 0) {
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + i);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const float r = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + i), g = *reinterpret_cast<const float4*>(w + i);
            *reinterpret_cast<float4*>(yr + i) = make_float4(g.x * (v.x * r), g.y * (v.y * r), g.z * (v.z * r), g.w * (v.w * r));
        }
    } else {
        for (int i = lane; i < d; i += 64) ss += xr[i] * xr[i];
        const float r = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
        for (int i = lane; i < d; i += 64) yr[i] = w[i] * (xr[i] * r);
    }
}

// ---------------------------------------------------------------------------------------------------------------- rotary
// In place on x[rows][H * dh], row = b * S + s, position pos0 + s: (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin) for the halves
// x1 = x[.. j], x2 = x[.. j + dh/2]; cos / sin tables [n_pos][dh/2].
__global__ __launch_bounds__(256) void rotary_kernel(float* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn, int64_t rows,
                                                     int S, int H, int dh, int pos0) {
    const int half = dh >> 1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * H * half) return;
    const int j = (int)(i % half);
    const int h = (int)((i / half) % H);
    const int64_t row = i / ((int64_t)half * H);
    const int p = pos0 + (int)(row % S);
    float* v = x + row * ((int64_t)H * dh) + h * dh + j;
    const float c = cs[(int64_t)p * half + j], s = sn[(int64_t)p * half + j];
    const float a = v[0], b = v[half];
    v[0] = a * c - b * s;
    v[half] = b * c + a * s;
}

// ---------------------------------------------------------------------------------------------------------------- routing
__global__ __launch_bounds__(256) void moe_route_kernel(const float* __restrict__ x, const float* __restrict__ wr, int64_t T, int d, int E, int top_k,
                                                        int* __restrict__ idx, float* __restrict__ wts) {
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= T) return;
    const float* xr = x + tok * d;
    float mine = -INFINITY;                                    // lane e keeps logit e
    for (int e = 0; e < E; ++e) {
        const float* w = wr + (int64_t)e * d;
        float s = 0.f;
        for (int i = lane; i < d; i += 64) s += xr[i] * w[i];
        s = wave_sum(s);
        if (lane == e) mine = s;
    }
    const float mx = wave_max(mine);
    const float ex = lane < E ? expf(mine - mx) : 0.f;
    float p = ex / wave_sum(ex);                               // router probability of expert `lane`
    float chosen_p = 0.f;
    int chosen_e = 0;
    float total = 0.f;
    for (int j = 0; j < top_k; ++j) {                          // the largest remaining probability, lowest expert index on ties
        const float best = wave_max(lane < E ? p : -1.f);
        const unsigned long long m = __ballot(lane < E && p == best);
        const int e = __ffsll((long long)m) - 1;
        if (lane == j) { chosen_p = best; chosen_e = e; }
        total += best;
        if (lane == e) p = -1.f;
    }
    if (lane < top_k) {
        idx[tok * top_k + lane] = chosen_e;
        wts[tok * top_k + lane] = chosen_p / total;
    }
}

// rank[entry] = number of earlier entries (token-major order) routed to the same expert; count[e] = entries of expert e.
__global__ __launch_bounds__(1024) void moe_rank_kernel(const int* __restrict__ idx, int64_t n_entries, int* __restrict__ rank, int* __restrict__ count) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_entries; c0 += 1024) {
        const int64_t n = c0 + tid;
        const bool hit = n < n_entries && idx[n] == e;
        const unsigned long long m = __ballot(hit);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (hit) rank[n] = off + before;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_tot[w]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) count[e] = base_s;
}

constexpr int TM = 64;        // rows of a tile of the grouped products

__global__ __launch_bounds__(256) void moe_layout_kernel(const int* __restrict__ idx, const int* __restrict__ rank, const int* __restrict__ count,
                                                         int64_t n_entries, int top_k, int E, int max_tiles, int* __restrict__ pos,
                                                         int* __restrict__ src_row, int* __restrict__ tile_expert, int* __restrict__ tile_rows) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n < n_entries) {
        const int e = idx[n];
        int off = 0;
        for (int q = 0; q < e; ++q) off += (count[q] + TM - 1) / TM * TM;
        const int dst = off + rank[n];
        pos[n] = dst;
        src_row[dst] = (int)(n / top_k);
    }
    if (n < max_tiles) {
        const int row0 = (int)n * TM;
        int off = 0, te = -1, tr = 0;
        for (int q = 0; q < E; ++q) {
            const int c = count[q], padded = (c + TM - 1) / TM * TM;
            if (row0 >= off && row0 < off + padded) { te = q; tr = min(TM, c - (row0 - off)); }
            off += padded;
        }
        tile_expert[n] = te;
        tile_rows[n] = tr;
    }
}

// ---------------------------------------------------------------------------------------------------------------- grouped products
constexpr int BN = 64, BK = 32;
constexpr int LDS_F32 = BK + 4;   // floats per LDS row (144 B: keeps 16-byte alignment, staggers banks)
constexpr int LDS_BF = BK + 8;    // bf16 per LDS row (80 B)

struct MoeGemmArgs {
    const float* A; int64_t lda;              // SWIGLU: token rows gathered through src_row; otherwise the packed rows themselves
    const void* W; const void* Wlo;           // [E][n_w][K]: fp32, or bf16 high halves (+ low halves under PREC_BF16X3)
    int n_w;                                  // rows of one expert's matrix (2 * N with SWIGLU: gate rows, then up rows)
    float* Y; int64_t ldy;                    // packed rows
    int N, K;
    const int* src_row; const int* tile_expert; const int* tile_rows;
};

template <int PREC, bool SWIGLU>
__global__ __launch_bounds__(256) void moe_gemm_kernel(const MoeGemmArgs a) {
    constexpr int NB = SWIGLU ? 2 : 1;
    constexpr int A_BYTES = PREC == PREC_F32 ? TM * LDS_F32 * 4 : (PREC == PREC_BF16X3 ? 2 : 1) * TM * LDS_BF * 2;
    constexpr int B_BYTES = PREC == PREC_F32 ? BN * LDS_F32 * 4 : (PREC == PREC_BF16X3 ? 2 : 1) * BN * LDS_BF * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + NB * B_BYTES];
    const int tile = blockIdx.x;
    const int e = a.tile_expert[tile];
    if (e < 0) return;                                          // beyond the table: uniform over the workgroup
    const int rows = a.tile_rows[tile];
    float* As = reinterpret_cast<float*>(smem);
    unsigned short* Ab = reinterpret_cast<unsigned short*>(smem);
    unsigned short* Al = Ab + TM * LDS_BF;
    auto Bs = [&](int nb) { return reinterpret_cast<float*>(smem + A_BYTES + nb * B_BYTES); };
    auto Bb = [&](int nb) { return reinterpret_cast<unsigned short*>(smem + A_BYTES + nb * B_BYTES); };
    auto Bl = [&](int nb) { return Bb(nb) + BN * LDS_BF; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.y * BN;
    const int N = a.N, K = a.K;
    const int ar = tid >> 3, akq = (tid & 7) * 4;              // A (and fp32 W): rows ar, ar + 32; 4 k's
    const float* arow[2];
    bool arow_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = ar + 32 * i;
        arow_ok[i] = r < rows;
        const int64_t src = SWIGLU ? (int64_t)a.src_row[tile * TM + (arow_ok[i] ? r : 0)] : (int64_t)tile * TM + r;
        arow[i] = a.A + src * a.lda;
    }
    const float* Wf = reinterpret_cast<const float*>(a.W) + (int64_t)e * a.n_w * K;
    const unsigned short* Wb = reinterpret_cast<const unsigned short*>(a.W) + (int64_t)e * a.n_w * K;
    const unsigned short* Wl = reinterpret_cast<const unsigned short*>(a.Wlo) + (int64_t)e * a.n_w * K;
    const int br = tid >> 2, bk8 = (tid & 3) * 8;              // bf16 W: row br, 8 k's

    f32x4 acc[NB][2][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[nb][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += BK) {                       // K is a multiple of 32 (checked on the host)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow_ok[i]) v = *reinterpret_cast<const float4*>(arow[i] + k0 + akq);
            const int r = ar + 32 * i;
            if (PREC == PREC_F32) {
                *reinterpret_cast<float4*>(&As[r * LDS_F32 + akq]) = v;
            } else {
                const ushort4 p = make_ushort4(f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w));
                *reinterpret_cast<ushort4*>(&Ab[r * LDS_BF + akq]) = p;
                if (PREC == PREC_BF16X3)
                    *reinterpret_cast<ushort4*>(&Al[r * LDS_BF + akq]) =
                        make_ushort4(f2bf(v.x - bf2f(p.x)), f2bf(v.y - bf2f(p.y)), f2bf(v.z - bf2f(p.z)), f2bf(v.w - bf2f(p.w)));
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int wrow0 = nb * N + n0;                      // gate rows first, then up rows
            if (PREC == PREC_F32) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = ar + 32 * i;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n0 + r < N) v = *reinterpret_cast<const float4*>(Wf + (int64_t)(wrow0 + r) * K + k0 + akq);
                    *reinterpret_cast<float4*>(&Bs(nb)[r * LDS_F32 + akq]) = v;
                }
            } else {
                uint4 v = make_uint4(0u, 0u, 0u, 0u), u = v;
                if (n0 + br < N) {
                    v = *reinterpret_cast<const uint4*>(Wb + (int64_t)(wrow0 + br) * K + k0 + bk8);
                    if (PREC == PREC_BF16X3) u = *reinterpret_cast<const uint4*>(Wl + (int64_t)(wrow0 + br) * K + k0 + bk8);
                }
                *reinterpret_cast<uint4*>(&Bb(nb)[br * LDS_BF + bk8]) = v;
                if (PREC == PREC_BF16X3) *reinterpret_cast<uint4*>(&Bl(nb)[br * LDS_BF + bk8]) = u;
            }
        }
        __syncthreads();
        if (PREC == PREC_F32) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 16) {
                float4 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(&As[(wr * 32 + i * 16 + (lane & 15)) * LDS_F32 + kk + 4 * (lane >> 4)]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float4 bfr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const float4*>(&Bs(nb)[(wc * 32 + j * 16 + (lane & 15)) * LDS_F32 + kk + 4 * (lane >> 4)]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bfr[j].x, acc[nb][i][j], 0, 0, 0);
                            acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bfr[j].y, acc[nb][i][j], 0, 0, 0);
                            acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bfr[j].z, acc[nb][i][j], 0, 0, 0);
                            acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bfr[j].w, acc[nb][i][j], 0, 0, 0);
                        }
                }
            }
        } else {
            bf16x8 af[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(&Ab[(wr * 32 + i * 16 + (lane & 15)) * LDS_BF + 8 * (lane >> 4)]);
                if (PREC == PREC_BF16X3) al[i] = *reinterpret_cast<const bf16x8*>(&Al[(wr * 32 + i * 16 + (lane & 15)) * LDS_BF + 8 * (lane >> 4)]);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bf16x8 bfr[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bfr[j] = *reinterpret_cast<const bf16x8*>(&Bb(nb)[(wc * 32 + j * 16 + (lane & 15)) * LDS_BF + 8 * (lane >> 4)]);
                    if (PREC == PREC_BF16X3) bl[j] = *reinterpret_cast<const bf16x8*>(&Bl(nb)[(wc * 32 + j * 16 + (lane & 15)) * LDS_BF + 8 * (lane >> 4)]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (PREC == PREC_BF16X3) {             // x w ~= x_lo w_hi + x_hi w_lo + x_hi w_hi (the two small terms first)
                            acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bfr[j], acc[nb][i][j], 0, 0, 0);
                            acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[nb][i][j], 0, 0, 0);
                        }
                        acc[nb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[nb][i][j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    // C layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = wr * 32 + i * 16 + (lane >> 4) * 4 + r;
            if (m >= rows) continue;
            float* yrow = a.Y + ((int64_t)tile * TM + m) * a.ldy;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wc * 32 + j * 16 + (lane & 15);
                if (n >= N) continue;
                float v = acc[0][i][j][r];
                if (SWIGLU) v = v / (1.f + expf(-v)) * acc[NB - 1][i][j][r];
                yrow[n] = v;
            }
        }
}

__global__ __launch_bounds__(256) void moe_combine_kernel(const float* __restrict__ y2, const float* __restrict__ res, const int* __restrict__ pos,
                                                          const float* __restrict__ wts, int64_t T, int d, int top_k, float* __restrict__ y) {
    const int d4 = d >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * d4) return;
    const int64_t tok = i / d4;
    const int c = (int)(i % d4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < top_k; ++j) {
        const float w = wts[tok * top_k + j];
        const float4 v = *reinterpret_cast<const float4*>(y2 + (int64_t)pos[tok * top_k + j] * d + c);
        s.x += v.x * w; s.y += v.y * w; s.z += v.z * w; s.w += v.w * w;
    }
    if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + tok * d + c);
        s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
    }
    *reinterpret_cast<float4*>(y + tok * d + c) = s;
}

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct MoeWorkspace {
    int* idx; float* wts; int* rank; int* pos; int* count; int* src_row; int* tile_expert; int* tile_rows; float* H; float* Y2;
    int max_tiles; size_t bytes;
};

// idx and wts come first (tests and callers read the routing back from the head of the workspace)
MoeWorkspace carve(void* base, int64_t T, int d, int inter, int E, int top_k) {
    MoeWorkspace w;
    const int64_t n = T * top_k;
    w.max_tiles = (int)((n + TM - 1) / TM) + E;
    const int64_t P = (int64_t)w.max_tiles * TM;
    size_t off = 0;
    auto take = [&](size_t b) { void* p = base ? static_cast<char*>(base) + off : nullptr; off += up256(b); return p; };
    w.idx = static_cast<int*>(take(n * 4));
    w.wts = static_cast<float*>(take(n * 4));
    w.rank = static_cast<int*>(take(n * 4));
    w.pos = static_cast<int*>(take(n * 4));
    w.count = static_cast<int*>(take((size_t)E * 4));
    w.src_row = static_cast<int*>(take(P * 4));
    w.tile_expert = static_cast<int*>(take((size_t)w.max_tiles * 4));
    w.tile_rows = static_cast<int*>(take((size_t)w.max_tiles * 4));
    w.H = static_cast<float*>(take((size_t)P * inter * 4));
    w.Y2 = static_cast<float*>(take((size_t)P * d * 4));
    w.bytes = off;
    return w;
}

template <bool SWIGLU>
void launch_gemm(const MoeGemmArgs& g, int prec, int max_tiles, hipStream_t s) {
    const dim3 grid(max_tiles, (g.N + BN - 1) / BN);
    if (prec == PREC_BF16X3) hipLaunchKernelGGL((moe_gemm_kernel<PREC_BF16X3, SWIGLU>), grid, dim3(256), 0, s, g);
    else if (prec == PREC_BF16) hipLaunchKernelGGL((moe_gemm_kernel<PREC_BF16, SWIGLU>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((moe_gemm_kernel<PREC_F32, SWIGLU>), grid, dim3(256), 0, s, g);
}

}  // namespace
}  // namespace adx

using namespace adx;

#define ADX_API extern "C" __attribute__((visibility("default")))

ADX_API int adx_op_rmsnorm(const float* x, const float* weight, float eps, int64_t rows, int d, float* y, void* stream) {
    if (!x || !weight || !y || rows <= 0 || d <= 0) { set_error("rmsnorm: null pointer or empty shape"); return ADX_EINVAL; }
    if ((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(weight)) & 15)) {
        set_error("rmsnorm: rows of a multiple of 4 floats need 16-byte aligned pointers"); return ADX_EINVAL;
    }
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, weight, eps, rows, d, y);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADX_OK : hip_fail(e, "rmsnorm");
}

ADX_API int adx_op_rotary(float* x, const float* cos_table, const float* sin_table, int B, int S, int H, int dh, int pos0, int n_pos, void* stream) {
    if (!x || !cos_table || !sin_table) { set_error("rotary: null pointer"); return ADX_EINVAL; }
    if (B <= 0 || S <= 0 || H <= 0 || dh <= 0 || (dh & 1) || pos0 < 0 || pos0 + S > n_pos) {
        set_error("rotary: needs an even head width and tables that cover positions pos0 .. pos0 + S - 1"); return ADX_EINVAL;
    }
    const int64_t n = (int64_t)B * S * H * (dh / 2);
    hipLaunchKernelGGL(rotary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, cos_table, sin_table, (int64_t)B * S, S, H,
                       dh, pos0);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADX_OK : hip_fail(e, "rotary");
}

ADX_API size_t adx_moe_workspace_bytes(int64_t T, int d, int inter, int E, int top_k) {
    if (T <= 0 || d <= 0 || inter <= 0 || E <= 0 || top_k <= 0) return 0;
    return carve(nullptr, T, d, inter, E, top_k).bytes;
}

ADX_API int adx_op_moe_block(const float* x, const float* res, const float* router_w, const void* w13, const void* w2, int64_t T, int d, int inter,
                             int E, int top_k, int precision, void* workspace, size_t workspace_bytes, float* y, void* stream) {
    if (!x || !router_w || !w13 || !w2 || !workspace || !y) { set_error("moe_block: null pointer"); return ADX_EINVAL; }
    if (T <= 0 || T * (int64_t)top_k > (1 << 30) || E < 1 || E > 64 || top_k < 1 || top_k > E || top_k > 8 || (d % 32) != 0 || (inter % 32) != 0) {
        set_error("moe_block: needs 1 <= top_k <= min(E, 8), E <= 64, and hidden / expert widths that are multiples of 32"); return ADX_EINVAL;
    }
    if (precision != PREC_F32 && precision != PREC_BF16 && precision != PREC_BF16X3) { set_error("moe_block: precision must be fp32, bf16 or bf16x3"); return ADX_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(w13) |
         reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(workspace)) & 15) { set_error("moe_block: pointers must be 16-byte aligned"); return ADX_EINVAL; }
    const MoeWorkspace w = carve(workspace, T, d, inter, E, top_k);
    if (workspace_bytes < w.bytes) { set_error("moe_block: workspace smaller than adx_moe_workspace_bytes"); return ADX_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = T * top_k;
    hipLaunchKernelGGL(moe_route_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, x, router_w, T, d, E, top_k, w.idx, w.wts);
    hipLaunchKernelGGL(moe_rank_kernel, dim3(E), dim3(1024), 0, s, w.idx, n, w.rank, w.count);
    const int64_t lay = n > w.max_tiles ? n : w.max_tiles;
    hipLaunchKernelGGL(moe_layout_kernel, dim3((unsigned)((lay + 255) / 256)), dim3(256), 0, s, w.idx, w.rank, w.count, n, top_k, E, w.max_tiles, w.pos,
                       w.src_row, w.tile_expert, w.tile_rows);
    const size_t e13 = (size_t)E * 2 * inter * d, e2 = (size_t)E * d * inter;      // elements of the stacked matrices (low halves follow the high ones)
    MoeGemmArgs g1{x, d, w13, precision == PREC_BF16X3 ? static_cast<const unsigned short*>(w13) + e13 : nullptr, 2 * inter, w.H, inter, inter, d,
                   w.src_row, w.tile_expert, w.tile_rows};
    launch_gemm<true>(g1, precision, w.max_tiles, s);
    MoeGemmArgs g2{w.H, inter, w2, precision == PREC_BF16X3 ? static_cast<const unsigned short*>(w2) + e2 : nullptr, d, w.Y2, d, d, inter,
                   w.src_row, w.tile_expert, w.tile_rows};
    launch_gemm<false>(g2, precision, w.max_tiles, s);
    hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)((T * (d / 4) + 255) / 256)), dim3(256), 0, s, w.Y2, res, w.pos, w.wts, T, d, top_k, y);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADX_OK : hip_fail(e, "moe_block");
}

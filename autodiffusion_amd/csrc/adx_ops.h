// Internal C++ interface of the HIP operator set (gfx950 only).
//
// Every launcher enqueues work on `stream` and returns; none allocates or
// synchronises, so a sequence of them can be captured into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adx {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_ERF = 2, ACT_GELU_TANH = 3, ACT_LEAKY = 4, ACT_SILU = 5 };
enum Prec : int { PREC_F32 = 0, PREC_BF16 = 1, PREC_BF16X3 = 2 };

// Row index map: logical row m of an (M x N) operand lives at physical row
//   (m / per_batch) * batch_stride + (m % per_batch) + off      (identity when per_batch == 0).
// It lets a GEMM read or write "the first r rows of every batch item's S-row block"
// or write its result into a slice of a token-concatenated buffer.
struct RowMap {
    int per_batch = 0, batch_stride = 0, off = 0;
};

// Y = res + gate * act(X W^T + bias + pos)        (each term optional)
struct LinearArgs {
    const float* X = nullptr; int64_t ldx = 0; RowMap xmap;
    const void* W = nullptr;            // [N][K] row-major; float (PREC_F32) or bf16 bits (PREC_BF16 / PREC_BF16X3: high halves)
    const void* Wlo = nullptr;          // PREC_BF16X3: bf16 bits of w - bf16(w)
    const float* bias = nullptr;        // [N]
    float* Y = nullptr; int64_t ldy = 0; RowMap ymap;
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    int in_act = ACT_NONE;              // activation applied to X while it is staged (SiLU for adaLN)
    const float* gate = nullptr; int64_t ldg = 0;   // [M][N], rows follow ymap ...
    int gate_group = 0, gate_period = 0;            // ... or, with gate_group > 0, row (r / gate_group) * gate_period + r % gate_period (shared conditioning)
    const float* res = nullptr;  int64_t ldr = 0;   // [M][N], rows follow ymap
    const float* pos = nullptr;  int64_t ldp = 0; int pos_period = 0; int pos_off = 0;  // + pos[(m % period) + off][n]
    // column map of Y: col n -> (n / col_group) * col_stride + (n % col_group) + col_off (identity when col_group == 0)
    int col_group = 0, col_stride = 0, col_off = 0;
    int prec = PREC_F32;
    // optional scratch (device memory, 16-byte aligned) for K-blocked bf16 planes of X (and of W unless Wblk is given):
    // at least linear_stage_bytes(M, Wblk ? 0 : N, K, prec) bytes; enables the LDS-DMA form of the large-tile kernel
    void* xstage = nullptr; size_t xstage_bytes = 0;
    const void* Wblk = nullptr; const void* Wblk_lo = nullptr;      // W already in the K-blocked layout [K/16][N][16]
    bool force_staged = false;          // take the staged (LDS-DMA) kernel whatever the shape, when xstage is large enough
    // X already as K-blocked bf16 planes [K/16][M][16] (high halves; low halves with PREC_BF16X3), rows in identity order, any
    // input activation already applied: the staged kernel then runs without its re-layout pass over X, and X itself may be null.
    // Only legal when linear_planes_ok(M, N, K, prec).
    const unsigned short* Xh = nullptr; const unsigned short* Xl = nullptr;
    // Also store the result as K-blocked planes of the NEXT product's operand, [N/16][M][16] (Yl may be null: high halves only);
    // Y may then be null (planes only).  Large-tile kernels only (linear_big_ok), 16-byte epilogue, no column map.
    unsigned short* Yh = nullptr; unsigned short* Yl = nullptr;
    // Planes this call reads or writes hold fp16 instead of bf16 halves, and a product whose X arrives as planes is an fp16
    // product with W rounded once to fp16, K-blocked in Wblk_h: x_hi w + x_lo w when Xl is given, x_hi w alone otherwise.  A call
    // whose X is fp32 runs as its `prec` says and only its output planes change format.
    bool f16 = false; const void* Wblk_h = nullptr;
};
// A call with pre-staged X planes (and a pre-blocked or stageable W) takes the staged kernel: bf16 precisions, K a multiple of 32,
// enough 256 x 128 tiles to fill the chip.
bool linear_planes_ok(int M, int N, int K, int prec);
// x[M][K] (row stride ldx, optional SiLU) -> K-blocked planes [K/16][M][16]: hi = bf16(x), lo (nullable) = bf16(x - hi).
hipError_t stage_planes(const float* x, int64_t ldx, int M, int K, int in_act, unsigned short* hi, unsigned short* lo, hipStream_t stream,
                        bool f16 = false);      // f16: fp16 halves (saturating) instead of bf16 ones
size_t linear_stage_bytes(int M, int N, int K, int prec);
hipError_t linear_staged_prepare();          // one-time, non-stream set-up of the staged product's kernels (call outside graph capture)
hipError_t linear_block_weight(const void* w, int N, int K, void* out, hipStream_t stream);   // [N][K] bf16 -> [K/16][N][16]
hipError_t linear(const LinearArgs& a, hipStream_t stream);        // picks linear_big when linear_big_ok(a)
// Two products in one kernel, the hidden layer kept on the chip: Y = epilogue2(act1(X W1^T + b1) W2^T + b2) for K = 256 inputs and
// 256 outputs (gemm.hip).  l1 carries X, the first weight (K-blocked copies required), bias and activation; l2 the second
// weight and the whole epilogue (bias, activation, gate, residual, Y and / or plane outputs); l1.Y and l2.X are not used.
bool mlp256_ok(const LinearArgs& l1, const LinearArgs& l2);
hipError_t mlp256(const LinearArgs& l1, const LinearArgs& l2, hipStream_t stream);
bool linear_big_ok(const LinearArgs& a);                             // large-tile bf16 kernel (gemm.hip) applies
hipError_t linear_big(const LinearArgs& a, hipStream_t stream);

// y = LN(x) [* gamma + beta] [* (1 + scale) + shift] [relu]; rows of length d.
struct LayerNormArgs {
    const float* x = nullptr; int64_t ldx = 0; RowMap xmap;
    float* y = nullptr; int64_t ldy = 0; RowMap ymap;
    const float* gamma = nullptr; const float* beta = nullptr;      // [d]
    const float* shift = nullptr; const float* scale = nullptr; int64_t ldm = 0;  // [rows][d] slices, identity row map
    const float* gate = nullptr; int64_t ldg = 0;   // y = res + gate * (...) ; [rows][d], identity row map
    // rows that share their conditioning (several samples of one scenario): shift / scale / gate of row r are read from row
    // (r / mod_group) * mod_period + r % mod_period of those tables (mod_group 0: row r itself)
    int mod_group = 0, mod_period = 0;
    const float* res = nullptr; int64_t ldr = 0;
    float eps = 1e-5f;
    int rows = 0, d = 0;
    int relu = 0;
    // also (or, with y == nullptr, only) store the result as K-blocked bf16 planes [d/16][rows][16] for a following product:
    // ph = bf16(v), pl (nullable) = bf16(v - ph).  d a multiple of 256 (the register-resident kernel), identity row order.
    unsigned short* ph = nullptr; unsigned short* pl = nullptr;
    bool pf16 = false;                  // the planes hold fp16 halves (saturating) instead of bf16 ones
};
hipError_t layernorm(const LayerNormArgs& a, hipStream_t stream);

// out[b, q, h*dv + c] = sum_j softmax_j(scale * <Q[b,q,h,:], K[b,j,h,:]>) V[b,j,h,c]
// mask_off < 0: no mask; otherwise key j is visible to query q iff j <= q + mask_off.
struct AttentionArgs {
    const float* Q = nullptr; int64_t ldq = 0; int64_t q_batch_stride = 0;   // element strides
    const float* K = nullptr; int64_t ldk = 0; int64_t k_batch_stride = 0;
    const float* V = nullptr; int64_t ldv = 0; int64_t v_batch_stride = 0;
    float* O = nullptr; int64_t ldo = 0; int64_t o_batch_stride = 0;
    int B = 0, H = 0, Sq = 0, Skv = 0, dh = 0, dv = 0;
    float scale = 1.f;
    int mask_off = -1;
    int prec = PREC_F32;        // PREC_BF16 / PREC_BF16X3: short unmasked sequences with head dims that are multiples of 32 use the bf16 matrix cores
    // Small-head kernels only.  state != nullptr: instead of O, store the running softmax state of every query,
    // {m, l, o[dh]} per (query, head) with m in log2 units of the scaled score and o un-normalised; row stride
    // state_ld floats (= H * (2 + dh)).  key0 != nullptr: key/value 0 of EVERY batch item are read from
    // key0[0:H*dh] / key0[H*dh:2*H*dh] instead of K/V (the per-step time token).
    float* state = nullptr; int64_t state_ld = 0; int64_t state_batch_stride = 0;
    const float* key0 = nullptr;
};
hipError_t attention(const AttentionArgs& a, hipStream_t stream);         // picks attention_mfma when attention_mfma_ok(a)
bool attention_mfma_ok(const AttentionArgs& a);                            // Skv <= 128, head dims multiples of 16, no mask
hipError_t attention_mfma(const AttentionArgs& a, hipStream_t stream);    // attn_mfma.hip
bool attention_long_ok(const AttentionArgs& a);                            // any length, optional mask, equal head dims that are multiples of 16 up to 128
hipError_t attention_long(const AttentionArgs& a, hipStream_t stream);    // attn_mfma.hip: keys in chunks of 64, online softmax

// Attention of a FEW queries (Sq <= 16 per launch group) against many keys, head_dim 4: one workgroup per
// (batch item, group of NQ queries), all heads; keys are spread over the threads and the per-thread softmax
// states are merged by shuffles and one LDS pass.  Honors a.key0; no mask.
hipError_t attention_few_queries(const AttentionArgs& a, hipStream_t stream);

// Fold ONE extra key (the time token: key0 = [k(H*dh), v(H*dh)], shared by all batch items) into cached softmax
// states: O[b, r, h, :] = (o * 2^(m - m') + 2^(s - m') v) / (l * 2^(m - m') + 2^(s - m')),  s = scale*log2e*<q, k>.
// Q/O/state address rows exactly as in AttentionArgs (Sq rows per batch item).  head_dim 4.
hipError_t attention_merge_key(const AttentionArgs& a, hipStream_t stream);

// Attention of a few query rows per item over Skv keys of the item, a thread per (head, query), keys walked through
// an LDS tile shared by 32 queries.  8 heads x 4.  Two uses:
//  * query rows shared by every item (q_batch_stride 0; the time token under each scheduler step): row s of Q holds
//    q | k | v (own = 1: the row's own key opens the softmax) and O[(s * B + b) * 32] receives the normalised output;
//  * per-item rows (own = 0) whose running softmax state {m, l, o[4]} is stored (state != nullptr), heads permuted as in
//    AttentionArgs::state.
struct RowsAttnArgs {
    const float* Q = nullptr; int64_t ldq = 0, q_batch_stride = 0; int own = 0;
    const float* K = nullptr; const float* V = nullptr; int64_t ldk = 0, k_batch_stride = 0;
    float* O = nullptr;
    float* state = nullptr; int64_t state_ld = 0, state_batch_stride = 0;
    int B = 0, NS = 0, Skv = 0;
    float scale = 1.f;
};
hipError_t attention_rows(const RowsAttnArgs& a, hipStream_t stream);

// state_out[b, r, h] = state_in[b, r, h] with NK more keys folded in (keys/values of rows K/V[b, 0..NK-1]); Q rows as in
// AttentionArgs, states in the permuted-head layout, row stride state_ld.  NK <= 64, 8 heads x 4.
struct FoldKeysArgs {
    const float* Q = nullptr; int64_t ldq = 0, q_batch_stride = 0;
    const float* K = nullptr; const float* V = nullptr; int64_t ldk = 0, k_batch_stride = 0;
    const float* state_in = nullptr; float* state_out = nullptr; int64_t state_ld = 0, state_batch_stride = 0;
    int B = 0, Sq = 0, NK = 0;
    float scale = 1.f;
};
hipError_t attention_fold_keys(const FoldKeysArgs& a, hipStream_t stream);

// Fused per-row pipeline of the variant-A encoder (d_model 32, 8 heads, FFN 128); see fused_a.hip.
// For every conditioning row: fold the time-token key (key0 = [k(32), v(32)]) into the cached layer-0 softmax state
// (row 0 of each item takes its attention output from att0 instead), out_proj + residual, LayerNorm, FFN + residual,
// then layer 1's LayerNorm and K/V projection.  Rows r < n_keep of every item also store the layer-1 residual input
// (x1) and its normalised copy (h1).
struct EncRowsArgs {
    const float* x0 = nullptr;                       // [B*L][32] encoder input rows (enc_rows reads row 0 of each item only)
    const float* q = nullptr; int64_t ldq = 0;       // layer-0 queries, row stride ldq           } read by enc_rows_pack,
    const float* state = nullptr;                    // [B*L][8][6] {m, l, o[4]}                  } not by enc_rows
    const float4* packed = nullptr;                  // x0, q, state of every row in lane order (enc_rows_packed_bytes)
    const float* att0 = nullptr;                     // [B][32] layer-0 attention output of row 0 (the time token) of each item
    const float* x0_time = nullptr;                  // [32] encoder input of row 0, the same for every item
    const float* key0 = nullptr;                     // [64]
    const void* image = nullptr;                     // weight fragment image from enc_rows_build_image
    const float* img_f32 = nullptr; const unsigned short* img_bf = nullptr;   // filled by the launcher
    const float* vecs = nullptr;                     // [416] bo, gamma2, beta2, b2, gamma1', beta1', b1(128), bkv(64), bq(32)
    float* kv = nullptr; int64_t ldkv = 0;           // K at kv[row*ldkv + 0..31], V at +32..63
    const float4* gates = nullptr;                   // when set (with gpk): this step's slice of the gate table (enc_rows_pack_gates), and
                                                     // `packed` holds the 8-vector layout that goes with it
    float4* gpk = nullptr;                           // when set: NO K/V projection; the layer-1 LayerNorm rows are stored instead, in
                                                     // lane order gpk[(tile * 4 + g) * 64 + lane] (128 B per row, enc_rows_g_bytes), for attention_g
    int prio = 0;                                    // wave priority the kernel raises itself to (0: leave it): the tail of the previous step runs beside it
    float* x1 = nullptr;                             // [B*L][32] layer-1 residual input, rows r < n_keep only
    float* q1 = nullptr; int64_t ldq1 = 0;           // layer-1 queries, rows r < n_keep only: row r of item b at q1[b * q1_item_stride + r * ldq1]
    int64_t q1_item_stride = 0;                      // 0: L * ldq1 (the rows of a [B*L][ldq1] buffer)
    int B = 0, L = 0, n_keep = 0;
    float scale = 1.f;
};
size_t enc_rows_image_bytes();
hipError_t enc_rows_build_image(const float* Wo, const float* W1, const float* W2, const float* Wkv, const float* Wq, void* image,
                                hipStream_t stream, int f16_mats = 0);   // f16_mats: bit m = the 16-bit fragments of matrix m (Wo, W1, W2, Wkv, Wq) hold fp16 (the mixed modes of enc_rows)
size_t enc_rows_packed_bytes(int B, int L);
hipError_t enc_rows_pack(const EncRowsArgs& a, hipStream_t stream);      // once per chunk: x0/q/state -> packed
hipError_t enc_rows(const EncRowsArgs& a, int prec, hipStream_t stream, int mix = 0);  // once per step; mix (split-bf16 mode): 7 = the three products on fp16 operands, 15 = also half-width inputs and output (gate table)
size_t enc_rows_g_bytes(int B, int L);
// Gate-table form (once per chunk instead of enc_rows_pack): packed <- x0 and o / l in lane order, gates[step][tile][lane] <- the
// fold gates of every step (tq: [n_steps][96] per-step q | k | v rows of the time token).  A step then passes
// gates + step * (enc_rows_gates_bytes(B, L, 1) / 16) in EncRowsArgs::gates.
size_t enc_rows_gates_bytes(int B, int L, int n_steps);
hipError_t enc_rows_pack_gates(const EncRowsArgs& a, float4* gates, const float* tq, int n_steps, hipStream_t stream, bool half = false);   // half: fp16 o / l and gates (enc_rows mix 15)

// Last encoder layer's attention of the n_keep visible rows of every item (n_keep * 8 heads <= 96) over ALL L rows of the item,
// straight from the LayerNorm rows that enc_rows stored (gpk): keys and values are projected per 32-row tile on the matrix
// cores and never exist in memory.  One workgroup of two wavefronts per item.  O[b * o_batch_stride + i * ldo + c].
struct AttnGArgs {
    const float4* gpk = nullptr;
    const void* image = nullptr;                     // the enc_rows weight image (its K | V projection tiles)
    const float* vecs = nullptr;                     // the enc_rows vectors (the V bias)
    const float* wk = nullptr;                       // [32][32] fp32 K rows of the layer's in_proj_weight
    const float* q1 = nullptr; int64_t ldq1 = 0;     // layer-1 query of visible row i of item b at q1[b * q1_item_stride + i * ldq1]
    int64_t q1_item_stride = 0;                      // 0: L * ldq1
    float* O = nullptr; int64_t ldo = 0, o_batch_stride = 0;
    int B = 0, L = 0, n_keep = 0;
    float scale = 1.f;
    int prio = 0;                                    // as EncRowsArgs::prio
    float lazy = 8.f;                                // a column's softmax reference moves only when a score exceeds it by more than this (log2 units; 0: on every new maximum)
};
bool attention_g_ok(int n_keep);
hipError_t attention_g(const AttnGArgs& a, int prec, hipStream_t stream, int mix = 0);     // mix (split-bf16 mode): 1 weighted sums in fp16, 3 scores and V too, 7 the rows arrive as fp16

// out[i, :] = [sin(t_i w), cos(t_i w)], w_j = 1e4^(-j/(dim/2-1))
hipError_t timestep_embedding(const int64_t* t, int n, int dim, float* out, int64_t ldo, hipStream_t stream);

// y[r, c] = act(x[r, c] + vec[(r / rows_per_vec) * cols + c])   (rows_per_vec <= 0: one vector for all rows)
hipError_t add_vec_act(const float* x, const float* vec, float* y, int64_t rows, int cols, int64_t rows_per_vec, int act,
                       hipStream_t stream);

// out[b * batch_stride + c] = a[b * a_stride + c] + bvec[c]   (one row per batch item, e.g. the time token)
hipError_t set_row(float* out, int64_t batch_stride, int B, const float* a, int64_t a_stride, const float* bvec, int cols,
                   hipStream_t stream);

// trajectory normalisation: ch0 = x/25 - 1, ch1 = y/5, others copied; C channels (4 or 2)
hipError_t normalize_traj(const float* x, float* y, int64_t n_points, int C, hipStream_t stream);
// inverse (+1)*25, *5, ch2 := 0, ch3 copied; optional residual add
hipError_t denormalize_traj(const float* y, const float* residual, float* x, int64_t n_points, int C, hipStream_t stream);
// y = x * std[c] + mean[c]
hipError_t affine_channels(const float* x, const float* mean, const float* std, float* y, int64_t n_points, int C,
                           hipStream_t stream);

// One scheduler update, elementwise over n values (see include/adx.h, adx_step_coef):
//   x0  = clamp(a0 * x + a1 * out, -clip, clip)
//   eps = eps_mode ? (b0 * x - x0) * b1 : out
//   x'  = c0 * x0 + c1 * x + c2 * eps + c3 * z
struct StepCoef { float t, a0, a1, clip, eps_mode, b0, b1, c0, c1, c2, c3, z_index; };
hipError_t sched_step(const float* x, const float* model_out, const float* z, float* x_next, float* x0_out,
                      int64_t n, const StepCoef* coef_dev, hipStream_t stream);

// Fused tail of a variant-A denoiser call + scheduler update; see tail_a.hip.  All offsets index the float buffer `w`
// (matrices packed [K/4][N][4] by tail_pack_matrix, vectors copied as they are).
struct TailMat { int w_off = 0, b_off = 0, N = 0, K = 0; };
struct TailLayer {
    int ln1_g = 0, ln1_b = 0, ln2_g = 0, ln2_b = 0, ln3_g = 0, ln3_b = 0;
    TailMat wqkv, wo, wq, wkv, wo2, w1, w2;
};
struct TailArgs {
    const float* w = nullptr;
    const float* att1 = nullptr;      // [B*L][32] last encoder layer attention output, rows 0..T of each item
    const float* x1 = nullptr;        // [B*L][32] its residual input, rows 0..T
    const float* x = nullptr;         // [B][T][C] x_t
    const float* z = nullptr;         // [B][T][C] noise of this step or nullptr
    const StepCoef* coef = nullptr;   // device
    float* x_next = nullptr;          // [B][T][C]
    float* model_out = nullptr;       // [B][T][C] or nullptr
    int B = 0, L = 0, T = 0, C = 0, n_dec = 0;
    int e_ln2_g = 0, e_ln2_b = 0, lnf_g = 0, lnf_b = 0, pos_off = 0;
    TailMat e_wo, e_w1, e_w2, d_in, d_head;
    TailLayer dec[4];
};
hipError_t tail_pack_matrix(const float* W, int N, int K, float* out, hipStream_t stream);
hipError_t tail_a(const TailArgs& a, hipStream_t stream);

// MFMA form of the same tail (one wavefront per item, fp32 16x16x4 MFMA); see tail_mfma.hip.  in_dim = out_dim = 4,
// T + 1 <= 16.  `image`: weight fragments in consumption order, block b (a 32x32 map) written by tail_mfma_pack_block:
//   0 enc.out_proj | 1..8 enc FFN (W1 slice i, W2 slice i) | per decoder layer 16 blocks: self q, k, v, out_proj,
//   cross q, memory k, v, out_proj, FFN (W1 slice i, W2 slice i); input embedding and head by tail_mfma_pack_ends.
// `vecs`: tail_mfma_vec_floats() floats laid out per tail_mfma_vec_layout() (pos: [16][32], head_b: zero-padded to 16).
struct TailMfmaArgs {
    const void* image = nullptr;
    const float* vecs = nullptr;
    const float* att1 = nullptr;      // as in TailArgs
    const float* x1 = nullptr;
    const float* x = nullptr;         // [B][T][4]
    const float* z = nullptr;
    const StepCoef* coef = nullptr;
    float* x_next = nullptr;
    float* model_out = nullptr;
    int B = 0, L = 0, T = 0, n_dec = 0;
    const float* pos = nullptr;       // pos_emb [T][32]: read from here when T + 1 > 16 (the table in `vecs` holds 16 rows)
    // arithmetic form the blocks were packed in (tail_mfma_pack_block(..., split)): `split` for the maps on memory rows, `traj` for
    // the maps on trajectory tokens.  0 fp32, 1 split-bf16 (memory rows only), 2 split-fp16; supported pairs: (0,0) (1,0) (2,0) (2,2)
    int split = 0, traj = 0;
    int prio = 0;                     // wave priority (s_setprio 0..3) against the kernels that run beside the tail on other streams
};
struct TailMfmaVecLayout {
    int e_bo, e_ln2_g, e_ln2_b, e_b1, e_b2, din_b, pos, layer0, layer_size;
    int ln1_g, ln1_b, ln2_g, ln2_b, ln3_g, ln3_b, bqkv, bo, bq, bkv, bo2, b1, b2;      // relative to layer0 + l * layer_size
    int lnf_g, lnf_b, head_b;
};
int tail_mfma_vec_floats();
int tail_mfma_blocks(int n_dec);
size_t tail_mfma_image_bytes(int n_dec);
TailMfmaVecLayout tail_mfma_vec_layout();
hipError_t tail_mfma_pack_block(const float* W, int ldw, int row0, int col0, void* image, int block, int split, hipStream_t stream);
hipError_t tail_mfma_pack_ends(const float* Win, const float* Whead, void* image, hipStream_t stream);
hipError_t tail_mfma(const TailMfmaArgs& a, hipStream_t stream);

// Reduce M sampled trajectories per scenario to k modes (farthest-point sampling on the end points + EM); see modes.hip.
struct SelectModesArgs {
    const float* traj = nullptr;      // [B][M][L][D]
    float* modes = nullptr;           // [B][k][L][D], ordered by weight (descending)
    float* weights = nullptr;         // [B][k]
    int B = 0, M = 0, L = 0, D = 0, k = 0, em_iter = 25;
    float org_sigma = 0.1f, init_sigma = 0.1f;
};
hipError_t select_modes(const SelectModesArgs& a, hipStream_t stream);

// Per-scenario ADE / FDE / heading errors / max displacement; out[B][18], layout in metrics.hip.  S divides 80, C in {2, 4}.
hipError_t trajectory_metrics(const float* pred, const float* label, int B, int S, int C, float* out, hipStream_t stream);

// Device-side Gaussian noise (extras.hip).  ctl_dev = {seed, global index of batch row 0}; out[n_lead][B][row_len], the value
// of an element depends only on (seed, global row, leading index, element / 4): Philox4x32-10 + Box-Muller.
hipError_t set_noise_ctl(uint64_t* ctl_dev, uint64_t seed, uint64_t row0, hipStream_t stream);
hipError_t randn_rows(const uint64_t* ctl_dev, float* out, int n_lead, int B, int row_len, hipStream_t stream);

// Forward diffusion of n_rows rows of row_len values with a timestep per row: xt = sqrt_acp[t] x0 + sqrt_1m_acp[t] noise;
// v (optional) = sqrt_acp[t] noise - sqrt_1m_acp[t] x0.
hipError_t q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, float* xt,
                    float* v, int n_rows, int row_len, hipStream_t stream);
// loss_rows[b] = weight[t[b]] * mean((out[b] - target[b])^2); loss_mean (optional) = mean over rows.
hipError_t traj_loss(const float* logits, const float* label, float* part, float* out, int B, int T, int C, int Cl, int smooth_l1,
                     hipStream_t stream);
hipError_t weighted_mse(const float* out, const float* target, const int64_t* t, const float* weight, float* loss_rows, float* loss_mean,
                        int n_rows, int row_len, hipStream_t stream);

// f32 -> bf16 (round to nearest even); ylo (optional) receives bf16(x - bf16(x))
hipError_t to_bf16(const float* x, uint16_t* y, uint16_t* ylo, int64_t n, hipStream_t stream);
hipError_t to_f16(const float* x, uint16_t* y, int64_t n, hipStream_t stream);      // fp16 bits, saturating
// strided 2-D copy: dst[r * ldd + c] = src[(r % src_rows) * lds + c]   (src_rows <= 0: no wrap; wrap = broadcast)
hipError_t copy2d(const float* src, int64_t lds, int64_t src_rows, float* dst, int64_t ldd, int64_t rows, int cols,
                  hipStream_t stream);

}  // namespace adx

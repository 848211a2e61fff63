// Variant-A engine: the encoder/decoder Transformer denoiser with its DDPM/DDIM sampling loop
// and the chunk loop of the generate() wrappers.
//
// Token layout of the conditioning sequence (length L = 1 + n_maps + 2P):
//   row 0            time token        sin/cos(t) + cond_pos[0]
//   rows 1..n_maps   map tokens        maps W_c^T + b_c + cond_pos[1+i]      (loop-invariant: built once per call)
//   next NP rows      transition_info   tr W_p^T + b_p + cond_pos[...]        (invariant within a chunk)
//   last NP rows      trajectory_prior  pr W_p^T + b_p + cond_pos[...]
// Only row 0 of the encoder input changes between diffusion steps.
//
// With cfg.elide the engine skips work the masks make unobservable: the decoder's memory mask lets query
// i see memory rows 0..i+1 only, so (a) cross-attention keys/values are projected for rows 0..T alone and
// (b) the last encoder layer evaluates queries, output projection and FFN for rows 0..T alone (its keys
// and values still cover all L rows).  Exact in real arithmetic; no summation is reordered.
//
// Precision policy with ADX_PREC_BF16: bf16 MFMA operands where the FLOPs are -- the map-token embedding and
// the encoder over the L conditioning rows.  Everything on the T-token trajectory path (the 4-wide input /
// prior embeddings, the whole decoder, the head) and the scheduler state stay fp32: they are <10 % of the work,
// and trajectory coordinates rounded to 8 mantissa bits would already cost decimetres.
#include "engine.h"

#include <stdlib.h>

#include <math.h>

using namespace adx;

namespace {

// What a sampling step hands to run_net so that the fused tail can also apply the scheduler update.
struct StepIO {
    int step = -1;                  // row of the schedule tables; < 0: plain denoiser call (no caches)
    const float* z = nullptr;       // noise of this step or nullptr
    float* x_next = nullptr;        // receives x_{t-1} when the tail kernel runs
};

struct EngineA final : adx_engine {
    int d, H, L, T, NP, Cin, Cout, Cp, nmaps, n_enc, n_dec, Lm;
    // workspace pointers
    float *X0 = nullptr, *XA = nullptr, *XB = nullptr, *H1 = nullptr, *QKV = nullptr, *ATT = nullptr, *FF = nullptr;
    float *Y = nullptr, *YN = nullptr, *DQKV = nullptr, *DATT = nullptr, *DFF = nullptr, *MKV = nullptr, *DQ = nullptr;
    float *XS0 = nullptr, *XS1 = nullptr, *MOUT = nullptr, *TRAN = nullptr, *PRIN = nullptr, *TRAW = nullptr, *PRAW = nullptr,
          *CHO = nullptr, *TE = nullptr, *ZBUF = nullptr;
    // elide level 2: layer-0 attention cache.  QKV0: in_proj(LN1(X0)) of every conditioning row, valid for a whole chunk
    // (rows >= 1 do not depend on the diffusion step); ST: softmax state of every query row >= 1 over key rows >= 1;
    // tqkv0: per-step q/k/v of the time token (identical for every batch item), [n_steps][3d].
    float *QKV0 = nullptr, *ST = nullptr, *PK = nullptr, *tqkv0 = nullptr;
    // fused path: tx0[s] = encoder input of the time token at step s; ATT0[s][b] = its layer-0 attention output
    float *tx0 = nullptr, *ATT0 = nullptr;
    // per-call map cache: states of the map queries over the map keys (the maps do not change between chunks)
    bool split_cache = false;
    float* STM = nullptr;
    float* KV1 = nullptr;      // fused path: layer-1 keys | values, [B*L][2d]
    bool fast4 = false;     // head_dim 4, 8 heads: the shapes the few-query / merge kernels are written for
    // elide level 3: the per-row pipeline between the two encoder attentions runs as one fused kernel
    bool fused = false;
    bool attn_g = false;       // fused path: the last encoder attention reads the stored LayerNorm rows (no K | V in memory)
    void* rows_image = nullptr;
    float* rows_vecs = nullptr;
    // fused tail (rest of the last encoder layer + decoder + head + scheduler update in one launch)
    bool tail_ok = false;
    float* tail_w = nullptr;
    TailArgs tail_tpl;
    bool tail_mfma_ok = false;         // the one-wavefront-per-item MFMA form applies (in_dim = out_dim = 4)
    int maps_batch = -1;               // batch size of the last sample() call that embedded maps_info (ADX_FLAG_KEEP_MAPS)
    int tail_split = 0;                // its maps run on split 16-bit operands (bf16 / fp16 precision modes) instead of fp32 MFMAs
    int tail_mem = 0, tail_traj = 0;   // arithmetic form of the MFMA tail's maps on memory rows / trajectory tokens (TailMfmaArgs::split / traj)
    void* tailm_image = nullptr;
    float* tailm_vecs = nullptr;
    // The decoder-side kernel (tail) of step k only depends on the encoder-side kernels of step k and on x_k, and the
    // encoder side does not depend on x at all: the tail of step k runs on a second stream under the encoder kernels of
    // step k + 1.  Its inputs (the last encoder layer's attention output and residual rows) are double-buffered by step parity.
    float *ATTb = nullptr;                    // second buffer of ATT (XB is the second buffer of XA)
    float *XC = nullptr, *XD = nullptr, *ATTc = nullptr, *ATTd = nullptr;      // third and fourth buffers of both (overlap depth 3 / 4)
    static constexpr int MAX_DEPTH = 4;
    int depth = 2;                            // steps the encoder kernels may run ahead of the tail (sets of per-step buffers in rotation; ADX_OVERLAP_DEPTH)
    float *GT = nullptr;                      // [n_steps][tiles][64] float4: the fold gates of every step of the chunk (gate-table form of the row kernel)
    bool gate_table = false;
    int mix = 0;               // products with both operands rounded once to fp16 (set at finalize): 1 | 16 | 32 the row kernel's second FFN product, out-projection and
                               // first FFN product, 2 the weighted sums of the last encoder attention, 4 its scores and V (with 2), 8 (with all of them and the gate table) fp16 o / l,
                               // gates and LayerNorm rows in memory
    float *Q1C = nullptr;                     // [depth][B][T + 1][d] layer-1 queries of the visible rows, by buffer set (LayerNorm-row attention)
    hipStream_t side_stream = nullptr, att_stream = nullptr;
    hipEvent_t ev_enc[MAX_DEPTH] = {}, ev_tail[MAX_DEPTH] = {}, ev_att[MAX_DEPTH] = {};
    int overlap_streams = 2;                  // 3: the last encoder attention of step k runs on a stream of its own, beside the row kernel of step k + 1
    bool overlap_ok = false;
    int overlap_max_b = 1536;

    ~EngineA() override {
        if (tqkv0) (void)hipFree(tqkv0);
        if (rows_image) (void)hipFree(rows_image);
        if (rows_vecs) (void)hipFree(rows_vecs);
        if (tail_w) (void)hipFree(tail_w);
        if (tx0) (void)hipFree(tx0);
        if (tailm_image) (void)hipFree(tailm_image);
        if (tailm_vecs) (void)hipFree(tailm_vecs);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (att_stream) (void)hipStreamDestroy(att_stream);
        for (hipEvent_t e : ev_att) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_enc) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_tail) if (e) (void)hipEventDestroy(e);
    }

    void on_workspace_realloc() override { maps_batch = -1; }     // the map embedding and its caches lived in the old workspace

    explicit EngineA(const adx_model_config& c) {
        cfg = c;
        d = c.d_model; H = c.n_head; T = c.horizon; NP = c.n_prior_tokens; Cin = c.in_dim; Cout = c.out_dim; Cp = c.prior_dim;
        nmaps = c.map_cond ? c.n_maps_token : 0;
        n_enc = c.n_enc_layers; n_dec = c.n_layers;
        L = 1 + nmaps + 2 * NP;
        Lm = c.elide ? T + 1 : L;
        fast4 = (H == 8 && d == 32);
        if (cfg.elide >= 2 && !fast4) cfg.elide = 1;
        fused = cfg.elide >= 3 && n_enc == 2;
        {
            const char* no_g = getenv("ADX_ATTN_G");              // fallback switch: 0 = few-query attention over stored K | V
            attn_g = fused && fast4 && attention_g_ok(T + 1) && !(no_g && no_g[0] == '0');
        }
        tail_ok = fused && T + 1 <= 12 && Cin == Cout && Cin <= 64 && (Cin % 4) == 0 && n_dec <= 4;
        const char* whole = getenv("ADX_CHUNK_CACHE_WHOLE");   // tuning knob: rebuild the whole layer-0 cache per chunk
        split_cache = fused && fast4 && cfg.map_cond && nmaps > 0 && 2 * NP <= 64 && !(whole && whole[0] == '1');
        const char* no_mfma = getenv("ADX_TAIL_VALU");       // tuning knob: force the VALU form of the tail
        tail_mfma_ok = fused && T + 1 <= 96 && Cin == 4 && Cout == 4 && n_dec <= 4 && !(no_mfma && no_mfma[0] == '1');
        {
            if (const char* dp = getenv("ADX_OVERLAP_DEPTH")) depth = std::min((int)MAX_DEPTH, std::max(2, atoi(dp)));      // tuning knob: sets of per-step buffers
            const char* f32_tail = getenv("ADX_TAIL_F32");       // tuning / fallback switch: fp32 MFMAs in every precision mode
            tail_split = cfg.precision != ADX_PREC_F32 && !(f32_tail && f32_tail[0] == '1');
        }
    }

    std::vector<ParamSpec> expected_params() const override {
        std::vector<ParamSpec> v;
        auto add = [&](const std::string& n, std::vector<int64_t> s) { v.push_back({n, std::move(s)}); };
        add("pos_emb", {1, T, d});
        add("cond_pos_emb", {1, L, d});
        add("input_emb.weight", {d, Cin}); add("input_emb.bias", {d});
        if (cfg.map_cond) { add("cond_obs_emb.weight", {d, cfg.maps_dim}); add("cond_obs_emb.bias", {d}); }
        add("prior_z_emb.weight", {d, Cp}); add("prior_z_emb.bias", {d});
        auto mha = [&](const std::string& p) {
            add(p + ".in_proj_weight", {3 * d, d}); add(p + ".in_proj_bias", {3 * d});
            add(p + ".out_proj.weight", {d, d}); add(p + ".out_proj.bias", {d});
        };
        auto ffn = [&](const std::string& p) {
            add(p + ".linear1.weight", {4 * d, d}); add(p + ".linear1.bias", {4 * d});
            add(p + ".linear2.weight", {d, 4 * d}); add(p + ".linear2.bias", {d});
        };
        auto ln = [&](const std::string& p) { add(p + ".weight", {d}); add(p + ".bias", {d}); };
        for (int i = 0; i < n_enc; ++i) {
            std::string p = "encoder.layers." + std::to_string(i);
            mha(p + ".self_attn"); ffn(p); ln(p + ".norm1"); ln(p + ".norm2");
        }
        for (int i = 0; i < n_dec; ++i) {
            std::string p = "decoder.layers." + std::to_string(i);
            mha(p + ".self_attn"); mha(p + ".multihead_attn"); ffn(p); ln(p + ".norm1"); ln(p + ".norm2"); ln(p + ".norm3");
        }
        ln("ln_f");
        add("head.weight", {Cout, d}); add("head.bias", {Cout});
        return v;
    }

    int carve_workspace(int B, bool dry, size_t* bytes) override {
        size_t off = 0;
        auto take = [&](float** p, size_t n) {
            if (!dry) *p = reinterpret_cast<float*>(ws + off);
            off += ((n * sizeof(float) + 255) / 256) * 256;
        };
        const size_t BL = (size_t)B * L, BT = (size_t)B * T;
        take(&X0, BL * d); take(&XA, BL * d); take(&XB, BL * d); take(&H1, BL * d);
        take(&QKV, BL * 3 * d); take(&ATT, BL * d); take(&FF, BL * 4 * d);
        take(&Y, BT * d); take(&YN, BT * d); take(&DQKV, BT * 3 * d); take(&DATT, BT * d); take(&DFF, BT * 4 * d);
        take(&MKV, (size_t)B * L * 2 * d); take(&DQ, BT * d);
        take(&XS0, BT * Cin); take(&XS1, BT * Cin); take(&MOUT, BT * Cout);
        take(&TRAN, (size_t)B * NP * Cp); take(&PRIN, (size_t)B * NP * Cp);
        take(&TRAW, (size_t)B * NP * Cp); take(&PRAW, (size_t)B * NP * Cp); take(&CHO, BT * Cin);
        take(&TE, (size_t)B * d);
        if (Cout != Cin) take(&ZBUF, BT * Cout);         // latent-carrying sampling: the latent the next chunk is conditioned on
        if (cfg.elide >= 2) { take(&QKV0, BL * 3 * d); take(&ST, BL * H * 6); }
        if (fused) {
            take(&PK, enc_rows_packed_bytes(B, L) / sizeof(float));
            take(&ATT0, (size_t)B * std::max<size_t>(steps.size(), 1) * d);
            if (split_cache) take(&STM, BL * H * 6);
            take(&KV1, std::max<int64_t>(BL * 2 * d, depth * (int64_t)(enc_rows_g_bytes(B, L) / sizeof(float))));      // K | V rows, or the sets of LayerNorm rows
            take(&Q1C, depth * (int64_t)B * (T + 1) * d);
            {
                // the gate table costs 32 B per row and step: used while it stays under a budget (long DDPM schedules at large
                // batches fall back to evaluating the gates inside the row kernel)
                const char* mb = getenv("ADX_GATE_TABLE_MAX_MB");      // tuning knob; 0 turns the table off
                const size_t budget = (size_t)(mb ? atoll(mb) : 16384) << 20;
                const size_t need = steps.empty() ? 0 : enc_rows_gates_bytes(B, L, (int)steps.size());
                gate_table = attn_g && need > 0 && need <= budget;
                if (gate_table) take(&GT, (int64_t)(need / sizeof(float)));
            }
            take(&ATTb, BL * d);
            if (depth >= 3) { take(&XC, BL * d); take(&ATTc, BL * d); }
            if (depth >= 4) { take(&XD, BL * d); take(&ATTd, BL * d); }
        }
        *bytes = off;
        return ADX_OK;
    }

    int build_time_table(hipStream_t s) override {
        const int n = (int)steps.size();
        ADX_HIP(timestep_embedding(steps_t_dev, n, d, temb_table, d, s));
        if (cfg.elide >= 2) {
            // q/k/v of the time token under encoder layer 0, one row per step
            if (tqkv0) (void)hipFree(tqkv0);
            tqkv0 = nullptr;
            if (tx0) (void)hipFree(tx0);
            tx0 = nullptr;
            float* tmp = nullptr;
            ADX_HIP(hipMalloc(&tqkv0, sizeof(float) * (size_t)n * 3 * d));
            ADX_HIP(hipMalloc(&tx0, sizeof(float) * (size_t)n * d));
            ADX_HIP(hipMalloc(&tmp, sizeof(float) * (size_t)n * d));
            const std::string p = "encoder.layers.0";
            hipError_t e = add_vec_act(temb_table, P("cond_pos_emb"), tx0, n, d, 0, ACT_NONE, s);
            int rc = ADX_OK;
            if (e == hipSuccess) rc = ln_rows(tx0, tmp, p + ".norm1", n, RowMap{}, s);
            if (e == hipSuccess && rc == ADX_OK)
                e = linear(lin(tmp, d, p + ".self_attn.in_proj_weight", 0, 3 * d, d, p + ".self_attn.in_proj_bias", tqkv0, 3 * d, n), s);
            (void)hipStreamSynchronize(s);
            (void)hipFree(tmp);
            if (e != hipSuccess) return hip_fail(e, "build_time_table");
            if (rc != ADX_OK) return rc;
        }
        if (fused) {
            if (!rows_image) ADX_HIP(hipMalloc(&rows_image, enc_rows_image_bytes()));
            if (!rows_vecs) ADX_HIP(hipMalloc(&rows_vecs, sizeof(float) * 416));
            const std::string p0 = "encoder.layers.0", p1 = "encoder.layers.1";
            // engines created with ADX_PREC_F16 (cfg.precision reads ADX_PREC_BF16X3): every product of the two per-step kernels that
            // walk all conditioning rows with both operands rounded once to fp16.  ADX_A_MIX (measurement knob, read here) picks a subset.
            mix = f16_single ? 63 : 0;
            if (const char* mx = getenv("ADX_A_MIX")) { if (cfg.precision == ADX_PREC_BF16X3) mix = atoi(mx); }
            if ((mix & 0x31) && (mix & 0x31) != 0x31) mix &= ~0x31;       // the row kernel has one mixed instance: all three products or none
            ADX_HIP(enc_rows_build_image(P(p0 + ".self_attn.out_proj.weight"), P(p0 + ".linear1.weight"), P(p0 + ".linear2.weight"),
                                         P(p1 + ".self_attn.in_proj_weight") + (int64_t)d * d, P(p1 + ".self_attn.in_proj_weight"),
                                         rows_image, s, ((mix & 1) ? 4 : 0) | ((mix & 16) ? 1 : 0) | ((mix & 32) ? 2 : 0)));
            struct { const float* src; int off, n; } parts[] = {
                {P(p0 + ".self_attn.out_proj.bias"), 0, 32}, {P(p0 + ".norm2.weight"), 32, 32}, {P(p0 + ".norm2.bias"), 64, 32},
                {P(p0 + ".linear2.bias"), 96, 32}, {P(p1 + ".norm1.weight"), 128, 32}, {P(p1 + ".norm1.bias"), 160, 32},
                {P(p0 + ".linear1.bias"), 192, 128}, {P(p1 + ".self_attn.in_proj_bias") + d, 320, 64},
                {P(p1 + ".self_attn.in_proj_bias"), 384, 32}};
            for (const auto& q : parts)
                ADX_HIP(hipMemcpyAsync(rows_vecs + q.off, q.src, sizeof(float) * q.n, hipMemcpyDeviceToDevice, s));
            ADX_HIP(hipStreamSynchronize(s));
        }
        if (tail_ok) ADX_TRY(build_tail_pack(s));
        if (tail_mfma_ok) ADX_TRY(build_tail_mfma(s));
        return ADX_OK;
    }

    // Weight-fragment image and vector table of the MFMA tail.
    int build_tail_mfma(hipStream_t s) {
        if (!tailm_image) ADX_HIP(hipMalloc(&tailm_image, tail_mfma_image_bytes(n_dec)));
        if (!tailm_vecs) ADX_HIP(hipMalloc(&tailm_vecs, sizeof(float) * tail_mfma_vec_floats()));
        ADX_HIP(hipMemsetAsync(tailm_image, 0, tail_mfma_image_bytes(n_dec), s));
        ADX_HIP(hipMemsetAsync(tailm_vecs, 0, sizeof(float) * tail_mfma_vec_floats(), s));
        const TailMfmaVecLayout v = tail_mfma_vec_layout();
        int blk = 0;
        // Forms (0 fp32, 1 split-bf16, 2 split-fp16): the fp32 engine (0, 0); the bf16 and fp16 engines (2, 2).
        // ADX_TAIL_FORMS = "<memory><trajectory>" (two digits, tuning knob, read here) picks another supported pair: "10" is memory
        // rows split-bf16 + trajectory tokens fp32, "20" the most accurate of the reduced forms.
        tail_mem = tail_traj = tail_split ? 2 : 0;
        if (const char* tf = getenv("ADX_TAIL_FORMS")) {
            const int m = tf[0] - '0', t = tf[0] ? tf[1] - '0' : -1;
            const bool supported = (m == 0 && t == 0) || (m == 1 && t == 0) || (m == 2 && t == 0) || (m == 2 && t == 2);      // the instances tail_mfma() has
            if (tail_split && supported) { tail_mem = m; tail_traj = t; }
        }
        bool mem_side = true;              // blocks that act on memory rows
        auto block = [&](const std::string& w, int ldw, int row0, int col0) {
            return tail_mfma_pack_block(P(w), ldw, row0, col0, tailm_image, blk++, mem_side ? tail_mem : tail_traj, s);
        };
        auto vec = [&](const float* src, int off, int n) {
            return hipMemcpyAsync(tailm_vecs + off, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s);
        };
        auto ffn = [&](const std::string& p) {
            for (int i = 0; i < 4; ++i) {
                hipError_t e = block(p + ".linear1.weight", d, 32 * i, 0);
                if (e == hipSuccess) e = block(p + ".linear2.weight", 4 * d, 0, 32 * i);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        const std::string e1 = "encoder.layers." + std::to_string(n_enc - 1);
        ADX_HIP(block(e1 + ".self_attn.out_proj.weight", d, 0, 0));
        ADX_HIP(ffn(e1));
        ADX_HIP(vec(P(e1 + ".self_attn.out_proj.bias"), v.e_bo, d));
        ADX_HIP(vec(P(e1 + ".norm2.weight"), v.e_ln2_g, d)); ADX_HIP(vec(P(e1 + ".norm2.bias"), v.e_ln2_b, d));
        ADX_HIP(vec(P(e1 + ".linear1.bias"), v.e_b1, 4 * d)); ADX_HIP(vec(P(e1 + ".linear2.bias"), v.e_b2, d));
        ADX_HIP(vec(P("input_emb.bias"), v.din_b, d));
        ADX_HIP(vec(P("pos_emb"), v.pos, std::min(T, 16) * d));
        for (int l = 0; l < n_dec; ++l) {
            const std::string p = "decoder.layers." + std::to_string(l);
            const int o = v.layer0 + l * v.layer_size;
            mem_side = false;
            for (int i = 0; i < 3; ++i) ADX_HIP(block(p + ".self_attn.in_proj_weight", d, 32 * i, 0));
            ADX_HIP(block(p + ".self_attn.out_proj.weight", d, 0, 0));
            for (int i = 0; i < 3; ++i) {
                mem_side = i > 0;          // query from the trajectory tokens, keys / values from the memory rows
                ADX_HIP(block(p + ".multihead_attn.in_proj_weight", d, 32 * i, 0));
            }
            mem_side = false;
            ADX_HIP(block(p + ".multihead_attn.out_proj.weight", d, 0, 0));
            ADX_HIP(ffn(p));
            ADX_HIP(vec(P(p + ".norm1.weight"), o + v.ln1_g, d)); ADX_HIP(vec(P(p + ".norm1.bias"), o + v.ln1_b, d));
            ADX_HIP(vec(P(p + ".norm2.weight"), o + v.ln2_g, d)); ADX_HIP(vec(P(p + ".norm2.bias"), o + v.ln2_b, d));
            ADX_HIP(vec(P(p + ".norm3.weight"), o + v.ln3_g, d)); ADX_HIP(vec(P(p + ".norm3.bias"), o + v.ln3_b, d));
            ADX_HIP(vec(P(p + ".self_attn.in_proj_bias"), o + v.bqkv, 3 * d));
            ADX_HIP(vec(P(p + ".self_attn.out_proj.bias"), o + v.bo, d));
            ADX_HIP(vec(P(p + ".multihead_attn.in_proj_bias"), o + v.bq, d));
            ADX_HIP(vec(P(p + ".multihead_attn.in_proj_bias") + d, o + v.bkv, 2 * d));
            ADX_HIP(vec(P(p + ".multihead_attn.out_proj.bias"), o + v.bo2, d));
            ADX_HIP(vec(P(p + ".linear1.bias"), o + v.b1, 4 * d)); ADX_HIP(vec(P(p + ".linear2.bias"), o + v.b2, d));
        }
        ADX_HIP(vec(P("ln_f.weight"), v.lnf_g, d)); ADX_HIP(vec(P("ln_f.bias"), v.lnf_b, d));
        ADX_HIP(vec(P("head.bias"), v.head_b, Cout));
        ADX_HIP(tail_mfma_pack_ends(P("input_emb.weight"), P("head.weight"), tailm_image, s));
        if (blk != tail_mfma_blocks(n_dec)) { set_error("build_tail_mfma: block count mismatch"); return ADX_ESTATE; }
        ADX_HIP(hipStreamSynchronize(s));
        return ADX_OK;
    }

    float* x1_buf(int par) const { return par == 0 ? XA : par == 1 ? XB : par == 2 ? XC : XD; }
    float* att_buf(int par) const { return par == 0 ? ATT : par == 1 ? ATTb : par == 2 ? ATTc : ATTd; }

    // Rest of the last encoder layer + decoder + head + scheduler update, one launch.
    int launch_tail(int B, const float* x, const float* z, const StepCoef* coef, float* x_next, float* out, hipStream_t s, int par = 0) {
        if (tail_mfma_ok) {
            TailMfmaArgs t;
            t.image = tailm_image; t.vecs = tailm_vecs; t.att1 = att_buf(par); t.x1 = x1_buf(par); t.x = x; t.z = z; t.coef = coef;
            t.x_next = x_next; t.model_out = out; t.B = B; t.L = L; t.T = T; t.n_dec = n_dec; t.split = tail_mem; t.traj = tail_traj; t.prio = tail_prio(B);
            t.pos = P("pos_emb");
            ADX_HIP(tail_mfma(t, s));
            return ADX_OK;
        }
        TailArgs t = tail_tpl;
        t.w = tail_w; t.att1 = att_buf(par); t.x1 = x1_buf(par); t.x = x; t.z = z; t.coef = coef;
        t.x_next = x_next; t.model_out = out;
        t.B = B; t.L = L; t.T = T; t.C = Cin; t.n_dec = n_dec;
        ADX_HIP(tail_a(t, s));
        return ADX_OK;
    }

    // Pack every matrix / vector the fused tail reads into one float buffer and record the offsets.
    int build_tail_pack(hipStream_t s) {
        std::vector<std::pair<const float*, int>> vec_copies;      // (src, n) in buffer order
        struct MatJob { const float* W; int N, K, off; };
        std::vector<MatJob> mats;
        int off = 0;
        std::vector<int> vec_dst;
        auto vec = [&](const float* src, int n) { int o = off; vec_copies.push_back({src, n}); vec_dst.push_back(o); off += (n + 3) / 4 * 4; return o; };
        auto mat = [&](const std::string& wname, int row0, int N, int K, const std::string& bname) {
            TailMat m;
            m.N = N; m.K = K;
            m.b_off = vec(P(bname) + row0, N);
            m.w_off = off;
            mats.push_back({P(wname) + (int64_t)row0 * K, N, K, off});
            off += N * K;
            return m;
        };
        TailArgs t;
        const std::string e1 = "encoder.layers." + std::to_string(n_enc - 1);
        t.e_wo = mat(e1 + ".self_attn.out_proj.weight", 0, d, d, e1 + ".self_attn.out_proj.bias");
        t.e_ln2_g = vec(P(e1 + ".norm2.weight"), d); t.e_ln2_b = vec(P(e1 + ".norm2.bias"), d);
        t.e_w1 = mat(e1 + ".linear1.weight", 0, 4 * d, d, e1 + ".linear1.bias");
        t.e_w2 = mat(e1 + ".linear2.weight", 0, d, 4 * d, e1 + ".linear2.bias");
        t.d_in = mat("input_emb.weight", 0, d, Cin, "input_emb.bias");
        t.pos_off = vec(P("pos_emb"), T * d);
        for (int i = 0; i < n_dec; ++i) {
            const std::string p = "decoder.layers." + std::to_string(i);
            TailLayer& l = t.dec[i];
            l.ln1_g = vec(P(p + ".norm1.weight"), d); l.ln1_b = vec(P(p + ".norm1.bias"), d);
            l.ln2_g = vec(P(p + ".norm2.weight"), d); l.ln2_b = vec(P(p + ".norm2.bias"), d);
            l.ln3_g = vec(P(p + ".norm3.weight"), d); l.ln3_b = vec(P(p + ".norm3.bias"), d);
            l.wqkv = mat(p + ".self_attn.in_proj_weight", 0, 3 * d, d, p + ".self_attn.in_proj_bias");
            l.wo = mat(p + ".self_attn.out_proj.weight", 0, d, d, p + ".self_attn.out_proj.bias");
            l.wq = mat(p + ".multihead_attn.in_proj_weight", 0, d, d, p + ".multihead_attn.in_proj_bias");
            l.wkv = mat(p + ".multihead_attn.in_proj_weight", d, 2 * d, d, p + ".multihead_attn.in_proj_bias");
            l.wo2 = mat(p + ".multihead_attn.out_proj.weight", 0, d, d, p + ".multihead_attn.out_proj.bias");
            l.w1 = mat(p + ".linear1.weight", 0, 4 * d, d, p + ".linear1.bias");
            l.w2 = mat(p + ".linear2.weight", 0, d, 4 * d, p + ".linear2.bias");
        }
        t.lnf_g = vec(P("ln_f.weight"), d); t.lnf_b = vec(P("ln_f.bias"), d);
        t.d_head = mat("head.weight", 0, Cout, d, "head.bias");
        if (tail_w) (void)hipFree(tail_w);
        tail_w = nullptr;
        ADX_HIP(hipMalloc(&tail_w, sizeof(float) * (size_t)off));
        ADX_HIP(hipMemsetAsync(tail_w, 0, sizeof(float) * (size_t)off, s));
        for (const MatJob& j : mats) ADX_HIP(tail_pack_matrix(j.W, j.N, j.K, tail_w + j.off, s));
        for (size_t i = 0; i < vec_copies.size(); ++i)
            ADX_HIP(hipMemcpyAsync(tail_w + vec_dst[i], vec_copies[i].first, sizeof(float) * vec_copies[i].second, hipMemcpyDeviceToDevice, s));
        ADX_HIP(hipStreamSynchronize(s));
        tail_tpl = t;
        return ADX_OK;
    }

    // Layer-0 q/k/v of `rows` conditioning rows per item starting at row `row0` (LayerNorm + in_proj)
    int project_rows0(int B, int row0, int rows, hipStream_t s) {
        const std::string p = "encoder.layers.0";
        const RowMap map{rows, L, row0};
        ADX_TRY(ln_rows(X0, H1, p + ".norm1", B * rows, map, s));
        LinearArgs a = lin(H1, d, p + ".self_attn.in_proj_weight", 0, 3 * d, d, p + ".self_attn.in_proj_bias", QKV0, 3 * d, B * rows);
        a.xmap = map; a.ymap = map;
        ADX_HIP(linear(a, s));
        return ADX_OK;
    }

    // Once per call (the map tokens do not change between chunks): their layer-0 q/k/v and the softmax state of every map
    // query over all map keys.
    int build_map_cache(int B, hipStream_t s) {
        if (!split_cache) return ADX_OK;
        const int dh = d / H;
        ADX_TRY(project_rows0(B, 1, nmaps, s));
        AttentionArgs at;
        at.Q = QKV0 + 3 * d; at.K = QKV0 + 3 * d + d; at.V = QKV0 + 3 * d + 2 * d;
        at.ldq = at.ldk = at.ldv = 3 * d;
        at.q_batch_stride = at.k_batch_stride = at.v_batch_stride = (int64_t)L * 3 * d;
        at.B = B; at.H = H; at.Sq = nmaps; at.Skv = nmaps; at.dh = dh; at.dv = dh; at.scale = 1.0f / sqrtf((float)dh); at.mask_off = -1;
        at.state = STM + H * 6; at.state_ld = H * 6; at.state_batch_stride = (int64_t)L * H * 6;
        ADX_HIP(attention(at, s));
        return ADX_OK;
    }

    // Per chunk: q/k/v of the conditioning rows under layer 0 and the softmax state of rows >= 1 over keys >= 1.
    // With the per-call map cache only the 2 NP transition / prior rows are new: their keys are folded into the map
    // queries' states and their own states are computed over all keys.
    int build_chunk_cache(int B, hipStream_t s) {
        const int dh = d / H;
        const float scale = 1.0f / sqrtf((float)dh);
        if (split_cache) {
            const int r0 = 1 + nmaps, nr = 2 * NP;
            ADX_TRY(project_rows0(B, r0, nr, s));
            FoldKeysArgs f;
            f.Q = QKV0 + 3 * d; f.ldq = 3 * d; f.q_batch_stride = (int64_t)L * 3 * d;
            f.K = QKV0 + (int64_t)r0 * 3 * d + d; f.V = QKV0 + (int64_t)r0 * 3 * d + 2 * d; f.ldk = 3 * d; f.k_batch_stride = (int64_t)L * 3 * d;
            f.state_in = STM + H * 6; f.state_out = ST + H * 6; f.state_ld = H * 6; f.state_batch_stride = (int64_t)L * H * 6;
            f.B = B; f.Sq = nmaps; f.NK = nr; f.scale = scale;
            ADX_HIP(attention_fold_keys(f, s));
            RowsAttnArgs ra;
            ra.Q = QKV0 + (int64_t)r0 * 3 * d; ra.ldq = 3 * d; ra.q_batch_stride = (int64_t)L * 3 * d; ra.own = 0;
            ra.K = QKV0 + 3 * d + d; ra.V = QKV0 + 3 * d + 2 * d; ra.ldk = 3 * d; ra.k_batch_stride = (int64_t)L * 3 * d;
            ra.state = ST + (int64_t)r0 * H * 6; ra.state_ld = H * 6; ra.state_batch_stride = (int64_t)L * H * 6;
            ra.B = B; ra.NS = nr; ra.Skv = L - 1; ra.scale = scale;
            ADX_HIP(attention_rows(ra, s));
        } else {
            ADX_TRY(project_rows0(B, 0, L, s));
            AttentionArgs at;
            at.Q = QKV0 + 3 * d; at.K = QKV0 + 3 * d + d; at.V = QKV0 + 3 * d + 2 * d;
            at.ldq = at.ldk = at.ldv = 3 * d;
            at.q_batch_stride = at.k_batch_stride = at.v_batch_stride = (int64_t)L * 3 * d;
            at.B = B; at.H = H; at.Sq = L - 1; at.Skv = L - 1; at.dh = dh; at.dv = dh; at.scale = scale; at.mask_off = -1;
            at.state = ST + H * 6; at.state_ld = H * 6; at.state_batch_stride = (int64_t)L * H * 6;
            ADX_HIP(attention(at, s));
        }
        if (fused) {
            if (gate_table) ADX_HIP(enc_rows_pack_gates(rows_args(B, 0), reinterpret_cast<float4*>(GT), tqkv0, (int)steps.size(), s, half_io()));
            else ADX_HIP(enc_rows_pack(rows_args(B, 0), s));
            // the time token's attention for every step of the chunk in one pass over the keys
            RowsAttnArgs tr;
            tr.Q = tqkv0; tr.ldq = 3 * d; tr.q_batch_stride = 0; tr.own = 1;
            tr.K = QKV0 + 3 * d + d; tr.V = QKV0 + 3 * d + 2 * d; tr.ldk = 3 * d; tr.k_batch_stride = (int64_t)L * 3 * d;
            tr.O = ATT0; tr.B = B; tr.NS = (int)steps.size(); tr.Skv = L - 1; tr.scale = scale;
            ADX_HIP(attention_rows(tr, s));
        }
        return ADX_OK;
    }

    float4* g_rows(int B, int par) const { return reinterpret_cast<float4*>(KV1) + (int64_t)par * (int64_t)(enc_rows_g_bytes(B, L) / sizeof(float4)); }

    // Wave priorities (s_setprio) of the two kernels of a step that walk all rows (ADX_A_PRIO) and of the tail beside them on the second
    // stream (ADX_TAIL_PRIO; tuning knobs, read per call).  Which side should issue first depends on which one closes the step's critical
    // cycle tail(k - 2) -> row kernel(k) -> attention(k) -> tail(k) (profiles/step_gaps.py).  fp16 engine, ms per generate with the row
    // kernels / the tail favoured: 256 scenarios 27.9 / 23.7, 512: 38.1 / 35.1, 1024: 52.6 / 50.3, 2048: 98.0 / 100.0 -- up to the batch
    // at which the row kernels alone fill the chip the tail's waves go first, above it the row kernels'.  Split-bf16 engine (its row
    // kernels are a third longer and leave the tail no slack): no priorities (1024 scenarios: 77.7, tail favoured 79.9, rows favoured 82.8).
    bool tail_first(int B) const { return mix != 0 && B <= 1280; }       // 1536 scenarios: 79.5 ms with the tail favoured, 76.5 with the row kernels
    int step_prio(int B) const { const char* e = getenv("ADX_A_PRIO"); return e ? atoi(e) : (mix && !tail_first(B) ? 3 : 0); }
    int tail_prio(int B) const { const char* e = getenv("ADX_TAIL_PRIO"); return e ? atoi(e) : (tail_first(B) ? 3 : 0); }
    bool half_io() const { return (mix & 8) && (mix & 0x31) == 0x31 && (mix & 6) == 6 && gate_table && attn_g; }
    int rows_mix() const { return ((mix & 0x31) == 0x31 ? 7 : 0) | (half_io() ? 8 : 0); }

    EncRowsArgs rows_args(int B, int step, int par = 0) const {
        EncRowsArgs r;
        r.x0 = X0; r.q = QKV0; r.ldq = 3 * d; r.state = ST; r.packed = reinterpret_cast<const float4*>(PK);
        r.att0 = ATT0 + (int64_t)step * B * d;     // [step][B][d] for the batch of this call
        r.x0_time = tx0 + (int64_t)step * d;
        r.key0 = tqkv0 + (int64_t)step * 3 * d + d;
        r.image = rows_image; r.vecs = rows_vecs;
        r.kv = KV1; r.ldkv = 2 * d; r.x1 = x1_buf(par); r.q1 = QKV; r.ldq1 = 3 * d;     // layer-1 K | V compact: both kernels stream whole rows
        if (attn_g) {       // the LayerNorm rows (128 B per row) take the place of K | V (256 B); they and the queries are double-buffered by step parity
            r.gpk = g_rows(B, par);
            if (gate_table) r.gates = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(GT) + (int64_t)step * (int64_t)(enc_rows_gates_bytes(B, L, 1) / (half_io() ? 2 : 1)));
            r.q1 = Q1C + (int64_t)par * B * (T + 1) * d; r.ldq1 = d; r.q1_item_stride = (int64_t)(T + 1) * d;
        }
        r.B = B; r.L = L; r.n_keep = T + 1; r.scale = 1.0f / sqrtf((float)(d / H));
        r.prio = step_prio(B);
        return r;
    }

    // ---------------------------------------------------------------------------------- conditioning rows
    int embed_maps(int B, const float* maps, hipStream_t s) {
        if (!cfg.map_cond) return ADX_OK;
        LinearArgs a = lin(maps, cfg.maps_dim, "cond_obs_emb.weight", 0, d, cfg.maps_dim, "cond_obs_emb.bias", X0, d, B * nmaps);
        a.ymap = RowMap{nmaps, L, 1};
        a.pos = P("cond_pos_emb"); a.ldp = d; a.pos_period = nmaps; a.pos_off = 1;
        ADX_HIP(linear(a, s));
        return ADX_OK;
    }

    int embed_prior(int B, const float* tran_n, const float* prior_n, hipStream_t s) {
        for (int which = 0; which < 2; ++which) {
            const int off = 1 + nmaps + which * NP;
            LinearArgs a = lin(which ? prior_n : tran_n, Cp, "prior_z_emb.weight", 0, d, Cp, "prior_z_emb.bias", X0, d, B * NP, ACT_NONE, PREC_F32);
            a.ymap = RowMap{NP, L, off};
            a.pos = P("cond_pos_emb"); a.ldp = d; a.pos_period = NP; a.pos_off = off;
            ADX_HIP(linear(a, s));
        }
        return ADX_OK;
    }

    // time rows: X0[b, 0, :] = te[b * te_stride : ] + cond_pos[0]
    int set_time(int B, const float* te, int64_t te_stride, hipStream_t s) {
        ADX_HIP(set_row(X0, (int64_t)L * d, B, te, te_stride, P("cond_pos_emb"), d, s));
        return ADX_OK;
    }

    int ln_rows(const float* x, float* y, const std::string& p, int rows, RowMap map, hipStream_t s) {
        LayerNormArgs a;
        a.x = x; a.ldx = d; a.xmap = map; a.y = y; a.ldy = d; a.ymap = map;
        a.gamma = P(p + ".weight"); a.beta = P(p + ".bias"); a.eps = 1e-5f; a.rows = rows; a.d = d;
        ADX_HIP(layernorm(a, s));
        return ADX_OK;
    }

    // ---------------------------------------------------------------------------------- the network
    // last encoder layer on the fused path: queries of the T+1 visible rows (in QKV) against the compact K | V rows
    // last encoder layer's attention on the fused path: from the stored LayerNorm rows when the visible rows fit the
    // matrix-core form (keys and values never reach memory), else few-query attention over the stored K | V
    int last_attention(int B, int par, hipStream_t s) {
        if (attn_g) {
            AttnGArgs g;
            g.gpk = g_rows(B, par); g.image = rows_image; g.vecs = rows_vecs;
            g.wk = P("encoder.layers.1.self_attn.in_proj_weight") + (int64_t)d * d;
            g.q1 = Q1C + (int64_t)par * B * (T + 1) * d; g.ldq1 = d; g.q1_item_stride = (int64_t)(T + 1) * d;
            g.O = att_buf(par); g.ldo = d; g.o_batch_stride = (int64_t)L * d;
            g.B = B; g.L = L; g.n_keep = T + 1; g.scale = 1.0f / sqrtf((float)(d / H));
            g.prio = step_prio(B);
            if (const char* lz = getenv("ADX_ATTN_LAZY")) g.lazy = (float)atof(lz);      // tuning knob, read per call
            ADX_HIP(attention_g(g, cfg.precision, s, ((mix >> 1) & 3) | (half_io() ? 4 : 0)));
        } else {
            ADX_HIP(attention_few_queries(fused_attn_args(B, par), s));
        }
        return ADX_OK;
    }

    AttentionArgs fused_attn_args(int B, int par = 0) const {
        AttentionArgs at = enc_attn_args(B, T + 1);
        if (par) at.O = att_buf(par);
        at.K = KV1; at.V = KV1 + d; at.ldk = at.ldv = 2 * d; at.k_batch_stride = at.v_batch_stride = (int64_t)L * 2 * d;
        return at;
    }

    AttentionArgs enc_attn_args(int B, int rq) const {
        const int dh = d / H;
        AttentionArgs at;
        at.Q = QKV; at.ldq = 3 * d; at.q_batch_stride = (int64_t)L * 3 * d;
        at.K = QKV + d; at.ldk = 3 * d; at.k_batch_stride = (int64_t)L * 3 * d;
        at.V = QKV + 2 * d; at.ldv = 3 * d; at.v_batch_stride = (int64_t)L * 3 * d;
        at.O = ATT; at.ldo = d; at.o_batch_stride = (int64_t)L * d;
        at.B = B; at.H = H; at.Sq = rq; at.Skv = L; at.dh = dh; at.dv = dh; at.scale = 1.0f / sqrtf((float)dh); at.mask_off = -1;
        return at;
    }

    // attention of the time token (row 0) under layer 0, against the cached keys plus its own per-step key
    int time_token_attention(int B, int step, hipStream_t s) {
        const int dh = d / H;
        const float* tq = tqkv0 + (int64_t)step * 3 * d;
        AttentionArgs a0;
        a0.Q = tq; a0.ldq = 3 * d; a0.q_batch_stride = 0;
        a0.K = QKV0 + d; a0.V = QKV0 + 2 * d; a0.ldk = a0.ldv = 3 * d; a0.k_batch_stride = a0.v_batch_stride = (int64_t)L * 3 * d;
        a0.O = ATT; a0.ldo = d; a0.o_batch_stride = (int64_t)L * d;
        a0.B = B; a0.H = H; a0.Sq = 1; a0.Skv = L; a0.dh = dh; a0.dv = dh; a0.scale = 1.0f / sqrtf((float)dh); a0.key0 = tq + d;
        ADX_HIP(attention_few_queries(a0, s));
        return ADX_OK;
    }

    // X0 must hold the conditioning rows; x is (B,T,Cin); out is (B,T,Cout).
    // Returns ADX_OK; *stepped is set when the scheduler update was applied too (fused tail).
    int run_net(int B, const float* x, float* out, hipStream_t s, const StepIO& io = StepIO(), bool* stepped = nullptr) {
        const int dh = d / H;
        const float scale = 1.0f / sqrtf((float)dh);
        const int step = io.step;
        if (stepped) *stepped = false;
        const float* cur = X0;
        int first_layer = 0;
        bool last_attn_done = false;      // ATT already holds the last encoder layer's attention for rows 0..T
        if (step >= 0 && fused) {
            // layer 0 and layer 1 up to its attention: three launches
            prof_begin(1, s);
            ADX_HIP(enc_rows(rows_args(B, step), cfg.precision, s, rows_mix()));
            prof_end(s);
            prof_begin(2, s);
            ADX_TRY(last_attention(B, 0, s));
            prof_end(s);
            if ((tail_ok || tail_mfma_ok) && io.x_next && stepped) {
                prof_begin(3, s);
                ADX_TRY(launch_tail(B, x, io.z, steps_dev + step, io.x_next, out, s));
                prof_end(s);
                *stepped = true;
                return ADX_OK;
            }
            cur = XA;
            first_layer = 1;
            last_attn_done = true;
        }
        // ---- encoder (pre-norm, GELU)
        for (int i = first_layer; i < n_enc; ++i) {
            const std::string p = "encoder.layers." + std::to_string(i);
            float* nxt = (i & 1) ? XB : XA;
            const bool last = cfg.elide && i == n_enc - 1;
            const int rq = last ? T + 1 : L;                      // query rows per batch item
            const RowMap qmap = last ? RowMap{rq, L, 0} : RowMap{};
            if (i == 0 && step >= 0 && cfg.elide >= 2 && !last) {
                // layer-0 attention from the chunk cache: only the time token (row 0) changed since the cache was built
                ADX_TRY(time_token_attention(B, step, s));
                const float* tq = tqkv0 + (int64_t)step * 3 * d;
                AttentionArgs a1;
                a1.Q = QKV0 + 3 * d; a1.ldq = 3 * d; a1.q_batch_stride = (int64_t)L * 3 * d;
                a1.O = ATT + d; a1.ldo = d; a1.o_batch_stride = (int64_t)L * d;
                a1.B = B; a1.H = H; a1.Sq = L - 1; a1.dh = dh; a1.dv = dh; a1.scale = scale; a1.key0 = tq + d;
                a1.state = ST + H * 6; a1.state_ld = H * 6; a1.state_batch_stride = (int64_t)L * H * 6;
                ADX_HIP(attention_merge_key(a1, s));
            } else if (!(last && last_attn_done)) {
                ADX_TRY(ln_rows(cur, H1, p + ".norm1", B * L, RowMap{}, s));
                if (!last) {
                    ADX_HIP(linear(lin(H1, d, p + ".self_attn.in_proj_weight", 0, 3 * d, d, p + ".self_attn.in_proj_bias", QKV, 3 * d, B * L), s));
                } else {
                    LinearArgs kv = lin(H1, d, p + ".self_attn.in_proj_weight", d, 2 * d, d, p + ".self_attn.in_proj_bias", QKV + d, 3 * d, B * L);
                    ADX_HIP(linear(kv, s));
                    LinearArgs q = lin(H1, d, p + ".self_attn.in_proj_weight", 0, d, d, p + ".self_attn.in_proj_bias", QKV, 3 * d, B * rq);
                    q.xmap = qmap; q.ymap = qmap;
                    ADX_HIP(linear(q, s));
                }
                const AttentionArgs at = enc_attn_args(B, rq);
                if (last && fast4) ADX_HIP(attention_few_queries(at, s));
                else ADX_HIP(attention(at, s));
            }
            LinearArgs o = lin(ATT, d, p + ".self_attn.out_proj.weight", 0, d, d, p + ".self_attn.out_proj.bias", nxt, d, B * rq);
            o.xmap = qmap; o.ymap = qmap; o.res = cur; o.ldr = d;
            ADX_HIP(linear(o, s));
            ADX_TRY(ln_rows(nxt, H1, p + ".norm2", B * rq, qmap, s));
            LinearArgs f1 = lin(H1, d, p + ".linear1.weight", 0, 4 * d, d, p + ".linear1.bias", FF, 4 * d, B * rq, ACT_GELU_ERF);
            f1.xmap = qmap; f1.ymap = qmap;
            ADX_HIP(linear(f1, s));
            LinearArgs f2 = lin(FF, 4 * d, p + ".linear2.weight", 0, d, 4 * d, p + ".linear2.bias", nxt, d, B * rq);
            f2.xmap = qmap; f2.ymap = qmap; f2.res = nxt; f2.ldr = d;
            ADX_HIP(linear(f2, s));
            cur = nxt;
        }
        const float* memory = cur;   // (B, L, d); with elide only rows 0..T of each item are defined
        // ---- decoder (pre-norm, GELU): causal self-attention, masked cross-attention, FFN.  Always fp32.
        {
            LinearArgs e = lin(x, Cin, "input_emb.weight", 0, d, Cin, "input_emb.bias", Y, d, B * T, ACT_NONE, PREC_F32);
            e.pos = P("pos_emb"); e.ldp = d; e.pos_period = T; e.pos_off = 0;
            ADX_HIP(linear(e, s));
        }
        const RowMap mmap = RowMap{Lm, L, 0};
        for (int i = 0; i < n_dec; ++i) {
            const std::string p = "decoder.layers." + std::to_string(i);
            ADX_TRY(ln_rows(Y, YN, p + ".norm1", B * T, RowMap{}, s));
            ADX_HIP(linear(lin(YN, d, p + ".self_attn.in_proj_weight", 0, 3 * d, d, p + ".self_attn.in_proj_bias", DQKV, 3 * d, B * T, ACT_NONE, PREC_F32), s));
            AttentionArgs sa;
            sa.Q = DQKV; sa.K = DQKV + d; sa.V = DQKV + 2 * d;
            sa.ldq = sa.ldk = sa.ldv = 3 * d;
            sa.q_batch_stride = sa.k_batch_stride = sa.v_batch_stride = (int64_t)T * 3 * d;
            sa.O = DATT; sa.ldo = d; sa.o_batch_stride = (int64_t)T * d;
            sa.B = B; sa.H = H; sa.Sq = T; sa.Skv = T; sa.dh = dh; sa.dv = dh; sa.scale = scale; sa.mask_off = 0;
            ADX_HIP(attention(sa, s));
            LinearArgs o = lin(DATT, d, p + ".self_attn.out_proj.weight", 0, d, d, p + ".self_attn.out_proj.bias", Y, d, B * T, ACT_NONE, PREC_F32);
            o.res = Y; o.ldr = d;
            ADX_HIP(linear(o, s));
            // cross attention
            ADX_TRY(ln_rows(Y, YN, p + ".norm2", B * T, RowMap{}, s));
            ADX_HIP(linear(lin(YN, d, p + ".multihead_attn.in_proj_weight", 0, d, d, p + ".multihead_attn.in_proj_bias", DQ, d, B * T, ACT_NONE, PREC_F32), s));
            LinearArgs kv = lin(memory, d, p + ".multihead_attn.in_proj_weight", d, 2 * d, d, p + ".multihead_attn.in_proj_bias", MKV, 2 * d, B * Lm, ACT_NONE, PREC_F32);
            kv.xmap = mmap;                       // memory rows 0..Lm-1 of each item -> dense (B*Lm) rows
            ADX_HIP(linear(kv, s));
            AttentionArgs ca;
            ca.Q = DQ; ca.ldq = d; ca.q_batch_stride = (int64_t)T * d;
            ca.K = MKV; ca.V = MKV + d; ca.ldk = ca.ldv = 2 * d; ca.k_batch_stride = ca.v_batch_stride = (int64_t)Lm * 2 * d;
            ca.O = DATT; ca.ldo = d; ca.o_batch_stride = (int64_t)T * d;
            ca.B = B; ca.H = H; ca.Sq = T; ca.Skv = Lm; ca.dh = dh; ca.dv = dh; ca.scale = scale; ca.mask_off = 1;
            ADX_HIP(attention(ca, s));
            LinearArgs o2 = lin(DATT, d, p + ".multihead_attn.out_proj.weight", 0, d, d, p + ".multihead_attn.out_proj.bias", Y, d, B * T, ACT_NONE, PREC_F32);
            o2.res = Y; o2.ldr = d;
            ADX_HIP(linear(o2, s));
            // FFN
            ADX_TRY(ln_rows(Y, YN, p + ".norm3", B * T, RowMap{}, s));
            ADX_HIP(linear(lin(YN, d, p + ".linear1.weight", 0, 4 * d, d, p + ".linear1.bias", DFF, 4 * d, B * T, ACT_GELU_ERF, PREC_F32), s));
            LinearArgs f2 = lin(DFF, 4 * d, p + ".linear2.weight", 0, d, 4 * d, p + ".linear2.bias", Y, d, B * T, ACT_NONE, PREC_F32);
            f2.res = Y; f2.ldr = d;
            ADX_HIP(linear(f2, s));
        }
        ADX_TRY(ln_rows(Y, YN, "ln_f", B * T, RowMap{}, s));
        ADX_HIP(linear(lin(YN, d, "head.weight", 0, Cout, d, "head.bias", out, Cout, B * T, ACT_NONE, PREC_F32), s));
        return ADX_OK;
    }

    // Launch ONE stage of a sampling step on the current workspace contents, `iters` times (for timing a kernel in
    // isolation with events on `s`).  Stages: 0 time-token attention, 1 fused encoder rows, 2 last-layer attention,
    // 3 fused tail, 4 per-chunk cache build, 5 per-call map cache build.  Needs a previous generate/sample call with the same B.
    int debug_stage(int B, int stage, int iters, hipStream_t s) {
        if (!finalized || steps.empty() || B > max_batch || !fused) { set_error("debug_stage: engine not ready (needs elide 3, a schedule and a prior call)"); return ADX_ESTATE; }
        for (int it = 0; it < iters; ++it) {
            if (stage == 0) ADX_TRY(time_token_attention(B, 0, s));
            else if (stage == 1) {
                ADX_HIP(enc_rows(rows_args(B, 0), cfg.precision, s, rows_mix()));
            } else if (stage == 2) ADX_TRY(last_attention(B, 0, s));
            else if (stage == 3) {
                if (!tail_ok && !tail_mfma_ok) { set_error("debug_stage: tail kernel not available for this shape"); return ADX_EINVAL; }
                ADX_TRY(launch_tail(B, XS0, nullptr, steps_dev, XS1, MOUT, s));
            } else if (stage == 4) ADX_TRY(build_chunk_cache(B, s));
            else if (stage == 5) ADX_TRY(build_map_cache(B, s));
            else { set_error("debug_stage: unknown stage"); return ADX_EINVAL; }
        }
        return ADX_OK;
    }

    int check_ready(int B, const char* who) {
        if (!finalized) { set_error(std::string(who) + ": engine is not finalized"); return ADX_ESTATE; }
        if (B <= 0) { set_error(std::string(who) + ": batch must be positive"); return ADX_EINVAL; }
        ADX_HIP(hipSetDevice(device));
        ADX_TRY(reserve(B));
        if (fused && (tail_ok || tail_mfma_ok) && !side_stream) {        // never created inside a capture
            ADX_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
            for (int i = 0; i < MAX_DEPTH; ++i) {
                ADX_HIP(hipEventCreateWithFlags(&ev_enc[i], hipEventDisableTiming));
                ADX_HIP(hipEventCreateWithFlags(&ev_tail[i], hipEventDisableTiming));
            }
            const char* no = getenv("ADX_NO_OVERLAP");                   // fallback switch: every kernel of a step on one stream
            overlap_ok = !(no && no[0] == '1');
            ADX_HIP(hipStreamCreateWithFlags(&att_stream, hipStreamNonBlocking));
            for (int i = 0; i < MAX_DEPTH; ++i) ADX_HIP(hipEventCreateWithFlags(&ev_att[i], hipEventDisableTiming));
            const char* ns = getenv("ADX_OVERLAP_STREAMS");               // tuning knob: 2 = tail on a second stream, 3 = attention on a third
            if (ns) overlap_streams = atoi(ns);
            const char* mb = getenv("ADX_OVERLAP_MAX_B");                // tuning knob: largest batch whose tail runs on the second stream
            if (mb) overlap_max_b = atoi(mb);
            else if (f16_single) overlap_max_b = 3072;                   // fp16 engine, trajectories/s with / without: 2048 scenarios 21.1 k / 20.5 k, 4096: 21.0 k / 22.0 k
        }
        return ADX_OK;
    }

    int denoise(int B, const float* x, const int64_t* t, const float* maps, const float* tran, const float* prior, float* out,
                hipStream_t s) {
        ADX_TRY(check_ready(B, "denoise_a"));
        maps_batch = -1;                    // the map caches of a previous sampling call no longer match the embedding
        ADX_TRY(embed_maps(B, maps, s));
        ADX_TRY(embed_prior(B, tran, prior, s));
        ADX_HIP(timestep_embedding(t, B, d, TE, d, s));
        ADX_TRY(set_time(B, TE, d, s));
        return run_net(B, x, out, s);
    }

    // One chunk: transition/prior raw (B,NP,Cp); noise (n_noise,B,T,Cin); result (B,T,Cin) written with row stride ld_out.
    int sample_body(int B, const float* tran_raw, const float* prior_raw, const float* noise, int n_noise, float* out,
                    int64_t ld_out, float* trace_x, float* trace_out, unsigned flags, hipStream_t s) {
        const int64_t n = (int64_t)B * T * Cin;
        const int64_t npts_p = (int64_t)B * NP * Cp / 4;
        const float* tn = tran_raw; const float* pn = prior_raw;
        if (flags & ADX_FLAG_NORMALIZE) {
            ADX_HIP(normalize_traj(tran_raw, TRAN, npts_p, 4, s));
            ADX_HIP(normalize_traj(prior_raw, PRIN, npts_p, 4, s));
            tn = TRAN; pn = PRIN;
        }
        ADX_TRY(embed_prior(B, tn, pn, s));
        const bool cached = cfg.elide >= 2 && n_enc >= 2;
        if (cached) {
            ADX_TRY(set_time(B, temb_table, 0, s));          // defined (if irrelevant) values in row 0 before the cache pass
            prof_begin(4, s);
            ADX_TRY(build_chunk_cache(B, s));
            prof_end(s);
        }
        float* xc = XS0; float* xn = XS1;
        if (flags & ADX_FLAG_INIT_ZERO) ADX_HIP(hipMemsetAsync(xc, 0, sizeof(float) * n, s));
        else ADX_HIP(copy2d(noise, n, 0, xc, n, 1, (int)n, s));
        // Measured (frame_stack 10, one MI355X, trajectories/s with / without): B=512 9810 / 8040, B=1024 11100 / 9130, B=2048 10830 / 10950,
        // B=4096 10940 / 11530 -- below ~2k scenarios the one-wavefront-per-scenario tail leaves the chip mostly idle and hides
        // under the encoder kernels; above, both fill the chip and only disturb each other.
        const bool overlap = cached && fused && (tail_ok || tail_mfma_ok) && overlap_ok && !profiling && side_stream && B <= overlap_max_b;
        if (overlap) {
            // stream s: encoder rows + last-layer attention of step k (buffer set k % depth);
            // side stream: tail of step k, after its encoder kernels and (stream order) after the tail of step k - 1.
            // The encoder kernels of step k reuse the buffers the tail of step k - depth read: they wait for it.  With two sets that wait
            // closes the step's critical cycle (tail(k - 2) -> row kernel(k) -> attention(k) -> tail(k), with a dependency latency of
            // 9-16 us on each edge: profiles/step_gaps.py).  Three or four sets (ADX_OVERLAP_DEPTH) take the wait off the cycle and gain
            // 2 % -- but favouring the tail's waves (tail_first) gains 4.5 % with two sets and nothing more with four, so two it is.
            const size_t n_steps = steps.size();
            for (size_t k = 0; k < n_steps; ++k) {
                const int par = (int)(k % (size_t)depth);
                const int zi = (int)steps[k].z_index;
                const float* z = nullptr;
                if (zi >= 0 && steps[k].c3 != 0.f) {
                    if (1 + zi >= n_noise) { set_error("sample_a: schedule draws more noise tensors than were supplied"); return ADX_EINVAL; }
                    z = noise + (int64_t)(1 + zi) * n;
                }
                if (k >= (size_t)depth) ADX_HIP(hipStreamWaitEvent(s, ev_tail[par], 0));
                ADX_HIP(enc_rows(rows_args(B, (int)k, par), cfg.precision, s, rows_mix()));
                if (overlap_streams >= 3 && attn_g) {
                    // the row kernel of step k + 1 does not wait for this attention: its outputs (rows, queries, residual rows)
                    // go to the other parity, and the buffers of this parity are free again once the tail of step k is done
                    ADX_HIP(hipEventRecord(ev_enc[par], s));
                    ADX_HIP(hipStreamWaitEvent(att_stream, ev_enc[par], 0));
                    ADX_TRY(last_attention(B, par, att_stream));
                    ADX_HIP(hipEventRecord(ev_att[par], att_stream));
                    ADX_HIP(hipStreamWaitEvent(side_stream, ev_att[par], 0));
                } else {
                    ADX_TRY(last_attention(B, par, s));
                    ADX_HIP(hipEventRecord(ev_enc[par], s));
                    ADX_HIP(hipStreamWaitEvent(side_stream, ev_enc[par], 0));
                }
                ADX_TRY(launch_tail(B, xc, z, steps_dev + k, xn, MOUT, side_stream, par));
                if (trace_x) ADX_HIP(copy2d(xc, n, 0, trace_x + k * n, n, 1, (int)n, side_stream));
                if (trace_out) ADX_HIP(copy2d(MOUT, n, 0, trace_out + k * n, n, 1, (int)n, side_stream));
                ADX_HIP(hipEventRecord(ev_tail[par], side_stream));
                float* tmp = xc; xc = xn; xn = tmp;
            }
            // join: everything after the loop (un-normalise, the next chunk) is ordered behind the last tails
            for (size_t j = 1; j <= (size_t)depth && j <= n_steps; ++j) ADX_HIP(hipStreamWaitEvent(s, ev_tail[(n_steps - j) % (size_t)depth], 0));
        }
        for (size_t k = 0; !overlap && k < steps.size(); ++k) {
            if (!fused) ADX_TRY(set_time(B, temb_table + k * d, 0, s));   // the fused path reads the time token from tables
            const int zi = (int)steps[k].z_index;
            const float* z = nullptr;
            if (zi >= 0 && steps[k].c3 != 0.f) {
                if (1 + zi >= n_noise) { set_error("sample_a: schedule draws more noise tensors than were supplied"); return ADX_EINVAL; }
                z = noise + (int64_t)(1 + zi) * n;
            }
            StepIO io;
            io.step = cached ? (int)k : -1; io.z = z; io.x_next = xn;
            bool stepped = false;
            ADX_TRY(run_net(B, xc, MOUT, s, io, &stepped));
            if (trace_x) ADX_HIP(copy2d(xc, n, 0, trace_x + k * n, n, 1, (int)n, s));
            if (trace_out) ADX_HIP(copy2d(MOUT, n, 0, trace_out + k * n, n, 1, (int)n, s));
            if (!stepped) ADX_HIP(sched_step(xc, MOUT, z, xn, nullptr, n, steps_dev + k, s));
            float* tmp = xc; xc = xn; xn = tmp;
        }
        // un-normalise (+ residual) into CHO, then place the chunk
        const float* res = xc;
        if (flags & ADX_FLAG_NORMALIZE) {
            ADX_HIP(denormalize_traj(xc, (flags & ADX_FLAG_RESIDUAL) ? prior_raw : nullptr, CHO, n / 4, 4, s));
            res = CHO;
        } else if (flags & ADX_FLAG_RESIDUAL) {
            if (NP * Cp != T * Cin) { set_error("sample_a: residual needs prior and sample of the same size"); return ADX_EINVAL; }
            ADX_HIP(add_vec_act(xc, prior_raw, CHO, B, T * Cin, 1, ACT_NONE, s));
            res = CHO;
        }
        ADX_HIP(copy2d(res, (int64_t)T * Cin, 0, out, ld_out, B, T * Cin, s));
        return ADX_OK;
    }

    int validate_sampling(int n_noise, const char* who) {
        if (steps.empty()) { set_error(std::string(who) + ": no schedule set"); return ADX_ESTATE; }
        if (Cin != Cout) { set_error(std::string(who) + ": sampling needs in_dim == out_dim"); return ADX_EINVAL; }
        if ((Cin % 4) || (Cp % 4)) { set_error(std::string(who) + ": channel counts must be multiples of 4"); return ADX_EINVAL; }
        if (n_noise < 0) { set_error(std::string(who) + ": negative noise count"); return ADX_EINVAL; }
        return ADX_OK;
    }

    // Noise for a call that passes none: [n_lead][1 + draws][B][T * Cin], filled by the generator kernel inside the graph.
    int device_noise(int n_lead, int B, const float** noise, int* n_noise) {
        *n_noise = 1 + n_schedule_draws();
        ADX_TRY(ensure_noise(sizeof(float) * (size_t)n_lead * *n_noise * B * T * Cin));
        *noise = noise_buf;
        return ADX_OK;
    }

    // Bump carving of the staging buffer (256-byte slots); `take` returns the offset of a slot of `bytes` bytes.
    struct StagePlan {
        size_t total = 0;
        size_t take(size_t bytes) { const size_t o = total; total += (bytes + 255) / 256 * 256; return o; }
    };

    int sample(int B, const float* maps, const float* tran, const float* prior, const float* noise, int n_noise, float* out,
               float* trace_x, float* trace_out, unsigned flags, hipStream_t s) {
        ADX_TRY(check_ready(B, "sample_a"));
        ADX_TRY(validate_sampling(n_noise, "sample_a"));
        const bool keep = (flags & ADX_FLAG_KEEP_MAPS) != 0;
        if (keep && maps_batch != B) { set_error("sample_a: ADX_FLAG_KEEP_MAPS without a previous call at this batch size"); return ADX_EINVAL; }
        const bool dev_noise = !noise && !(flags & ADX_FLAG_INIT_ZERO);
        if (dev_noise) ADX_TRY(device_noise(1, B, &noise, &n_noise));
        char shape[160], ptrs[200];
        snprintf(shape, sizeof shape, "s|%d|%d|%u|%d|%d", B, n_noise, flags, schedule_version, (int)dev_noise);
        snprintf(ptrs, sizeof ptrs, "|%p|%p|%p|%p|%p|%p|%p", keep ? nullptr : (const void*)maps, (const void*)tran, (const void*)prior,
                 dev_noise ? nullptr : (const void*)noise, (void*)out, (void*)trace_x, (void*)trace_out);
        const bool staged = !trace_x && !trace_out && use_staging(shape, ptrs, flags);
        const size_t maps_b = (cfg.map_cond && !keep) ? sizeof(float) * (size_t)B * nmaps * cfg.maps_dim : 0;
        const size_t tp_b = sizeof(float) * (size_t)B * NP * Cp, out_b = sizeof(float) * (size_t)B * T * Cin;
        const size_t noise_b = (noise && !dev_noise) ? sizeof(float) * (size_t)n_noise * B * T * Cin : 0;
        float* g_out = out;
        if (staged) {
            StagePlan pl;
            const size_t o_maps = pl.take(maps_b), o_tran = pl.take(tp_b), o_prior = pl.take(tp_b), o_noise = pl.take(noise_b), o_out = pl.take(out_b);
            ADX_TRY(ensure_stage(pl.total));
            auto stage_in = [&](const float*& p, size_t off, size_t bytes) -> hipError_t {
                if (!bytes || !p) return hipSuccess;
                hipError_t e = hipMemcpyAsync(stage_buf + off, p, bytes, hipMemcpyDeviceToDevice, s);
                p = reinterpret_cast<const float*>(stage_buf + off);
                return e;
            };
            ADX_HIP(stage_in(maps, o_maps, maps_b));
            ADX_HIP(stage_in(tran, o_tran, tp_b));
            ADX_HIP(stage_in(prior, o_prior, tp_b));
            ADX_HIP(stage_in(noise, o_noise, noise_b));
            g_out = reinterpret_cast<float*>(stage_buf + o_out);
        }
        if (dev_noise) ADX_TRY(push_noise_ctl(s));
        maps_batch = B;
        const std::string key = std::string(shape) + (staged ? "|staged" : ptrs);
        const int rc = run(key, flags, s, [&](hipStream_t cs) -> int {
            if (dev_noise) ADX_HIP(randn_rows(noise_ctl, noise_buf, n_noise, B, T * Cin, cs));
            if (!keep) {
                ADX_TRY(embed_maps(B, maps, cs));
                ADX_TRY(build_map_cache(B, cs));
            }
            return sample_body(B, tran, prior, noise, n_noise, g_out, (int64_t)T * Cin, trace_x, trace_out, flags, cs);
        });
        if (rc == ADX_OK && staged) ADX_HIP(hipMemcpyAsync(out, g_out, out_b, hipMemcpyDeviceToDevice, s));
        return rc;
    }

    int generate(int B, int S, int fs, const float* maps, const float* prior_full, const float* init_x, const float* noise,
                 int n_noise, float* out, unsigned flags, hipStream_t s) {
        ADX_TRY(check_ready(B, "generate_a"));
        ADX_TRY(validate_sampling(n_noise, "generate_a"));
        if (fs <= 0 || S % fs) { set_error("Number of frames must be divisible by frame stack size"); return ADX_EINVAL; }
        if (fs * 4 != T * Cin || NP * Cp != fs * 4) { set_error("generate_a: frame_stack does not match the engine's token layout"); return ADX_EINVAL; }
        const int n_chunks = S / fs;
        const int chunk = fs * 4;                       // floats per batch item per chunk
        const bool dev_noise = !noise && !(flags & ADX_FLAG_INIT_ZERO);
        if (dev_noise) ADX_TRY(device_noise(n_chunks, B, &noise, &n_noise));
        char shape[160], ptrs[160];
        snprintf(shape, sizeof shape, "g|%d|%d|%d|%d|%u|%d|%d", B, S, fs, n_noise, flags, schedule_version, (int)dev_noise);
        snprintf(ptrs, sizeof ptrs, "|%p|%p|%p|%p|%p", (const void*)maps, (const void*)prior_full, (const void*)init_x,
                 dev_noise ? nullptr : (const void*)noise, (void*)out);
        const bool staged = use_staging(shape, ptrs, flags);
        const size_t maps_b = cfg.map_cond ? sizeof(float) * (size_t)B * nmaps * cfg.maps_dim : 0;
        const size_t traj_b = sizeof(float) * (size_t)B * S * 4, init_b = sizeof(float) * (size_t)chunk;
        const size_t noise_b = (noise && !dev_noise) ? sizeof(float) * (size_t)n_chunks * n_noise * B * chunk : 0;
        float* g_out = out;
        if (staged) {
            StagePlan pl;
            const size_t o_maps = pl.take(maps_b), o_prior = pl.take(traj_b), o_init = pl.take(init_b), o_noise = pl.take(noise_b), o_out = pl.take(traj_b);
            ADX_TRY(ensure_stage(pl.total));
            auto stage_in = [&](const float*& p, size_t off, size_t bytes) -> hipError_t {
                if (!bytes || !p) return hipSuccess;
                hipError_t e = hipMemcpyAsync(stage_buf + off, p, bytes, hipMemcpyDeviceToDevice, s);
                p = reinterpret_cast<const float*>(stage_buf + off);
                return e;
            };
            ADX_HIP(stage_in(maps, o_maps, maps_b));
            ADX_HIP(stage_in(prior_full, o_prior, traj_b));
            ADX_HIP(stage_in(init_x, o_init, init_b));
            ADX_HIP(stage_in(noise, o_noise, noise_b));
            g_out = reinterpret_cast<float*>(stage_buf + o_out);
        }
        if (dev_noise) ADX_TRY(push_noise_ctl(s));
        maps_batch = -1;                    // generate() embeds the maps itself; a later KEEP_MAPS sample call must not trust them
        const std::string key = std::string(shape) + (staged ? "|staged" : ptrs);
        const int rc = run(key, flags, s, [&](hipStream_t cs) -> int {
            if (dev_noise) ADX_HIP(randn_rows(noise_ctl, noise_buf, n_chunks * n_noise, B, chunk, cs));
            ADX_TRY(embed_maps(B, maps, cs));
            prof_begin(5, cs);
            ADX_TRY(build_map_cache(B, cs));
            prof_end(cs);
            for (int c = 0; c < n_chunks; ++c) {
                ADX_HIP(copy2d(prior_full + (int64_t)c * chunk, (int64_t)S * 4, 0, PRAW, chunk, B, chunk, cs));
                if (c == 0) ADX_HIP(copy2d(init_x, chunk, 1, TRAW, chunk, B, chunk, cs));
                else ADX_HIP(copy2d(g_out + (int64_t)(c - 1) * chunk, (int64_t)S * 4, 0, TRAW, chunk, B, chunk, cs));
                const float* nz = noise ? noise + (int64_t)c * n_noise * B * chunk : nullptr;
                ADX_TRY(sample_body(B, TRAW, PRAW, nz, n_noise, g_out + (int64_t)c * chunk, (int64_t)S * 4, nullptr, nullptr, flags, cs));
            }
            return ADX_OK;
        });
        if (rc == ADX_OK && staged) ADX_HIP(hipMemcpyAsync(out, g_out, traj_b, hipMemcpyDeviceToDevice, s));
        return rc;
    }
    // Latent-carrying chunk loop (StrDiffDiT.generate, models/backbone/StrDiffDiT.py:202-235, with DiffusionTransitionModel.
    // conditional_sample, models/diffusion_loss/diffusion_transition.py:43-78): the denoiser returns a latent z (B, T, Cout);
    // x_from_z maps it to the Cin-wide model output the scheduler step consumes; every chunk starts from its prior (no initial
    // draw); the latent of a chunk's last step, through z_emb, is the next chunk's transition; the prior's first frame is zeroed;
    // out = sample * std + mean.  All chunks x steps in one captured graph.  noise: [n_chunks][n_noise][B][T * Cin].
    int generate_transition(int B, int S, int fs, const float* maps, const float* prior_full, const float* init_z, const float* noise, int n_noise,
                            const float* mean_std, float* out, unsigned flags, hipStream_t s) {
        ADX_TRY(check_ready(B, "generate_transition"));
        if (steps.empty()) { set_error("generate_transition: no schedule set"); return ADX_ESTATE; }
        if (Cin != 4 || Cp != 4 || Cout == Cin || !ZBUF) { set_error("generate_transition: needs in_dim = prior_dim = 4 and a latent head (out_dim != in_dim)"); return ADX_EINVAL; }
        if (fs <= 0 || S % fs) { set_error("Number of frames must be divisible by frame stack size"); return ADX_EINVAL; }
        if (fs != T || NP != fs) { set_error("generate_transition: frame_stack does not match the engine's token layout"); return ADX_EINVAL; }
        const Param *ze = find("z_emb.weight"), *zb = find("z_emb.bias"), *xw = find("x_from_z.weight"), *xb = find("x_from_z.bias");
        if (!ze || !zb || !xw || !xb || ze->shape != std::vector<int64_t>{Cp, Cout} || xw->shape != std::vector<int64_t>{Cin, Cout}) {
            set_error("generate_transition: z_emb.weight (prior_dim, z_dim) / z_emb.bias / x_from_z.weight (in_dim, z_dim) / x_from_z.bias must be set");
            return ADX_EINVAL;
        }
        const int draws = n_schedule_draws();
        if (draws > 0 && (!noise || n_noise < draws)) { set_error("generate_transition: the schedule draws more noise tensors than were supplied"); return ADX_EINVAL; }
        const int n_chunks = S / fs, chunk = fs * 4;
        const int64_t n = (int64_t)B * chunk;
        char shape[160], ptrs[200];
        snprintf(shape, sizeof shape, "t|%d|%d|%d|%d|%u|%d", B, S, fs, n_noise, flags, schedule_version);
        snprintf(ptrs, sizeof ptrs, "|%p|%p|%p|%p|%p|%p", (const void*)maps, (const void*)prior_full, (const void*)init_z, (const void*)noise,
                 (const void*)mean_std, (void*)out);
        const bool staged = use_staging(shape, ptrs, flags);
        const size_t maps_b = cfg.map_cond ? sizeof(float) * (size_t)B * nmaps * cfg.maps_dim : 0;
        const size_t traj_b = sizeof(float) * (size_t)B * S * 4, z_b = sizeof(float) * (size_t)T * Cout, ms_b = sizeof(float) * 8;
        const size_t noise_b = noise ? sizeof(float) * (size_t)n_chunks * n_noise * n : 0;
        float* g_out = out;
        if (staged) {
            StagePlan pl;
            const size_t o_maps = pl.take(maps_b), o_prior = pl.take(traj_b), o_z = pl.take(z_b), o_noise = pl.take(noise_b), o_ms = pl.take(ms_b),
                         o_out = pl.take(traj_b);
            ADX_TRY(ensure_stage(pl.total));
            auto stage_in = [&](const float*& p, size_t off, size_t bytes) -> hipError_t {
                if (!bytes || !p) return hipSuccess;
                hipError_t e = hipMemcpyAsync(stage_buf + off, p, bytes, hipMemcpyDeviceToDevice, s);
                p = reinterpret_cast<const float*>(stage_buf + off);
                return e;
            };
            ADX_HIP(stage_in(maps, o_maps, maps_b));
            ADX_HIP(stage_in(prior_full, o_prior, traj_b));
            ADX_HIP(stage_in(init_z, o_z, z_b));
            ADX_HIP(stage_in(noise, o_noise, noise_b));
            ADX_HIP(stage_in(mean_std, o_ms, ms_b));
            g_out = reinterpret_cast<float*>(stage_buf + o_out);
        }
        maps_batch = -1;
        const bool cached = cfg.elide >= 2 && n_enc >= 2;
        const std::string key = std::string(shape) + (staged ? "|staged" : ptrs);
        const int rc = run(key, flags, s, [&](hipStream_t cs) -> int {
            ADX_TRY(embed_maps(B, maps, cs));
            if (cached) ADX_TRY(build_map_cache(B, cs));
            ADX_HIP(copy2d(init_z, (int64_t)T * Cout, 1, ZBUF, (int64_t)T * Cout, B, T * Cout, cs));          // the initial latent, one copy per item
            for (int c = 0; c < n_chunks; ++c) {
                ADX_HIP(copy2d(prior_full + (int64_t)c * chunk, (int64_t)S * 4, 0, PRAW, chunk, B, chunk, cs));
                if (c == 0) ADX_HIP(hipMemset2DAsync(PRAW, sizeof(float) * chunk, 0, sizeof(float) * 4, B, cs));   // first frame of the prior := 0
                LinearArgs zl = lin(ZBUF, Cout, "z_emb.weight", 0, Cp, Cout, "z_emb.bias", TRAW, Cp, B * T, ACT_NONE, PREC_F32);
                ADX_HIP(linear(zl, cs));
                ADX_TRY(embed_prior(B, TRAW, PRAW, cs));
                if (cached) {
                    ADX_TRY(set_time(B, temb_table, 0, cs));
                    ADX_TRY(build_chunk_cache(B, cs));
                }
                float* xc = XS0; float* xn = XS1;
                ADX_HIP(copy2d(PRAW, n, 0, xc, n, 1, (int)n, cs));                                           // the sample starts from the prior
                const float* nz = noise ? noise + (int64_t)c * n_noise * n : nullptr;
                for (size_t k = 0; k < steps.size(); ++k) {
                    if (!fused) ADX_TRY(set_time(B, temb_table + k * d, 0, cs));
                    const int zi = (int)steps[k].z_index;
                    const float* z = (zi >= 0 && steps[k].c3 != 0.f) ? nz + (int64_t)zi * n : nullptr;
                    StepIO io;
                    io.step = cached ? (int)k : -1;
                    ADX_TRY(run_net(B, xc, MOUT, cs, io));
                    ADX_HIP(linear(lin(MOUT, Cout, "x_from_z.weight", 0, Cin, Cout, "x_from_z.bias", CHO, Cin, B * T, ACT_NONE, PREC_F32), cs));
                    ADX_HIP(sched_step(xc, CHO, z, xn, nullptr, n, steps_dev + k, cs));
                    float* tmp = xc; xc = xn; xn = tmp;
                }
                if (c + 1 < n_chunks) ADX_HIP(copy2d(MOUT, (int64_t)B * T * Cout, 0, ZBUF, (int64_t)B * T * Cout, 1, B * T * Cout, cs));
                if (mean_std) ADX_HIP(affine_channels(xc, mean_std, mean_std + 4, CHO, n / 4, 4, cs));
                ADX_HIP(copy2d(mean_std ? CHO : xc, chunk, 0, g_out + (int64_t)c * chunk, (int64_t)S * 4, B, chunk, cs));
            }
            return ADX_OK;
        });
        if (rc == ADX_OK && staged) ADX_HIP(hipMemcpyAsync(out, g_out, traj_b, hipMemcpyDeviceToDevice, s));
        return rc;
    }
};

}  // namespace

adx_engine* make_engine_a(const adx_model_config& cfg) {
    if (cfg.n_prior_tokens <= 0 || cfg.prior_dim <= 0 || cfg.n_enc_layers <= 0) {
        set_error("engine_create: variant A needs n_prior_tokens, prior_dim and n_enc_layers > 0");
        return nullptr;
    }
    if (cfg.map_cond && (cfg.n_maps_token <= 0 || cfg.maps_dim <= 0)) {
        set_error("engine_create: map_cond needs n_maps_token and maps_dim");
        return nullptr;
    }
    const int dh = cfg.d_model / cfg.n_head;
    if (dh != 4 && dh != 8 && dh != 16 && cfg.n_maps_token + 2 * cfg.n_prior_tokens + 1 > 800) {
        set_error("engine_create: variant A with head_dim outside {4,8,16} supports at most 800 conditioning tokens");
        return nullptr;
    }
    return new EngineA(cfg);
}

#define ADX_API extern "C" __attribute__((visibility("default")))

static EngineA* as_a(adx_engine* e, const char* who) {
    if (!e || e->cfg.variant != ADX_VARIANT_A) { set_error(std::string(who) + ": not a variant-A engine"); return nullptr; }
    return static_cast<EngineA*>(e);
}

ADX_API int adx_denoise_a(adx_engine* e, int B, const float* x, const int64_t* t, const float* maps, const float* transition,
                          const float* prior, float* out, void* stream) {
    EngineA* a = as_a(e, "denoise_a");
    if (!a) return ADX_EINVAL;
    if (!x || !t || !transition || !prior || !out || (a->cfg.map_cond && !maps)) { set_error("denoise_a: null pointer"); return ADX_EINVAL; }
    return a->denoise(B, x, t, maps, transition, prior, out, (hipStream_t)stream);
}

ADX_API int adx_debug_stage_a(adx_engine* e, int B, int stage, int iters, void* stream) {
    EngineA* a = as_a(e, "debug_stage_a");
    if (!a) return ADX_EINVAL;
    return a->debug_stage(B, stage, iters, (hipStream_t)stream);
}

ADX_API int adx_sample_a(adx_engine* e, int B, const float* maps, const float* transition, const float* prior, const float* noise,
                         int n_noise, float* out, float* trace_x, float* trace_out, unsigned flags, void* stream) {
    EngineA* a = as_a(e, "sample_a");
    if (!a) return ADX_EINVAL;
    if (!transition || !prior || !out || (a->cfg.map_cond && !maps)) { set_error("sample_a: null pointer"); return ADX_EINVAL; }
    return a->sample(B, maps, transition, prior, noise, n_noise, out, trace_x, trace_out, flags, (hipStream_t)stream);
}

ADX_API int adx_generate_a(adx_engine* e, int B, int S, int frame_stack, const float* maps, const float* prior_full,
                           const float* init_x, const float* noise, int n_noise, float* out, unsigned flags, void* stream) {
    EngineA* a = as_a(e, "generate_a");
    if (!a) return ADX_EINVAL;
    if (!prior_full || !init_x || !out || (a->cfg.map_cond && !maps)) { set_error("generate_a: null pointer"); return ADX_EINVAL; }
    return a->generate(B, S, frame_stack, maps, prior_full, init_x, noise, n_noise, out, flags, (hipStream_t)stream);
}

ADX_API int adx_generate_transition(adx_engine* e, int B, int S, int frame_stack, const float* maps, const float* prior_full, const float* init_z,
                                    const float* noise, int n_noise, const float* mean_std, float* out, unsigned flags, void* stream) {
    EngineA* a = as_a(e, "generate_transition");
    if (!a) return ADX_EINVAL;
    if (!prior_full || !init_z || !out || (a->cfg.map_cond && !maps)) { set_error("generate_transition: null pointer"); return ADX_EINVAL; }
    return a->generate_transition(B, S, frame_stack, maps, prior_full, init_z, noise, n_noise, mean_std, out, flags, (hipStream_t)stream);
}

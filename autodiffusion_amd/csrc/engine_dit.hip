// Variant-B engine: the DiT / MM-DiT trajectory denoiser and its DDPM sampling loop.
//
// Loop-invariant work is done once per call: state_encoder(state) and, per schedule, the time-encoder MLP
// of every step (timesteps are uniform over the batch while sampling, so the time conditioning is one
// d-vector per step).  Per step: x_encoder (four residual-concat MLP stages), adaLN modulation, the blocks,
// the final layer and the scheduler update.
//
// Residual-concat stage (DecoderResCat): fc(cat[x, relu(LN(W x + b))]).  The concatenation is never copied:
// every stage owns a buffer of width in+hidden whose left columns are written by the previous stage's fc
// GEMM and whose right columns are written by the LayerNorm kernel.
#include "engine.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>

using namespace adx;

namespace {

struct Stage { std::string name; int hidden, in, out; };

struct EngineDiT final : adx_engine {
    int d, H, NL, S, C, n_inner, connect;
    bool mm;
    std::vector<Stage> s_stages, x_stages;
    // workspace
    std::vector<float*> s_cat, x_cat;
    float *HT = nullptr, *SE = nullptr, *HX = nullptr, *CC = nullptr, *SIL = nullptr, *MOD = nullptr, *MODC = nullptr, *HN = nullptr,
          *HN2 = nullptr, *QB = nullptr, *KB = nullptr, *VB = nullptr, *ATT = nullptr, *Z0 = nullptr, *Z1 = nullptr, *FF = nullptr,
          *OB = nullptr, *XA = nullptr, *CA = nullptr, *MODF = nullptr, *XS0 = nullptr, *XS1 = nullptr, *MOUT = nullptr, *TEB = nullptr,
          *TEH1 = nullptr, *TEH2 = nullptr;
    // bf16 operand planes ([K/16][rows][16], high halves then low halves) written by the producing kernels for the large GEMMs
    unsigned short *SILP = nullptr, *HNP = nullptr, *Z1P = nullptr, *FFP = nullptr;
    bool wants_blocked_weights() const override { return cfg.precision != ADX_PREC_F32; }

    explicit EngineDiT(const adx_model_config& c) {
        cfg = c;
        d = c.d_model; H = c.n_head; NL = c.n_layers; S = c.horizon; C = c.in_dim;
        n_inner = c.n_inner; connect = c.connect_dim;
        mm = c.variant == ADX_VARIANT_MMDIT;
        s_stages = {{"state_encoder.0", n_inner, d, connect}, {"state_encoder.1", 2 * connect, connect, d}};
        x_stages = {{"x_encoder.0", 64, C, 128}, {"x_encoder.1", 256, 128, 512}, {"x_encoder.2", 1024, 512, 768},
                    {"x_encoder.3", 2048, 768, d}};
    }

    std::vector<ParamSpec> expected_params() const override {
        std::vector<ParamSpec> v;
        auto add = [&](const std::string& n, std::vector<int64_t> s) { v.push_back({n, std::move(s)}); };
        auto lin2 = [&](const std::string& p, int o, int i) { add(p + ".weight", {o, i}); add(p + ".bias", {o}); };
        add("position_embedding", {1, S, d});
        for (const auto* st : {&s_stages, &x_stages})
            for (const Stage& g : *st) {
                lin2(g.name + ".mlp.linear", g.hidden, g.in);
                add(g.name + ".mlp.layer_norm.weight", {g.hidden}); add(g.name + ".mlp.layer_norm.bias", {g.hidden});
                lin2(g.name + ".fc", g.out, g.hidden + g.in);
            }
        lin2("time_encoder.1", 2 * d, d); lin2("time_encoder.3", 2 * d, 2 * d); lin2("time_encoder.5", d, 2 * d);
        for (int i = 0; i < NL; ++i) {
            const std::string p = "blocks." + std::to_string(i);
            if (!mm) {
                add(p + ".attn.self_attn.in_proj_weight", {3 * d, d}); add(p + ".attn.self_attn.in_proj_bias", {3 * d});
                lin2(p + ".attn.self_attn.out_proj", d, d);
                lin2(p + ".attn.linear1", 4 * d, d); lin2(p + ".attn.linear2", d, 4 * d);
                add(p + ".attn.norm1.weight", {d}); add(p + ".attn.norm1.bias", {d});
                add(p + ".attn.norm2.weight", {d}); add(p + ".attn.norm2.bias", {d});
                lin2(p + ".mlp.0", 4 * d, d); lin2(p + ".mlp.2", d, 4 * d);
                lin2(p + ".adaLN_modulation.1", 6 * d, d);
            } else {
                for (const char* sfx : {"x", "c"}) {
                    const std::string q = sfx;
                    lin2(p + ".adaLN_modulation_" + q + ".1", 6 * d, d);
                    lin2(p + ".linear_" + q + ".0", d, d);
                    lin2(p + ".mlp_" + q + ".0", 4 * d, d); lin2(p + ".mlp_" + q + ".2", d, 4 * d);
                    for (const char* w : {"Wq_", "Wk_", "Wv_"}) add(p + ".attn." + w + q + ".weight", {(int64_t)d * H, d});
                }
                lin2(p + ".attn.proj", d, 2 * d * H);
            }
        }
        lin2("final_layer.linear", C, d);
        lin2("final_layer.adaLN_modulation.1", 2 * d, d);
        return v;
    }

    int carve_workspace(int B, bool dry, size_t* bytes) override {
        size_t off = 0;
        auto take = [&](float** p, size_t n) {
            if (!dry) *p = reinterpret_cast<float*>(ws + off);
            off += ((n * sizeof(float) + 255) / 256) * 256;
        };
        const size_t R = (size_t)B * S;
        int maxh = 0;
        s_cat.resize(s_stages.size()); x_cat.resize(x_stages.size());
        for (size_t i = 0; i < s_stages.size(); ++i) { take(&s_cat[i], R * (s_stages[i].in + s_stages[i].hidden)); maxh = std::max(maxh, s_stages[i].hidden); }
        for (size_t i = 0; i < x_stages.size(); ++i) { take(&x_cat[i], R * (x_stages[i].in + x_stages[i].hidden)); maxh = std::max(maxh, x_stages[i].hidden); }
        take(&HT, R * maxh);
        take(&SE, R * d); take(&HX, R * d); take(&SIL, R * d); take(&MOD, R * 6 * d); take(&HN, R * d);
        take(&FF, R * 4 * d); take(&MODF, R * 2 * d);
        if (!mm) {
            take(&QB, R * 3 * d); take(&ATT, R * d); take(&Z0, R * d); take(&Z1, R * d);
        } else {
            take(&CC, R * d); take(&MODC, R * 6 * d); take(&HN2, R * d);
            take(&QB, R * 2 * d * H); take(&KB, R * 2 * d * H); take(&VB, R * 2 * d * H); take(&ATT, R * 2 * d * H);
            take(&OB, R * d); take(&XA, R * d); take(&CA, R * d);
        }
        take(&XS0, R * C); take(&XS1, R * C); take(&MOUT, R * C);
        if (!mm && cfg.precision != ADX_PREC_F32) {      // operand planes: 2 planes x 2 bytes = one float per element
            float* p = nullptr;
            take(&p, R * d); SILP = reinterpret_cast<unsigned short*>(p);
            take(&p, R * d); HNP = reinterpret_cast<unsigned short*>(p);
            take(&p, R * d); Z1P = reinterpret_cast<unsigned short*>(p);
            take(&p, R * 4 * d); FFP = reinterpret_cast<unsigned short*>(p);
        }
        {   // bf16 planes of the largest GEMM operands (X: R x 4d, W: 6d x 4d at most) for the staged kernel
            float* stage = nullptr;
            const size_t sb = cfg.precision == ADX_PREC_F32 ? 0 : adx::linear_stage_bytes((int)R, 6 * d, 4 * d, cfg.precision);
            take(&stage, sb / sizeof(float));
            xstage = dry ? nullptr : stage; xstage_bytes = sb;
        }
        take(&TEB, (size_t)B * d); take(&TEH1, (size_t)std::max(B, 4096) * 2 * d); take(&TEH2, (size_t)std::max(B, 4096) * 2 * d);
        *bytes = off;
        return ADX_OK;
    }

    // time_encoder on n rows of int64 timesteps -> out (n, d).  Scratch: TEH1/TEH2 (n, 2d).
    int time_encoder(const int64_t* t, int n, float* out, float* h1, float* h2, hipStream_t s) {
        ADX_HIP(timestep_embedding(t, n, d, out, d, s));
        ADX_HIP(linear(lin(out, d, "time_encoder.1.weight", 0, 2 * d, d, "time_encoder.1.bias", h1, 2 * d, n, ACT_LEAKY), s));
        ADX_HIP(linear(lin(h1, 2 * d, "time_encoder.3.weight", 0, 2 * d, 2 * d, "time_encoder.3.bias", h2, 2 * d, n, ACT_LEAKY), s));
        ADX_HIP(linear(lin(h2, 2 * d, "time_encoder.5.weight", 0, d, 2 * d, "time_encoder.5.bias", out, d, n), s));
        return ADX_OK;
    }

    int build_time_table(hipStream_t s) override {
        const int n = (int)steps.size();
        float *h1 = nullptr, *h2 = nullptr;
        ADX_HIP(hipMalloc(&h1, sizeof(float) * (size_t)n * 2 * d));
        ADX_HIP(hipMalloc(&h2, sizeof(float) * (size_t)n * 2 * d));
        int rc = time_encoder(steps_t_dev, n, temb_table, h1, h2, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(h1); (void)hipFree(h2);
        return rc;
    }

    // chain of residual-concat stages; input already sits in cat[0][:, :in0]; last fc writes `out` (+ pos)
    // Stages whose input is raw trajectory coordinates (in <= 8 channels) always run in fp32.
    // A stage both of whose products the staged GEMM takes keeps its concatenated input [x | relu(LN(Linear x))] as operand
    // planes ONLY, in the bytes of the fp32 buffer (two bf16 planes = one float per element; K-blocked planes concatenate
    // along K by appending blocks): the previous stage's fc and the stage's own LayerNorm write them, no re-layout pass runs
    // and the fp32 concatenation never exists.
    int run_stages(const std::vector<Stage>& st, std::vector<float*>& cat, int R, float* out, const float* pos, hipStream_t s) {
        const int eprec = cfg.precision;
        const bool x3 = eprec == ADX_PREC_BF16X3 && !f16_single;     // the planes carry low halves
        const char* pl_env = getenv("ADX_DIT_PLANES");               // read per call: 0 = fp32 activations between all kernels
        const char* st_env = getenv("ADX_DIT_STAGE_PLANES");         // 0 = only these stages keep fp32 concatenations (A/B switch)
        const bool planes_on = eprec != ADX_PREC_F32 && !(pl_env && pl_env[0] == '0') && !(st_env && st_env[0] == '0');
        auto as_planes = [&](size_t i) {
            const Stage& g = st[i];
            return planes_on && i > 0 /* the chain's input arrives as fp32 */ && g.in > 8 && (g.in % 16) == 0 && (g.hidden % 256) == 0 && (g.out % 16) == 0 &&
                   linear_planes_ok(R, g.hidden, g.in, eprec) && linear_planes_ok(R, g.out, g.in + g.hidden, eprec);
        };
        for (size_t i = 0; i < st.size(); ++i) {
            const Stage& g = st[i];
            const int ld = g.in + g.hidden;
            const int prec = g.in <= 8 ? (int)PREC_F32 : -1;
            const bool pk = as_planes(i);
            unsigned short* ph = reinterpret_cast<unsigned short*>(cat[i]);
            unsigned short* pl = x3 ? ph + (size_t)R * ld : nullptr;
            LinearArgs l1 = lin(cat[i], ld, g.name + ".mlp.linear.weight", 0, g.hidden, g.in, g.name + ".mlp.linear.bias", HT, g.hidden, R, ACT_NONE, prec);
            if (pk) { l1.Xh = ph; l1.Xl = pl; }                       // the first in/16 blocks of the stage's planes
            ADX_HIP(linear(l1, s));
            LayerNormArgs ln;
            ln.x = HT; ln.ldx = g.hidden; ln.y = cat[i] + g.in; ln.ldy = ld;
            if (pk) { ln.y = nullptr; ln.ph = ph + (size_t)g.in * R; ln.pl = x3 ? pl + (size_t)g.in * R : nullptr; ln.pf16 = f16x2; }
            ln.gamma = P(g.name + ".mlp.layer_norm.weight"); ln.beta = P(g.name + ".mlp.layer_norm.bias");
            ln.eps = 1e-5f; ln.rows = R; ln.d = g.hidden; ln.relu = 1;
            ADX_HIP(layernorm(ln, s));
            const bool lastst = i + 1 == st.size();
            float* y = lastst ? out : cat[i + 1];
            const int ldy = lastst ? g.out : st[i + 1].in + st[i + 1].hidden;
            LinearArgs fc = lin(cat[i], ld, g.name + ".fc.weight", 0, g.out, ld, g.name + ".fc.bias", y, ldy, R, ACT_NONE, prec);
            if (pk) { fc.Xh = ph; fc.Xl = pl; }
            if (!lastst && as_planes(i + 1)) {
                fc.Y = nullptr;
                fc.Yh = reinterpret_cast<unsigned short*>(cat[i + 1]);
                fc.Yl = x3 ? fc.Yh + (size_t)R * ldy : nullptr;
            }
            if (lastst && pos) { fc.pos = pos; fc.ldp = g.out; fc.pos_period = S; fc.pos_off = 0; }
            ADX_HIP(linear(fc, s));
        }
        return ADX_OK;
    }

    int encode_state(int B, const float* state, hipStream_t s) {
        const int R = B * S;
        ADX_HIP(copy2d(state, d, 0, s_cat[0], s_stages[0].in + s_stages[0].hidden, R, d, s));
        return run_stages(s_stages, s_cat, R, SE, nullptr, s);
    }

    // ph: also (y == nullptr: only) store the result as operand planes of the next product (low halves with split operands)
    void set_planes(LayerNormArgs& a, unsigned short* ph, int R) const {
        a.ph = ph;
        a.pl = (ph && cfg.precision == ADX_PREC_BF16X3 && !f16_single) ? ph + (size_t)R * d : nullptr;
        a.pf16 = f16x2;
    }

    // Samples of one scenario share their conditioning: with a sample group G > 1, rows (b G + g) S + tok of the token buffers
    // read the modulation row b S + tok -- every conditioning-only tensor (state embedding, SiLU input, adaLN outputs) has
    // B / G items instead of B.
    int group = 1;                          // set per call by sample(); 1 everywhere else
    int mod_group() const { return group > 1 ? group * S : 0; }

    int ln_mod(const float* x, float* y, const float* mod, int ld_mod, int shift_chunk, int R, hipStream_t s, unsigned short* ph = nullptr) {
        LayerNormArgs a;
        set_planes(a, ph, R);
        a.x = x; a.ldx = d; a.y = y; a.ldy = d; a.eps = 1e-6f; a.rows = R; a.d = d;
        a.mod_group = mod_group(); a.mod_period = S;
        a.shift = mod + (int64_t)shift_chunk * d; a.scale = mod + (int64_t)(shift_chunk + 1) * d; a.ldm = ld_mod;
        ADX_HIP(layernorm(a, s));
        return ADX_OK;
    }

    int ln_affine(const float* x, float* y, const std::string& p, int R, const float* gate, int ldg, const float* res, hipStream_t s,
                  unsigned short* ph = nullptr) {
        LayerNormArgs a;
        set_planes(a, ph, R);
        a.x = x; a.ldx = d; a.y = y; a.ldy = d; a.eps = 1e-5f; a.rows = R; a.d = d;
        a.gamma = P(p + ".weight"); a.beta = P(p + ".bias");
        a.gate = gate; a.ldg = ldg; a.res = res; a.ldr = d;
        if (gate) { a.mod_group = mod_group(); a.mod_period = S; }
        ADX_HIP(layernorm(a, s));
        return ADX_OK;
    }

    // One denoiser evaluation.  x (B,S,C); te: time conditioning (rows_per_vec = S: one vector per batch item,
    // 0: one vector for all); SE must hold state_encoder(state).  out (B,S,C).
    int run_net(int B, const float* x, const float* te, int64_t rows_per_vec, float* out, hipStream_t s) {
        const int R = B * S;
        const int Rc = R / group;               // conditioning rows (group > 1: several samples per scenario)
        ADX_HIP(copy2d(x, C, 0, x_cat[0], x_stages[0].in + x_stages[0].hidden, R, C, s));
        ADX_TRY(run_stages(x_stages, x_cat, R, HX, P("position_embedding"), s));
        // SiLU(t_emb + state_emb): the only use of the conditioning sum is through SiLU (adaLN inputs)
        ADX_HIP(add_vec_act(SE, te, SIL, Rc, d, rows_per_vec, ACT_SILU, s));
        // hidden size 256: the two-layer MLPs of a block run as one kernel each, their 4d-wide hidden layers stay on the chip
        const char* fm_env = getenv("ADX_DIT_FUSED_MLP");              // read per call: 0 = two products through memory
        const bool fuse_mlp = d == 256 && cfg.precision != ADX_PREC_F32 && R >= 4096 && !(fm_env && fm_env[0] == '0');
        if (!mm) {
            const int dh = d / H;
            // Operand planes: where the staged GEMM will take a product (enough tiles, long enough K), the kernel that PRODUCES
            // its X writes bf16 planes in the layout that GEMM copies from -- no re-layout pass, and activations that only feed
            // a product (the 4d-wide hidden layers, the second modulated LayerNorm) never exist in fp32.
            const int prec = cfg.precision;
            const char* pl_env = getenv("ADX_DIT_PLANES");           // fallback / measurement switch, read per call: 0 = fp32 activations between all kernels
            const bool planes = prec != ADX_PREC_F32 && HNP && (d % 256) == 0 && !(pl_env && pl_env[0] == '0');
            const bool pk_d = planes && linear_planes_ok(R, 3 * d, d, prec) && linear_planes_ok(R, 4 * d, d, prec);
            const bool pk_4d = planes && linear_planes_ok(R, d, 4 * d, prec);
            const bool pk_sil = planes && linear_planes_ok(Rc, 6 * d, d, prec);
            // fp16 engines also take the out-projection through planes (one re-layout pass over the attention output, then the
            // one- or two-instruction product); with split-bf16 operands the tiled kernel on fp32 input measured no slower
            const char* po_env = getenv("ADX_DIT_OUT_PLANES");       // measurement switch, read per call: 0 = tiled kernel
            const bool pk_o = planes && f16x2 && linear_planes_ok(R, d, d, prec) && !(po_env && po_env[0] == '0');
            const bool x3 = prec == ADX_PREC_BF16X3 && !f16_single;      // the planes carry low halves
            auto use_planes = [&](LinearArgs& a, const unsigned short* ph, int K) { a.Xh = ph; a.Xl = x3 ? ph + (size_t)R * K : nullptr; };
            auto out_planes = [&](LinearArgs& a, unsigned short* ph, int N) { a.Y = nullptr; a.Yh = ph; a.Yl = x3 ? ph + (size_t)R * N : nullptr; };
            if (pk_sil) ADX_HIP(stage_planes(SIL, d, Rc, d, ACT_NONE, SILP, x3 ? SILP + (size_t)Rc * d : nullptr, s, f16x2));
            for (int i = 0; i < NL; ++i) {
                const std::string p = "blocks." + std::to_string(i);
                LinearArgs ada = lin(SIL, d, p + ".adaLN_modulation.1.weight", 0, 6 * d, d, p + ".adaLN_modulation.1.bias", MOD, 6 * d, Rc);
                if (pk_sil) { ada.Xh = SILP; ada.Xl = x3 ? SILP + (size_t)Rc * d : nullptr; }
                ADX_HIP(linear(ada, s));
                ADX_TRY(ln_mod(HX, HN, MOD, 6 * d, 0, R, s, pk_d ? HNP : nullptr));
                // post-norm ReLU encoder layer on HN
                LinearArgs qkv = lin(HN, d, p + ".attn.self_attn.in_proj_weight", 0, 3 * d, d, p + ".attn.self_attn.in_proj_bias", QB, 3 * d, R);
                if (pk_d) use_planes(qkv, HNP, d);
                ADX_HIP(linear(qkv, s));
                AttentionArgs at;
                at.Q = QB; at.K = QB + d; at.V = QB + 2 * d; at.ldq = at.ldk = at.ldv = 3 * d;
                at.q_batch_stride = at.k_batch_stride = at.v_batch_stride = (int64_t)S * 3 * d;
                at.O = ATT; at.ldo = d; at.o_batch_stride = (int64_t)S * d;
                at.B = B; at.H = H; at.Sq = S; at.Skv = S; at.dh = dh; at.dv = dh; at.scale = 1.0f / sqrtf((float)dh); at.mask_off = -1;
                at.prec = prec;
                ADX_HIP(attention(at, s));
                LinearArgs o = lin(ATT, d, p + ".attn.self_attn.out_proj.weight", 0, d, d, p + ".attn.self_attn.out_proj.bias", Z0, d, R);
                o.res = HN; o.ldr = d;
                if (pk_o) {             // Z1P is free here: norm1 writes it after this product, linear1 reads it after that
                    ADX_HIP(stage_planes(ATT, d, R, d, ACT_NONE, Z1P, x3 ? Z1P + (size_t)R * d : nullptr, s, true));
                    use_planes(o, Z1P, d);
                }
                ADX_HIP(linear(o, s));
                ADX_TRY(ln_affine(Z0, Z1, p + ".attn.norm1", R, nullptr, 0, nullptr, s, pk_d ? Z1P : nullptr));
                LinearArgs f1 = lin(Z1, d, p + ".attn.linear1.weight", 0, 4 * d, d, p + ".attn.linear1.bias", FF, 4 * d, R, ACT_RELU);
                LinearArgs f2 = lin(FF, 4 * d, p + ".attn.linear2.weight", 0, d, 4 * d, p + ".attn.linear2.bias", Z0, d, R);
                f2.res = Z1; f2.ldr = d;
                if (fuse_mlp && mlp256_ok(f1, f2)) {
                    ADX_HIP(mlp256(f1, f2, s));
                } else {
                    if (pk_d) use_planes(f1, Z1P, d);
                    if (pk_4d) { out_planes(f1, FFP, 4 * d); use_planes(f2, FFP, 4 * d); }
                    ADX_HIP(linear(f1, s));
                    ADX_HIP(linear(f2, s));
                }
                // x = x + gate_msa * norm2(...)
                ADX_TRY(ln_affine(Z0, HX, p + ".attn.norm2", R, MOD + 2 * d, 6 * d, HX, s));
                // x = x + gate_mlp * gelu(W2 gelu(W1 modulate(LN(x))))
                ADX_TRY(ln_mod(HX, (pk_d && !fuse_mlp) ? nullptr : HN, MOD, 6 * d, 3, R, s, pk_d ? HNP : nullptr));
                LinearArgs m1 = lin(HN, d, p + ".mlp.0.weight", 0, 4 * d, d, p + ".mlp.0.bias", FF, 4 * d, R, ACT_GELU_TANH);
                LinearArgs m2 = lin(FF, 4 * d, p + ".mlp.2.weight", 0, d, 4 * d, p + ".mlp.2.bias", HX, d, R, ACT_GELU_TANH);
                m2.gate = MOD + 5 * d; m2.ldg = 6 * d; m2.res = HX; m2.ldr = d;
                m2.gate_group = mod_group(); m2.gate_period = S;
                if (fuse_mlp && mlp256_ok(m1, m2)) {
                    ADX_HIP(mlp256(m1, m2, s));
                } else {
                    if (pk_d) use_planes(m1, HNP, d);
                    if (pk_4d) { out_planes(m1, FFP, 4 * d); use_planes(m2, FFP, 4 * d); }
                    ADX_HIP(linear(m1, s));
                    ADX_HIP(linear(m2, s));
                }
            }
        } else {
            const int kd = d;                   // k_dim = v_dim = d_model per head, per stream
            const int hd = 2 * kd;              // joint head width (x part then c part)
            // the condition stream starts from the state embedding of the sample's scenario (it then evolves per sample; only the
            // modulation tables, functions of the conditioning alone, stay shared inside a sample group)
            if (group == 1) ADX_HIP(copy2d(SE, d, 0, CC, d, R, d, s));
            else
                for (int g = 0; g < group; ++g)     // item b G + g <- item b: whole items as rows of S d floats
                    ADX_HIP(copy2d(SE, (int64_t)S * d, 0, CC + (size_t)g * S * d, (int64_t)group * S * d, B / group, S * d, s));
            for (int i = 0; i < NL; ++i) {
                const std::string p = "blocks." + std::to_string(i);
                ADX_HIP(linear(lin(SIL, d, p + ".adaLN_modulation_x.1.weight", 0, 6 * d, d, p + ".adaLN_modulation_x.1.bias", MOD, 6 * d, Rc), s));
                ADX_HIP(linear(lin(SIL, d, p + ".adaLN_modulation_c.1.weight", 0, 6 * d, d, p + ".adaLN_modulation_c.1.bias", MODC, 6 * d, Rc), s));
                ADX_TRY(ln_mod(HX, HN, MOD, 6 * d, 0, R, s));
                ADX_TRY(ln_mod(CC, HN2, MODC, 6 * d, 0, R, s));
                const char* names[3] = {"Wq_", "Wk_", "Wv_"};
                float* dst[3] = {QB, KB, VB};
                for (int w = 0; w < 3; ++w)
                    for (int st = 0; st < 2; ++st) {
                        const std::string wn = p + ".attn." + names[w] + (st ? "c" : "x") + ".weight";
                        LinearArgs a = lin(st ? HN2 : HN, d, wn, 0, kd * H, d, "", dst[w], (int64_t)hd * H, R);
                        a.col_group = kd; a.col_stride = hd; a.col_off = st * kd;
                        ADX_HIP(linear(a, s));
                    }
                AttentionArgs at;
                at.Q = QB; at.K = KB; at.V = VB; at.ldq = at.ldk = at.ldv = (int64_t)hd * H;
                at.q_batch_stride = at.k_batch_stride = at.v_batch_stride = (int64_t)S * hd * H;
                at.O = ATT; at.ldo = (int64_t)hd * H; at.o_batch_stride = (int64_t)S * hd * H;
                at.B = B; at.H = H; at.Sq = S; at.Skv = S; at.dh = hd; at.dv = hd; at.scale = 1.0f / sqrtf((float)kd); at.mask_off = -1;
                at.prec = cfg.precision;
                ADX_HIP(attention(at, s));
                ADX_HIP(linear(lin(ATT, (int64_t)hd * H, p + ".attn.proj.weight", 0, d, hd * H, p + ".attn.proj.bias", OB, d, R), s));
                for (int st = 0; st < 2; ++st) {
                    const std::string q = st ? "c" : "x";
                    float* stream_buf = st ? CC : HX;
                    float* attn_buf = st ? CA : XA;
                    const float* mod = st ? MODC : MOD;
                    float* hn = st ? HN2 : HN;
                    LinearArgs l1 = lin(OB, d, p + ".linear_" + q + ".0.weight", 0, d, d, p + ".linear_" + q + ".0.bias", attn_buf, d, R, ACT_GELU_TANH);
                    l1.gate = mod + 2 * d; l1.ldg = 6 * d; l1.res = stream_buf; l1.ldr = d;
                    l1.gate_group = mod_group(); l1.gate_period = S;
                    ADX_HIP(linear(l1, s));
                    ADX_TRY(ln_mod(attn_buf, hn, mod, 6 * d, 3, R, s));
                    LinearArgs m1 = lin(hn, d, p + ".mlp_" + q + ".0.weight", 0, 4 * d, d, p + ".mlp_" + q + ".0.bias", FF, 4 * d, R, ACT_GELU_TANH);
                    // residual from the stream's block INPUT, not from attn_buf
                    LinearArgs m2 = lin(FF, 4 * d, p + ".mlp_" + q + ".2.weight", 0, d, 4 * d, p + ".mlp_" + q + ".2.bias", stream_buf, d, R, ACT_GELU_TANH);
                    m2.gate = mod + 5 * d; m2.ldg = 6 * d; m2.res = stream_buf; m2.ldr = d;
                    m2.gate_group = mod_group(); m2.gate_period = S;
                    if (fuse_mlp && mlp256_ok(m1, m2)) {
                        ADX_HIP(mlp256(m1, m2, s));
                    } else {
                        ADX_HIP(linear(m1, s));
                        ADX_HIP(linear(m2, s));
                    }
                }
            }
        }
        // final layer
        ADX_HIP(linear(lin(SIL, d, "final_layer.adaLN_modulation.1.weight", 0, 2 * d, d, "final_layer.adaLN_modulation.1.bias", MODF, 2 * d, Rc), s));
        ADX_TRY(ln_mod(HX, HN, MODF, 2 * d, 0, R, s));
        ADX_HIP(linear(lin(HN, d, "final_layer.linear.weight", 0, C, d, "final_layer.linear.bias", out, C, R, ACT_NONE, PREC_F32), s));
        return ADX_OK;
    }

    int check_ready(int B, const char* who) {
        if (!finalized) { set_error(std::string(who) + ": engine is not finalized"); return ADX_ESTATE; }
        if (B <= 0) { set_error(std::string(who) + ": batch must be positive"); return ADX_EINVAL; }
        ADX_HIP(hipSetDevice(device));
        ADX_TRY(reserve(B));
        return ADX_OK;
    }

    int denoise(int B, const float* x, const int64_t* t, const float* state, float* out, hipStream_t s) {
        ADX_TRY(check_ready(B, "denoise_dit"));
        ADX_TRY(encode_state(B, state, s));
        ADX_TRY(time_encoder(t, B, TEB, TEH1, TEH2, s));
        return run_net(B, x, TEB, S, out, s);
    }

    // G > 1: the B sample rows come in groups of G consecutive rows that share their conditioning; `state` then holds B / G items
    int sample(int B, const float* state, const float* noise, int n_noise, float* out, unsigned flags, hipStream_t s, int G = 1) {
        ADX_TRY(check_ready(B, "sample_dit"));
        if (G < 1 || B % G != 0) { set_error("sample_dit: the batch must be a multiple of the sample group"); return ADX_EINVAL; }
        if (steps.empty()) { set_error("sample_dit: no schedule set"); return ADX_ESTATE; }
        if (cfg.in_dim != cfg.out_dim) { set_error("sample_dit: in_dim must equal out_dim"); return ADX_EINVAL; }
        const int64_t n = (int64_t)B * S * C;
        const bool dev_noise = !noise && !(flags & ADX_FLAG_INIT_ZERO);
        if (dev_noise) {
            n_noise = 1 + n_schedule_draws();
            ADX_TRY(ensure_noise(sizeof(float) * (size_t)n_noise * n));
            noise = noise_buf;
        }
        for (const adx_step_coef& k : steps)
            if (k.c3 != 0.f && k.z_index >= 0 && (!noise || 1 + (int)k.z_index >= n_noise)) {
                set_error("sample_dit: schedule draws more noise tensors than were supplied"); return ADX_EINVAL;
            }
        char shape[160], ptrs[120];
        snprintf(shape, sizeof shape, "d|%d|%d|%u|%d|%d|g%d", B, n_noise, flags, schedule_version, (int)dev_noise, G);
        snprintf(ptrs, sizeof ptrs, "|%p|%p|%p", (const void*)state, dev_noise ? nullptr : (const void*)noise, (void*)out);
        const bool staged = use_staging(shape, ptrs, flags);
        const size_t state_b = sizeof(float) * (size_t)(B / G) * S * d, out_b = sizeof(float) * (size_t)n;
        const size_t noise_b = (noise && !dev_noise) ? sizeof(float) * (size_t)n_noise * n : 0;
        float* g_out = out;
        if (staged) {
            auto slot = [](size_t b) { return (b + 255) / 256 * 256; };
            const size_t o_state = 0, o_noise = slot(state_b), o_out = o_noise + slot(noise_b);
            ADX_TRY(ensure_stage(o_out + slot(out_b)));
            ADX_HIP(hipMemcpyAsync(stage_buf + o_state, state, state_b, hipMemcpyDeviceToDevice, s));
            state = reinterpret_cast<const float*>(stage_buf + o_state);
            if (noise_b) {
                ADX_HIP(hipMemcpyAsync(stage_buf + o_noise, noise, noise_b, hipMemcpyDeviceToDevice, s));
                noise = reinterpret_cast<const float*>(stage_buf + o_noise);
            }
            g_out = reinterpret_cast<float*>(stage_buf + o_out);
        }
        if (dev_noise) ADX_TRY(push_noise_ctl(s));
        const std::string key = std::string(shape) + (staged ? "|staged" : ptrs);
        const int rc = run(key, flags, s, [&](hipStream_t cs) -> int {
            if (dev_noise) ADX_HIP(randn_rows(noise_ctl, noise_buf, n_noise, B, S * C, cs));
            ADX_TRY(encode_state(B / G, state, cs));
            group = G;
            float* xc = XS0; float* xn = XS1;
            if (flags & ADX_FLAG_INIT_ZERO) ADX_HIP(hipMemsetAsync(xc, 0, sizeof(float) * n, cs));
            else ADX_HIP(copy2d(noise, n, 0, xc, n, 1, (int)n, cs));
            for (size_t k = 0; k < steps.size(); ++k) {
                ADX_TRY(run_net(B, xc, temb_table + k * d, 0, MOUT, cs));
                const int zi = (int)steps[k].z_index;
                const float* z = (zi >= 0 && steps[k].c3 != 0.f) ? noise + (int64_t)(1 + zi) * n : nullptr;
                ADX_HIP(sched_step(xc, MOUT, z, xn, nullptr, n, steps_dev + k, cs));
                std::swap(xc, xn);
            }
            group = 1;
            if (flags & ADX_FLAG_NORMALIZE) ADX_HIP(denormalize_traj(xc, nullptr, g_out, n / C, C, cs));
            else ADX_HIP(copy2d(xc, n, 0, g_out, n, 1, (int)n, cs));
            return ADX_OK;
        });
        group = 1;
        if (rc == ADX_OK && staged) ADX_HIP(hipMemcpyAsync(out, g_out, out_b, hipMemcpyDeviceToDevice, s));
        return rc;
    }
};

}  // namespace

adx_engine* make_engine_dit(const adx_model_config& cfg) {
    if (cfg.n_inner <= 0 || cfg.connect_dim <= 0) { set_error("engine_create: DiT needs n_inner and connect_dim"); return nullptr; }
    if (cfg.in_dim != 4 && cfg.in_dim != 2) { set_error("engine_create: DiT in_dim must be 4 or 2"); return nullptr; }
    const int hd = cfg.variant == ADX_VARIANT_MMDIT ? 2 * cfg.d_model : cfg.d_model / cfg.n_head;
    (void)hd;
    return new EngineDiT(cfg);
}

#define ADX_API extern "C" __attribute__((visibility("default")))

static EngineDiT* as_dit(adx_engine* e, const char* who) {
    if (!e || (e->cfg.variant != ADX_VARIANT_DIT && e->cfg.variant != ADX_VARIANT_MMDIT)) {
        set_error(std::string(who) + ": not a DiT engine");
        return nullptr;
    }
    return static_cast<EngineDiT*>(e);
}

ADX_API int adx_denoise_dit(adx_engine* e, int B, const float* x, const int64_t* t, const float* state, float* out, void* stream) {
    EngineDiT* a = as_dit(e, "denoise_dit");
    if (!a) return ADX_EINVAL;
    if (!x || !t || !state || !out) { set_error("denoise_dit: null pointer"); return ADX_EINVAL; }
    return a->denoise(B, x, t, state, out, (hipStream_t)stream);
}

ADX_API int adx_sample_dit(adx_engine* e, int B, const float* state, const float* noise, int n_noise, float* out, unsigned flags,
                           void* stream) {
    EngineDiT* a = as_dit(e, "sample_dit");
    if (!a) return ADX_EINVAL;
    if (!state || !out) { set_error("sample_dit: null pointer"); return ADX_EINVAL; }
    return a->sample(B, state, noise, n_noise, out, flags, (hipStream_t)stream);
}

ADX_API int adx_sample_dit_grouped(adx_engine* e, int B, int group, const float* state, const float* noise, int n_noise, float* out,
                                   unsigned flags, void* stream) {
    EngineDiT* a = as_dit(e, "sample_dit_grouped");
    if (!a) return ADX_EINVAL;
    if (!state || !out) { set_error("sample_dit_grouped: null pointer"); return ADX_EINVAL; }
    return a->sample(B, state, noise, n_noise, out, flags, (hipStream_t)stream, group);
}

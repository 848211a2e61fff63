// Engine base implementation and the extern "C" surface of libadx.so.
#include "engine.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace adx {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return ADX_EHIP;
}

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace adx

using namespace adx;

// ------------------------------------------------------------------------------------------ engine base
adx_engine::~adx_engine() {
    drop_graphs();
    for (const ProfSpan& p : prof_spans) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : prof_pool) (void)hipEventDestroy(e);
    for (auto& kv : params) {
        if (kv.second.f32) (void)hipFree(kv.second.f32);
        if (kv.second.bf16) (void)hipFree(kv.second.bf16);
        if (kv.second.bf16lo) (void)hipFree(kv.second.bf16lo);
        if (kv.second.blk) (void)hipFree(kv.second.blk);
        if (kv.second.blk_lo) (void)hipFree(kv.second.blk_lo);
        if (kv.second.blk_h) (void)hipFree(kv.second.blk_h);
    }
    if (ws) (void)hipFree(ws);
    if (stage_buf) (void)hipFree(stage_buf);
    if (noise_buf) (void)hipFree(noise_buf);
    if (noise_ctl) (void)hipFree(noise_ctl);
    if (steps_dev) (void)hipFree(steps_dev);
    if (steps_t_dev) (void)hipFree(steps_t_dev);
    if (temb_table) (void)hipFree(temb_table);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
}

void adx_engine::drop_graphs() {
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
}

const Param* adx_engine::find(const std::string& name) const {
    auto it = params.find(name);
    return it == params.end() ? nullptr : &it->second;
}

const float* adx_engine::P(const std::string& name) const {
    const Param* p = find(name);
    return p ? p->f32 : nullptr;
}

const void* adx_engine::W(const std::string& name, int64_t row0, int prec) const {
    const Param* p = find(name);
    if (!p) return nullptr;
    const int64_t K = p->shape.back();
    if (prec == ADX_PREC_BF16 || prec == ADX_PREC_BF16X3) return p->bf16 + row0 * K;
    return p->f32 + row0 * K;
}

LinearArgs adx_engine::lin(const float* X, int64_t ldx, const std::string& wname, int64_t row0, int N, int K,
                           const std::string& bname, float* Y, int64_t ldy, int M, int act, int prec) const {
    if (prec < 0) prec = cfg.precision;
    LinearArgs a;
    a.X = X; a.ldx = ldx;
    a.W = W(wname, row0, prec);
    if (prec == ADX_PREC_BF16X3) {
        const Param* p = find(wname);
        a.Wlo = p ? p->bf16lo + row0 * p->shape.back() : nullptr;
    }
    a.bias = bname.empty() ? nullptr : P(bname) + row0;
    a.Y = Y; a.ldy = ldy;
    a.M = M; a.N = N; a.K = K;
    a.act = act;
    a.prec = prec;
    a.xstage = xstage; a.xstage_bytes = xstage_bytes;
    if (prec != ADX_PREC_F32) {
        const Param* p = find(wname);
        if (p && p->blk && row0 == 0 && p->shape.size() == 2 && N == p->shape[0] && K == p->shape[1]) { a.Wblk = p->blk; a.Wblk_lo = p->blk_lo; }
        if (f16x2 && prec == ADX_PREC_BF16X3) { a.f16 = true; a.Wblk_h = a.Wblk ? p->blk_h : nullptr; }
    }
    return a;
}

int adx_engine::set_tensor(const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!name || !data || !shape || ndim < 1 || ndim > 4) return fail(ADX_EINVAL, "set_tensor: bad argument");
    int64_t numel = 1;
    std::vector<int64_t> shp(shape, shape + ndim);
    for (int64_t s : shp) {
        if (s <= 0) return fail(ADX_EINVAL, std::string("set_tensor: non-positive dimension in ") + name);
        numel *= s;
    }
    ADX_HIP(hipSetDevice(device));
    Param& p = params[name];
    if (p.numel != numel) {
        if (p.f32) (void)hipFree(p.f32);
        if (p.bf16) (void)hipFree(p.bf16);
        if (p.bf16lo) (void)hipFree(p.bf16lo);
        if (p.blk) (void)hipFree(p.blk);
        if (p.blk_lo) (void)hipFree(p.blk_lo);
        if (p.blk_h) (void)hipFree(p.blk_h);
        p.f32 = nullptr; p.bf16 = nullptr; p.bf16lo = nullptr; p.blk = nullptr; p.blk_lo = nullptr; p.blk_h = nullptr;
        ADX_HIP(hipMalloc(&p.f32, sizeof(float) * numel));
        if (ndim >= 2) {
            ADX_HIP(hipMalloc(&p.bf16, sizeof(uint16_t) * numel));
            ADX_HIP(hipMalloc(&p.bf16lo, sizeof(uint16_t) * numel));
        }
        drop_graphs();                      // captured graphs hold the old addresses
    }
    p.shape = shp;
    p.numel = numel;
    ADX_HIP(hipMemcpyAsync(p.f32, data, sizeof(float) * numel, hipMemcpyDeviceToDevice, nullptr));
    if (p.bf16) ADX_HIP(to_bf16(p.f32, p.bf16, p.bf16lo, numel, nullptr));
    finalized = false;
    return ADX_OK;
}

int adx_engine::finalize() {
    for (const ParamSpec& s : expected_params()) {
        const Param* p = find(s.name);
        if (!p) return fail(ADX_ENOTFOUND, "finalize: missing parameter '" + s.name + "'");
        if (p->shape != s.shape) {
            std::string got, want;
            for (int64_t v : p->shape) got += std::to_string(v) + ",";
            for (int64_t v : s.shape) want += std::to_string(v) + ",";
            return fail(ADX_EINVAL, "finalize: parameter '" + s.name + "' has shape [" + got + "] expected [" + want + "]");
        }
    }
    if (wants_blocked_weights()) {
        ADX_HIP(hipSetDevice(device));
        for (auto& kv : params) {
            Param& p = kv.second;
            if (p.shape.size() != 2 || !p.bf16) continue;
            const int64_t N = p.shape[0], K = p.shape[1];
            if ((K % 32) != 0 || N * K >= ((int64_t)1 << 31)) continue;
            if (!p.blk) {
                ADX_HIP(hipMalloc(&p.blk, sizeof(uint16_t) * p.numel));
                ADX_HIP(hipMalloc(&p.blk_lo, sizeof(uint16_t) * p.numel));
            }
            ADX_HIP(linear_block_weight(p.bf16, (int)N, (int)K, p.blk, nullptr));
            ADX_HIP(linear_block_weight(p.bf16lo, (int)N, (int)K, p.blk_lo, nullptr));
            if (f16x2) {                    // fp16 copy, K-blocked: row-major fp16 bits through a scratch buffer first
                uint16_t* tmp = nullptr;
                if (!p.blk_h) ADX_HIP(hipMalloc(&p.blk_h, sizeof(uint16_t) * p.numel));
                ADX_HIP(hipMalloc(&tmp, sizeof(uint16_t) * p.numel));
                hipError_t e = to_f16(p.f32, tmp, p.numel, nullptr);
                if (e == hipSuccess) e = linear_block_weight(tmp, (int)N, (int)K, p.blk_h, nullptr);
                if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
                (void)hipFree(tmp);
                ADX_HIP(e);
            }
        }
    }
    ADX_HIP(hipStreamSynchronize(nullptr));
    finalized = true;
    if (!steps.empty()) {
        // The schedule tables and the packed weight images of the fused kernels are derived from the parameters: rebuild
        // them for the weights just loaded (their buffers may move, so captured graphs go too).
        ADX_HIP(hipSetDevice(device));
        ADX_HIP(hipDeviceSynchronize());
        drop_graphs();
        ADX_TRY(build_time_table(nullptr));
        ADX_HIP(hipStreamSynchronize(nullptr));
    }
    return ADX_OK;
}

int adx_engine::ensure_stage(size_t bytes) {
    if (bytes <= stage_bytes && stage_buf) return ADX_OK;
    ADX_HIP(hipSetDevice(device));
    if (stage_buf) {
        ADX_HIP(hipDeviceSynchronize());            // a replay that reads the old buffer may still be running
        (void)hipFree(stage_buf);
        stage_buf = nullptr; stage_bytes = 0;
        drop_graphs();                              // staged graphs hold the old addresses
    }
    hipError_t e = hipMalloc(&stage_buf, bytes);
    if (e != hipSuccess) { stage_buf = nullptr; return fail(ADX_ENOMEM, "staging buffer: hipMalloc of " + std::to_string(bytes) + " bytes failed"); }
    stage_bytes = bytes;
    return ADX_OK;
}

bool adx_engine::use_staging(const std::string& shape_key, const std::string& ptr_key, unsigned flags) {
    if ((flags & ADX_FLAG_NO_GRAPH) || profiling) return false;      // eager calls read the caller's buffers directly
    if (io_modes.size() > 256) io_modes.clear();
    IoMode& m = io_modes[shape_key];
    if (m.staged) return true;
    if (m.ptr_key.empty()) { m.ptr_key = ptr_key; return false; }
    if (m.ptr_key == ptr_key) return false;
    m.staged = true;                                                  // the pointers of this shape change between calls
    return true;
}

int adx_engine::ensure_noise(size_t bytes) {
    ADX_HIP(hipSetDevice(device));
    if (!noise_ctl) {
        ADX_HIP(hipMalloc(&noise_ctl, 2 * sizeof(uint64_t)));
        ADX_HIP(hipMemset(noise_ctl, 0, 2 * sizeof(uint64_t)));
    }
    if (bytes <= noise_bytes && noise_buf) return ADX_OK;
    if (noise_buf) {
        ADX_HIP(hipDeviceSynchronize());
        (void)hipFree(noise_buf);
        noise_buf = nullptr; noise_bytes = 0;
        drop_graphs();
    }
    hipError_t e = hipMalloc(&noise_buf, bytes);
    if (e != hipSuccess) { noise_buf = nullptr; return fail(ADX_ENOMEM, "noise buffer: hipMalloc of " + std::to_string(bytes) + " bytes failed"); }
    noise_bytes = bytes;
    return ADX_OK;
}

int adx_engine::push_noise_ctl(hipStream_t s) {
    ADX_HIP(set_noise_ctl(noise_ctl, noise_seed, noise_row0, s));
    return ADX_OK;
}

int adx_engine::n_schedule_draws() const {
    int n = 0;
    for (const adx_step_coef& k : steps)
        if (k.c3 != 0.f && k.z_index >= 0.f) n = std::max(n, (int)k.z_index + 1);
    return n;
}

int adx_engine::reserve(int batch) {
    if (batch <= 0) return fail(ADX_EINVAL, "reserve: batch must be positive");
    if (batch <= max_batch && ws) return ADX_OK;
    ADX_HIP(hipSetDevice(device));
    size_t bytes = 0;
    ADX_TRY(carve_workspace(batch, true, &bytes));
    drop_graphs();
    if (ws) { ADX_HIP(hipDeviceSynchronize()); (void)hipFree(ws); ws = nullptr; }
    hipError_t e = hipMalloc(&ws, bytes);
    if (e != hipSuccess) {
        ws = nullptr; ws_bytes = 0; max_batch = 0;
        return fail(ADX_ENOMEM, "reserve: hipMalloc of " + std::to_string(bytes) + " bytes failed");
    }
    ws_bytes = bytes;
    max_batch = batch;
    on_workspace_realloc();
    return carve_workspace(batch, false, &bytes);
}

void adx_engine::prof_begin(int stage, hipStream_t s) {
    if (!profiling) return;
    auto take = [&]() {
        hipEvent_t e = nullptr;
        if (!prof_pool.empty()) { e = prof_pool.back(); prof_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    };
    ProfSpan p{take(), take(), stage};
    if (!p.a || !p.b) return;
    (void)hipEventRecord(p.a, s);
    prof_spans.push_back(p);
}

void adx_engine::prof_end(hipStream_t s) {
    if (!profiling || prof_spans.empty()) return;
    (void)hipEventRecord(prof_spans.back().b, s);
}

int adx_engine::profile_read(int stage, double* total_ms, int* count) {
    ADX_HIP(hipSetDevice(device));
    ADX_HIP(hipDeviceSynchronize());
    double tot = 0.0;
    int n = 0;
    for (const ProfSpan& p : prof_spans) {
        if (p.stage != stage) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { tot += ms; ++n; }
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    if (stage < 0) {
        for (const ProfSpan& p : prof_spans) { prof_pool.push_back(p.a); prof_pool.push_back(p.b); }
        prof_spans.clear();
    }
    return ADX_OK;
}

int adx_engine::set_schedule(const adx_step_coef* s, int n) {
    if (!s || n <= 0 || n > 100000) return fail(ADX_EINVAL, "set_schedule: bad argument");
    if (!finalized) return fail(ADX_ESTATE, "set_schedule: engine is not finalized");
    ADX_HIP(hipSetDevice(device));
    drop_graphs();
    steps.assign(s, s + n);
    if (steps_dev) (void)hipFree(steps_dev);
    if (steps_t_dev) (void)hipFree(steps_t_dev);
    if (temb_table) (void)hipFree(temb_table);
    steps_dev = nullptr; steps_t_dev = nullptr; temb_table = nullptr;
    static_assert(sizeof(StepCoef) == sizeof(adx_step_coef), "coefficient layouts must match");
    ADX_HIP(hipMalloc(&steps_dev, sizeof(StepCoef) * n));
    ADX_HIP(hipMalloc(&steps_t_dev, sizeof(int64_t) * n));
    ADX_HIP(hipMalloc(&temb_table, sizeof(float) * (size_t)n * cfg.d_model));
    std::vector<int64_t> tt(n);
    for (int i = 0; i < n; ++i) tt[i] = (int64_t)s[i].t;
    ADX_HIP(hipMemcpy(steps_dev, s, sizeof(StepCoef) * n, hipMemcpyHostToDevice));
    ADX_HIP(hipMemcpy(steps_t_dev, tt.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice));
    ++schedule_version;
    ADX_TRY(build_time_table(nullptr));
    ADX_HIP(hipStreamSynchronize(nullptr));
    if (ws) {                                   // per-step buffers in the workspace scale with the schedule length
        const int mb = max_batch;
        max_batch = 0;
        ADX_TRY(reserve(mb));
    }
    return ADX_OK;
}

// ------------------------------------------------------------------------------------------ C ABI
#define ADX_API extern "C" __attribute__((visibility("default")))

ADX_API int adx_abi_version(void) { return ADX_ABI_VERSION; }
ADX_API const char* adx_last_error(void) { return g_err.c_str(); }

static int done(hipError_t e, const char* what) { return e == hipSuccess ? ADX_OK : hip_fail(e, what); }

ADX_API int adx_op_timestep_embedding(const int64_t* t, int n, int dim, float* out, void* stream) {
    if (!t || !out) return fail(ADX_EINVAL, "timestep_embedding: null pointer");
    return done(timestep_embedding(t, n, dim, out, dim, (hipStream_t)stream), "timestep_embedding");
}

ADX_API int adx_op_linear(const float* x, const void* w, const float* bias, float* y, int M, int N, int K, int act,
                          int prec, void* stream) {
    if (!x || !w || !y) return fail(ADX_EINVAL, "linear: null pointer");
    if (prec != ADX_PREC_F32 && prec != ADX_PREC_BF16 && prec != ADX_PREC_BF16X3) return fail(ADX_EINVAL, "linear: unknown precision");
    if (M <= 0 || N <= 0 || K <= 0) return fail(ADX_EINVAL, "linear: empty shape");
    LinearArgs a;
    if (prec == ADX_PREC_BF16X3) a.Wlo = reinterpret_cast<const uint16_t*>(w) + (int64_t)N * K;
    a.X = x; a.ldx = K; a.W = w; a.bias = bias; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.act = act; a.prec = prec;
    return done(linear(a, (hipStream_t)stream), "linear");
}

ADX_API int adx_op_linear_staged(const float* x, const void* w, const float* bias, float* y, int M, int N, int K, int act,
                                 int prec, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !y) return fail(ADX_EINVAL, "linear_staged: null pointer");
    if (prec != ADX_PREC_BF16 && prec != ADX_PREC_BF16X3) return fail(ADX_EINVAL, "linear_staged: bf16 precisions only");
    if (M <= 0 || N <= 0 || K <= 0) return fail(ADX_EINVAL, "linear_staged: empty shape");
    if (!workspace || workspace_bytes < linear_stage_bytes(M, N, K, prec) || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return fail(ADX_EINVAL, "linear_staged: workspace missing, misaligned or smaller than adx_linear_stage_bytes(M, N, K, prec)");
    LinearArgs a;
    if (prec == ADX_PREC_BF16X3) a.Wlo = reinterpret_cast<const uint16_t*>(w) + (int64_t)N * K;
    a.X = x; a.ldx = K; a.W = w; a.bias = bias; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.act = act; a.prec = prec;
    a.xstage = workspace; a.xstage_bytes = workspace_bytes;
    return done(linear(a, (hipStream_t)stream), "linear_staged");
}

ADX_API size_t adx_linear_stage_bytes(int M, int N, int K, int prec) { return linear_stage_bytes(M, N, K, prec); }

// Operands kept in the staged kernel's own layout across calls: K-blocked bf16 planes [K/16][rows][16], the high-half plane
// followed by the low-half plane under ADX_PREC_BF16X3.  A caller that multiplies the same X or W more than once (or whose
// producer can write the planes) pays the re-layout once and then launches the product kernel alone.
ADX_API size_t adx_linear_planes_bytes(int rows, int K, int prec) {
    return (size_t)rows * K * 2 * ((prec == ADX_PREC_BF16X3 || prec == ADX_PREC_F16X2) ? 2 : 1);       // ADX_PREC_F16: one plane
}

ADX_API int adx_op_stage_planes(const float* x, int M, int K, int prec, void* planes, void* stream) {
    if (!x || !planes) return fail(ADX_EINVAL, "stage_planes: null pointer");
    if (prec != ADX_PREC_BF16 && prec != ADX_PREC_BF16X3 && prec != ADX_PREC_F16X2 && prec != ADX_PREC_F16)
        return fail(ADX_EINVAL, "stage_planes: bf16 / fp16 precisions only");
    if (M <= 0 || K <= 0 || (K % 16) != 0 || (reinterpret_cast<uintptr_t>(planes) & 15) || (reinterpret_cast<uintptr_t>(x) & 15))
        return fail(ADX_EINVAL, "stage_planes: K must be a multiple of 16 and both pointers 16-byte aligned");
    unsigned short* hi = reinterpret_cast<unsigned short*>(planes);
    const bool two = prec == ADX_PREC_BF16X3 || prec == ADX_PREC_F16X2;
    return done(stage_planes(x, K, M, K, ACT_NONE, hi, two ? hi + (size_t)M * K : nullptr, (hipStream_t)stream,
                             prec == ADX_PREC_F16X2 || prec == ADX_PREC_F16), "stage_planes");
}

ADX_API int adx_op_block_weight(const void* w, int N, int K, int prec, void* blocked, void* stream) {
    if (!w || !blocked) return fail(ADX_EINVAL, "block_weight: null pointer");
    if (prec != ADX_PREC_BF16 && prec != ADX_PREC_BF16X3 && prec != ADX_PREC_F16X2 && prec != ADX_PREC_F16)
        return fail(ADX_EINVAL, "block_weight: bf16 / fp16 precisions only");
    if (N <= 0 || K <= 0 || (K % 32) != 0 || (reinterpret_cast<uintptr_t>(blocked) & 15) || (reinterpret_cast<uintptr_t>(w) & 15))
        return fail(ADX_EINVAL, "block_weight: K must be a multiple of 32 and both pointers 16-byte aligned");
    const uint16_t* src = reinterpret_cast<const uint16_t*>(w);
    uint16_t* dst = reinterpret_cast<uint16_t*>(blocked);
    hipError_t e = linear_block_weight(src, N, K, dst, (hipStream_t)stream);
    if (e == hipSuccess && prec == ADX_PREC_BF16X3) e = linear_block_weight(src + (size_t)N * K, N, K, dst + (size_t)N * K, (hipStream_t)stream);
    return done(e, "block_weight");
}

ADX_API int adx_op_linear_planes(const void* x_planes, const void* w_blocked, const float* bias, float* y, int M, int N, int K, int act,
                                 int prec, void* stream) {
    if (!x_planes || !w_blocked || !y) return fail(ADX_EINVAL, "linear_planes: null pointer");
    const bool f16 = prec == ADX_PREC_F16X2 || prec == ADX_PREC_F16;
    if (f16) prec = prec == ADX_PREC_F16X2 ? ADX_PREC_BF16X3 : ADX_PREC_BF16;      // the fp16 forms are instances of the split / single-pass kernels
    if (!linear_planes_ok(M, N, K, prec))
        return fail(ADX_EINVAL, "linear_planes: needs a bf16 / fp16 precision, M >= 256, N >= 64 and a multiple of 4, K >= 512 and a multiple of 64, "
                                "and at least 320 tiles of 256 x 128");
    if ((reinterpret_cast<uintptr_t>(x_planes) & 15) || (reinterpret_cast<uintptr_t>(w_blocked) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
        return fail(ADX_EINVAL, "linear_planes: operands must be 16-byte aligned");
    LinearArgs a;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x_planes);
    const uint16_t* wh = reinterpret_cast<const uint16_t*>(w_blocked);
    a.Xh = xh; a.Wblk = wh; a.W = wh;                  // W / Wlo only satisfy the shape checks: with both operands staged nothing reads them
    if (prec == ADX_PREC_BF16X3) { a.Xl = xh + (size_t)M * K; a.Wblk_lo = wh + (size_t)N * K; a.Wlo = a.Wblk_lo; }
    if (f16) { a.f16 = true; a.Wblk_h = wh; a.Wblk_lo = nullptr; a.Wlo = wh; }     // one fp16 plane of W (X: two planes or one)
    a.ldx = K; a.bias = bias; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.act = act; a.prec = prec;
#ifdef ADX_MEASURE          // measurement library only: the epilogue with other outputs (results are not the documented ones)
    if (const char* o = getenv("ADX_PLANES_OUT")) {     // read per call
        if (o[0] == 'n') a.Y = nullptr;                                                     // "none": the epilogue computes and stores nothing
        if (o[0] == 'p') { a.Y = nullptr; a.Yh = reinterpret_cast<unsigned short*>(y);      // "planes": high halves (+ "planes2": low halves)
                           if (o[6] == '2') a.Yl = a.Yh + (size_t)M * N; }                  //            of the result as the next product's operand
    }
#endif
    return done(linear(a, (hipStream_t)stream), "linear_planes");
}

ADX_API int adx_op_mlp256(const float* x, const void* w1_blocked, const float* b1, int act1, const void* w2_blocked, const float* b2, int act2,
                          const float* gate, const float* res, float* y, int M, int N1, int prec, void* stream) {
    if (!x || !w1_blocked || !w2_blocked || !y) return fail(ADX_EINVAL, "mlp256: null pointer");
    if (prec != ADX_PREC_BF16 && prec != ADX_PREC_BF16X3) return fail(ADX_EINVAL, "mlp256: bf16 precisions only");
    if (M <= 0 || N1 <= 0) return fail(ADX_EINVAL, "mlp256: empty shape");
    const int D = 256;
    LinearArgs l1, l2;
    const uint16_t* w1 = reinterpret_cast<const uint16_t*>(w1_blocked);
    const uint16_t* w2 = reinterpret_cast<const uint16_t*>(w2_blocked);
    l1.X = x; l1.ldx = D; l1.W = w1; l1.Wblk = w1; l1.bias = b1; l1.act = act1; l1.M = M; l1.N = N1; l1.K = D; l1.prec = prec;
    l2.W = w2; l2.Wblk = w2; l2.bias = b2; l2.act = act2; l2.M = M; l2.N = D; l2.K = N1; l2.prec = prec;
    l2.gate = gate; l2.ldg = D; l2.res = res; l2.ldr = D; l2.Y = y; l2.ldy = D;
    if (prec == ADX_PREC_BF16X3) {
        l1.Wblk_lo = w1 + (size_t)N1 * D; l1.Wlo = l1.Wblk_lo;
        l2.Wblk_lo = w2 + (size_t)N1 * D; l2.Wlo = l2.Wblk_lo;
    }
    if (!mlp256_ok(l1, l2))
        return fail(ADX_EINVAL, "mlp256: needs M >= 128, a hidden width that is a multiple of 64 (at most 4096) and 16-byte aligned operands");
    return done(mlp256(l1, l2, (hipStream_t)stream), "mlp256");
}

ADX_API int adx_op_select_modes(const float* traj, int B, int M, int L, int D, int k, int em_iter, float* modes, float* weights,
                                void* stream) {
    if (!traj || !modes || !weights) return fail(ADX_EINVAL, "select_modes: null pointer");
    if (k <= 0 || k > M) return fail(ADX_EINVAL, "select_modes: k must be in 1..M");
    SelectModesArgs a;
    a.traj = traj; a.modes = modes; a.weights = weights; a.B = B; a.M = M; a.L = L; a.D = D; a.k = k; a.em_iter = em_iter;
    return done(select_modes(a, (hipStream_t)stream), "select_modes");
}

ADX_API int adx_op_trajectory_metrics(const float* pred, const float* label, int B, int S, int C, float* out, void* stream) {
    if (!pred || !label || !out) return fail(ADX_EINVAL, "trajectory_metrics: null pointer");
    if (S <= 0 || S > 80 || (80 % S) != 0 || (C != 2 && C != 4)) return fail(ADX_EINVAL, "trajectory_metrics: S must divide 80 and C be 2 or 4");
    return done(trajectory_metrics(pred, label, B, S, C, out, (hipStream_t)stream), "trajectory_metrics");
}

ADX_API int adx_op_to_f16(const float* x, uint16_t* y, int64_t n, void* stream) {
    if (!x || !y) return fail(ADX_EINVAL, "to_f16: null pointer");
    return done(to_f16(x, y, n, (hipStream_t)stream), "to_f16");
}

ADX_API int adx_op_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream) {
    if (!x || !y) return fail(ADX_EINVAL, "to_bf16: null pointer");
    return done(to_bf16(x, y, nullptr, n, (hipStream_t)stream), "to_bf16");
}

ADX_API int adx_op_layernorm(const float* x, const float* gamma, const float* beta, const float* shift, const float* scale,
                             int64_t ld_mod, float eps, int relu, int64_t rows, int d, float* y, void* stream) {
    if (!x || !y) return fail(ADX_EINVAL, "layernorm: null pointer");
    if ((gamma == nullptr) != (beta == nullptr) || (shift == nullptr) != (scale == nullptr))
        return fail(ADX_EINVAL, "layernorm: gamma/beta and shift/scale come in pairs");
    LayerNormArgs a;
    a.x = x; a.ldx = d; a.y = y; a.ldy = d; a.gamma = gamma; a.beta = beta; a.shift = shift; a.scale = scale; a.ldm = ld_mod;
    a.eps = eps; a.rows = (int)rows; a.d = d; a.relu = relu;
    return done(layernorm(a, (hipStream_t)stream), "layernorm");
}

static int op_attention(const float* q, const float* k, const float* v, float* o, int B, int H, int Sq, int Skv, int dh, int dv, float scale,
                        int mask_off, int prec, void* stream);

ADX_API int adx_op_attention(const float* q, const float* k, const float* v, float* o, int B, int H, int Sq, int Skv, int dh,
                             int dv, float scale, int mask_off, void* stream) {
    return op_attention(q, k, v, o, B, H, Sq, Skv, dh, dv, scale, mask_off, ADX_PREC_F32, stream);
}

ADX_API int adx_op_attention_prec(const float* q, const float* k, const float* v, float* o, int B, int H, int Sq, int Skv, int dh,
                                  int dv, float scale, int mask_off, int precision, void* stream) {
    if (precision != ADX_PREC_F32 && precision != ADX_PREC_BF16 && precision != ADX_PREC_BF16X3) return fail(ADX_EINVAL, "attention: unknown precision");
    return op_attention(q, k, v, o, B, H, Sq, Skv, dh, dv, scale, mask_off, precision, stream);
}

static int op_attention(const float* q, const float* k, const float* v, float* o, int B, int H, int Sq, int Skv, int dh, int dv, float scale,
                        int mask_off, int prec, void* stream) {
    if (!q || !k || !v || !o) return fail(ADX_EINVAL, "attention: null pointer");
    AttentionArgs a;
    a.prec = prec;
    a.Q = q; a.ldq = (int64_t)H * dh; a.q_batch_stride = (int64_t)Sq * H * dh;
    a.K = k; a.ldk = (int64_t)H * dh; a.k_batch_stride = (int64_t)Skv * H * dh;
    a.V = v; a.ldv = (int64_t)H * dv; a.v_batch_stride = (int64_t)Skv * H * dv;
    a.O = o; a.ldo = (int64_t)H * dv; a.o_batch_stride = (int64_t)Sq * H * dv;
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.dh = dh; a.dv = dv; a.scale = scale; a.mask_off = mask_off;
    hipError_t e = attention(a, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(ADX_EINVAL, "attention: unsupported shape");
    return done(e, "attention");
}

ADX_API int adx_op_normalize(const float* x, float* y, int64_t n_points, int C, void* stream) {
    if (!x || !y || (C != 2 && C != 4)) return fail(ADX_EINVAL, "normalize: bad argument");
    return done(normalize_traj(x, y, n_points, C, (hipStream_t)stream), "normalize");
}

ADX_API int adx_op_denormalize(const float* y, const float* residual, float* x, int64_t n_points, int C, void* stream) {
    if (!x || !y || (C != 2 && C != 4)) return fail(ADX_EINVAL, "denormalize: bad argument");
    return done(denormalize_traj(y, residual, x, n_points, C, (hipStream_t)stream), "denormalize");
}

ADX_API int adx_op_sched_step(const float* x, const float* model_out, const float* z, float* x_next, float* x0_out, int64_t n,
                              const adx_step_coef* coef_dev, void* stream) {
    if (!x || !model_out || !x_next || !coef_dev) return fail(ADX_EINVAL, "sched_step: null pointer");
    return done(sched_step(x, model_out, z, x_next, x0_out, n, reinterpret_cast<const StepCoef*>(coef_dev), (hipStream_t)stream),
                "sched_step");
}

ADX_API int adx_engine_create(const adx_model_config* cfg, adx_engine** out) {
    if (!cfg || !out) return fail(ADX_EINVAL, "engine_create: null pointer");
    const bool f16_single = cfg->precision == ADX_PREC_F16;
    if (cfg->precision != ADX_PREC_F32 && cfg->precision != ADX_PREC_BF16 && cfg->precision != ADX_PREC_BF16X3 && cfg->precision != ADX_PREC_F16X2 &&
        !f16_single)
        return fail(ADX_EINVAL, "engine_create: unknown precision");
    if (cfg->precision == ADX_PREC_F16X2 && cfg->variant != ADX_VARIANT_DIT)
        return fail(ADX_EINVAL, "engine_create: ADX_PREC_F16X2 is a precision of the DiT engine (its operand-plane products)");
    if (f16_single && cfg->variant == ADX_VARIANT_MMDIT)
        return fail(ADX_EINVAL, "engine_create: ADX_PREC_F16 is a precision of the DiT engine (its operand-plane products) and of the refiner engine (its per-step kernels)");
    adx_model_config base = *cfg;
    const bool f16x2 = cfg->precision == ADX_PREC_F16X2 || f16_single;
    if (f16x2) base.precision = ADX_PREC_BF16X3;
    cfg = &base;
    if (cfg->d_model <= 0 || cfg->n_head <= 0 || cfg->d_model % cfg->n_head) return fail(ADX_EINVAL, "engine_create: d_model must be a positive multiple of n_head");
    if (cfg->horizon <= 0 || cfg->in_dim <= 0 || cfg->out_dim <= 0 || cfg->n_layers <= 0) return fail(ADX_EINVAL, "engine_create: bad dimensions");
    adx_engine* e = nullptr;
    if (cfg->variant == ADX_VARIANT_A) e = make_engine_a(*cfg);
    else if (cfg->variant == ADX_VARIANT_DIT || cfg->variant == ADX_VARIANT_MMDIT) e = make_engine_dit(*cfg);
    else return fail(ADX_EINVAL, "engine_create: unknown variant");
    if (!e) return ADX_EINVAL;   // message set by the factory
    e->f16x2 = f16x2;
    e->f16_single = f16_single;
    hipError_t he = hipGetDevice(&e->device);
    if (he != hipSuccess) { delete e; return hip_fail(he, "hipGetDevice"); }
    *out = e;
    return ADX_OK;
}

ADX_API int adx_engine_destroy(adx_engine* e) {
    if (!e) return ADX_OK;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    delete e;
    return ADX_OK;
}

ADX_API int adx_engine_set_tensor(adx_engine* e, const char* name, const float* data, const int64_t* shape_host, int ndim) {
    if (!e) return fail(ADX_EINVAL, "set_tensor: null engine");
    return e->set_tensor(name, data, shape_host, ndim);
}

ADX_API int adx_engine_finalize(adx_engine* e) {
    if (!e) return fail(ADX_EINVAL, "finalize: null engine");
    return e->finalize();
}

ADX_API int adx_engine_profile(adx_engine* e, int enable) {
    if (!e) return fail(ADX_EINVAL, "profile: null engine");
    e->profiling = enable != 0;
    return ADX_OK;
}

ADX_API int adx_engine_profile_read(adx_engine* e, int stage, double* total_ms, int* count) {
    if (!e) return fail(ADX_EINVAL, "profile_read: null engine");
    return e->profile_read(stage, total_ms, count);
}

ADX_API int64_t adx_engine_capture_count(const adx_engine* e) { return e ? e->n_captures : 0; }

ADX_API int adx_engine_set_noise_seed(adx_engine* e, uint64_t seed, int64_t row0) {
    if (!e) return fail(ADX_EINVAL, "set_noise_seed: null engine");
    if (row0 < 0) return fail(ADX_EINVAL, "set_noise_seed: negative row offset");
    e->noise_seed = seed;
    e->noise_row0 = (uint64_t)row0;
    return ADX_OK;
}

ADX_API int adx_op_randn(float* out, int n_lead, int B, int row_len, uint64_t seed, int64_t row0, uint64_t* ctl_scratch, void* stream) {
    if (!out || !ctl_scratch) return fail(ADX_EINVAL, "randn: null pointer");
    if (n_lead <= 0 || B <= 0 || row_len <= 0 || row0 < 0) return fail(ADX_EINVAL, "randn: bad shape");
    hipError_t he = set_noise_ctl(ctl_scratch, seed, (uint64_t)row0, (hipStream_t)stream);
    if (he == hipSuccess) he = randn_rows(ctl_scratch, out, n_lead, B, row_len, (hipStream_t)stream);
    return done(he, "randn");
}

ADX_API int adx_op_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp,
                            float* xt, float* v, int n_rows, int row_len, void* stream) {
    if (!x0 || !noise || !t || !sqrt_acp || !sqrt_1m_acp || !xt) return fail(ADX_EINVAL, "q_sample: null pointer");
    if (n_rows <= 0 || row_len <= 0) return fail(ADX_EINVAL, "q_sample: empty shape");
    return done(q_sample(x0, noise, t, sqrt_acp, sqrt_1m_acp, xt, v, n_rows, row_len, (hipStream_t)stream), "q_sample");
}

ADX_API int adx_op_weighted_mse(const float* out, const float* target, const int64_t* t, const float* weight, float* loss_rows,
                                float* loss_mean, int n_rows, int row_len, void* stream) {
    if (!out || !target || !t || !weight || !loss_rows) return fail(ADX_EINVAL, "weighted_mse: null pointer");
    if (n_rows <= 0 || row_len <= 0) return fail(ADX_EINVAL, "weighted_mse: empty shape");
    return done(weighted_mse(out, target, t, weight, loss_rows, loss_mean, n_rows, row_len, (hipStream_t)stream), "weighted_mse");
}

ADX_API int adx_op_traj_loss(const float* logits, const float* label, int B, int T, int C, int C_label, int smooth_l1, float* part, float* out,
                             void* stream) {
    if (!logits || !label || !part || !out) return fail(ADX_EINVAL, "traj_loss: null pointer");
    if (B <= 0 || T <= 0 || C < 2 || C_label < 2) return fail(ADX_EINVAL, "traj_loss: needs B, T > 0 and at least two channels");
    return done(traj_loss(logits, label, part, out, B, T, C, C_label, smooth_l1 != 0, (hipStream_t)stream), "traj_loss");
}

ADX_API int adx_engine_reserve(adx_engine* e, int max_batch) {
    if (!e) return fail(ADX_EINVAL, "reserve: null engine");
    return e->reserve(max_batch);
}

ADX_API int64_t adx_engine_workspace_bytes(const adx_engine* e) { return e ? (int64_t)e->ws_bytes : 0; }

ADX_API int adx_engine_set_schedule(adx_engine* e, const adx_step_coef* steps_host, int n_steps) {
    if (!e) return fail(ADX_EINVAL, "set_schedule: null engine");
    return e->set_schedule(steps_host, n_steps);
}

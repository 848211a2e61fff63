// Engine base: parameter store, activation workspace, schedule tables and hipGraph cache.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/adx.h"
#include "adx_ops.h"

namespace adx {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define ADX_HIP(call)                                   \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) return ::adx::hip_fail(e__, #call); \
    } while (0)
#define ADX_TRY(call)               \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != ADX_OK) return rc__; \
    } while (0)

struct Param {
    float* f32 = nullptr;
    uint16_t* bf16 = nullptr;
    uint16_t* bf16lo = nullptr;   // bf16(w - bf16(w)), for the split-operand GEMM
    uint16_t* blk = nullptr;      // the same two planes in the K-blocked layout [K/16][N][16] the staged GEMM copies from (built by
    uint16_t* blk_lo = nullptr;   // finalize() for engines that ask for it), so that no call re-lays the weights
    uint16_t* blk_h = nullptr;    // the weight rounded once to fp16 in the same K-blocked layout (engines created with ADX_PREC_F16X2)
    std::vector<int64_t> shape;
    int64_t numel = 0;
};

struct ParamSpec {
    std::string name;
    std::vector<int64_t> shape;
};

}  // namespace adx

// The opaque handle of the C ABI.
struct adx_engine {
    adx_model_config cfg{};
    bool f16x2 = false;              // created with ADX_PREC_F16X2: cfg.precision reads ADX_PREC_BF16X3 (what every kernel but the
                                     // operand-plane products runs as), planes hold fp16 halves and those products take two terms
    bool f16_single = false;         // created with ADX_PREC_F16: as f16x2, but the planes hold high halves only (one term per product)
    int device = 0;
    bool finalized = false;
    std::map<std::string, adx::Param> params;

    // activation workspace (one allocation, carved by a bump pointer at reserve())
    char* ws = nullptr;
    size_t ws_bytes = 0;
    void* xstage = nullptr;          // scratch for the staged (LDS-DMA) form of the large GEMMs, carved from ws by the engine
    size_t xstage_bytes = 0;
    int max_batch = 0;

    // schedule
    std::vector<adx_step_coef> steps;
    adx::StepCoef* steps_dev = nullptr;
    int64_t* steps_t_dev = nullptr;     // int64 timesteps, one per step
    float* temb_table = nullptr;        // [n_steps][d_model] time-conditioning vector per step
    int schedule_version = 0;

    // per-kernel timing of a call (adx_engine_profile): while enabled, calls run without graph capture and the marked
    // launches are bracketed by events on the call's stream
    bool profiling = false;
    struct ProfSpan { hipEvent_t a, b; int stage; };
    std::vector<ProfSpan> prof_spans;
    std::vector<hipEvent_t> prof_pool;
    void prof_begin(int stage, hipStream_t s);
    void prof_end(hipStream_t s);
    int profile_read(int stage, double* total_ms, int* count);     // synchronises; clears the collected spans when stage < 0

    // graphs
    hipStream_t cap_stream = nullptr;
    std::map<std::string, hipGraphExec_t> graphs;
    int64_t n_captures = 0;              // graphs captured since the engine was created (adx_engine_capture_count)

    // Callers that hand in the same buffers on every call get a graph keyed on those pointers.  A caller whose pointers
    // change between calls of the same shape (fresh tensors per batch) is switched to STAGED mode for that shape: its
    // inputs are copied into engine-owned buffers on the call's stream, ONE graph captured on those buffers is replayed,
    // and the result is copied out -- so the capture happens once, not per call.
    struct IoMode { std::string ptr_key; bool staged = false; };
    std::map<std::string, IoMode> io_modes;
    char* stage_buf = nullptr;
    size_t stage_bytes = 0;
    int ensure_stage(size_t bytes);       // (re)allocates stage_buf; never inside a capture
    // decides the mode of this call; returns true when the call must go through the staging buffers
    bool use_staging(const std::string& shape_key, const std::string& ptr_key, unsigned flags);

    // device-side noise for callers that pass none: {seed, global row of batch row 0} in device memory (read by the
    // generator kernel inside the graph, rewritten before every launch), and the buffer the generator fills
    uint64_t noise_seed = 0, noise_row0 = 0;
    uint64_t* noise_ctl = nullptr;
    float* noise_buf = nullptr;
    size_t noise_bytes = 0;
    int ensure_noise(size_t bytes);
    int push_noise_ctl(hipStream_t s);    // enqueue the current seed / row offset
    int n_schedule_draws() const;         // noise tensors the schedule reads after the initial one

    virtual ~adx_engine();
    virtual std::vector<adx::ParamSpec> expected_params() const = 0;
    virtual int carve_workspace(int max_batch, bool dry_run, size_t* bytes) = 0;
    virtual int build_time_table(hipStream_t s) = 0;   // fills temb_table and every weight-derived table for the current schedule
    virtual void on_workspace_realloc() {}             // caches that live in the workspace are gone
    virtual bool wants_blocked_weights() const { return false; }   // keep K-blocked copies of the 2-D parameters (staged GEMM operands)

    int set_tensor(const char* name, const float* data, const int64_t* shape, int ndim);
    int finalize();
    int reserve(int batch);
    int set_schedule(const adx_step_coef* steps_host, int n);
    void drop_graphs();

    // parameter access
    const adx::Param* find(const std::string& name) const;
    const float* P(const std::string& name) const;                         // f32 pointer (biases, gains, tables)
    const void* W(const std::string& name, int64_t row0, int prec) const;     // GEMM operand: f32 or bf16 copy
    bool has(const std::string& name) const { return params.count(name) != 0; }

    // y = act(x W[row0:row0+N]^T + b[row0:row0+N]) helper; prec < 0 selects the engine precision
    adx::LinearArgs lin(const float* X, int64_t ldx, const std::string& wname, int64_t row0, int N, int K,
                        const std::string& bname, float* Y, int64_t ldy, int M, int act = adx::ACT_NONE, int prec = -1) const;

    template <class F>
    int run(const std::string& key, unsigned flags, hipStream_t stream, F&& body);
};

template <class F>
int adx_engine::run(const std::string& key, unsigned flags, hipStream_t stream, F&& body) {
    if ((flags & ADX_FLAG_NO_GRAPH) || profiling) return body(stream);
    auto it = graphs.find(key);
    if (it == graphs.end()) {
        if (graphs.size() >= 16) {      // a caller that never reuses its buffers: bound the cache (a replay may still be running)
            ADX_HIP(hipDeviceSynchronize());
            drop_graphs();
        }
        if (!cap_stream) ADX_HIP(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
        ADX_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = body(cap_stream);
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(cap_stream, &graph);
        if (rc != ADX_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess) return adx::hip_fail(e, "hipStreamEndCapture");
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return adx::hip_fail(e, "hipGraphInstantiate");
        ++n_captures;
        it = graphs.emplace(key, exec).first;
    }
    ADX_HIP(hipGraphLaunch(it->second, stream));
    return ADX_OK;
}

adx_engine* make_engine_a(const adx_model_config& cfg);
adx_engine* make_engine_dit(const adx_model_config& cfg);

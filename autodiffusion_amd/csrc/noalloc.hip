// An entry point that allocates nothing, for callers that cannot call hipMalloc (code under hipGraph capture, tight serving
// loops, harnesses that forbid allocation): adx_op_linear_noalloc is adx_op_linear_staged with the staging workspace taken
// from a fixed device array of this library instead of from the caller.
//
// The array (`scratch`, ADX_NOALLOC_SCRATCH_BYTES, 256-byte aligned) exists once per GPU the library runs on and stays
// until the process ends.  In host code the array's NAME is a host-side stand-in, not a device address; its address on a
// device is obtained once per device with hipGetSymbolAddress (that device current) and kept in g_scratch_ptr -- only
// that pointer is ever handed to kernels.  Every call on a GPU shares that GPU's array: the staged GEMM overwrites the
// part it uses before reading it (nothing has to start at zero), and calls that overlap in time on one GPU race.
#include "engine.h"

#include <mutex>

using namespace adx;

#define ADX_NOALLOC_SCRATCH_BYTES (64u << 20)

static __device__ __align__(256) unsigned char scratch[ADX_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

// Address of `scratch` on the current device (looked up on the first call per device; no allocation).
int scratch_on_current_device(void** out) {
    int dev = 0;
    ADX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) { set_error("linear_noalloc: device index out of range"); return ADX_EINVAL; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        ADX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { set_error("linear_noalloc: hipGetSymbolAddress returned a null address"); return ADX_EHIP; }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return ADX_OK;
}

}  // namespace

#define ADX_API extern "C" __attribute__((visibility("default")))

ADX_API size_t adx_noalloc_scratch_bytes(void) { return ADX_NOALLOC_SCRATCH_BYTES; }

// Per-device set-up of adx_op_linear_noalloc, to be called once with the device current and with NO stream capture in
// progress: looks up the fixed array's address on this device (the first lookup loads the library's code object) and sets the
// staged kernels' attributes.  adx_op_linear_noalloc does the same on its first call on a device, so a caller that never
// captures may skip this; a caller that captures must call it (or make one eager call) first.
ADX_API int adx_noalloc_init(void) {
    void* ptr = nullptr;
    ADX_TRY(scratch_on_current_device(&ptr));
    const hipError_t e = linear_staged_prepare();
    return e == hipSuccess ? ADX_OK : hip_fail(e, "noalloc_init");
}

ADX_API int adx_op_linear_noalloc(const float* x, const void* w, const float* bias, float* y, int M, int N, int K, int act, int prec,
                                  void* stream) {
    if (!x || !w || !y) { set_error("linear_noalloc: null pointer"); return ADX_EINVAL; }
    if (prec != ADX_PREC_BF16 && prec != ADX_PREC_BF16X3) { set_error("linear_noalloc: bf16 precisions only"); return ADX_EINVAL; }
    if (M <= 0 || N <= 0 || K <= 0) { set_error("linear_noalloc: empty shape"); return ADX_EINVAL; }
    const size_t need = linear_stage_bytes(M, N, K, prec);
    if (need > ADX_NOALLOC_SCRATCH_BYTES) {
        set_error("linear_noalloc: the bf16 planes of this problem need " + std::to_string(need) + " bytes of scratch, the fixed device array holds " +
                  std::to_string((size_t)ADX_NOALLOC_SCRATCH_BYTES) + " (use adx_op_linear_staged with a workspace of your own)");
        return ADX_EINVAL;
    }
    LinearArgs a;
    if (prec == ADX_PREC_BF16X3) a.Wlo = reinterpret_cast<const uint16_t*>(w) + (int64_t)N * K;
    a.X = x; a.ldx = K; a.W = w; a.bias = bias; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.act = act; a.prec = prec;
    if (!linear_big_ok(a)) {
        set_error("linear_noalloc: the staged kernel needs M >= 256, N >= 64, K a multiple of 64 and 16-byte aligned operands");
        return ADX_EINVAL;
    }
    void* ptr = nullptr;
    ADX_TRY(scratch_on_current_device(&ptr));
    a.xstage = ptr; a.xstage_bytes = ADX_NOALLOC_SCRATCH_BYTES;
    a.force_staged = true;
    const hipError_t e = linear(a, (hipStream_t)stream);
    return e == hipSuccess ? ADX_OK : hip_fail(e, "linear_noalloc");
}

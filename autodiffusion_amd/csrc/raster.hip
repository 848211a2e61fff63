// Raster -> patch matrix for the ViT conditioning encoder (phase 1 of generate(): NuplanRasterizeEncoder.forward,
// transformer4planning/models/encoder/nuplan_raster_encoder.py:178-187, with cat_raster_seq, models/utils.py:14-41).
//
// The reference permutes the channels-last raster (B, H, W, C) to (B, C, W, H) -- pixel_values[b][c][i][j] = raster[b][j][i][c] --,
// rebuilds one image per context frame out of the shared channels (route, 20 road types, 4 traffic-light types) and that frame's
// 8 agent channels (agent channel a of frame f sits at base + f + a * framenum), and hands (B * framenum, C', W, H) images to a
// ViT whose patch embedding is a Conv2d(C', hidden, kernel = stride = 16).  A convolution whose stride equals its kernel is a
// product of the (c', ky, kx)-flattened patch with the flattened filter, so the three steps become ONE gather into the patch
// matrix  P[(b, f, py, px)][(c' * 16 + ky) * 16 + kx] = raster[b][px * 16 + kx][py * 16 + ky][chan(c', f)]  followed by the
// library's GEMM.
//
// One workgroup per (b, py, px) patch position: the 16 x 16 pixels x C channels of the patch are read once, as 16 contiguous
// runs of 16 C floats (all frames share them), into LDS, and written out as framenum rows of C' * 256 contiguous floats.
#include "adx_ops.h"
#include "engine.h"

namespace adx {
namespace {

constexpr int PATCH = 16;

__global__ __launch_bounds__(256) void raster_patches_kernel(const float* __restrict__ raster, int H, int W, int C, int framenum, int traffic,
                                                             int route, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [kx][ky][C] (+1 float per kx: the gather below walks kx fastest): pixel (px*16 + kx, py*16 + ky)
    const int nx = H / PATCH, ny = W / PATCH;                               // px runs over the raster's first spatial axis
    const int px = blockIdx.x % nx, py = (blockIdx.x / nx) % ny, b = blockIdx.x / (nx * ny);
    const int run = PATCH * C;                                              // floats of one kx: 16 consecutive ky, all channels
    for (int kx = 0; kx < PATCH; ++kx) {
        const float* src = raster + (((int64_t)b * H + px * PATCH + kx) * W + py * PATCH) * C;
        for (int i = threadIdx.x; i < run; i += 256) tile[kx * (run + 1) + i] = src[i];
    }
    __syncthreads();
    const int shared = route + 20 + (traffic ? 4 : 0);                     // channels every frame's image starts with
    const int cf = shared + 8;                                              // channels of one frame's image
    const int agent0 = route + 24;                                          // first agent channel of the raster
    const int K = cf * PATCH * PATCH;
    for (int f = 0; f < framenum; ++f) {
        float* dst = out + ((((int64_t)b * framenum + f) * ny + py) * nx + px) * K;
        for (int i = threadIdx.x; i < K; i += 256) {
            const int kx = i & 15, ky = (i >> 4) & 15, c = i >> 8;
            const int ch = c < shared ? c : agent0 + f + (c - shared) * framenum;      // the shared channels are the raster's first ones
            dst[i] = tile[kx * (run + 1) + ky * C + ch];
        }
    }
}

}  // namespace

hipError_t raster_patches(const float* raster, int B, int H, int W, int C, int framenum, int traffic, float* out, hipStream_t stream) {
    const int route = C - 8 * framenum - 24;
    if (!raster || !out || B <= 0 || framenum <= 0 || route < 0 || (H % PATCH) != 0 || (W % PATCH) != 0) return hipErrorInvalidValue;
    const size_t smem = (size_t)PATCH * (PATCH * C + 1) * sizeof(float);
    if (smem > 150 * 1024) return hipErrorInvalidValue;
    const int64_t blocks = (int64_t)B * (H / PATCH) * (W / PATCH);
    if (blocks > (1 << 30)) return hipErrorInvalidValue;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&raster_patches_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(raster_patches_kernel, dim3((unsigned)blocks), dim3(256), smem, stream, raster, H, W, C, framenum, traffic, route, out);
    return hipGetLastError();
}

}  // namespace adx

using namespace adx;

#define ADX_API extern "C" __attribute__((visibility("default")))

ADX_API int adx_op_raster_patches(const float* raster, int B, int H, int W, int C, int framenum, int traffic, float* out, void* stream) {
    if (!raster || !out) { set_error("raster_patches: null pointer"); return ADX_EINVAL; }
    if (B <= 0 || framenum <= 0 || C - 8 * framenum - 24 < 0 || (H % 16) != 0 || (W % 16) != 0 || C > 149) {
        set_error("raster_patches: needs H and W multiples of 16, C >= 8 * framenum + 24 (route + 20 road + 4 traffic-light + 8 agent channels per "
                  "frame) and at most 149 channels");
        return ADX_EINVAL;
    }
    const hipError_t e = raster_patches(raster, B, H, W, C, framenum, traffic, out, (hipStream_t)stream);
    return e == hipSuccess ? ADX_OK : hip_fail(e, "raster_patches");
}

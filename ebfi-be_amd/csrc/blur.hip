// Blur-level maps computed on the device (the reference round-trips through host OpenCV inside
// model.forward: myutils/utils.py:34-49 Frame2Lap, :15-31 Frame2DCP).
// Arithmetic restated from OpenCV's 8-bit algorithms (4.x constants); PARITY UNPINNED: cv2 is not
// in this image and the reference pins no version -- see oracle/blur_ref.py.
#include "common.hpp"

using namespace ebfi;

namespace {

__device__ __forceinline__ int gray_u8(const float *__restrict__ f, int64_t plane, int64_t idx) {
    // (frame*255).astype(uint8) then BGR2GRAY on RGB-ordered data: channel 0 gets the "B" weight
    const int c0 = (int)(unsigned char)(int)(f[idx] * 255.f);
    const int c1 = (int)(unsigned char)(int)(f[idx + plane] * 255.f);
    const int c2 = (int)(unsigned char)(int)(f[idx + 2 * plane] * 255.f);
    return (c0 * 3735 + c1 * 19235 + c2 * 9798 + (1 << 14)) >> 15;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

__global__ void frame2lap_kernel(const float *__restrict__ frame, float *__restrict__ out, int B, int H, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = (int64_t)H * W;
    if (idx >= B * plane) return;
    const int b = (int)(idx / plane);
    const int y = (int)((idx - b * plane) / W), x = (int)(idx - b * plane - (int64_t)y * W);
    const float *f = frame + (int64_t)b * 3 * plane;
    const int yu = reflect101(y - 1, H), yd = reflect101(y + 1, H);
    const int xl = reflect101(x - 1, W), xr = reflect101(x + 1, W);
    const int c = gray_u8(f, plane, (int64_t)y * W + x);
    const int lap = gray_u8(f, plane, (int64_t)yu * W + x) + gray_u8(f, plane, (int64_t)yd * W + x) +
                    gray_u8(f, plane, (int64_t)y * W + xl) + gray_u8(f, plane, (int64_t)y * W + xr) - 4 * c;
    out[idx] = (float)lap;
}

// pass 1: channel minimum + horizontal clipped-window minimum
__global__ void dcp_rows_kernel(const float *__restrict__ frame, float *__restrict__ tmp, int B, int H, int W, int r) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = (int64_t)H * W;
    if (idx >= B * plane) return;
    const int b = (int)(idx / plane);
    const int y = (int)((idx - b * plane) / W), x = (int)(idx - b * plane - (int64_t)y * W);
    const float *f = frame + (int64_t)b * 3 * plane + (int64_t)y * W;
    float m = INFINITY;
    const int x0 = max(0, x - r), x1 = min(W - 1, x + r);
    for (int xx = x0; xx <= x1; ++xx) m = fminf(m, fminf(fminf(f[xx], f[xx + plane]), f[xx + 2 * plane]));
    tmp[idx] = m;
}

__global__ void dcp_cols_kernel(const float *__restrict__ tmp, float *__restrict__ out, int B, int H, int W, int r) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = (int64_t)H * W;
    if (idx >= B * plane) return;
    const int b = (int)(idx / plane);
    const int y = (int)((idx - b * plane) / W), x = (int)(idx - b * plane - (int64_t)y * W);
    const float *t = tmp + (int64_t)b * plane + x;
    float m = INFINITY;
    const int y0 = max(0, y - r), y1 = min(H - 1, y + r);
    for (int yy = y0; yy <= y1; ++yy) m = fminf(m, t[(int64_t)yy * W]);
    out[idx] = m;
}

}  // namespace

extern "C" int ebfi_frame2lap(const float *frame, float *out, int B, int H, int W, void *stream) {
    if (!frame || !out || B < 0 || H <= 0 || W <= 0) return fail(EBFI_ERR_ARG, "frame2lap: bad argument");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * W;
    {
        ProfScope ps("frame2lap", st, 0.0, 16.0 * total);
        hipLaunchKernelGGL(frame2lap_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, frame, out, B, H, W);
    }
    return check_launch("frame2lap");
}

extern "C" int ebfi_frame2dcp(const float *frame, float *out, float *scratch, int B, int H, int W, int window,
                              void *stream) {
    if (!frame || !out || !scratch || B < 0 || H <= 0 || W <= 0 || window < 1 || window % 2 == 0)
        return fail(EBFI_ERR_ARG, "frame2dcp: bad argument (window must be odd)");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * W;
    {
        ProfScope ps("frame2dcp_rows", st);
        hipLaunchKernelGGL(dcp_rows_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, frame, scratch, B, H, W,
                           window / 2);
    }
    if (int rc = check_launch("frame2dcp_rows")) return rc;
    {
        ProfScope ps("frame2dcp_cols", st);
        hipLaunchKernelGGL(dcp_cols_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, scratch, out, B, H, W,
                           window / 2);
    }
    return check_launch("frame2dcp_cols");
}

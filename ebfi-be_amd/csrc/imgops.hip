// Small image operators of the training loss, as single kernels instead of chains of shifted-slice ops.
//
// gauss5: the 5x5 binomial blur ([1,4,6,4,1]/16 twice, x `factor`) with reflect padding that the reference's
// LaplacianPyramid applies 8 times per pyramid through a depthwise conv2d (loss/restore.py:149-199, GaussianConv).
// Forward gathers the 25 taps through the reflected index; backward is the exact adjoint: an input pixel receives
// from its own position and -- within 2 pixels of a border -- from the mirrored virtual position as well.
#include "common.hpp"

using namespace ebfi;

namespace {

__device__ __forceinline__ int reflect(int i, int n) {   // torch 'reflect' (no edge repeat); |overshoot| <= 2 < n
    if (i < 0) return -i;
    if (i >= n) return 2 * (n - 1) - i;
    return i;
}

__global__ void gauss5_fwd_kernel(const float *__restrict__ x, float *__restrict__ out, int64_t planes, int H, int W,
                                  float factor) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hw = (int64_t)H * W;
    if (idx >= planes * hw) return;
    const int64_t p = idx / hw;
    const int y = (int)((idx - p * hw) / W), xx = (int)(idx - p * hw - (int64_t)y * W);
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    const float *src = x + p * hw;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float *row = src + (int64_t)reflect(y + i - 2, H) * W;
        float h = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) h += k[j] * row[reflect(xx + j - 2, W)];
        acc += k[i] * h;
    }
    out[idx] = acc * factor;
}

// adjoint: gin[y,x] = factor * sum over virtual rows vy in V(y), virtual cols vx in V(x), taps (i,j) of
//          k[i] k[j] gout[vy - i + 2, vx - j + 2]   (positions outside the image contribute nothing)
__global__ void gauss5_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin, int64_t planes, int H, int W,
                                  float factor) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hw = (int64_t)H * W;
    if (idx >= planes * hw) return;
    const int64_t p = idx / hw;
    const int y = (int)((idx - p * hw) / W), xx = (int)(idx - p * hw - (int64_t)y * W);
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    const float *src = gout + p * hw;
    // virtual positions that the reflect padding maps onto this pixel
    int vy[3], vx[3], ny = 0, nx = 0;
    vy[ny++] = y;
    if (y >= 1 && y <= 2) vy[ny++] = -y;
    if (y >= H - 3 && y <= H - 2) vy[ny++] = 2 * (H - 1) - y;
    vx[nx++] = xx;
    if (xx >= 1 && xx <= 2) vx[nx++] = -xx;
    if (xx >= W - 3 && xx <= W - 2) vx[nx++] = 2 * (W - 1) - xx;
    float acc = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int i = 0; i < 5; ++i) {
            const int oy = vy[a] - i + 2;
            if (oy < 0 || oy >= H) continue;
            const float *row = src + (int64_t)oy * W;
            float h = 0.f;
            for (int b = 0; b < nx; ++b)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int ox = vx[b] - j + 2;
                    if (ox >= 0 && ox < W) h += k[j] * row[ox];
                }
            acc += k[i] * h;
        }
    gin[idx] = acc * factor;
}

}  // namespace

extern "C" int ebfi_gauss5_forward(const float *input, float *output, int64_t planes, int H, int W, float factor,
                                   void *stream) {
    if (!input || !output || planes < 0 || H < 3 || W < 3) return fail(EBFI_ERR_ARG, "gauss5_forward: bad argument (H, W >= 3)");
    if (planes == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = planes * H * W;
    {
        ProfScope ps("gauss5_fwd", st, 0.0, 8.0 * total);
        hipLaunchKernelGGL(gauss5_fwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, input, output, planes, H, W,
                           factor);
    }
    return check_launch("gauss5_fwd");
}

extern "C" int ebfi_gauss5_backward(const float *grad_output, float *grad_input, int64_t planes, int H, int W, float factor,
                                    void *stream) {
    if (!grad_output || !grad_input || planes < 0 || H < 3 || W < 3)
        return fail(EBFI_ERR_ARG, "gauss5_backward: bad argument (H, W >= 3)");
    if (planes == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = planes * H * W;
    {
        ProfScope ps("gauss5_bwd", st, 0.0, 8.0 * total);
        hipLaunchKernelGGL(gauss5_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, grad_output, grad_input,
                           planes, H, W, factor);
    }
    return check_launch("gauss5_bwd");
}

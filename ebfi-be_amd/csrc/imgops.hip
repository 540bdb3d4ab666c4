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

// ---------------------------------------------------------------------------------------------------------
// census (Ternary) loss, reference loss/restore.py:108-145 with patch 7:
//   g = mean_c(img);  d_k(q) = g(q + k) - g(q) over the 49 taps k (zero outside the image);
//   t = d / sqrt(0.81 + d^2);  u = t_x - t_y;  dist(q) = mean_k u^2 / (0.1 + u^2);
//   loss = sum_{q interior by 3} dist(q) / (B*H*W).
// The reference materialises five [B,49,H,W] tensors per image; here a 16x16 pixel tile keeps the two gray
// images (halo 3, zeros outside the image) in LDS, the forward writes one partial sum per tile (reduced in
// fixed order by the caller) and the backward recomputes the 49-tap terms instead of storing them.
constexpr int CT = 16, CR = 3, CTW = CT + 2 * CR;   // tile, halo, LDS tile width

__device__ __forceinline__ void census_stage(const float *__restrict__ x, const float *__restrict__ y, int C, int H, int W,
                                             int b, int ty0, int tx0, float (*gx)[CTW + 1], float (*gy)[CTW + 1]) {
    const int64_t hw = (int64_t)H * W;
    const float invc = 1.f / (float)C;
    for (int i = threadIdx.x; i < CTW * CTW; i += CT * CT) {
        const int ly = i / CTW, lx = i - ly * CTW;
        const int py = ty0 + ly - CR, px = tx0 + lx - CR;
        float sx = 0.f, sy = 0.f;
        if (py >= 0 && py < H && px >= 0 && px < W) {
            const int64_t o = (int64_t)b * C * hw + (int64_t)py * W + px;
            for (int c = 0; c < C; ++c) {
                sx += x[o + c * hw];
                sy += y[o + c * hw];
            }
            sx *= invc;
            sy *= invc;
        }
        gx[ly][lx] = sx;
        gy[ly][lx] = sy;
    }
}

__device__ __forceinline__ float census_t(float d) { return d * rsqrtf(0.81f + d * d); }

__global__ __launch_bounds__(CT *CT) void census_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            float *__restrict__ partial, int C, int H, int W) {
    __shared__ float gx[CTW][CTW + 1], gy[CTW][CTW + 1];
    __shared__ float red[4];
    const int b = blockIdx.z, ty0 = blockIdx.y * CT, tx0 = blockIdx.x * CT;
    census_stage(x, y, C, H, W, b, ty0, tx0, gx, gy);
    __syncthreads();
    const int ly = threadIdx.x / CT, lx = threadIdx.x % CT;
    const int py = ty0 + ly, px = tx0 + lx;
    float acc = 0.f;
    if (py >= CR && py < H - CR && px >= CR && px < W - CR) {
        const float cx = gx[ly + CR][lx + CR], cy = gy[ly + CR][lx + CR];
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float u = census_t(gx[ly + i][lx + j] - cx) - census_t(gy[ly + i][lx + j] - cy);
                const float q = u * u;
                acc += q / (0.1f + q);
            }
        acc *= (1.f / 49.f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dL/dd of one (pixel q, tap) term given the two differences; scale applied by the caller
__device__ __forceinline__ float census_dterm(float dx, float dy) {
    const float rx = rsqrtf(0.81f + dx * dx);
    const float u = dx * rx - census_t(dy);
    const float den = 0.1f + u * u;
    return (0.2f * u / (den * den)) * (0.81f * rx * rx * rx);
}

// grad_x[b,c,p] = (1/C) * scale * ( sum_k G(p-k, k) - sum_k G(p, k) ),  scale = grad_loss / (49*B*H*W),
// G(q,k) = [q interior] * census_dterm(gx(q+k) - gx(q), gy(q+k) - gy(q)).
// census_dterm is odd in (dx, dy) and the 49 taps are symmetric, so the first sum -- this pixel as tap k of the centre p - k --
// is -sum_k [p+k interior] D(k) with D(k) = census_dterm(gx(p+k) - gx(p), gy(p+k) - gy(p)), the same 49 terms as the second:
// grad = -scale/C * sum_k D(k) * ([p+k interior] + [p interior]) -- 49 evaluations (two rsqrt and a division each) instead of 98.
__global__ __launch_bounds__(CT *CT) void census_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ grad_loss, float *__restrict__ grad_x,
                                                            int C, int H, int W, float norm) {
    __shared__ float gx[CTW][CTW + 1], gy[CTW][CTW + 1];
    const int b = blockIdx.z, ty0 = blockIdx.y * CT, tx0 = blockIdx.x * CT;
    census_stage(x, y, C, H, W, b, ty0, tx0, gx, gy);
    __syncthreads();
    const int ly = threadIdx.x / CT, lx = threadIdx.x % CT;
    const int py = ty0 + ly, px = tx0 + lx;
    if (py >= H || px >= W) return;
    const float cx = gx[ly + CR][lx + CR], cy = gy[ly + CR][lx + CR];
    const float self_in = (py >= CR && py < H - CR && px >= CR && px < W - CR) ? 1.f : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int qy = py + (i - CR);
        const bool row_in = qy >= CR && qy < H - CR;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int qx = px + (j - CR);
            const float wgt = self_in + ((row_in && qx >= CR && qx < W - CR) ? 1.f : 0.f);
            // (a tap outside the image has weight 0 whenever p is not interior, and the staged zero otherwise: as in the forward)
            acc -= wgt * census_dterm(gx[ly + i][lx + j] - cx, gy[ly + i][lx + j] - cy);
        }
    }
    const float g = acc * norm * grad_loss[0] / (float)C;
    const int64_t hw = (int64_t)H * W;
    for (int c = 0; c < C; ++c) grad_x[((int64_t)b * C + c) * hw + (int64_t)py * W + px] = g;
}

// Adjoints of nn.ReflectionPad2d(p) (the detail branch's output conv, model_singleframe.py:207: ReflectionPad2d(3) before the
// 7x7 convolution) and nn.ReplicationPad2d(p) (the FAC module, KernelConv2D.py:82-86): grad_in[i][j] = sum of the padded
// gradient over the padded positions that map onto (i, j).  One thread per output pixel, a GATHER in a fixed order:
// bit-reproducible, where torch's reflection_pad2d_backward / replication_pad2d_backward scatter with atomic adds and made the
// training step's trajectory differ from run to run in the last bit.
//   reflect:   the position itself plus at most one mirror image per axis (own, row mirror, column mirror, corner)
//   replicate: the position itself; the first / last row (column) also collects the p rows (columns) outside it
template <bool REPLICATE>
__global__ __launch_bounds__(256) void pad2d_bwd_kernel(const float *__restrict__ gp, float *__restrict__ gx, int64_t planes,
                                                        int H, int W, int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const int64_t total = planes * H * W;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % W);
    const int i = (int)((idx / W) % H);
    const int64_t pl = idx / ((int64_t)W * H);
    const float *g = gp + pl * Hp * Wp;
    float acc = 0.f;
    if constexpr (REPLICATE) {
        const int ylo = i == 0 ? 0 : i + p, yhi = i == H - 1 ? H - 1 + 2 * p : i + p;
        const int xlo = j == 0 ? 0 : j + p, xhi = j == W - 1 ? W - 1 + 2 * p : j + p;
        for (int y = ylo; y <= yhi; ++y)
            for (int x = xlo; x <= xhi; ++x) acc += g[(int64_t)y * Wp + x];
    } else {
        // padded rows that reflect onto i: i + p; p - i (for 1 <= i <= p); 2 (H - 1) - i + p (for H - 1 - p <= i <= H - 2)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = i + p;
        if (i >= 1 && i <= p) ys[ny++] = p - i;
        if (i <= H - 2 && i >= H - 1 - p) ys[ny++] = 2 * (H - 1) - i + p;
        xs[nx++] = j + p;
        if (j >= 1 && j <= p) xs[nx++] = p - j;
        if (j <= W - 2 && j >= W - 1 - p) xs[nx++] = 2 * (W - 1) - j + p;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) acc += g[(int64_t)ys[a] * Wp + xs[b]];
    }
    gx[idx] = acc;
}

}  // namespace

extern "C" int ebfi_pad2d_backward(const float *grad_padded, float *grad_input, int64_t planes, int H, int W, int pad, int replicate,
                                   void *stream) {
    if (!grad_padded || !grad_input) return fail(EBFI_ERR_ARG, "pad2d_backward: null argument");
    if (planes < 0 || H < 1 || W < 1 || pad < 0 || (!replicate && (pad >= H || pad >= W)))
        return fail(EBFI_ERR_ARG, "pad2d_backward: pad %d on %d x %d (reflection needs pad < H, W)", pad, H, W);
    if (planes == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = planes * H * W;
    {
        ProfScope ps(replicate ? "replicate_pad_bwd" : "reflect_pad_bwd", st, 0.0, 4.0 * ((double)total + (double)planes * (H + 2 * pad) * (W + 2 * pad)));
        if (replicate)
            hipLaunchKernelGGL(pad2d_bwd_kernel<true>, dim3((unsigned)ceil_div(total, (int64_t)256)), dim3(256), 0, st, grad_padded, grad_input,
                               planes, H, W, pad);
        else
            hipLaunchKernelGGL(pad2d_bwd_kernel<false>, dim3((unsigned)ceil_div(total, (int64_t)256)), dim3(256), 0, st, grad_padded, grad_input,
                               planes, H, W, pad);
    }
    return check_launch("pad2d_bwd");
}

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

extern "C" int64_t ebfi_census_partials(int B, int H, int W) { return (int64_t)B * ceil_div(H, CT) * ceil_div(W, CT); }

// partial[ebfi_census_partials(B,H,W)]: per-tile sums of dist over interior pixels; loss = sum(partial) / (B*H*W)
extern "C" int ebfi_census_forward(const float *x, const float *y, float *partial, int B, int C, int H, int W, void *stream) {
    if (!x || !y || !partial) return fail(EBFI_ERR_ARG, "census_forward: null argument");
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return fail(EBFI_ERR_ARG, "census_forward: bad dimensions");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("census_fwd", st, 0.0, 8.0 * B * C * (double)H * W);
        hipLaunchKernelGGL(census_fwd_kernel, dim3(ceil_div(W, CT), ceil_div(H, CT), B), dim3(CT * CT), 0, st, x, y, partial, C, H, W);
    }
    return check_launch("census_fwd");
}

// grad_x = d loss / d x given grad_loss (a 1-element device array); y is treated as a constant (detached target)
extern "C" int ebfi_census_backward(const float *x, const float *y, const float *grad_loss, float *grad_x, int B, int C, int H,
                                    int W, void *stream) {
    if (!x || !y || !grad_loss || !grad_x) return fail(EBFI_ERR_ARG, "census_backward: null argument");
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return fail(EBFI_ERR_ARG, "census_backward: bad dimensions");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float norm = (float)(1.0 / (49.0 * (double)B * (double)H * (double)W));
    {
        ProfScope ps("census_bwd", st, 0.0, 12.0 * B * C * (double)H * W);
        hipLaunchKernelGGL(census_bwd_kernel, dim3(ceil_div(W, CT), ceil_div(H, CT), B), dim3(CT * CT), 0, st, x, y, grad_loss,
                           grad_x, C, H, W, norm);
    }
    return check_launch("census_bwd");
}

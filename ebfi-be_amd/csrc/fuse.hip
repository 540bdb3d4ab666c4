// Fused elementwise stages between the convolutions of the hot path (each replaces a chain of PyTorch
// elementwise / cat / reduction launches over full feature maps).
//
// scale_residual_cat: one round of ResidualControl (reference models/Ours/model_singleframe.py:79-136)
//     by_ex = Conv1(ex) * Conv3(x) + x ;  by_t = Conv2(t) * Conv4(x) + x ;  cat([by_ex, by_t], 1)
// with a0 = Conv3(x), a1 = Conv4(x) [B,C,H,W] and the per-sample channel scales s0 = Conv1(ex), s1 = Conv2(t) [B,C].
#include "common.hpp"

using namespace ebfi;

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void src_fwd_kernel(const float *__restrict__ a0, const float *__restrict__ s0,
                                                      const float *__restrict__ a1, const float *__restrict__ s1,
                                                      const float *__restrict__ x, float *__restrict__ out, int C, int64_t HW4,
                                                      int64_t total4, int64_t a_bs4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int64_t bc = i / HW4, r = i - bc * HW4;
    const int64_t b = bc / C, c = bc - b * C;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const f4 xv = x ? reinterpret_cast<const f4 *>(x)[i] : zero;          // x == NULL: no residual
    const int64_t ia = b * a_bs4 + c * HW4 + r;                          // a0 / a1 may be channel slices of a wider tensor
    const f4 v0 = reinterpret_cast<const f4 *>(a0)[ia], v1 = reinterpret_cast<const f4 *>(a1)[ia];
    const float k0 = s0[bc], k1 = s1 ? s1[bc] : 1.f;                     // s1 == NULL: second half unscaled
    f4 *o = reinterpret_cast<f4 *>(out);
    o[(b * 2 * C + c) * HW4 + r] = v0 * k0 + xv;
    o[(b * 2 * C + C + c) * HW4 + r] = v1 * k1 + xv;
}

// one workgroup per (b, c) plane: grad_a0 = g0*s0, grad_a1 = g1*s1, grad_x = g0 + g1,
// grad_s0[b,c] = sum g0*a0, grad_s1[b,c] = sum g1*a1   (fixed-order reduction -> deterministic)
__global__ __launch_bounds__(256) void src_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ a0,
                                                      const float *__restrict__ s0, const float *__restrict__ a1,
                                                      const float *__restrict__ s1, float *__restrict__ ga0,
                                                      float *__restrict__ ga1, float *__restrict__ gx, float *__restrict__ gs0,
                                                      float *__restrict__ gs1, int C, int64_t HW4, int64_t a_bs4, int64_t ga_bs4,
                                                      int mask_leaky, float mask_slope) {
    __shared__ float red[2][4];
    const int64_t bc = blockIdx.x;
    const int64_t b = bc / C, c = bc - b * C;
    const f4 *g0 = reinterpret_cast<const f4 *>(gout) + (b * 2 * C + c) * HW4;
    const f4 *g1 = reinterpret_cast<const f4 *>(gout) + (b * 2 * C + C + c) * HW4;
    // a0 / a1 and grad_a0 / grad_a1 may be channel slices of wider tensors (batch strides a_bs4 / ga_bs4)
    const f4 *p0 = reinterpret_cast<const f4 *>(a0) + b * a_bs4 + c * HW4, *p1 = reinterpret_cast<const f4 *>(a1) + b * a_bs4 + c * HW4;
    f4 *o0 = reinterpret_cast<f4 *>(ga0) + b * ga_bs4 + c * HW4, *o1 = reinterpret_cast<f4 *>(ga1) + b * ga_bs4 + c * HW4;
    f4 *ox = reinterpret_cast<f4 *>(gx) + bc * HW4;
    const float k0 = s0[bc], k1 = s1 ? s1[bc] : 1.f;
    float d0 = 0.f, d1 = 0.f;
    // mask_leaky: a0 / a1 are outputs of LeakyReLU(mask_slope) layers and the gradients leave as PRE-activation gradients
    auto dm = [&](const f4 &v) {
        f4 m;
        m.x = v.x > 0.f ? 1.f : mask_slope; m.y = v.y > 0.f ? 1.f : mask_slope;
        m.z = v.z > 0.f ? 1.f : mask_slope; m.w = v.w > 0.f ? 1.f : mask_slope;
        return m;
    };
    for (int64_t i = threadIdx.x; i < HW4; i += 256) {
        const f4 u0 = g0[i], u1 = g1[i], v0 = p0[i];
        if (mask_leaky) {
            o0[i] = u0 * k0 * dm(v0);
            o1[i] = u1 * k1 * dm(p1[i]);
        } else {
            o0[i] = u0 * k0;
            o1[i] = u1 * k1;
        }
        if (gx) ox[i] = u0 + u1;
        d0 += (u0.x * v0.x + u0.y * v0.y) + (u0.z * v0.z + u0.w * v0.w);
        if (gs1) {
            const f4 v1 = p1[i];
            d1 += (u1.x * v1.x + u1.y * v1.y) + (u1.z * v1.z + u1.w * v1.w);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        d0 += __shfl_xor(d0, d, 64);
        d1 += __shfl_xor(d1, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = d0;
        red[1][threadIdx.x >> 6] = d1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        gs0[bc] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        if (gs1) gs1[bc] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// out[i] = sum_r src[idx[i*R + r]]  (idx < 0: structural zero).  Used for the weight re-layouts ("folds") of the
// depth-2 3-D convolutions: forward map with R = 1, its transpose with R = max fan-out of a source element.
__global__ void gather_sum_kernel(const float *__restrict__ src, const int *__restrict__ idx, float *__restrict__ out,
                                  int64_t n_out, int R) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    float acc = 0.f;
    for (int r = 0; r < R; ++r) {
        const int j = idx[i * R + r];
        if (j >= 0) acc += src[j];
    }
    out[i] = acc;
}

// product-mean: out[plane] = mean_i a[plane][i] * b[plane][i]   (AdaptiveAvgPool2d(1) of a product of two maps:
// ExposureDecision's correlation of the normalised event / blur features, model_singleframe.py:66-68) and its adjoint
// ga = (g[plane] / n) * b, gb = (g[plane] / n) * a.  One workgroup per plane, fixed-order reduction.
__global__ __launch_bounds__(256) void prodmean_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           float *__restrict__ out, int64_t HW4) {
    __shared__ float red[4];
    const f4 *pa = reinterpret_cast<const f4 *>(a) + (int64_t)blockIdx.x * HW4;
    const f4 *pb = reinterpret_cast<const f4 *>(b) + (int64_t)blockIdx.x * HW4;
    float d = 0.f;
    for (int64_t i = threadIdx.x; i < HW4; i += 256) {
        const f4 u = pa[i], v = pb[i];
        d += (u.x * v.x + u.y * v.y) + (u.z * v.z + u.w * v.w);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(HW4 * 4);
}

__global__ __launch_bounds__(256) void prodmean_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           const float *__restrict__ g, float *__restrict__ ga,
                                                           float *__restrict__ gb, int64_t HW4, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float k = g[i / HW4] / (float)(HW4 * 4);
    const f4 u = reinterpret_cast<const f4 *>(a)[i], v = reinterpret_cast<const f4 *>(b)[i];
    reinterpret_cast<f4 *>(ga)[i] = v * k;
    reinterpret_cast<f4 *>(gb)[i] = u * k;
}

int check_planes(const char *who, int B, int C, int64_t HW) {
    if (B < 0 || C <= 0 || HW <= 0) return fail(EBFI_ERR_ARG, "%s: bad dimensions", who);
    if (HW % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "%s: H*W must be a multiple of 4 (got %lld)", who, (long long)HW);
    if ((int64_t)B * C > 2147483647LL) return fail(EBFI_ERR_ARG, "%s: too many planes", who);
    return EBFI_OK;
}

}  // namespace

// out [B,2C,H,W]: out[:, :C] = s0[b,c]*a0 + x, out[:, C:] = s1[b,c]*a1 + x   (x == NULL: no residual; s1 == NULL: s1 = 1)
extern "C" int ebfi_scale_residual_cat_forward_ex(const float *a0, const float *s0, const float *a1, const float *s1,
                                                  const float *x, float *out, int B, int C, int64_t HW, int64_t a_batch_stride,
                                                  void *stream);
extern "C" int ebfi_scale_residual_cat_forward(const float *a0, const float *s0, const float *a1, const float *s1,
                                               const float *x, float *out, int B, int C, int64_t HW, void *stream) {
    return ebfi_scale_residual_cat_forward_ex(a0, s0, a1, s1, x, out, B, C, HW, (int64_t)C * HW, stream);
}

// same with a0 / a1 given as channel slices of wider [B, *, H, W] tensors: a_batch_stride elements between samples
extern "C" int ebfi_scale_residual_cat_forward_ex(const float *a0, const float *s0, const float *a1, const float *s1,
                                                  const float *x, float *out, int B, int C, int64_t HW, int64_t a_batch_stride,
                                                  void *stream) {
    if (!a0 || !s0 || !a1 || !out) return fail(EBFI_ERR_ARG, "scale_residual_cat_forward: null argument");
    if (a_batch_stride % 4 != 0 || a_batch_stride < (int64_t)C * HW) return fail(EBFI_ERR_ARG, "scale_residual_cat_forward: batch stride");
    if (int rc = check_planes("scale_residual_cat_forward", B, C, HW)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total4 = (int64_t)B * C * HW / 4;
    {
        ProfScope ps("scale_residual_cat_fwd", st, 0.0, 20.0 * B * C * (double)HW);
        hipLaunchKernelGGL(src_fwd_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, a0, s0, a1, s1, x, out, C, HW / 4,
                           total4, a_batch_stride / 4);
    }
    return check_launch("scale_residual_cat_fwd");
}

// adjoint of the above for grad_out [B,2C,H,W]: grad_a0, grad_a1, grad_x [B,C,H,W]; grad_s0, grad_s1 [B,C]
// (grad_x == NULL when there was no residual; s1 == NULL: a1 and grad_s1 are not touched)
extern "C" int ebfi_scale_residual_cat_backward_ex(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                                   const float *s1, float *grad_a0, float *grad_a1, float *grad_x,
                                                   float *grad_s0, float *grad_s1, int B, int C, int64_t HW,
                                                   int64_t a_batch_stride, int64_t grad_a_batch_stride, int mask_leaky,
                                                   float mask_slope, void *stream);
extern "C" int ebfi_scale_residual_cat_backward(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                                const float *s1, float *grad_a0, float *grad_a1, float *grad_x,
                                                float *grad_s0, float *grad_s1, int B, int C, int64_t HW, void *stream) {
    return ebfi_scale_residual_cat_backward_ex(grad_out, a0, s0, a1, s1, grad_a0, grad_a1, grad_x, grad_s0, grad_s1, B, C, HW,
                                               (int64_t)C * HW, (int64_t)C * HW, 0, 0.f, stream);
}

// same with a0 / a1 and grad_a0 / grad_a1 as channel slices of wider tensors (batch strides in elements); mask_leaky != 0:
// a0 / a1 came out of LeakyReLU(mask_slope) layers and grad_a* leave multiplied by that derivative (pre-activation gradients)
extern "C" int ebfi_scale_residual_cat_backward_ex(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                                   const float *s1, float *grad_a0, float *grad_a1, float *grad_x,
                                                   float *grad_s0, float *grad_s1, int B, int C, int64_t HW,
                                                   int64_t a_batch_stride, int64_t grad_a_batch_stride, int mask_leaky,
                                                   float mask_slope, void *stream) {
    if (a_batch_stride % 4 != 0 || grad_a_batch_stride % 4 != 0 || a_batch_stride < (int64_t)C * HW ||
        grad_a_batch_stride < (int64_t)C * HW || (mask_leaky && !a1))
        return fail(EBFI_ERR_ARG, "scale_residual_cat_backward: batch stride / mask");
    if (!grad_out || !a0 || !s0 || !grad_a0 || !grad_a1 || !grad_s0 || (s1 && (!a1 || !grad_s1)))
        return fail(EBFI_ERR_ARG, "scale_residual_cat_backward: null argument");
    if (int rc = check_planes("scale_residual_cat_backward", B, C, HW)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("scale_residual_cat_bwd", st, 0.0, 28.0 * B * C * (double)HW);
        hipLaunchKernelGGL(src_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, st, grad_out, a0, s0, a1, s1, grad_a0, grad_a1,
                           grad_x, grad_s0, grad_s1, C, HW / 4, a_batch_stride / 4, grad_a_batch_stride / 4, mask_leaky, mask_slope);
    }
    return check_launch("scale_residual_cat_bwd");
}

// out[i] = sum_{r<R} src[idx[i*R+r]] with negative indices skipped (idx [n_out*R] int32; caller guarantees idx < len(src))
extern "C" int ebfi_gather_sum(const float *src, const int32_t *idx, float *out, int64_t n_out, int R, void *stream) {
    if (!src || !idx || !out) return fail(EBFI_ERR_ARG, "gather_sum: null argument");
    if (n_out < 0 || R <= 0) return fail(EBFI_ERR_ARG, "gather_sum: bad dimensions");
    if (n_out == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gather_sum_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, st, src, idx, out, n_out, R);
    return check_launch("gather_sum");
}

// out[planes] = mean over HW of a*b (a, b [planes, HW] contiguous); HW a multiple of 4
extern "C" int ebfi_prodmean_forward(const float *a, const float *b, float *out, int64_t planes, int64_t HW, void *stream) {
    if (!a || !b || !out) return fail(EBFI_ERR_ARG, "prodmean_forward: null argument");
    if (planes < 0 || planes > 2147483647LL || HW <= 0) return fail(EBFI_ERR_ARG, "prodmean_forward: bad dimensions");
    if (HW % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "prodmean_forward: H*W must be a multiple of 4");
    if (planes == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("prodmean_fwd", st, 0.0, 8.0 * planes * (double)HW);
        hipLaunchKernelGGL(prodmean_fwd_kernel, dim3((unsigned)planes), dim3(256), 0, st, a, b, out, HW / 4);
    }
    return check_launch("prodmean_fwd");
}

// grad_a = (grad_out[plane] / HW) * b, grad_b = (grad_out[plane] / HW) * a
extern "C" int ebfi_prodmean_backward(const float *a, const float *b, const float *grad_out, float *grad_a, float *grad_b,
                                      int64_t planes, int64_t HW, void *stream) {
    if (!a || !b || !grad_out || !grad_a || !grad_b) return fail(EBFI_ERR_ARG, "prodmean_backward: null argument");
    if (planes < 0 || HW <= 0) return fail(EBFI_ERR_ARG, "prodmean_backward: bad dimensions");
    if (HW % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "prodmean_backward: H*W must be a multiple of 4");
    if (planes == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total4 = planes * HW / 4;
    {
        ProfScope ps("prodmean_bwd", st, 0.0, 16.0 * planes * (double)HW);
        hipLaunchKernelGGL(prodmean_bwd_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, a, b, grad_out, grad_a, grad_b,
                           HW / 4, total4);
    }
    return check_launch("prodmean_bwd");
}

// GroupNorm over NCHW fp32 for the exposure-decision head (nn.GroupNorm(4, 64) applied to two
// [B,64,H,W] full-resolution maps per forward: reference models/Ours/model_singleframe.py:36,66-67).
// PyTorch's row-wise moments kernel runs one workgroup per (sample, group) row -- 32 workgroups for 134 MB --
// so the op is split into wide partial reductions (one workgroup per (b, c, slice)), a tiny fixed-order
// finalisation in double precision, and vectorised apply kernels.  Deterministic.
#include "common.hpp"

using namespace ebfi;

namespace {

constexpr int NSL = 16;   // slices per (b, c) plane in the partial reductions

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// partial[(b*C + c)*NSL + s] = {sum x, sum x^2} (forward)  or  {sum gy, sum gy*xhat} (backward)
template <bool BWD>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ a, const float *__restrict__ x,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         double *__restrict__ partial, int C, int G, int64_t HW) {
    __shared__ double red[2][4];
    const int bc = blockIdx.x, s = blockIdx.y;
    const int64_t chunk = (HW + NSL - 1) / NSL;
    const int64_t lo = s * chunk, hi = (lo + chunk < HW) ? lo + chunk : HW;
    const float *pa = a + (int64_t)bc * HW;
    float m = 0.f, r = 0.f;
    const float *px = nullptr;
    if (BWD) {
        const int b = bc / C, c = bc - b * C, g = c / (C / G);
        m = mean[b * G + g];
        r = rstd[b * G + g];
        px = x + (int64_t)bc * HW;
    }
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = pa[i];
        if (BWD) {
            s0 += v;
            s1 += (double)v * (double)((px[i] - m) * r);
        } else {
            s0 += v;
            s1 += (double)v * v;
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((int64_t)bc * NSL + s) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[((int64_t)bc * NSL + s) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// forward finalisation: one thread per (b, g)
__global__ void gn_finalize_fwd_kernel(const double *__restrict__ partial, float *__restrict__ mean,
                                       float *__restrict__ rstd, int B, int C, int G, int64_t HW, float eps) {
    const int bg = blockIdx.x * blockDim.x + threadIdx.x;
    if (bg >= B * G) return;
    const int b = bg / G, g = bg - b * G, cpg = C / G;
    double s0 = 0.0, s1 = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c)
        for (int s = 0; s < NSL; ++s) {
            s0 += partial[(((int64_t)b * C + c) * NSL + s) * 2];
            s1 += partial[(((int64_t)b * C + c) * NSL + s) * 2 + 1];
        }
    const double n = (double)cpg * (double)HW;
    const double mu = s0 / n;
    double var = s1 / n - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[bg] = (float)mu;
    rstd[bg] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ void gn_apply_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, const float *__restrict__ mean,
                                    const float *__restrict__ rstd, float *__restrict__ y, int C, int G, int64_t HW,
                                    int64_t total4) {
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const int64_t e = i4 * 4;
    const int64_t bc = e / HW;                       // HW % 4 == 0: the 4 elements share a plane
    const int b = (int)(bc / C), c = (int)(bc - (int64_t)b * C), g = c / (C / G);
    const float m = mean[b * G + g], r = rstd[b * G + g];
    const float sc = r * (gamma ? gamma[c] : 1.f), sh = (beta ? beta[c] : 0.f) - m * sc;
    const float4 v = *reinterpret_cast<const float4 *>(x + e);
    *reinterpret_cast<float4 *>(y + e) = make_float4(v.x * sc + sh, v.y * sc + sh, v.z * sc + sh, v.w * sc + sh);
}

// backward finalisation: thread per (b, g) for the two group sums; thread per c for ggamma / gbeta
__global__ void gn_finalize_bwd_kernel(const double *__restrict__ partial, const float *__restrict__ gamma,
                                       float *__restrict__ s1g, float *__restrict__ s2g, float *__restrict__ ggamma,
                                       float *__restrict__ gbeta, int B, int C, int G) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int cpg = C / G;
    if (t < B * G) {
        const int b = t / G, g = t - b * G;
        double a1 = 0.0, a2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            double d0 = 0.0, d1 = 0.0;
            for (int s = 0; s < NSL; ++s) {
                d0 += partial[(((int64_t)b * C + c) * NSL + s) * 2];
                d1 += partial[(((int64_t)b * C + c) * NSL + s) * 2 + 1];
            }
            const double gm = gamma ? (double)gamma[c] : 1.0;
            a1 += gm * d1;       // sum gy * gamma * xhat
            a2 += gm * d0;       // sum gy * gamma
        }
        s1g[t] = (float)a1;
        s2g[t] = (float)a2;
    }
    if (t < C && (ggamma || gbeta)) {
        double d0 = 0.0, d1 = 0.0;
        for (int b = 0; b < B; ++b)
            for (int s = 0; s < NSL; ++s) {
                d0 += partial[(((int64_t)b * C + t) * NSL + s) * 2];
                d1 += partial[(((int64_t)b * C + t) * NSL + s) * 2 + 1];
            }
        if (ggamma) ggamma[t] = (float)d1;
        if (gbeta) gbeta[t] = (float)d0;
    }
}

// gx = rstd * (gy*gamma - (s2 + xhat*s1)/N)
__global__ void gn_apply_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                    const float *__restrict__ gamma, const float *__restrict__ mean,
                                    const float *__restrict__ rstd, const float *__restrict__ s1g,
                                    const float *__restrict__ s2g, float *__restrict__ gx, int C, int G, int64_t HW,
                                    int64_t total4) {
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const int64_t e = i4 * 4;
    const int64_t bc = e / HW;
    const int b = (int)(bc / C), c = (int)(bc - (int64_t)b * C), g = c / (C / G);
    const float m = mean[b * G + g], r = rstd[b * G + g];
    const float invn = 1.f / ((float)(C / G) * (float)HW);
    const float gm = gamma ? gamma[c] : 1.f;
    const float k1 = s1g[b * G + g] * invn, k2 = s2g[b * G + g] * invn;
    const float4 go = *reinterpret_cast<const float4 *>(gy + e);
    const float4 xv = *reinterpret_cast<const float4 *>(x + e);
    float4 o;
    o.x = r * (go.x * gm - k2 - (xv.x - m) * r * k1);
    o.y = r * (go.y * gm - k2 - (xv.y - m) * r * k1);
    o.z = r * (go.z * gm - k2 - (xv.z - m) * r * k1);
    o.w = r * (go.w * gm - k2 - (xv.w - m) * r * k1);
    *reinterpret_cast<float4 *>(gx + e) = o;
}

int check_gn(int B, int C, int G, int64_t HW) {
    if (B < 0 || C <= 0 || G <= 0 || HW <= 0 || C % G != 0) return fail(EBFI_ERR_ARG, "groupnorm: bad dimensions");
    if (HW % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "groupnorm: H*W must be a multiple of 4 (got %lld)", (long long)HW);
    if ((int64_t)B * C > 2147483647LL) return fail(EBFI_ERR_ARG, "groupnorm: too many planes");
    return EBFI_OK;
}

}  // namespace

extern "C" size_t ebfi_groupnorm_workspace(int B, int C) { return (size_t)B * C * NSL * 2 * sizeof(double) + 64; }

// y = (x - mean_bg) * rstd_bg * gamma_c + beta_c; mean / rstd [B*G] are outputs kept for the backward
extern "C" int ebfi_groupnorm_forward(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                                      float *rstd, int B, int C, int64_t HW, int groups, float eps, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    if (!x || !y || !mean || !rstd) return fail(EBFI_ERR_ARG, "groupnorm_forward: null argument");
    if (int rc = check_gn(B, C, groups, HW)) return rc;
    if (!workspace || workspace_bytes < ebfi_groupnorm_workspace(B, C)) return fail(EBFI_ERR_WORKSPACE, "groupnorm_forward: workspace too small");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(workspace);
    const int64_t total4 = (int64_t)B * C * HW / 4;
    {
        ProfScope ps("groupnorm_fwd", st, 0.0, 12.0 * B * C * (double)HW);
        hipLaunchKernelGGL((gn_partial_kernel<false>), dim3((unsigned)(B * C), NSL), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                           partial, C, groups, HW);
        hipLaunchKernelGGL(gn_finalize_fwd_kernel, dim3((unsigned)ceil_div(B * groups, 64)), dim3(64), 0, st, partial, mean, rstd,
                           B, C, groups, HW, eps);
        hipLaunchKernelGGL(gn_apply_fwd_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, x, gamma, beta, mean, rstd,
                           y, C, groups, HW, total4);
    }
    return check_launch("groupnorm_fwd");
}

// grad_x (required), grad_gamma / grad_beta [C] (optional, NULL to skip)
extern "C" int ebfi_groupnorm_backward(const float *grad_y, const float *x, const float *gamma, const float *mean,
                                       const float *rstd, float *grad_x, float *grad_gamma, float *grad_beta, int B,
                                       int C, int64_t HW, int groups, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (!grad_y || !x || !mean || !rstd || !grad_x) return fail(EBFI_ERR_ARG, "groupnorm_backward: null argument");
    if (int rc = check_gn(B, C, groups, HW)) return rc;
    const size_t need = ebfi_groupnorm_workspace(B, C) + (size_t)2 * B * groups * sizeof(float);
    if (!workspace || workspace_bytes < need) return fail(EBFI_ERR_WORKSPACE, "groupnorm_backward: workspace too small");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(workspace);
    float *s1g = reinterpret_cast<float *>(static_cast<char *>(workspace) + ebfi_groupnorm_workspace(B, C));
    float *s2g = s1g + (size_t)B * groups;
    const int64_t total4 = (int64_t)B * C * HW / 4;
    const int nfin = (B * groups > C ? B * groups : C);
    {
        ProfScope ps("groupnorm_bwd", st, 0.0, 20.0 * B * C * (double)HW);
        hipLaunchKernelGGL((gn_partial_kernel<true>), dim3((unsigned)(B * C), NSL), dim3(256), 0, st, grad_y, x, mean, rstd, partial,
                           C, groups, HW);
        hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3((unsigned)ceil_div(nfin, 64)), dim3(64), 0, st, partial, gamma, s1g, s2g,
                           grad_gamma, grad_beta, B, C, groups);
        hipLaunchKernelGGL(gn_apply_bwd_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, grad_y, x, gamma, mean,
                           rstd, s1g, s2g, grad_x, C, groups, HW, total4);
    }
    return check_launch("groupnorm_bwd");
}

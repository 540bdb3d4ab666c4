// ExposureDecision head (models/Ours/model_singleframe.py:66-72):
//     atten = sigmoid(AVGPool(GN(ev) * GN(bl)));   out = cat([ev * atten, bl], 1)
// with ONE shared GroupNorm applied to two full-resolution maps (134 MB each at B=8, 256x256).
//
// GN(x) is affine in x per (sample, channel): GN(x)_c = a_c x + p_c with a_c = gamma_c rstd_g, p_c = beta_c - a_c mu_g.
// The pooled product therefore needs only the five plane moments (sum x, x^2, z, z^2, xz): one pass over the two maps
// and neither normalised map is ever written.  The backward of the whole head is likewise closed-form in the moments:
//     grad_ev = A_c bl + B_g ev + C_c + atten_c gout[:, :C],      grad_bl = A'_c ev + B'_g bl + C'_c + gout[:, C:]
// (per-plane coefficients from a tiny double-precision kernel), i.e. one reduction (gout . ev per plane) and one
// elementwise pass instead of GroupNorm backward x2, the product-mean backward, the concat-stage backward and two
// 134 MB gradient accumulations.  Sums: fp32 inside a slice, double across slices / channels, fixed order (deterministic).
#include "common.hpp"

using namespace ebfi;

namespace {

constexpr int ET = 256;

struct Plan {
    int slices;
    int64_t chunk;   // elements per slice, multiple of 4
};

Plan ed_plan(int64_t planes, int64_t HW) {
    int64_t s = ceil_div(4096, planes);
    const int64_t max_s = std::max<int64_t>(1, HW / 4096);
    s = std::min<int64_t>(std::max<int64_t>(s, 1), std::min<int64_t>(max_s, 64));
    const int64_t chunk = ceil_div(ceil_div(HW, s), 4) * 4;
    return {(int)ceil_div(HW, chunk), chunk};
}

template <int NV> __device__ __forceinline__ void block_reduce(float (&v)[NV], float *out) {
    __shared__ float red[ET / 64][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.f;
        for (int w = 0; w < ET / 64; ++w) s += red[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// partial[(plane * S + slice) * 5 + {x, xx, z, zz, xz}]
__global__ __launch_bounds__(ET) void ed_stats_kernel(const float *__restrict__ x, const float *__restrict__ z,
                                                      float *__restrict__ partial, int64_t HW, int S, int64_t chunk) {
    const int64_t plane = blockIdx.x / S;
    const int sl = blockIdx.x - (int)(plane * S);
    const int64_t begin = sl * chunk, end = min(HW, begin + chunk);
    const float *px = x + plane * HW, *pz = z + plane * HW;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i = begin + threadIdx.x * 4; i < end; i += ET * 4) {
        const float4 a = *reinterpret_cast<const float4 *>(px + i), b = *reinterpret_cast<const float4 *>(pz + i);
        v[0] += (a.x + a.y) + (a.z + a.w);
        v[1] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        v[2] += (b.x + b.y) + (b.z + b.w);
        v[3] += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
        v[4] += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    block_reduce<5>(v, partial + (int64_t)blockIdx.x * 5);
}

// stats: [B*C][5] plane means (x, xx, z, zz, xz), then [B*G][4] = (mu_x, rstd_x, mu_z, rstd_z)
// one workgroup of >= C threads per sample; dynamic LDS: (5 C + 4 G) doubles
__global__ void ed_fwd_finalize_kernel(const float *__restrict__ partial, const float *__restrict__ gamma,
                                       const float *__restrict__ beta, double *__restrict__ stats, float *__restrict__ atten, int B,
                                       int C, int G, int64_t HW, int S, float eps) {
    extern __shared__ double sh[];
    double *M = sh, *grp = sh + 5 * C;
    const int c = threadIdx.x, cpg = C / G;
    (void)B;
    {
        const int b = blockIdx.x;          // one workgroup per sample (samples are independent)
        if (c < C) {
            const float *pp = partial + ((int64_t)(b * C + c) * S) * 5;
            for (int k = 0; k < 5; ++k) {
                double s = 0.0;
                for (int sl = 0; sl < S; ++sl) s += (double)pp[sl * 5 + k];
                s /= (double)HW;
                M[c * 5 + k] = s;
                stats[((int64_t)b * C + c) * 5 + k] = s;
            }
        }
        __syncthreads();
        if (c < G) {
            double mx = 0, mxx = 0, mz = 0, mzz = 0;
            for (int j = c * cpg; j < (c + 1) * cpg; ++j) {
                mx += M[j * 5 + 0];
                mxx += M[j * 5 + 1];
                mz += M[j * 5 + 2];
                mzz += M[j * 5 + 3];
            }
            mx /= cpg, mxx /= cpg, mz /= cpg, mzz /= cpg;
            const double vx = fmax(mxx - mx * mx, 0.0), vz = fmax(mzz - mz * mz, 0.0);
            grp[c * 4 + 0] = mx;
            grp[c * 4 + 1] = 1.0 / sqrt(vx + (double)eps);
            grp[c * 4 + 2] = mz;
            grp[c * 4 + 3] = 1.0 / sqrt(vz + (double)eps);
            for (int k = 0; k < 4; ++k) stats[(int64_t)gridDim.x * C * 5 + ((int64_t)b * G + c) * 4 + k] = grp[c * 4 + k];
        }
        __syncthreads();
        if (c < C) {
            const int g = c / cpg;
            const double mux = grp[g * 4], rx = grp[g * 4 + 1], muz = grp[g * 4 + 2], rz = grp[g * 4 + 3];
            const double ga = gamma[c], be = beta[c];
            const double Mx = M[c * 5], Mz = M[c * 5 + 2], Mxz = M[c * 5 + 4];
            // mean((ga rx (x - mux) + be) (ga rz (z - muz) + be))
            const double s = ga * ga * rx * rz * (Mxz - mux * Mz - muz * Mx + mux * muz) + be * ga * rx * (Mx - mux) +
                             be * ga * rz * (Mz - muz) + be * be;
            atten[b * C + c] = (float)(1.0 / (1.0 + exp(-s)));
        }
    }
}

// out[b, c] = ev[b, c] * atten[b, c];  out[b, C + c] = bl[b, c]
__global__ __launch_bounds__(ET) void ed_cat_kernel(const float *__restrict__ x, const float *__restrict__ z,
                                                    const float *__restrict__ atten, float *__restrict__ out, int C, int64_t HW,
                                                    int S, int64_t chunk) {
    const int64_t plane = blockIdx.x / S;
    const int sl = blockIdx.x - (int)(plane * S);
    const int64_t b = plane / C, c = plane - b * C;
    const int64_t begin = sl * chunk, end = min(HW, begin + chunk);
    const float *px = x + plane * HW, *pz = z + plane * HW;
    float *o0 = out + (b * 2 * C + c) * HW, *o1 = o0 + (int64_t)C * HW;
    const float s = atten[plane];
    for (int64_t i = begin + threadIdx.x * 4; i < end; i += ET * 4) {
        float4 a = *reinterpret_cast<const float4 *>(px + i);
        const float4 bb = *reinterpret_cast<const float4 *>(pz + i);
        a.x *= s, a.y *= s, a.z *= s, a.w *= s;
        *reinterpret_cast<float4 *>(o0 + i) = a;
        *reinterpret_cast<float4 *>(o1 + i) = bb;
    }
}

// partial[plane * S + slice] = sum over the slice of gout[b, c] * ev[b, c]
__global__ __launch_bounds__(ET) void ed_plane_dot_kernel(const float *__restrict__ gout, const float *__restrict__ x,
                                                          float *__restrict__ partial, int C, int64_t HW, int S, int64_t chunk) {
    const int64_t plane = blockIdx.x / S;
    const int sl = blockIdx.x - (int)(plane * S);
    const int64_t b = plane / C, c = plane - b * C;
    const int64_t begin = sl * chunk, end = min(HW, begin + chunk);
    const float *px = x + plane * HW, *pg = gout + (b * 2 * C + c) * HW;
    float v[1] = {0.f};
    for (int64_t i = begin + threadIdx.x * 4; i < end; i += ET * 4) {
        const float4 a = *reinterpret_cast<const float4 *>(px + i), g = *reinterpret_cast<const float4 *>(pg + i);
        v[0] += (a.x * g.x + a.y * g.y) + (a.z * g.z + a.w * g.w);
    }
    block_reduce<1>(v, partial + blockIdx.x);
}

// coef[plane][8] = {A, Bg, Cc, A', Bg', Cc', atten, 0}; grad_gamma / grad_beta summed over samples in order
// one workgroup of >= C threads; dynamic LDS: 4 C + 4 G doubles
__global__ void ed_bwd_coeff_kernel(const float *__restrict__ dot_partial, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, const float *__restrict__ atten, const double *__restrict__ stats,
                                    float *__restrict__ coef, float *__restrict__ grad_gamma, float *__restrict__ grad_beta, int B,
                                    int C, int G, int64_t HW, int S) {
    extern __shared__ double sh[];
    double *T = sh, *D = sh + 4 * C;
    const int c = threadIdx.x, cpg = C / G, g = c < C ? c / cpg : 0;
    double ggam = 0.0, gbet = 0.0;
    for (int b = 0; b < B; ++b) {
        double kn = 0, a = 0, p = 0, q = 0, r = 0, mux = 0, rx = 0, muz = 0, rz = 0, att = 0;
        if (c < C) {
            const int64_t plane = (int64_t)b * C + c;
            double gatt = 0.0;
            for (int sl = 0; sl < S; ++sl) gatt += (double)dot_partial[plane * S + sl];
            att = atten[plane];
            const double gS = gatt * att * (1.0 - att);
            const double ga = gamma[c], be = beta[c];
            const double *gs = stats + (int64_t)B * C * 5 + ((int64_t)b * G + g) * 4;
            mux = gs[0], rx = gs[1], muz = gs[2], rz = gs[3];
            const double *M = stats + plane * 5;
            const double Mx = M[0], Mz = M[2], Mxz = M[4];
            a = ga * rx, p = be - a * mux, q = ga * rz, r = be - q * muz;
            kn = ga * gS;
            const double vx = rx * (q * (Mxz - mux * Mz) + r * (Mx - mux));    // mean(GN(bl) * xhat)
            const double uz = rz * (a * (Mxz - muz * Mx) + p * (Mz - muz));    // mean(GN(ev) * zhat)
            const double vm = q * Mz + r, um = a * Mx + p;                      // mean(GN(bl)), mean(GN(ev))
            T[c * 4 + 0] = kn * vm;
            T[c * 4 + 1] = kn * vx;
            T[c * 4 + 2] = kn * um;
            T[c * 4 + 3] = kn * uz;
            ggam += gS * (vx + uz);
            gbet += gS * (vm + um);
        }
        __syncthreads();
        if (c < G) {
            double s[4] = {0, 0, 0, 0};
            for (int j = c * cpg; j < (c + 1) * cpg; ++j)
                for (int k = 0; k < 4; ++k) s[k] += T[j * 4 + k];
            const double inv = 1.0 / ((double)cpg * (double)HW);
            for (int k = 0; k < 4; ++k) D[c * 4 + k] = s[k] * inv;
        }
        __syncthreads();
        if (c < C) {
            const double k = kn / (double)HW;
            const double D1 = D[g * 4], D2 = D[g * 4 + 1], E1 = D[g * 4 + 2], E2 = D[g * 4 + 3];
            float *o = coef + ((int64_t)b * C + c) * 8;
            o[0] = (float)(rx * k * q);
            o[1] = (float)(-rx * rx * D2);
            o[2] = (float)(rx * (k * r - D1) + rx * rx * mux * D2);
            o[3] = (float)(rz * k * a);
            o[4] = (float)(-rz * rz * E2);
            o[5] = (float)(rz * (k * p - E1) + rz * rz * muz * E2);
            o[6] = (float)att;
            o[7] = 0.f;
        }
        __syncthreads();
    }
    if (c < C) {
        if (grad_gamma) grad_gamma[c] = (float)ggam;
        if (grad_beta) grad_beta[c] = (float)gbet;
    }
}

__global__ __launch_bounds__(ET) void ed_bwd_apply_kernel(const float *__restrict__ gout, const float *__restrict__ x,
                                                          const float *__restrict__ z, const float *__restrict__ coef,
                                                          float *__restrict__ gx, float *__restrict__ gz, int C, int64_t HW, int S,
                                                          int64_t chunk) {
    const int64_t plane = blockIdx.x / S;
    const int sl = blockIdx.x - (int)(plane * S);
    const int64_t b = plane / C, c = plane - b * C;
    const int64_t begin = sl * chunk, end = min(HW, begin + chunk);
    const float *px = x + plane * HW, *pz = z + plane * HW;
    const float *g0 = gout + (b * 2 * C + c) * HW, *g1 = g0 + (int64_t)C * HW;
    float *ox = gx + plane * HW, *oz = gz + plane * HW;
    const float *k = coef + plane * 8;
    const float A = k[0], Bx = k[1], Cx = k[2], Az = k[3], Bz = k[4], Cz = k[5], att = k[6];
    for (int64_t i = begin + threadIdx.x * 4; i < end; i += ET * 4) {
        const float4 a = *reinterpret_cast<const float4 *>(px + i), bb = *reinterpret_cast<const float4 *>(pz + i);
        const float4 ga = *reinterpret_cast<const float4 *>(g0 + i), gb = *reinterpret_cast<const float4 *>(g1 + i);
        float4 rx, rz;
        rx.x = (A * bb.x + Bx * a.x + Cx) + att * ga.x;
        rx.y = (A * bb.y + Bx * a.y + Cx) + att * ga.y;
        rx.z = (A * bb.z + Bx * a.z + Cx) + att * ga.z;
        rx.w = (A * bb.w + Bx * a.w + Cx) + att * ga.w;
        rz.x = (Az * a.x + Bz * bb.x + Cz) + gb.x;
        rz.y = (Az * a.y + Bz * bb.y + Cz) + gb.y;
        rz.z = (Az * a.z + Bz * bb.z + Cz) + gb.z;
        rz.w = (Az * a.w + Bz * bb.w + Cz) + gb.w;
        *reinterpret_cast<float4 *>(ox + i) = rx;
        *reinterpret_cast<float4 *>(oz + i) = rz;
    }
}

const char *ed_check(const void *a, const void *b, int B, int C, int64_t HW, int groups) {
    if (B <= 0 || C <= 0 || HW <= 0 || groups <= 0) return "bad dimensions";
    if (C % groups || C > 1024) return "channels must divide into the groups and be <= 1024";
    if (HW % 4) return "H*W must be a multiple of 4";
    if (!aligned16(a) || !aligned16(b)) return "maps must be 16-byte aligned";
    return nullptr;
}

}  // namespace

extern "C" size_t ebfi_ed_head_workspace(int B, int C, int64_t HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    const int64_t planes = (int64_t)B * C;
    const Plan pl = ed_plan(planes, HW);
    return (size_t)(planes * pl.slices * 5 + planes * 8) * sizeof(float);
}

extern "C" int ebfi_ed_head_forward(const float *ev, const float *bl, const float *gamma, const float *beta, float *out,
                                    float *atten, double *stats, int B, int C, int64_t HW, int groups, float eps, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    if (!ev || !bl || !gamma || !beta || !out || !atten || !stats || !workspace) return fail(EBFI_ERR_ARG, "ed_head_forward: null argument");
    if (const char *why = ed_check(ev, bl, B, C, HW, groups)) return fail(EBFI_ERR_ARG, "ed_head: %s", why);
    if (!aligned16(out)) return fail(EBFI_ERR_ARG, "ed_head_forward: out must be 16-byte aligned");
    if (workspace_bytes < ebfi_ed_head_workspace(B, C, HW)) return fail(EBFI_ERR_ARG, "ed_head_forward: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t planes = (int64_t)B * C;
    const Plan pl = ed_plan(planes, HW);
    float *partial = static_cast<float *>(workspace);
    const unsigned grid = (unsigned)(planes * pl.slices);
    const int fin_threads = (int)(ceil_div(std::max(C, groups), 64) * 64);
    const size_t fin_lds = (size_t)(5 * C + 4 * groups) * sizeof(double);
    {
        ProfScope ps("ed_stats", st, 0.0, 8.0 * planes * (double)HW);
        hipLaunchKernelGGL(ed_stats_kernel, dim3(grid), dim3(ET), 0, st, ev, bl, partial, HW, pl.slices, pl.chunk);
    }
    {
        ProfScope ps("ed_finalize", st);
        hipLaunchKernelGGL(ed_fwd_finalize_kernel, dim3((unsigned)B), dim3(fin_threads), fin_lds, st, partial, gamma, beta, stats, atten, B, C,
                           groups, HW, pl.slices, eps);
    }
    {
        ProfScope ps("ed_cat", st, 0.0, 16.0 * planes * (double)HW);
        hipLaunchKernelGGL(ed_cat_kernel, dim3(grid), dim3(ET), 0, st, ev, bl, atten, out, C, HW, pl.slices, pl.chunk);
    }
    return check_launch("ed_head_forward");
}

extern "C" int ebfi_ed_head_backward(const float *grad_out, const float *ev, const float *bl, const float *gamma, const float *beta,
                                     const float *atten, const double *stats, float *grad_ev, float *grad_bl, float *grad_gamma,
                                     float *grad_beta, int B, int C, int64_t HW, int groups, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    if (!grad_out || !ev || !bl || !gamma || !beta || !atten || !stats || !grad_ev || !grad_bl || !workspace)
        return fail(EBFI_ERR_ARG, "ed_head_backward: null argument");
    if (const char *why = ed_check(ev, bl, B, C, HW, groups)) return fail(EBFI_ERR_ARG, "ed_head: %s", why);
    if (!aligned16(grad_out) || !aligned16(grad_ev) || !aligned16(grad_bl)) return fail(EBFI_ERR_ARG, "ed_head_backward: 16-byte alignment");
    if (workspace_bytes < ebfi_ed_head_workspace(B, C, HW)) return fail(EBFI_ERR_ARG, "ed_head_backward: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t planes = (int64_t)B * C;
    const Plan pl = ed_plan(planes, HW);
    float *partial = static_cast<float *>(workspace);
    float *coef = partial + planes * pl.slices * 5;
    const unsigned grid = (unsigned)(planes * pl.slices);
    const int fin_threads = (int)(ceil_div(std::max(C, groups), 64) * 64);
    const size_t fin_lds = (size_t)(4 * C + 4 * groups) * sizeof(double);
    {
        ProfScope ps("ed_plane_dot", st, 0.0, 8.0 * planes * (double)HW);
        hipLaunchKernelGGL(ed_plane_dot_kernel, dim3(grid), dim3(ET), 0, st, grad_out, ev, partial, C, HW, pl.slices, pl.chunk);
    }
    {
        ProfScope ps("ed_bwd_coeff", st);
        hipLaunchKernelGGL(ed_bwd_coeff_kernel, dim3(1), dim3(fin_threads), fin_lds, st, partial, gamma, beta, atten, stats, coef,
                           grad_gamma, grad_beta, B, C, groups, HW, pl.slices);
    }
    {
        ProfScope ps("ed_bwd_apply", st, 0.0, 24.0 * planes * (double)HW);
        hipLaunchKernelGGL(ed_bwd_apply_kernel, dim3(grid), dim3(ET), 0, st, grad_out, ev, bl, coef, grad_ev, grad_bl, C, HW, pl.slices,
                           pl.chunk);
    }
    return check_launch("ed_head_backward");
}

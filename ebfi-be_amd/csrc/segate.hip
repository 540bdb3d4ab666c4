// Squeeze-excite gate of the detail branch as TWO launches forward and THREE backward.
//
// Reference: SEGating (models/model_misc/resnet_3D.py:89-105): x * sigmoid(Conv3d_1x1x1(AdaptiveAvgPool3d(1)(x))), used
// after every conv pair of BasicBlock (:108-141, followed by `+ residual` and ReLU) and after the decoder convs
// (Conv_3d / upConv3D :382-417, followed by LeakyReLU(0.2) in UNet3d_18.forward, model_singleframe.py:200-223).
// Through PyTorch that is mean / tiny GEMM / sigmoid / broadcast multiply / add / activation = 6 small launches forward and
// ~12 backward per gate, 13 gates per step; the tensors are small (<= 8 MB), so each launch is mostly latency.  Here:
//
//   forward   plane_mean:  mean[b,c] over the N = D*H*W elements of plane (b,c)       (one workgroup per plane)
//             se_apply:    gate[b,c] = sigmoid(bias[c] + sum_k W[c,k] mean[b,k]);  out = act(x * gate (+ res))
//   backward  se_bwd_reduce:  ggate[b,c] = sum_n g'[n] x[n],  g' = grad_out * act'(out)
//             se_bwd_small:   gz = ggate * gate (1 - gate);  grad_W = gz^T mean, grad_b = sum_b gz, gmean = gz W   (one workgroup)
//             se_bwd_apply:   grad_x = g' * gate + gmean[b,c] / N;  grad_res = g'
//
// Planes are contiguous [B*C][N] fp32 (a [B,C,D,H,W] tensor as it stands), N a multiple of 4.  Reductions run in a fixed
// order (deterministic).  act: 0 none, 1 LeakyReLU(slope) (slope 0 = ReLU).
#include "common.hpp"

using namespace ebfi;

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum(float v, float *red) {      // 256 threads -> every thread gets the sum
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Round 3: the two per-plane reductions ran ONE workgroup per (b, c) plane -- 128 workgroups streaming 128-384 KB each on the
// largest gates (0.4 TB/s, 48 / 129 us).  A plane is now cut into `S` slices (grid = S x planes, like the apply kernels);
// the slice sums land in a scratch area and are added in slice order by a tiny finalising kernel (forward) / by
// se_bwd_small (backward): same fixed-order determinism, ten times the workgroups.  The scratch is the head of a tensor the
// NEXT kernel of the gate overwrites anyway (`out` forward, `grad_x` backward: planes * S <= planes * N floats).
__global__ __launch_bounds__(256) void plane_mean_kernel(const float *__restrict__ x, float *__restrict__ part, int64_t N4) {
    __shared__ float red[4];
    const int plane = blockIdx.y, S = gridDim.x;
    const f4 *p = reinterpret_cast<const f4 *>(x) + (int64_t)plane * N4;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)S * 256) {
        const f4 v = p[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[(int64_t)plane * S + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void plane_mean_finish_kernel(const float *__restrict__ part, float *__restrict__ mean, int planes,
                                                                int S, float inv_n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= planes) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(int64_t)i * S + k];
    mean[i] = s * inv_n;
}

__device__ __forceinline__ float gate_of(const float *__restrict__ mean, const float *__restrict__ W, const float *__restrict__ bias,
                                         int b, int c, int C) {
    float z = bias ? bias[c] : 0.f;
    for (int k = 0; k < C; ++k) z = fmaf(W[c * C + k], mean[b * C + k], z);      // uniform per workgroup: scalar loads
    return 1.f / (1.f + __expf(-z));
}

__global__ __launch_bounds__(256) void se_apply_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           const float *__restrict__ res, float *__restrict__ out,
                                                           float *__restrict__ gate_out, int C, int64_t N4, int act, float slope) {
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const float gate = gate_of(mean, W, bias, b, c, C);
    if (blockIdx.x == 0 && threadIdx.x == 0) gate_out[plane] = gate;
    const f4 *px = reinterpret_cast<const f4 *>(x) + (int64_t)plane * N4;
    const f4 *pr = res ? reinterpret_cast<const f4 *>(res) + (int64_t)plane * N4 : nullptr;
    f4 *po = reinterpret_cast<f4 *>(out) + (int64_t)plane * N4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)gridDim.x * 256) {
        f4 v = px[i] * gate;
        if (pr) v += pr[i];
        if (act == 1) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        }
        po[i] = v;
    }
}

__device__ __forceinline__ f4 act_mask(const f4 &g, const f4 &o, int act, float slope) {
    if (act != 1) return g;
    f4 r;
    r.x = o.x > 0.f ? g.x : g.x * slope; r.y = o.y > 0.f ? g.y : g.y * slope;
    r.z = o.z > 0.f ? g.z : g.z * slope; r.w = o.w > 0.f ? g.w : g.w * slope;
    return r;
}

__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const float *__restrict__ gout, const float *__restrict__ out,
                                                            const float *__restrict__ x, float *__restrict__ ggate, int64_t N4,
                                                            int act, float slope) {
    __shared__ float red[4];
    const int plane = blockIdx.y, S = gridDim.x;
    const int64_t base = (int64_t)plane * N4;
    const f4 *pg = reinterpret_cast<const f4 *>(gout) + base, *po = reinterpret_cast<const f4 *>(out) + base;
    const f4 *px = reinterpret_cast<const f4 *>(x) + base;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)S * 256) {
        const f4 g = act_mask(pg[i], po[i], act, slope), v = px[i];
        s += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) ggate[(int64_t)plane * S + blockIdx.x] = s;      // slice sums; se_bwd_small adds them in order
}

// one workgroup: gz[b,c] = ggate*gate*(1-gate) -> grad_W[c,k] = sum_b gz[b,c] mean[b,k], grad_b[c] = sum_b gz[b,c],
// gmean[b,k] = sum_c gz[b,c] W[c,k]      (B*C <= 4096)
__global__ __launch_bounds__(256) void se_bwd_small_kernel(const float *__restrict__ ggate, const float *__restrict__ gate,
                                                           const float *__restrict__ mean, const float *__restrict__ W,
                                                           float *__restrict__ gW, float *__restrict__ gb, float *__restrict__ gmean,
                                                           int B, int C, int S) {
    __shared__ float gz[4096];
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const float s = gate[i];
        float gg = 0.f;
        for (int k = 0; k < S; ++k) gg += ggate[(int64_t)i * S + k];          // slice sums of se_bwd_reduce, fixed order
        gz[i] = gg * s * (1.f - s);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) {
        const int c = i / C, k = i - c * C;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a = fmaf(gz[b * C + c], mean[b * C + k], a);
        gW[i] = a;
    }
    if (gb)
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f;
            for (int b = 0; b < B; ++b) a += gz[b * C + c];
            gb[c] = a;
        }
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C, k = i - b * C;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(gz[b * C + c], W[c * C + k], a);
        gmean[i] = a;
    }
}

__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const float *__restrict__ gout, const float *__restrict__ out,
                                                           const float *__restrict__ gate, const float *__restrict__ gmean,
                                                           float *__restrict__ gx, float *__restrict__ gres, int64_t N4, int act,
                                                           float slope) {
    const int plane = blockIdx.y;
    const float s = gate[plane], m = gmean[plane] / (float)(N4 * 4);
    const int64_t base = (int64_t)plane * N4;
    const f4 *pg = reinterpret_cast<const f4 *>(gout) + base, *po = reinterpret_cast<const f4 *>(out) + base;
    f4 *ox = reinterpret_cast<f4 *>(gx) + base;
    f4 *orr = gres ? reinterpret_cast<f4 *>(gres) + base : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)gridDim.x * 256) {
        const f4 g = act_mask(pg[i], po[i], act, slope);
        ox[i] = g * s + m;
        if (orr) orr[i] = g;
    }
}

int check(const char *who, int B, int C, int64_t N) {
    if (B < 0 || C <= 0 || N <= 0) return fail(EBFI_ERR_ARG, "%s: bad dimensions", who);
    if (N % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "%s: plane size must be a multiple of 4 (got %lld)", who, (long long)N);
    if ((int64_t)B * C > 4096) return fail(EBFI_ERR_UNSUPPORTED, "%s: B*C = %lld > 4096", who, (long long)B * C);
    return EBFI_OK;
}
unsigned slices(int64_t N4) {
    const int64_t s = (N4 + 256 * 8 - 1) / (256 * 8);       // ~8 vectors per thread
    return (unsigned)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

}  // namespace

extern "C" int ebfi_se_gate_forward(const float *x, const float *weight, const float *bias, const float *res, float *out,
                                    float *mean, float *gate, int B, int C, int64_t N, int act, float slope, void *stream) {
    if (!x || !weight || !out || !mean || !gate) return fail(EBFI_ERR_ARG, "se_gate_forward: null argument");
    if (act < 0 || act > 1) return fail(EBFI_ERR_ARG, "se_gate_forward: activation %d (0 none, 1 leaky)", act);
    if (int rc = check("se_gate_forward", B, C, N)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("se_gate_fwd", st, 0.0, 4.0 * B * C * (double)N * (res ? 4 : 3));
        const unsigned S = slices(N / 4);
        hipLaunchKernelGGL(plane_mean_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, x, out, N / 4);   // slice sums -> head of `out`
        hipLaunchKernelGGL(plane_mean_finish_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, st, out, mean, B * C, (int)S,
                           1.f / (float)N);
        hipLaunchKernelGGL(se_apply_fwd_kernel, dim3(slices(N / 4), (unsigned)(B * C)), dim3(256), 0, st, x, mean, weight, bias, res,
                           out, gate, C, N / 4, act, slope);
    }
    return check_launch("se_gate_fwd");
}

// workspace: 2*B*C floats
extern "C" int ebfi_se_gate_backward(const float *grad_out, const float *out, const float *x, const float *weight,
                                     const float *gate, const float *mean, float *grad_x, float *grad_res, float *grad_weight,
                                     float *grad_bias, float *workspace, int B, int C, int64_t N, int act, float slope,
                                     void *stream) {
    if (!grad_out || !x || !weight || !gate || !mean || !grad_x || !grad_weight || !workspace || (act == 1 && !out))
        return fail(EBFI_ERR_ARG, "se_gate_backward: null argument");
    if (act < 0 || act > 1) return fail(EBFI_ERR_ARG, "se_gate_backward: activation %d", act);
    if (int rc = check("se_gate_backward", B, C, N)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *gmean = workspace + (size_t)B * C;      // (workspace[0 .. B*C) is no longer used: the slice sums live in grad_x)
    const float *o = out ? out : grad_out;      // act == 0: `out` is not read through the mask
    {
        ProfScope ps("se_gate_bwd", st, 0.0, 4.0 * B * C * (double)N * (grad_res ? 7 : 6));
        const unsigned S = slices(N / 4);
        // slice sums into the head of grad_x (B*C*S <= B*C*N floats; se_bwd_apply overwrites it afterwards); grad_x must not
        // alias grad_out / out / x, which the callers guarantee (freshly allocated)
        hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, grad_out, o, x, grad_x, N / 4, act, slope);
        hipLaunchKernelGGL(se_bwd_small_kernel, dim3(1), dim3(256), 0, st, grad_x, gate, mean, weight, grad_weight, grad_bias, gmean, B, C,
                           (int)S);
        hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(slices(N / 4), (unsigned)(B * C)), dim3(256), 0, st, grad_out, o, gate, gmean,
                           grad_x, grad_res, N / 4, act, slope);
    }
    return check_launch("se_gate_bwd");
}

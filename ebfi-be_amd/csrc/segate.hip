// Squeeze-excite gate of the detail branch as TWO launches forward and TWO backward.
//
// Reference: SEGating (models/model_misc/resnet_3D.py:89-105): x * sigmoid(Conv3d_1x1x1(AdaptiveAvgPool3d(1)(x))), used
// after every conv pair of BasicBlock (:108-141, followed by `+ residual` and ReLU) and after the decoder convs
// (Conv_3d / upConv3D :382-417, followed by LeakyReLU(0.2) in UNet3d_18.forward, model_singleframe.py:200-223).
// Through PyTorch that is mean / tiny GEMM / sigmoid / broadcast multiply / add / activation = 6 small launches forward and
// ~12 backward per gate, 13 gates per step; the tensors are small (<= 8 MB), so each launch is mostly latency.  Here:
//
//   forward   plane_mean:  slice sums of plane (b,c) over its N = D*H*W elements
//             se_apply:    mean from the slice sums; gate[b,c] = sigmoid(bias[c] + sum_k W[c,k] mean[b,k]);  out = act(x * gate (+ res))
//   backward  se_bwd_reduce:  slice sums of ggate[b,c] = sum_n g'[n] x[n],  g' = grad_out * act'(out)
//             se_bwd_apply:   gz = ggate * gate (1 - gate);  gmean = gz W;  grad_x = g' * gate + gmean[b,c] / N;  grad_res = g';
//                             its slice-0 workgroups of sample 0 write grad_W = gz^T mean and grad_b = sum_b gz
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
// the slice sums land in a caller-provided scratch area [planes][S] and every workgroup of the APPLY kernel adds the ones it
// needs in slice order in its prologue (C * S floats from L2) -- the tiny finalising launches in between (plane_mean_finish,
// se_bwd_small: 5-6 us each, 26 per step, pure launch latency) are gone.  Fixed summation order: deterministic.
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

__device__ __forceinline__ float slice_sum(const float *__restrict__ part, int plane, int S) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(int64_t)plane * S + k];
    return s;
}

// gate[b,c] = sigmoid(bias[c] + sum_k W[c,k] mean[b,k]) with mean[b,k] = inv_n * (slice sums of plane (b,k)); thread t takes
// k = t, t + 256, ...; the workgroup of slice 0 also publishes mean[b,c] and the gate for the backward pass
__global__ __launch_bounds__(256) void se_apply_fwd_kernel(const float *__restrict__ x, const float *__restrict__ part, int S,
                                                           float inv_n, const float *__restrict__ W, const float *__restrict__ bias,
                                                           const float *__restrict__ res, float *__restrict__ out,
                                                           float *__restrict__ mean_out, float *__restrict__ gate_out, int C,
                                                           int64_t N4, int act, float slope) {
    __shared__ float red[4];
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    float zp = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) {
        const float m = slice_sum(part, b * C + k, S) * inv_n;
        zp = fmaf(W[c * C + k], m, zp);
        if (k == c && blockIdx.x == 0) mean_out[plane] = m;
    }
    const float z = block_sum(zp, red) + (bias ? bias[c] : 0.f);
    const float gate = 1.f / (1.f + __expf(-z));
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
    if (threadIdx.x == 0) ggate[(int64_t)plane * S + blockIdx.x] = s;      // slice sums, added in order by se_bwd_apply
}

// gz[b,c] = ggate[b,c] * gate (1 - gate) with ggate = slice sums of se_bwd_reduce
__device__ __forceinline__ float gz_of(const float *__restrict__ part, const float *__restrict__ gate, int plane, int S) {
    const float s = gate[plane];
    return slice_sum(part, plane, S) * s * (1.f - s);
}

// grad_x = g' * gate + gmean[b,c] / N with gmean[b,c] = sum_c' gz[b,c'] W[c',c] formed in the prologue (thread t: c' = t, t + 256, ..);
// grad_res = g'.  The slice-0 workgroups of sample 0 also write row c of grad_W (= sum_b gz[b,c] mean[b,k]) and grad_b[c].
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const float *__restrict__ gout, const float *__restrict__ out,
                                                           const float *__restrict__ gate, const float *__restrict__ part, int S,
                                                           const float *__restrict__ mean, const float *__restrict__ W,
                                                           float *__restrict__ gx, float *__restrict__ gres, float *__restrict__ gW,
                                                           float *__restrict__ gb, int B, int C, int64_t N4, int act, float slope) {
    __shared__ float red[4];
    __shared__ float gzb[64];
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    float gp = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) gp = fmaf(gz_of(part, gate, b * C + k, S), W[k * C + c], gp);
    const float m = block_sum(gp, red) / (float)(N4 * 4);
    if (blockIdx.x == 0 && b == 0) {                       // (uniform per workgroup)
        for (int bb0 = 0; bb0 < B; bb0 += 64) {            // gz[., c] of up to 64 samples at a time
            __syncthreads();
            if (threadIdx.x < 64 && bb0 + (int)threadIdx.x < B) gzb[threadIdx.x] = gz_of(part, gate, (bb0 + threadIdx.x) * C + c, S);
            __syncthreads();
            const int nb = min(64, B - bb0);
            for (int k = threadIdx.x; k < C; k += 256) {
                float a = bb0 ? gW[c * C + k] : 0.f;
                for (int bb = 0; bb < nb; ++bb) a = fmaf(gzb[bb], mean[(bb0 + bb) * C + k], a);
                gW[c * C + k] = a;
            }
            if (gb && threadIdx.x == 0) {
                float a = bb0 ? gb[c] : 0.f;
                for (int bb = 0; bb < nb; ++bb) a += gzb[bb];
                gb[c] = a;
            }
        }
    }
    const float s = gate[plane];
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

// ---- round 6: the gate of an up-convolution stage reads the transposed convolution's output THROUGH the pixel shuffle -----------
// ConvTranspose3d (3,4,4)/(1,2,2) runs as a 3x3 convolution to 8*Co channels + PixelShuffle (ebfi_amd/fold3d.py): conv channel
// (c, q), q = (d*2 + py)*2 + px, pixel (y, x) is element (d, 2y + py, 2x + px) of plane c of the stage's [B, Co, 2, 2h, 2w] tensor.
// The eight conv channels of a plane are one contiguous block of 8*h*w floats -- the SAME block as the shuffled plane, in another
// order: the plane mean needs no change, and the apply kernels walk the SHUFFLED order (4 consecutive X = two conv pixels x two
// px phases) and address the unshuffled side with two 8-byte accesses.  The PixelShuffle copy (a full read + write of the
// stage's largest tensor, forward and again backward) is gone.  h, w even.
struct PsGeom {
    int h, w;        // conv-output spatial size (the plane is [2][2h][2w])
};
__device__ __forceinline__ void ps_offsets(const PsGeom &p, int64_t i, int64_t &u0, int64_t &u1) {
    const int wq = p.w >> 1;                          // f4 per shuffled row (2w / 4)
    const int xq = (int)(i % wq);
    const int64_t r = i / wq;
    const int Y = (int)(r % (2 * p.h)), d = (int)(r / (2 * p.h));
    const int q0 = (d * 2 + (Y & 1)) * 2;
    const int64_t hw = (int64_t)p.h * p.w;
    u0 = (int64_t)q0 * hw + (int64_t)(Y >> 1) * p.w + 2 * xq;      // conv channel q0 (px = 0), pixels (y, 2 xq), (y, 2 xq + 1)
    u1 = u0 + hw;                                                   // conv channel q0 + 1 (px = 1)
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4 ps_load(const float *__restrict__ plane, const PsGeom &p, int64_t i) {
    int64_t u0, u1;
    ps_offsets(p, i, u0, u1);
    const f2 a = *reinterpret_cast<const f2 *>(plane + u0), b = *reinterpret_cast<const f2 *>(plane + u1);
    return f4{a.x, b.x, a.y, b.y};
}
__device__ __forceinline__ void ps_store(float *__restrict__ plane, const PsGeom &p, int64_t i, const f4 &v) {
    int64_t u0, u1;
    ps_offsets(p, i, u0, u1);
    *reinterpret_cast<f2 *>(plane + u0) = f2{v.x, v.z};
    *reinterpret_cast<f2 *>(plane + u1) = f2{v.y, v.w};
}

__global__ __launch_bounds__(256) void se_apply_fwd_ps_kernel(const float *__restrict__ x, const float *__restrict__ part, int S,
                                                              float inv_n, const float *__restrict__ W, const float *__restrict__ bias,
                                                              float *__restrict__ out, float *__restrict__ mean_out,
                                                              float *__restrict__ gate_out, int C, int64_t N4, int act, float slope,
                                                              PsGeom ps) {
    __shared__ float red[4];
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    float zp = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) {
        const float m = slice_sum(part, b * C + k, S) * inv_n;
        zp = fmaf(W[c * C + k], m, zp);
        if (k == c && blockIdx.x == 0) mean_out[plane] = m;
    }
    const float z = block_sum(zp, red) + (bias ? bias[c] : 0.f);
    const float gate = 1.f / (1.f + __expf(-z));
    if (blockIdx.x == 0 && threadIdx.x == 0) gate_out[plane] = gate;
    const float *px = x + (int64_t)plane * N4 * 4;
    f4 *po = reinterpret_cast<f4 *>(out) + (int64_t)plane * N4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)gridDim.x * 256) {
        f4 v = ps_load(px, ps, i) * gate;
        if (act == 1) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        }
        po[i] = v;
    }
}

__global__ __launch_bounds__(256) void se_bwd_reduce_ps_kernel(const float *__restrict__ gout, const float *__restrict__ out,
                                                               const float *__restrict__ x, float *__restrict__ ggate, int64_t N4,
                                                               int act, float slope, PsGeom ps) {
    __shared__ float red[4];
    const int plane = blockIdx.y, S = gridDim.x;
    const int64_t base = (int64_t)plane * N4;
    const f4 *pg = reinterpret_cast<const f4 *>(gout) + base, *po = reinterpret_cast<const f4 *>(out) + base;
    const float *px = x + base * 4;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)S * 256) {
        const f4 g = act_mask(pg[i], po[i], act, slope), v = ps_load(px, ps, i);
        s += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) ggate[(int64_t)plane * S + blockIdx.x] = s;
}

// (the prologue -- gmean, grad_W, grad_b -- is se_bwd_apply_kernel's; grad_x leaves in the UNSHUFFLED layout: it is the gradient of
//  the convolution's output)
__global__ __launch_bounds__(256) void se_bwd_apply_ps_kernel(const float *__restrict__ gout, const float *__restrict__ out,
                                                              const float *__restrict__ gate, const float *__restrict__ part, int S,
                                                              const float *__restrict__ mean, const float *__restrict__ W,
                                                              float *__restrict__ gx, float *__restrict__ gW, float *__restrict__ gb,
                                                              int B, int C, int64_t N4, int act, float slope, PsGeom ps) {
    __shared__ float red[4];
    __shared__ float gzb[64];
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    float gp = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) gp = fmaf(gz_of(part, gate, b * C + k, S), W[k * C + c], gp);
    const float m = block_sum(gp, red) / (float)(N4 * 4);
    if (blockIdx.x == 0 && b == 0) {
        for (int bb0 = 0; bb0 < B; bb0 += 64) {
            __syncthreads();
            if (threadIdx.x < 64 && bb0 + (int)threadIdx.x < B) gzb[threadIdx.x] = gz_of(part, gate, (bb0 + threadIdx.x) * C + c, S);
            __syncthreads();
            const int nb = min(64, B - bb0);
            for (int k = threadIdx.x; k < C; k += 256) {
                float a = bb0 ? gW[c * C + k] : 0.f;
                for (int bb = 0; bb < nb; ++bb) a = fmaf(gzb[bb], mean[(bb0 + bb) * C + k], a);
                gW[c * C + k] = a;
            }
            if (gb && threadIdx.x == 0) {
                float a = bb0 ? gb[c] : 0.f;
                for (int bb = 0; bb < nb; ++bb) a += gzb[bb];
                gb[c] = a;
            }
        }
    }
    const float s = gate[plane];
    const int64_t base = (int64_t)plane * N4;
    const f4 *pg = reinterpret_cast<const f4 *>(gout) + base, *po = reinterpret_cast<const f4 *>(out) + base;
    float *ox = gx + base * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (int64_t)gridDim.x * 256) {
        const f4 g = act_mask(pg[i], po[i], act, slope);
        ps_store(ox, ps, i, g * s + m);
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

// scratch floats both directions need for the slice sums of (B, C, N)
extern "C" size_t ebfi_se_gate_workspace(int B, int C, int64_t N) {
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    return (size_t)B * C * slices(N / 4);
}

extern "C" int ebfi_se_gate_forward(const float *x, const float *weight, const float *bias, const float *res, float *out,
                                    float *mean, float *gate, float *workspace, int B, int C, int64_t N, int act, float slope,
                                    void *stream) {
    if (!x || !weight || !out || !mean || !gate || !workspace) return fail(EBFI_ERR_ARG, "se_gate_forward: null argument");
    if (act < 0 || act > 1) return fail(EBFI_ERR_ARG, "se_gate_forward: activation %d (0 none, 1 leaky)", act);
    if (int rc = check("se_gate_forward", B, C, N)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("se_gate_fwd", st, 0.0, 4.0 * B * C * (double)N * (res ? 4 : 3));
        const unsigned S = slices(N / 4);
        hipLaunchKernelGGL(plane_mean_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, x, workspace, N / 4);
        hipLaunchKernelGGL(se_apply_fwd_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, x, workspace, (int)S, 1.f / (float)N,
                           weight, bias, res, out, mean, gate, C, N / 4, act, slope);
    }
    return check_launch("se_gate_fwd");
}

// workspace: ebfi_se_gate_workspace(B, C, N) floats
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
    const float *o = out ? out : grad_out;      // act == 0: `out` is not read through the mask
    {
        ProfScope ps("se_gate_bwd", st, 0.0, 4.0 * B * C * (double)N * (grad_res ? 7 : 6));
        const unsigned S = slices(N / 4);
        hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, grad_out, o, x, workspace, N / 4, act, slope);
        hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, grad_out, o, gate, workspace, (int)S, mean,
                           weight, grad_x, grad_res, grad_weight, grad_bias, B, C, N / 4, act, slope);
    }
    return check_launch("se_gate_bwd");
}

// The same gate on the output of a transposed convolution that is still UNSHUFFLED (round 6; see PsGeom above): x = the 3x3
// convolution's output [B, 8*C, h, w]; out / grad_out = the stage's tensor [B, C, 2, 2h, 2w]; grad_x in x's layout.  No residual.
extern "C" int ebfi_se_gate_forward_ps(const float *x, const float *weight, const float *bias, float *out, float *mean, float *gate,
                                       float *workspace, int B, int C, int h, int w, int act, float slope, void *stream) {
    if (!x || !weight || !out || !mean || !gate || !workspace) return fail(EBFI_ERR_ARG, "se_gate_forward_ps: null argument");
    if (act < 0 || act > 1) return fail(EBFI_ERR_ARG, "se_gate_forward_ps: activation %d (0 none, 1 leaky)", act);
    if (h <= 0 || w <= 0 || (w & 1)) return fail(EBFI_ERR_UNSUPPORTED, "se_gate_forward_ps: %d x %d (w must be even)", h, w);
    const int64_t N = 8LL * h * w;
    if (int rc = check("se_gate_forward_ps", B, C, N)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("se_gate_fwd/shuffle", st, 0.0, 4.0 * B * C * (double)N * 3);
        const unsigned S = slices(N / 4);
        hipLaunchKernelGGL(plane_mean_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, x, workspace, N / 4);
        hipLaunchKernelGGL(se_apply_fwd_ps_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, x, workspace, (int)S, 1.f / (float)N,
                           weight, bias, out, mean, gate, C, N / 4, act, slope, PsGeom{h, w});
    }
    return check_launch("se_gate_fwd/shuffle");
}

extern "C" int ebfi_se_gate_backward_ps(const float *grad_out, const float *out, const float *x, const float *weight,
                                        const float *gate, const float *mean, float *grad_x, float *grad_weight, float *grad_bias,
                                        float *workspace, int B, int C, int h, int w, int act, float slope, void *stream) {
    if (!grad_out || !x || !weight || !gate || !mean || !grad_x || !grad_weight || !workspace || (act == 1 && !out))
        return fail(EBFI_ERR_ARG, "se_gate_backward_ps: null argument");
    if (act < 0 || act > 1) return fail(EBFI_ERR_ARG, "se_gate_backward_ps: activation %d", act);
    if (h <= 0 || w <= 0 || (w & 1)) return fail(EBFI_ERR_UNSUPPORTED, "se_gate_backward_ps: %d x %d (w must be even)", h, w);
    const int64_t N = 8LL * h * w;
    if (int rc = check("se_gate_backward_ps", B, C, N)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *o = out ? out : grad_out;
    {
        ProfScope ps("se_gate_bwd/shuffle", st, 0.0, 4.0 * B * C * (double)N * 6);
        const unsigned S = slices(N / 4);
        hipLaunchKernelGGL(se_bwd_reduce_ps_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, grad_out, o, x, workspace, N / 4, act, slope,
                           PsGeom{h, w});
        hipLaunchKernelGGL(se_bwd_apply_ps_kernel, dim3(S, (unsigned)(B * C)), dim3(256), 0, st, grad_out, o, gate, workspace, (int)S, mean,
                           weight, grad_x, grad_weight, grad_bias, B, C, N / 4, act, slope, PsGeom{h, w});
    }
    return check_launch("se_gate_bwd/shuffle");
}

// Laplacian-pyramid L1 loss of the training step as one pyramid of the DIFFERENCE image.
//
// Reference: loss/restore.py:149-213 (GaussianConv 5x5 reflect, LaplacianPyramid with avg_pool2d reduce and
// zero-insertion expand x4, LaplacianLoss = sum_i 2^i * L1sum(lap_i(x), lap_i(y))), applied to two predictions
// against the same target in train_ours.py:258-268.  Every pyramid operator is linear, so
// lap_i(x) - lap_i(y) = lap_i(x - y): ONE pyramid over the planes [a - t ; b - t] replaces the three pyramids of a
// step, and each level is two kernels (reduce; expand + subtract + |.| + partial sums) instead of ~15 elementwise
// passes.  Tap order inside the blur and its adjoint is the one of gauss5_fwd / gauss5_bwd (imgops.hip), so a level
// computes bit-for-bit what the operator-by-operator path computes on the same planes.
//
// Workspace (floats): level images cur_0 .. cur_{L-1}, level l at offset sum_{k<l} planes*H*W/4^k.  Forward
// overwrites cur_l with s_l = coef(plane) * 2^l * sign(lap_l) (what the backward needs); backward consumes the
// workspace in place (g_{l+1} -= 4 G^T(s_l) at even positions; s_l += G^T(P^T g_{l+1})) -> it can run once per forward.
#include "common.hpp"

using namespace ebfi;

namespace {

constexpr int LT = 256;

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) return -i;
    if (i >= n) return 2 * (n - 1) - i;
    return i;
}

struct Coef {
    float c[2];
    int64_t planes_per_term;
};

__device__ __forceinline__ float block_sum(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < LT / 64; ++w) s += red[w];
    return s;
}

// cur_0 = [a - t ; b - t] (b may be null: one term)
__global__ __launch_bounds__(LT) void lap_diff_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      const float *__restrict__ t, float *__restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * LT + threadIdx.x;
    if (i >= n) return;
    d[i] = a[i] - t[i];
    if (b) d[n + i] = b[i] - t[i];
}

// red = avg_pool2d(gauss5(cur), 2): the four blurred values of a 2x2 cell from one 6x6 window
__global__ __launch_bounds__(LT) void lap_reduce_kernel(const float *__restrict__ cur, float *__restrict__ red, int64_t planes,
                                                        int H, int W) {
    const int h = H >> 1, w = W >> 1;
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    const int64_t hw = (int64_t)h * w;
    if (idx >= planes * hw) return;
    const int64_t p = idx / hw;
    const int y = (int)((idx - p * hw) / w), x = (int)(idx - p * hw - (int64_t)y * w);
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    const float *src = cur + p * (int64_t)H * W;
    int cx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) cx[j] = reflect_idx(2 * x + j - 2, W);
    float h0[6], h1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float *row = src + (int64_t)reflect_idx(2 * y + i - 2, H) * W;
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = row[cx[j]];
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            a += k[j] * v[j];
            b += k[j] * v[j + 1];
        }
        h0[i] = a;
        h1[i] = b;
    }
    float g00 = 0.f, g01 = 0.f, g10 = 0.f, g11 = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        g00 += k[i] * h0[i];
        g01 += k[i] * h1[i];
        g10 += k[i] * h0[i + 1];
        g11 += k[i] * h1[i + 1];
    }
    red[idx] = (((g00 + g01) + g10) + g11) * 0.25f;
}

// lap = cur - 4 * gauss5(zero_insert(red));  partial[block] = sum coef * weight * |lap|;  cur <- coef * weight * sign(lap)
// (red == nullptr: last level, lap = cur)
__global__ __launch_bounds__(LT) void lap_level_kernel(float *__restrict__ cur, const float *__restrict__ red,
                                                       float *__restrict__ partial, int64_t planes, int H, int W, float weight,
                                                       Coef cf) {
    __shared__ float sred[LT / 64];
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    const int64_t hw = (int64_t)H * W;
    float contrib = 0.f;
    if (idx < planes * hw) {
        const int64_t p = idx / hw;
        const int y = (int)((idx - p * hw) / W), x = (int)(idx - p * hw - (int64_t)y * W);
        float lap = cur[idx];
        if (red) {
            const int w = W >> 1;
            const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
            const float *src = red + p * (hw >> 2);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int yy = reflect_idx(y + i - 2, H);
                float hsum = 0.f;
                if (!(yy & 1)) {
                    const float *row = src + (int64_t)(yy >> 1) * w;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int xx = reflect_idx(x + j - 2, W);
                        if (!(xx & 1)) hsum += k[j] * row[xx >> 1];
                    }
                }
                acc += k[i] * hsum;
            }
            lap -= acc * 4.f;
        }
        const float cw = cf.c[p >= cf.planes_per_term ? 1 : 0] * weight;
        contrib = cw * fabsf(lap);
        cur[idx] = lap > 0.f ? cw : (lap < 0.f ? -cw : 0.f);
    }
    const float s = block_sum(contrib, sred);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// adjoint of the reflect-padded blur at (y, x): sum over the virtual positions the padding maps onto (y, x)
template <class F> __device__ __forceinline__ float gauss5_adj(int y, int x, int H, int W, F at) {
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    int vy[3], vx[3], ny = 0, nx = 0;
    vy[ny++] = y;
    if (y >= 1 && y <= 2) vy[ny++] = -y;
    if (y >= H - 3 && y <= H - 2) vy[ny++] = 2 * (H - 1) - y;
    vx[nx++] = x;
    if (x >= 1 && x <= 2) vx[nx++] = -x;
    if (x >= W - 3 && x <= W - 2) vx[nx++] = 2 * (W - 1) - x;
    float acc = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int i = 0; i < 5; ++i) {
            const int oy = vy[a] - i + 2;
            if (oy < 0 || oy >= H) continue;
            float hsum = 0.f;
            for (int b = 0; b < nx; ++b)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int ox = vx[b] - j + 2;
                    if (ox >= 0 && ox < W) hsum += k[j] * at(oy, ox);
                }
            acc += k[i] * hsum;
        }
    return acc;
}

// g[y,x] (H/2 x W/2) += -4 * G^T(s)[2y, 2x]      (gradient reaching `red` through the expand path)
__global__ __launch_bounds__(LT) void lap_bwd_reduce_kernel(const float *__restrict__ s, float *__restrict__ g, int64_t planes,
                                                            int H, int W) {
    const int h = H >> 1, w = W >> 1;
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    const int64_t hw = (int64_t)h * w;
    if (idx >= planes * hw) return;
    const int64_t p = idx / hw;
    const int y = (int)((idx - p * hw) / w), x = (int)(idx - p * hw - (int64_t)y * w);
    const float *src = s + p * (int64_t)H * W;
    const float adj = gauss5_adj(2 * y, 2 * x, H, W, [&](int oy, int ox) { return src[(int64_t)oy * W + ox]; });
    g[idx] = g[idx] + (-(adj * 4.f));
}

// out[y,x] = (s[y,x] + G^T(P^T g)[y,x]) * scale,  P^T g = g[y/2, x/2] / 4       (out may alias s)
__global__ __launch_bounds__(LT) void lap_bwd_expand_kernel(const float *s, const float *__restrict__ g, float *out,
                                                            const float *__restrict__ scale, int64_t planes, int H, int W) {
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    const int64_t hw = (int64_t)H * W;
    if (idx >= planes * hw) return;
    const int64_t p = idx / hw;
    const int y = (int)((idx - p * hw) / W), x = (int)(idx - p * hw - (int64_t)y * W);
    const int w = W >> 1;
    const float *src = g + p * (hw >> 2);
    const float adj = gauss5_adj(y, x, H, W, [&](int oy, int ox) { return src[(int64_t)(oy >> 1) * w + (ox >> 1)] * 0.25f; });
    float v = s[idx] + adj;
    if (scale) v *= scale[0];
    out[idx] = v;
}

bool lap_dims_ok(int64_t planes_per_term, int H, int W, int levels) {
    if (planes_per_term <= 0 || levels < 2 || levels > 8 || H < 3 || W < 3) return false;
    const int m = 1 << (levels - 1);
    if (H % m || W % m) return false;
    return H / (m / 2) >= 3 && W / (m / 2) >= 3;   // the coarsest blurred level still reflects
}

}  // namespace

extern "C" int64_t ebfi_laploss_workspace_floats(int64_t planes, int H, int W, int levels) {
    int64_t n = 0;
    for (int l = 0; l < levels; ++l) n += planes * (int64_t)(H >> l) * (W >> l);
    return n;
}

extern "C" int64_t ebfi_laploss_partials(int64_t planes, int H, int W, int levels) {
    int64_t n = 0;
    for (int l = 0; l < levels; ++l) n += ceil_div(planes * (int64_t)(H >> l) * (W >> l), LT);
    return n;
}

extern "C" int ebfi_laploss_forward(const float *pred_a, const float *pred_b, const float *target, float coef_a, float coef_b,
                                    float *workspace, float *partial, int64_t planes_per_term, int H, int W, int levels,
                                    void *stream) {
    if (!pred_a || !target || !workspace || !partial) return fail(EBFI_ERR_ARG, "laploss_forward: null argument");
    if (!lap_dims_ok(planes_per_term, H, W, levels))
        return fail(EBFI_ERR_ARG, "laploss_forward: H, W must be multiples of 2^(levels-1) with >= 3 pixels on the coarsest blurred level");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t planes = planes_per_term * (pred_b ? 2 : 1);
    const int64_t n0 = planes_per_term * (int64_t)H * W;
    Coef cf{{coef_a, coef_b}, planes_per_term};
    {
        ProfScope ps("lap_diff", st, 0.0, 4.0 * n0 * (pred_b ? 5 : 3));
        hipLaunchKernelGGL(lap_diff_kernel, dim3((unsigned)ceil_div(n0, LT)), dim3(LT), 0, st, pred_a, pred_b, target, workspace, n0);
    }
    float *cur = workspace;
    for (int l = 0; l + 1 < levels; ++l) {
        const int h = H >> l, w = W >> l;
        const int64_t n = planes * (int64_t)h * w;
        ProfScope ps("lap_reduce", st, 0.0, 5.0 * n);
        hipLaunchKernelGGL(lap_reduce_kernel, dim3((unsigned)ceil_div(n / 4, LT)), dim3(LT), 0, st, cur, cur + n, planes, h, w);
        cur += n;
    }
    cur = workspace;
    float *part = partial;
    for (int l = 0; l < levels; ++l) {
        const int h = H >> l, w = W >> l;
        const int64_t n = planes * (int64_t)h * w;
        const bool last = l + 1 == levels;
        ProfScope ps("lap_level", st, 0.0, (last ? 8.0 : 9.0) * n);
        hipLaunchKernelGGL(lap_level_kernel, dim3((unsigned)ceil_div(n, LT)), dim3(LT), 0, st, cur, last ? nullptr : cur + n, part,
                           planes, h, w, (float)(1 << l), cf);
        part += ceil_div(n, LT);
        cur += n;
    }
    return check_launch("laploss_forward");
}

extern "C" int ebfi_laploss_backward(const float *grad_loss, float *workspace, float *grad_pred, int64_t planes, int H, int W,
                                     int levels, void *stream) {
    if (!grad_loss || !workspace || !grad_pred) return fail(EBFI_ERR_ARG, "laploss_backward: null argument");
    if (!lap_dims_ok(planes, H, W, levels)) return fail(EBFI_ERR_ARG, "laploss_backward: bad dimensions");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t off[9];
    off[0] = 0;
    for (int l = 0; l < levels; ++l) off[l + 1] = off[l] + planes * (int64_t)(H >> l) * (W >> l);
    for (int l = levels - 2; l >= 0; --l) {
        const int h = H >> l, w = W >> l;
        const int64_t n = planes * (int64_t)h * w;
        float *s = workspace + off[l], *g = workspace + off[l + 1];
        {
            ProfScope ps("lap_bwd_reduce", st, 0.0, 6.0 * n);
            hipLaunchKernelGGL(lap_bwd_reduce_kernel, dim3((unsigned)ceil_div(n / 4, LT)), dim3(LT), 0, st, s, g, planes, h, w);
        }
        {
            ProfScope ps("lap_bwd_expand", st, 0.0, 9.0 * n);
            hipLaunchKernelGGL(lap_bwd_expand_kernel, dim3((unsigned)ceil_div(n, LT)), dim3(LT), 0, st, s, g, l == 0 ? grad_pred : s,
                               l == 0 ? grad_loss : nullptr, planes, h, w);
        }
    }
    return check_launch("laploss_backward");
}

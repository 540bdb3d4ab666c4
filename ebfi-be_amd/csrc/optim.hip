// Adam update of the training step (train_ours.py:276-277: optimizer.step() of torch.optim.Adam, config train_ours.yml
// :59-65) over the ONE flat parameter buffer the engine trains (ebfi_amd.dp.FlatAdam): 5.7 M elements in one pass of
// 16-byte accesses.  PyTorch's fused multi-tensor kernel walks a single tensor in 64 K-element chunks -- 87 workgroups
// for this model, 112 us; this launch is bandwidth-bound (7 maps of 22.8 MB).
// Arithmetic follows torch's fused Adam: m <- m + (1-b1)(g - m); v <- b2 v + (1-b2) g^2; bias corrections 1 - b^t in
// double precision; p <- p - (lr / c1) m / (sqrt(v) / sqrt(c2) + eps).
#include "common.hpp"

using namespace ebfi;

namespace {

__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, float *__restrict__ step, int64_t n4,
                                                        int64_t n, double lr, double beta1, double beta2, double eps,
                                                        int *__restrict__ guard) {
    // guard (optional): guard[0] != 0 marks this step's gradient as unusable (an fp16 operand of the backward pass left its
    // range, conv2d_f16.inc.hpp): nothing is updated, the step count is taken back, guard[1] counts the skipped steps
    if (guard != nullptr && guard[0] != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            step[0] -= 1.f;
            guard[1] += 1;
        }
        return;
    }
    const double t = (double)step[0];
    const double c1 = 1.0 - pow(beta1, t), c2 = 1.0 - pow(beta2, t);
    const float step_size = (float)(lr / c1), c2s = (float)sqrt(c2);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2), e = (float)eps;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        mm = mm + w1 * (gg - mm);
        vv = b2 * vv + w2 * gg * gg;
        const float denom = sqrtf(vv) / c2s + e;
        pp -= step_size * mm / denom;
    };
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        float4 P = reinterpret_cast<float4 *>(p)[i], M = reinterpret_cast<float4 *>(m)[i], V = reinterpret_cast<float4 *>(v)[i];
        const float4 G = reinterpret_cast<const float4 *>(g)[i];
        upd(P.x, G.x, M.x, V.x);
        upd(P.y, G.y, M.y, V.y);
        upd(P.z, G.z, M.z, V.z);
        upd(P.w, G.w, M.w, V.w);
        reinterpret_cast<float4 *>(p)[i] = P;
        reinterpret_cast<float4 *>(m)[i] = M;
        reinterpret_cast<float4 *>(v)[i] = V;
    }
    if (i == 0)
        for (int64_t k = n4 * 4; k < n; ++k) upd(p[k], g[k], m[k], v[k]);
}

}  // namespace

extern "C" int ebfi_adam_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, int64_t n,
                                      double lr, double beta1, double beta2, double eps, int *guard, void *stream);

extern "C" int ebfi_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *step, int64_t n,
                              double lr, double beta1, double beta2, double eps, void *stream) {
    return ebfi_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, const_cast<float *>(step), n, lr, beta1, beta2, eps, nullptr, stream);
}

extern "C" int ebfi_adam_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, int64_t n,
                                      double lr, double beta1, double beta2, double eps, int *guard, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n < 0) return fail(EBFI_ERR_ARG, "adam_step: null argument");
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
        return fail(EBFI_ERR_ARG, "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n4 = n / 4;
    {
        ProfScope ps("adam_flat", st, 0.0, 28.0 * (double)n);
        hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)ceil_div(std::max<int64_t>(n4, 1), 256)), dim3(256), 0, st, param, grad,
                           exp_avg, exp_avg_sq, step, n4, n, lr, beta1, beta2, eps, guard);
    }
    return check_launch("adam_flat");
}

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
                                                        int *__restrict__ guard, const float *__restrict__ flag) {
    // guard (optional): guard[0] != 0 marks this step's gradient as unusable (an fp16 operand of the backward pass left its
    // range, conv2d_f16.inc.hpp): nothing is updated, the step count is taken back, guard[1] counts the skipped steps.
    // flag (optional, with guard): the ranks' guard flags summed by the gradient all-reduce (ebfi_grad_gather put this rank's
    // into the wire buffer); anything but an exact zero skips the step on every rank and is written back to guard[0].
    // (Every thread reads the same two words; thread 0 of block 0 is the only writer and writes only values that keep the
    // other threads' decision: guard[0] becomes non-zero only when the flag already said "skip".)
    if (guard != nullptr) {
        const bool by_flag = flag != nullptr && !(flag[0] == 0.f);
        if (guard[0] != 0 || by_flag) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                step[0] -= 1.f;
                guard[1] += 1;
                if (by_flag) guard[0] = 1;
            }
            return;
        }
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

// Gradient packing of the training step (ebfi_amd.dp.FlatGradBucket.gather): the per-parameter gradient tensors autograd
// produced are copied into the one flat buffer the all-reduce and the Adam launch work on.  The source pointers travel BY VALUE
// in the kernel arguments, 128 tensors per launch (3 launches for the model's 255 parameters) -- nothing to upload, and a
// captured launch replays with the addresses its capture saw.  A tensor is cut into chunks of 16384 elements, one workgroup
// each (torch.cat's batched copy gives every input the same number of workgroups: the 1.8 M-element KernelConv weight next to
// 64-element biases took 90 us for the 22.8 MB); the workgroup finds its tensor by bisection over the chunk prefix sums.
// The last `pad` floats of the buffer carry this rank's overflow flag (guard[0] != 0) and zeros, so the flag travels inside the
// gradient message.
constexpr int GG_BATCH = 128, GG_CHUNK = 16384;
struct GatherBatch {
    const float *src[GG_BATCH];        // NULL: the parameter has no gradient, its range is zero-filled
    long long dst[GG_BATCH];           // element offset in the flat buffer
    int n[GG_BATCH];                   // elements
    unsigned chunk0[GG_BATCH + 1];     // prefix sums of ceil(n / GG_CHUNK); chunk0[count] = workgroups of this launch
    int count;
    int trailer;                       // 1: one extra workgroup writes the trailer
    unsigned chunks;                   // = chunk0[count] (a field of its own: no dynamic index into the kernarg table)
};

__global__ __launch_bounds__(256) void grad_gather_kernel(GatherBatch bt, float *__restrict__ flat, int64_t numel, int pad,
                                                          const int *__restrict__ guard) {
    const unsigned wg = blockIdx.x;
    if (wg == bt.chunks) {                          // (only when bt.trailer: the grid has one workgroup more)
        if ((int)threadIdx.x < pad) flat[numel + threadIdx.x] = (threadIdx.x == 0 && guard != nullptr && guard[0] != 0) ? 1.f : 0.f;
        return;
    }
    // this workgroup's tensor: the last k with chunk0[k] <= wg.  A linear walk with CONSTANT indices on purpose: the arguments
    // live in the kernarg segment, and a bisection's dynamic index made the compiler copy the table into scratch memory first
    // (144 bytes per lane); the walk is 128 scalar compare-and-select steps, uniform over the workgroup.
    const float *src = nullptr;
    long long dst0 = 0;
    int cnt = 0;
    unsigned c0 = 0;
#pragma unroll
    for (int k = 0; k < GG_BATCH; ++k) {
        if (k < bt.count && wg >= bt.chunk0[k]) {
            src = bt.src[k];
            dst0 = bt.dst[k];
            cnt = bt.n[k];
            c0 = bt.chunk0[k];
        }
    }
    const int o = (int)(wg - c0) * GG_CHUNK;
    const int n = min(GG_CHUNK, cnt - o);
    float *__restrict__ d = flat + dst0 + o;
    if (src == nullptr) {
        for (int i = threadIdx.x; i < n; i += 256) d[i] = 0.f;
        return;
    }
    const float *__restrict__ a = src + o;
    // 16-byte LOADS from the source as soon as it is 16-byte aligned (gradient tensors are: `sh` skips to the boundary), eight
    // of them in flight per thread; 16-byte STORES when the destination happens to share that alignment, four dword stores per
    // load otherwise -- parameter offsets are arbitrary element counts, and behind the first 3-element bias every destination
    // is misaligned: the first version fell back to a dword-per-iteration loop there (64 dependent round trips, 33 us per launch).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int sh = min(n, (int)((16u - ((unsigned)reinterpret_cast<uintptr_t>(a) & 15u)) & 15u) >> 2);
    if ((int)threadIdx.x < sh) d[threadIdx.x] = a[threadIdx.x];
    const int n4 = (n - sh) >> 2;
    const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a + sh);
    float *dd = d + sh;
    const bool dst16 = (reinterpret_cast<uintptr_t>(dd) & 15u) == 0;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a4[min(i0 + 256 * k, n4 - 1)];          // (unconditional, clamped: stays in registers)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + 256 * k;
            if (i < n4) {
                if (dst16) {
                    reinterpret_cast<f32x4 *>(dd)[i] = v[k];
                } else {
                    dd[4 * i] = v[k].x; dd[4 * i + 1] = v[k].y; dd[4 * i + 2] = v[k].z; dd[4 * i + 3] = v[k].w;
                }
            }
        }
    }
    for (int i = sh + 4 * n4 + threadIdx.x; i < n; i += 256) d[i] = a[i];
}

}  // namespace

extern "C" int ebfi_grad_gather(const void *const *grads, const int64_t *numels, int count, float *flat, int64_t total, int pad,
                                const int *guard, void *stream) {
    if (!grads || !numels || !flat || count <= 0 || total <= 0 || pad < 1 || pad > 256) return fail(EBFI_ERR_ARG, "grad_gather: bad argument");
    int64_t sum = 0;
    for (int k = 0; k < count; ++k) {
        if (numels[k] < 0 || numels[k] > 0x7fffffff) return fail(EBFI_ERR_ARG, "grad_gather: tensor %d has %lld elements", k, (long long)numels[k]);
        if (grads[k] && (reinterpret_cast<uintptr_t>(grads[k]) & 3u)) return fail(EBFI_ERR_ARG, "grad_gather: tensor %d is not 4-byte aligned", k);
        sum += numels[k];
    }
    if (sum != total) return fail(EBFI_ERR_ARG, "grad_gather: the tensors hold %lld elements, the buffer %lld", (long long)sum, (long long)total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t off = 0;
    for (int k0 = 0; k0 < count; k0 += GG_BATCH) {
        GatherBatch bt;
        bt.count = std::min(GG_BATCH, count - k0);
        bt.trailer = k0 + GG_BATCH >= count ? 1 : 0;
        unsigned chunks = 0;
        double bytes = 0.0;
        for (int j = 0; j < bt.count; ++j) {
            bt.src[j] = static_cast<const float *>(grads[k0 + j]);
            bt.dst[j] = off;
            bt.n[j] = (int)numels[k0 + j];
            bt.chunk0[j] = chunks;
            chunks += (unsigned)ceil_div(numels[k0 + j], (int64_t)GG_CHUNK);
            off += numels[k0 + j];
            bytes += 8.0 * (double)numels[k0 + j];
        }
        bt.chunk0[bt.count] = chunks;
        bt.chunks = chunks;
        for (int j = bt.count; j < GG_BATCH; ++j) bt.src[j] = nullptr, bt.dst[j] = 0, bt.n[j] = 0, bt.chunk0[j + 1] = chunks;
        if (chunks + (unsigned)bt.trailer == 0) continue;
        ProfScope ps("grad_gather", st, 0.0, bytes);
        hipLaunchKernelGGL(grad_gather_kernel, dim3(chunks + (unsigned)bt.trailer), dim3(256), 0, st, bt, flat, total, pad, guard);
    }
    return check_launch("grad_gather");
}

extern "C" int ebfi_adam_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, int64_t n,
                                      double lr, double beta1, double beta2, double eps, int *guard, const float *flag, void *stream);

extern "C" int ebfi_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *step, int64_t n,
                              double lr, double beta1, double beta2, double eps, void *stream) {
    return ebfi_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, const_cast<float *>(step), n, lr, beta1, beta2, eps, nullptr, nullptr,
                                  stream);
}

extern "C" int ebfi_adam_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, int64_t n,
                                      double lr, double beta1, double beta2, double eps, int *guard, const float *flag, void *stream) {
    if (flag && !guard) return fail(EBFI_ERR_ARG, "adam_step: an overflow flag needs the guard words");
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n < 0) return fail(EBFI_ERR_ARG, "adam_step: null argument");
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
        return fail(EBFI_ERR_ARG, "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n4 = n / 4;
    {
        ProfScope ps("adam_flat", st, 0.0, 28.0 * (double)n);
        hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)ceil_div(std::max<int64_t>(n4, 1), 256)), dim3(256), 0, st, param, grad,
                           exp_avg, exp_avg_sq, step, n4, n, lr, beta1, beta2, eps, guard, flag);
    }
    return check_launch("adam_flat");
}

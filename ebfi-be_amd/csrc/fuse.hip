// Fused elementwise stages between the convolutions of the hot path (each replaces a chain of PyTorch
// elementwise / cat / reduction launches over full feature maps).
//
// scale_residual_cat: one round of ResidualControl (reference models/Ours/model_singleframe.py:79-136)
//     by_ex = Conv1(ex) * Conv3(x) + x ;  by_t = Conv2(t) * Conv4(x) + x ;  cat([by_ex, by_t], 1)
// with a0 = Conv3(x), a1 = Conv4(x) [B,C,H,W] and the per-sample channel scales s0 = Conv1(ex), s1 = Conv2(t) [B,C].
#include "common.hpp"
#include "c16.hpp"

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

// ------------------------------------------------------------------------------------------------
// fp16 operand storage (round 4, c16.hpp): the fused stages write the tensors the backward convolutions will stage as scaled
// fp16 images [B][C/16][HW][16] themselves.  A thread owns 4 consecutive pixels of 8 channels (one half of a 16-channel
// block): eight 16-byte plane loads in, four 16-byte pieces (pixel, 8 channels) out.
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void c16_store_quad(_Float16 *__restrict__ dst, int64_t piece0, const f4 (&v)[8], float s, float &amax) {
    // dst + piece0 * 8 halves: the piece of the FIRST pixel; the quad's four pixels lie in one image row (W % 4 == 0), so their
    // pieces are 64 contiguous bytes -- and consecutive lanes own consecutive quads: every store instruction writes one run
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u4 q;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float a = v[e][j], b = v[e + 1][j];
            amax = amax_acc(amax, a, b);
            q[e >> 1] = pack_f16(a * s, b * s);
        }
        *reinterpret_cast<u4 *>(dst + (piece0 + j) * 8) = q;
    }
}
// piece index of (block cb of a tensor with CB blocks per sample, sample b, flat pixel p = 4 q, channel half): image rows are W pieces
__device__ __forceinline__ int64_t c16_piece(int64_t b, int64_t CB, int64_t cb, int64_t HW4, int64_t q, int W4, int half) {
    const int64_t y = q / W4, xq = q - y * W4;
    return ((b * CB + cb) * (HW4 * 4) * 2) + ((y * 2 + half) * W4 + xq) * 4;
}

// fp32 [B, C, HW] -> c16 image (optionally times the LeakyReLU derivative of `mask_y`: the pre-activation gradient of a layer
// whose output is mask_y).  One thread per (b, 8-channel group, pixel quad).
__global__ __launch_bounds__(256) void to_c16_kernel(const float *__restrict__ src, const float *__restrict__ mask_y, float mask_slope,
                                                     _Float16 *__restrict__ dst, float *__restrict__ slot, int C, int64_t HW4,
                                                     int W4, int64_t total) {
    saturate_fp16_conversions();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float amax = 0.f;
    if (i < total) {
        const int64_t q = i % HW4, bg = i / HW4;                  // bg = b * (C / 8) + channel group
        const int64_t G8 = C / 8, b = bg / G8, cg = bg - b * G8;
        const f4 *p = reinterpret_cast<const f4 *>(src) + (b * C + cg * 8) * HW4 + q;
        f4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[e * HW4];
        if (mask_y) {
            const f4 *m = reinterpret_cast<const f4 *>(mask_y) + (b * C + cg * 8) * HW4 + q;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f4 y = m[e * HW4];
                v[e].x *= y.x > 0.f ? 1.f : mask_slope; v[e].y *= y.y > 0.f ? 1.f : mask_slope;
                v[e].z *= y.z > 0.f ? 1.f : mask_slope; v[e].w *= y.w > 0.f ? 1.f : mask_slope;
            }
        }
        c16_store_quad(dst, c16_piece(b, C / 16, cg / 2, HW4, q, W4, (int)(cg & 1)), v, slot[0], amax);
    }
    ScaleSlot{slot}.record(amax);
}

// The image of cat([src0, src1], 1) without the concatenated fp32 tensor (round 6; Modification: `cat([ev, FrameTensor])` has one
// consumer in the training step, the 128 -> 1600 KernelConv, which reads only its image): channel group cg of the image comes from
// src0 (C0 channels per sample) or src1 (C1); both multiples of 8, so a group never straddles.
__global__ __launch_bounds__(256) void to_c16_cat2_kernel(const float *__restrict__ src0, int C0, const float *__restrict__ src1, int C1,
                                                          _Float16 *__restrict__ dst, float *__restrict__ slot, int64_t HW4, int W4,
                                                          int64_t total) {
    saturate_fp16_conversions();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float amax = 0.f;
    if (i < total) {
        const int C = C0 + C1;
        const int64_t q = i % HW4, bg = i / HW4;                  // bg = b * (C / 8) + channel group
        const int64_t G8 = C / 8, b = bg / G8, cg = bg - b * G8;
        const int64_t c = cg * 8;
        const f4 *p = c < C0 ? reinterpret_cast<const f4 *>(src0) + (b * C0 + c) * HW4 + q
                             : reinterpret_cast<const f4 *>(src1) + (b * C1 + (c - C0)) * HW4 + q;
        f4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[e * HW4];
        c16_store_quad(dst, c16_piece(b, C / 16, cg / 2, HW4, q, W4, (int)(cg & 1)), v, slot[0], amax);
    }
    ScaleSlot{slot}.record(amax);
}

// src_fwd_kernel with the output additionally as a c16 image (the input of the weight gradient of the layer that follows)
__global__ __launch_bounds__(256) void src_fwd_c16_kernel(const float *__restrict__ a0, const float *__restrict__ s0,
                                                          const float *__restrict__ a1, const float *__restrict__ s1,
                                                          const float *__restrict__ x, float *__restrict__ out,
                                                          _Float16 *__restrict__ out16, float *__restrict__ slot, int C, int64_t HW4,
                                                          int W4, int64_t total, int64_t a_bs4) {
    saturate_fp16_conversions();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float amax = 0.f;
    if (i < total) {
        const int64_t q = i % HW4, bg = i / HW4;                  // bg = b * (2C / 8) + output channel group
        const int64_t G8 = 2 * C / 8, b = bg / G8, cg = bg - b * G8;
        const int64_t co = cg * 8;                                // first output channel of the group (never straddles C: C % 8 == 0)
        const bool second = co >= C;
        const int64_t c = second ? co - C : co;
        const f4 *pa = reinterpret_cast<const f4 *>(second ? a1 : a0) + b * a_bs4 + c * HW4 + q;
        const f4 *px = reinterpret_cast<const f4 *>(x) + (b * C + c) * HW4 + q;
        const float *ps = (second ? s1 : s0) + b * C + c;
        f4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = pa[e * HW4] * ps[e] + px[e * HW4];
        if (out) {                                                // (uniform: the fp32 copy is optional)
            f4 *o = reinterpret_cast<f4 *>(out) + (b * 2 * C + co) * HW4 + q;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e * HW4] = v[e];
        }
        c16_store_quad(out16, c16_piece(b, 2 * C / 16, cg / 2, HW4, q, W4, (int)(cg & 1)), v, slot[0], amax);
    }
    ScaleSlot{slot}.record(amax);
}

// src_bwd_kernel writing grad_a0 | grad_a1 (the pre-activation gradient of the two second layers) as ONE c16 image of 2C
// channels instead of two fp32 tensors.  A workgroup owns 8 channels of one sample and one of `nslice` pixel slices; the
// per-channel sums of grad*a (the gradients of the scales) leave as per-slice partials [nslice][B][C], summed in slice order by
// the caller (fixed order: deterministic).
// A16 (round 5): a0 is NOT an fp32 plane tensor but the c16 image of [a0 | a1] (2C channels, scaled by a_slot[0]) the forward
// convolution's epilogue wrote (EpiExtra::pre16): four 16-byte pieces per thread and quad instead of eight 16-byte plane loads,
// half the bytes, and the forward keeps no fp32 `a` at all.  The scale gradients see `a` rounded to fp16 (2^-11 per element).
template <bool A16>
__global__ __launch_bounds__(256) void src_bwd_c16_kernel(const float *__restrict__ gout, const float *__restrict__ a0,
                                                          const float *__restrict__ s0, const float *__restrict__ a1,
                                                          const float *__restrict__ s1, _Float16 *__restrict__ ga16,
                                                          float *__restrict__ slot, float *__restrict__ gx, float *__restrict__ gs0p,
                                                          float *__restrict__ gs1p, int B, int C, int64_t HW4, int W4, int64_t a_bs4,
                                                          int nslice, float mask_slope, const float *__restrict__ a_slot = nullptr) {
    saturate_fp16_conversions();
    __shared__ float red[4][16];
    const int G8 = C / 8;
    const int slice = blockIdx.x % nslice, bg = blockIdx.x / nslice;
    const int b = bg / G8, cg = bg - b * G8, c = cg * 8;
    const int64_t q0 = HW4 * slice / nslice, q1 = HW4 * (slice + 1) / nslice;
    float k0[8], k1[8], d0[8], d1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) k0[e] = s0[b * C + c + e], k1[e] = s1[b * C + c + e], d0[e] = 0.f, d1[e] = 0.f;
    const float sc = slot[0];
    float amax = 0.f;
    auto dm = [&](const f4 &v) {
        f4 m;
        m.x = v.x > 0.f ? 1.f : mask_slope; m.y = v.y > 0.f ? 1.f : mask_slope;
        m.z = v.z > 0.f ? 1.f : mask_slope; m.w = v.w > 0.f ? 1.f : mask_slope;
        return m;
    };
    const f4 *g0 = reinterpret_cast<const f4 *>(gout) + ((int64_t)b * 2 * C + c) * HW4;
    const f4 *g1 = reinterpret_cast<const f4 *>(gout) + ((int64_t)b * 2 * C + C + c) * HW4;
    const f4 *p0 = reinterpret_cast<const f4 *>(a0) + (int64_t)b * a_bs4 + (int64_t)c * HW4;
    const f4 *p1 = reinterpret_cast<const f4 *>(a1) + (int64_t)b * a_bs4 + (int64_t)c * HW4;
    [[maybe_unused]] const _Float16 *img = reinterpret_cast<const _Float16 *>(a0);
    [[maybe_unused]] const float ainv = A16 ? 1.f / a_slot[0] : 1.f;
    // the four pieces (pixel j, 8 channels) of a quad -> per channel e the quad's four pixels, de-scaled
    auto quad_from_image = [&](int64_t piece0, f4 (&v)[8]) {
        u4 pc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pc[j] = *reinterpret_cast<const u4 *>(img + (piece0 + j) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned wd = pc[j][e >> 1];
                const _Float16 hv = __builtin_bit_cast(_Float16, (unsigned short)((e & 1) ? (wd >> 16) : (wd & 0xffffu)));
                v[e][j] = (float)hv * ainv;
            }
        }
    };
    f4 *ox = reinterpret_cast<f4 *>(gx) + ((int64_t)b * C + c) * HW4;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += 256) {
        f4 u0[8], u1[8], w[8], va[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u0[e] = g0[e * HW4 + q], u1[e] = g1[e * HW4 + q];
#pragma unroll
        for (int e = 0; e < 8; ++e) ox[e * HW4 + q] = u0[e] + u1[e];
        const int64_t pc0 = c16_piece(b, 2 * C / 16, cg / 2, HW4, q, W4, cg & 1);
        const int64_t pc1 = c16_piece(b, 2 * C / 16, (C + c) / 16, HW4, q, W4, ((C + c) >> 3) & 1);
        if constexpr (A16) quad_from_image(pc0, va);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f4 v = A16 ? va[e] : p0[e * HW4 + q];
            d0[e] += (u0[e].x * v.x + u0[e].y * v.y) + (u0[e].z * v.z + u0[e].w * v.w);
            w[e] = u0[e] * k0[e] * dm(v);
        }
        c16_store_quad(ga16, pc0, w, sc, amax);
        if constexpr (A16) quad_from_image(pc1, va);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f4 v = A16 ? va[e] : p1[e * HW4 + q];
            d1[e] += (u1[e].x * v.x + u1[e].y * v.y) + (u1[e].z * v.z + u1[e].w * v.w);
            w[e] = u1[e] * k1[e] * dm(v);
        }
        c16_store_quad(ga16, pc1, w, sc, amax);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            d0[e] += __shfl_xor(d0[e], d, 64);
            d1[e] += __shfl_xor(d1[e], d, 64);
        }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x >> 6][e] = d0[e], red[threadIdx.x >> 6][8 + e] = d1[e];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        float *dst = threadIdx.x < 8 ? gs0p : gs1p;
        dst[((int64_t)slice * B + b) * C + c + (threadIdx.x & 7)] = v;
    }
    ScaleSlot{slot}.record(amax);
}

int check_planes(const char *who, int B, int C, int64_t HW) {
    if (B < 0 || C <= 0 || HW <= 0) return fail(EBFI_ERR_ARG, "%s: bad dimensions", who);
    if (HW % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "%s: H*W must be a multiple of 4 (got %lld)", who, (long long)HW);
    if ((int64_t)B * C > 2147483647LL) return fail(EBFI_ERR_ARG, "%s: too many planes", who);
    return EBFI_OK;
}


// ---- scalar-conditioned channel scales (ebfi_scalar_conv_*): a bank of S 1x1 convolutions on [B,K,1,1] inputs --------------
constexpr int SC_MAX_LAYERS = 32, SC_MAX_K = 8;
struct ScalarBank {                        // by value in the kernel arguments: a captured launch carries the pointers
    const float *w[SC_MAX_LAYERS];
    const float *b[SC_MAX_LAYERS];
};

__global__ __launch_bounds__(256) void scalar_conv_fwd_kernel(ScalarBank bank, const float *__restrict__ v, float *__restrict__ out,
                                                              int B, int K, int C, float slope) {
    const int s = blockIdx.x;
    // (constant-index walk over the by-value table: a runtime index would put the struct in scratch, see optim.hip)
    const float *w = nullptr, *bias = nullptr;
#pragma unroll
    for (int j = 0; j < SC_MAX_LAYERS; ++j)
        if (j == s) { w = bank.w[j]; bias = bank.b[j]; }
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C, c = i - b * C;
        float acc = bias ? bias[c] : 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(v[b * K + k], w[c * K + k], acc);
        out[((int64_t)s * B + b) * C + c] = acc > 0.f ? acc : acc * slope;
    }
}

// one workgroup for the whole bank (S*B*C values: 6144 for the default model): every sum in a fixed order
__global__ __launch_bounds__(256) void scalar_conv_bwd_kernel(ScalarBank bank, const float *__restrict__ v, const float *__restrict__ out,
                                                              const float *__restrict__ gout, float *__restrict__ gw, float *__restrict__ gb,
                                                              float *__restrict__ gv, int S, int B, int K, int C, float slope) {
    // grad_weight / grad_bias: thread = (layer, channel), walks the samples
    for (int i = threadIdx.x; i < S * C; i += 256) {
        const int s = i / C, c = i - s * C;
        float aw[SC_MAX_K], ab = 0.f;
#pragma unroll
        for (int k = 0; k < SC_MAX_K; ++k) aw[k] = 0.f;
        for (int b = 0; b < B; ++b) {
            const int64_t o = ((int64_t)s * B + b) * C + c;
            const float gp = gout[o] * (out[o] > 0.f ? 1.f : slope);
            ab += gp;
#pragma unroll
            for (int k = 0; k < SC_MAX_K; ++k)
                if (k < K) aw[k] = fmaf(gp, v[b * K + k], aw[k]);
        }
        if (gb) gb[i] = ab;
        if (gw) {
#pragma unroll
            for (int k = 0; k < SC_MAX_K; ++k)
                if (k < K) gw[(int64_t)i * K + k] = aw[k];
        }
    }
    // grad_v[b][k] = sum over (layer, channel): one wave per (b, k) pair, lanes stride the S*C terms, butterfly in fixed order
    if (gv) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int p = wave; p < B * K; p += 4) {
            const int b = p / K, k = p - b * K;
            float acc = 0.f;
            for (int s = 0; s < S; ++s) {
                const float *w = nullptr;
#pragma unroll
                for (int j = 0; j < SC_MAX_LAYERS; ++j)
                    if (j == s) w = bank.w[j];
                for (int c = lane; c < C; c += 64) {
                    const int64_t o = ((int64_t)s * B + b) * C + c;
                    acc = fmaf(gout[o] * (out[o] > 0.f ? 1.f : slope), w[c * K + k], acc);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
            if (lane == 0) gv[p] = acc;
        }
    }
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

// fp16 c16 images (include/ebfi_hip.h, csrc/c16.hpp) ---------------------------------------------------------------
extern "C" int ebfi_to_c16(const float *src, const float *mask_y, float mask_slope, void *dst16, void *slot, int B, int C, int H, int W,
                           void *stream) {
    const int64_t HW = (int64_t)H * W;
    if (!src || !dst16 || !slot) return fail(EBFI_ERR_ARG, "to_c16: null argument");
    if (W % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "to_c16: W %% 4 != 0 (W = %d)", W);
    if (C % 16 != 0) return fail(EBFI_ERR_UNSUPPORTED, "to_c16: %d channels (multiples of 16)", C);
    if (int rc = check_planes("to_c16", B, C, HW)) return rc;
    if (!aligned16(src) || !aligned16(dst16) || (mask_y && !aligned16(mask_y))) return fail(EBFI_ERR_ARG, "to_c16: 16-byte aligned tensors");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * (C / 8) * (HW / 4);
    {
        ProfScope ps("to_c16", st, 0.0, (mask_y ? 10.0 : 6.0) * B * C * (double)HW);
        hipLaunchKernelGGL(to_c16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, src, mask_y, mask_slope,
                           static_cast<_Float16 *>(dst16), static_cast<float *>(slot), C, HW / 4, W / 4, total);
    }
    return check_launch("to_c16");
}

extern "C" int ebfi_to_c16_cat2(const float *src0, int C0, const float *src1, int C1, void *dst16, void *slot, int B, int H, int W,
                                void *stream) {
    const int64_t HW = (int64_t)H * W;
    if (!src0 || !src1 || !dst16 || !slot) return fail(EBFI_ERR_ARG, "to_c16_cat2: null argument");
    if (W % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "to_c16_cat2: W %% 4 != 0 (W = %d)", W);
    if (C0 < 8 || C1 < 8 || C0 % 8 != 0 || C1 % 8 != 0 || (C0 + C1) % 16 != 0)
        return fail(EBFI_ERR_UNSUPPORTED, "to_c16_cat2: %d + %d channels (multiples of 8, a multiple of 16 together)", C0, C1);
    if (int rc = check_planes("to_c16_cat2", B, C0 + C1, HW)) return rc;
    if (!aligned16(src0) || !aligned16(src1) || !aligned16(dst16)) return fail(EBFI_ERR_ARG, "to_c16_cat2: 16-byte aligned tensors");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * ((C0 + C1) / 8) * (HW / 4);
    {
        ProfScope ps("to_c16/cat2", st, 0.0, 6.0 * B * (C0 + C1) * (double)HW);
        hipLaunchKernelGGL(to_c16_cat2_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, src0, C0, src1, C1,
                           static_cast<_Float16 *>(dst16), static_cast<float *>(slot), HW / 4, W / 4, total);
    }
    return check_launch("to_c16/cat2");
}

extern "C" int ebfi_scale_residual_cat_forward_c16(const float *a0, const float *s0, const float *a1, const float *s1, const float *x,
                                                   float *out, void *out16, void *slot, int B, int C, int H, int W,
                                                   int64_t a_batch_stride, void *stream) {
    const int64_t HW = (int64_t)H * W;
    // `out` may be NULL: the image alone (the convolution that follows reads fp16 operand images in its forward pass too)
    if (!a0 || !s0 || !a1 || !s1 || !x || !out16 || !slot) return fail(EBFI_ERR_ARG, "scale_residual_cat_forward_c16: null argument");
    if (W % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "scale_residual_cat_forward_c16: W %% 4 != 0 (W = %d)", W);
    if (a_batch_stride % 4 != 0 || a_batch_stride < (int64_t)C * HW) return fail(EBFI_ERR_ARG, "scale_residual_cat_forward_c16: batch stride");
    if (C % 8 != 0) return fail(EBFI_ERR_UNSUPPORTED, "scale_residual_cat_forward_c16: %d channels (multiples of 8)", C);
    if (int rc = check_planes("scale_residual_cat_forward_c16", B, C, HW)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * (2 * C / 8) * (HW / 4);
    {
        ProfScope ps("scale_residual_cat_fwd", st, 0.0, (out ? 24.0 : 16.0) * B * C * (double)HW);
        hipLaunchKernelGGL(src_fwd_c16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, a0, s0, a1, s1, x, out,
                           static_cast<_Float16 *>(out16), static_cast<float *>(slot), C, HW / 4, W / 4, total, a_batch_stride / 4);
    }
    return check_launch("scale_residual_cat_fwd");
}

extern "C" int ebfi_scale_residual_cat_backward_slices(void) { return 8; }

// grad_a16: c16 image of [grad_a0 | grad_a1] (2C channels) times the LeakyReLU(mask_slope) derivative of a0 / a1, scaled by
// slot[0]; grad_x [B,C,HW] fp32; grad_s0_part / grad_s1_part [ebfi_scale_residual_cat_backward_slices()][B][C] partial sums
extern "C" int ebfi_scale_residual_cat_backward_c16(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                                    const float *s1, void *grad_a16, void *slot, float *grad_x, float *grad_s0_part,
                                                    float *grad_s1_part, int B, int C, int H, int W, int64_t a_batch_stride,
                                                    float mask_slope, void *stream) {
    const int64_t HW = (int64_t)H * W;
    if (W % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "scale_residual_cat_backward_c16: W %% 4 != 0 (W = %d)", W);
    if (!grad_out || !a0 || !s0 || !a1 || !s1 || !grad_a16 || !slot || !grad_x || !grad_s0_part || !grad_s1_part)
        return fail(EBFI_ERR_ARG, "scale_residual_cat_backward_c16: null argument");
    if (a_batch_stride % 4 != 0 || a_batch_stride < (int64_t)C * HW) return fail(EBFI_ERR_ARG, "scale_residual_cat_backward_c16: batch stride");
    if (C % 16 != 0) return fail(EBFI_ERR_UNSUPPORTED, "scale_residual_cat_backward_c16: %d channels (multiples of 16)", C);
    if (int rc = check_planes("scale_residual_cat_backward_c16", B, C, HW)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslice = ebfi_scale_residual_cat_backward_slices();
    {
        ProfScope ps("scale_residual_cat_bwd", st, 0.0, 24.0 * B * C * (double)HW);
        hipLaunchKernelGGL(src_bwd_c16_kernel<false>, dim3((unsigned)(B * (C / 8) * nslice)), dim3(256), 0, st, grad_out, a0, s0, a1, s1,
                           static_cast<_Float16 *>(grad_a16), static_cast<float *>(slot), grad_x, grad_s0_part, grad_s1_part, B, C,
                           HW / 4, W / 4, a_batch_stride / 4, nslice, mask_slope, nullptr);
    }
    return check_launch("scale_residual_cat_bwd");
}

// The same stage with `a` given as the c16 IMAGE of [a0 | a1] (2C channels, scaled by a_slot[0]) that the forward convolution's
// epilogue wrote (ebfi_conv2d_packed_x3_rc pre16): the forward keeps no fp32 copy of `a`.
extern "C" int ebfi_scale_residual_cat_backward_c16a(const float *grad_out, const void *a16, const void *a_slot, const float *s0,
                                                     const float *s1, void *grad_a16, void *slot, float *grad_x, float *grad_s0_part,
                                                     float *grad_s1_part, int B, int C, int H, int W, float mask_slope, void *stream) {
    const int64_t HW = (int64_t)H * W;
    if (W % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "scale_residual_cat_backward_c16a: W %% 4 != 0 (W = %d)", W);
    if (!grad_out || !a16 || !a_slot || !s0 || !s1 || !grad_a16 || !slot || !grad_x || !grad_s0_part || !grad_s1_part)
        return fail(EBFI_ERR_ARG, "scale_residual_cat_backward_c16a: null argument");
    if (C % 16 != 0) return fail(EBFI_ERR_UNSUPPORTED, "scale_residual_cat_backward_c16a: %d channels (multiples of 16)", C);
    if (!aligned16(a16) || !aligned16(grad_a16)) return fail(EBFI_ERR_ARG, "scale_residual_cat_backward_c16a: 16-byte aligned images");
    if (int rc = check_planes("scale_residual_cat_backward_c16a", B, C, HW)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslice = ebfi_scale_residual_cat_backward_slices();
    {
        ProfScope ps("scale_residual_cat_bwd", st, 0.0, 20.0 * B * C * (double)HW);
        hipLaunchKernelGGL(src_bwd_c16_kernel<true>, dim3((unsigned)(B * (C / 8) * nslice)), dim3(256), 0, st, grad_out,
                           static_cast<const float *>(a16), s0, static_cast<const float *>(a16), s1, static_cast<_Float16 *>(grad_a16),
                           static_cast<float *>(slot), grad_x, grad_s0_part, grad_s1_part, B, C, HW / 4, W / 4, (int64_t)0, nslice,
                           mask_slope, static_cast<const float *>(a_slot));
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

static int scalar_bank(ScalarBank &bank, const void *const *weights, const void *const *biases, int S, int B, int K, int C, const char *what) {
    if (!weights || S < 1 || S > SC_MAX_LAYERS || K < 1 || K > SC_MAX_K || B < 0 || C < 1 || (int64_t)S * B * C > (1 << 20))
        return fail(EBFI_ERR_ARG, "%s: S = %d layers (<= %d), K = %d (<= %d), B = %d, C = %d", what, S, SC_MAX_LAYERS, K, SC_MAX_K, B, C);
    for (int j = 0; j < SC_MAX_LAYERS; ++j) {
        bank.w[j] = j < S ? static_cast<const float *>(weights[j]) : nullptr;
        bank.b[j] = (j < S && biases) ? static_cast<const float *>(biases[j]) : nullptr;
        if (j < S && !bank.w[j]) return fail(EBFI_ERR_ARG, "%s: weight %d is NULL", what, j);
    }
    return EBFI_OK;
}

extern "C" int ebfi_scalar_conv_forward(const float *v, const void *const *weights, const void *const *biases, float *out, int S, int B,
                                        int K, int C, float slope, void *stream) {
    ScalarBank bank;
    if (int rc = scalar_bank(bank, weights, biases, S, B, K, C, "scalar_conv_forward")) return rc;
    if (!v || !out) return fail(EBFI_ERR_ARG, "scalar_conv_forward: null argument");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("scalar_conv_fwd", st, 2.0 * S * B * C * K, 4.0 * (S * (double)C * (K + 1) + B * K + (double)S * B * C));
        hipLaunchKernelGGL(scalar_conv_fwd_kernel, dim3((unsigned)S), dim3(256), 0, st, bank, v, out, B, K, C, slope);
    }
    return check_launch("scalar_conv_fwd");
}

extern "C" int ebfi_scalar_conv_backward(const float *v, const void *const *weights, const float *out, const float *grad_out,
                                         float *grad_weight, float *grad_bias, float *grad_v, int S, int B, int K, int C, float slope,
                                         void *stream) {
    ScalarBank bank;
    if (int rc = scalar_bank(bank, weights, nullptr, S, B, K, C, "scalar_conv_backward")) return rc;
    if (!v || !out || !grad_out) return fail(EBFI_ERR_ARG, "scalar_conv_backward: null argument");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("scalar_conv_bwd", st, 4.0 * S * B * C * K, 4.0 * (2.0 * S * B * C + 2.0 * S * C * (K + 1) + 2.0 * B * K));
        hipLaunchKernelGGL(scalar_conv_bwd_kernel, dim3(1), dim3(256), 0, st, bank, v, out, grad_out, grad_weight, grad_bias, grad_v,
                           S, B, K, C, slope);
    }
    return check_launch("scalar_conv_bwd");
}

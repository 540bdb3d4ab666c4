// Filter-Adaptive Convolution (FAC) for gfx950.
//
// Semantics: reference models/FAC/kernelconv2d/KernelConv2D_kernel.cu:25-53 (forward),
// :91-125 (grad_input), :128-150 (grad_kernel); entry points replace KernelConv2D_cuda.cpp:10-61.
//
// The op is pure HBM streaming (AI ~0.46 FLOP/B): per (b, c) plane it reads K*K filter planes once,
// an input plane K*K times (on-chip reuse) and writes one plane.  Design:
//   forward   one 256-thread workgroup per TH x TW output tile of a (b,c) plane; the (TH+K-1) x
//             (TW+K-1) input halo tile is staged once in LDS; every thread owns 4 consecutive x and
//             issues K*K independent 16-byte coalesced loads of the filter planes (all in flight
//             together), fp32 FMA in the reference's (ky,kx) order, one 16-byte store.
//   backward  ONE fused kernel for both gradients so the filter planes and grad_output are read
//             once: a thread owns 4 consecutive x of one row Y of the PADDED grid, walks the K rows
//             y = Y - ky that feed it, writes grad_kernel = in * gout (16-byte streaming stores) and
//             keeps 4+K-1 private partial sums of grad_input; the K-1 partials that belong to the
//             next thread's columns move one lane up with a wave shuffle (carried across x-chunks),
//             so grad_input needs no atomics, no LDS and no zero-fill.
//   generic   per-element kernels (any stride, any K, 64-bit indexing) for everything the fast
//             paths do not cover; still HIP -- there is no CPU fallback.
#include "common.hpp"
#include "c16.hpp"

using namespace ebfi;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// streamed-once data: non-temporal 16-byte accesses
__device__ __forceinline__ f32x4 ld_stream4(const float *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
}
__device__ __forceinline__ void st_stream4(float *p, f32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
}

// fp16 filter / grad_kernel storage (round 4): the same PLANAR [B, C*K*K, Ho, Wo] tensors as halves, multiplied by the
// power-of-two scale of a slot (c16.hpp ScaleSlot): 8-byte accesses of 4 pixels instead of 16-byte ones, half the bytes of
// the op's two big streams.  `inv` undoes the scale on the way in.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 ld_stream4h(const _Float16 *p, float inv) {
    // (one 64-bit scalar: hipcc 7.2 turns a non-temporal load of a 2 x 32-bit ext vector into a single dword load and hands the
    // same word out twice -- the first version of this function returned pixels 0, 1, 0, 1)
    const unsigned long long q64 = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(p));
    const f16x2 a = __builtin_bit_cast(f16x2, (unsigned)q64), b = __builtin_bit_cast(f16x2, (unsigned)(q64 >> 32));
    return f32x4{(float)a[0] * inv, (float)a[1] * inv, (float)b[0] * inv, (float)b[1] * inv};
}
__device__ __forceinline__ void st_stream4h(_Float16 *p, f32x4 v, float s) {
    const unsigned long long q64 = (unsigned long long)pack_f16(v[0] * s, v[1] * s) | ((unsigned long long)pack_f16(v[2] * s, v[3] * s) << 32);
    __builtin_nontemporal_store(q64, reinterpret_cast<unsigned long long *>(p));
}
template <bool H16>
__device__ __forceinline__ f32x4 ld_filter4(const void *base, int64_t elem, float inv) {
    if constexpr (H16) return ld_stream4h(static_cast<const _Float16 *>(base) + elem, inv);
    else return ld_stream4(static_cast<const float *>(base) + elem);
}

struct Str4 {
    int64_t s0, s1, s2, s3;
};
static inline Str4 str4(const int64_t *p) { return Str4{p[0], p[1], p[2], p[3]}; }

// ------------------------------------------------------------------------------------------------
// forward, vectorised tile kernel.  Requirements (checked by the launcher): innermost stride 1 for
// all three tensors, Wo % 4 == 0, filter/output rows 16-byte aligned.
// CLAMP (round 5): `in` is the UNPADDED [B, C, Ho, Wo] tensor and the replicate padding of KernelConv2D.py:82-86 happens here,
// as clamped reads while the halo tile is staged -- no padded copy of the input exists (the F.pad launch and its 35 MB).
template <int K, int TH, int TW, bool H16 = false, bool CLAMP = false>
__global__ __launch_bounds__(256) void fac_fwd_tile_f32(const float *__restrict__ in, Str4 is,
                                                        const void *__restrict__ kern, Str4 ks,
                                                        float *__restrict__ out, Str4 os, int C, int Ho,
                                                        int Wo, const float *__restrict__ f_slot = nullptr) {
    static_assert(TH * (TW / 4) == 256, "tile must map onto 256 threads");
    constexpr int IH = TH + K - 1;
    constexpr int IW = TW + K - 1;
    constexpr int NV = (4 + K - 1 + 3) / 4;   // float4 reads per thread per tile row
    constexpr int IWP = TW - 4 + 4 * NV;      // padded LDS row: every thread's NV reads stay inside
    static_assert(IWP >= IW, "LDS row too short");
    __shared__ __attribute__((aligned(16))) float tile[IH * IWP];

    const int tid = threadIdx.x;
    const int b = blockIdx.z / C, c = blockIdx.z % C;
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const int Hi = Ho + K - 1, Wi = Wo + K - 1;

    const float *inp = in + (int64_t)b * is.s0 + (int64_t)c * is.s1;
    for (int i = tid; i < IH * IWP; i += 256) {
        const int r = i / IWP, col = i - r * IWP;
        const int yy = y0 + r, xx = x0 + col;
        float v = 0.f;
        if (col < IW && yy < Hi && xx < Wi) {
            if constexpr (CLAMP) v = inp[(int64_t)min(max(yy - K / 2, 0), Ho - 1) * is.s2 + min(max(xx - K / 2, 0), Wo - 1)];
            else v = inp[(int64_t)yy * is.s2 + xx];
        }
        tile[i] = v;
    }
    __syncthreads();

    const int ty = tid / (TW / 4), tx = tid % (TW / 4);
    const int y = y0 + ty, x = x0 + 4 * tx;
    if (y >= Ho || x >= Wo) return;

    const int64_t kp = (int64_t)b * ks.s0 + (int64_t)c * K * K * ks.s1 + (int64_t)y * ks.s2 + x;
    const float finv = H16 ? 1.f / f_slot[0] : 1.f;
    f32x4 kv[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) kv[t] = ld_filter4<H16>(kern, kp + (int64_t)t * ks.s1, finv);

    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        float r[4 * NV];
        const float4 *row = reinterpret_cast<const float4 *>(&tile[(ty + ky) * IWP + 4 * tx]);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 q = row[v];
            r[4 * v + 0] = q.x; r[4 * v + 1] = q.y; r[4 * v + 2] = q.z; r[4 * v + 3] = q.w;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 w = kv[ky * K + kx];
            a0 = fmaf(r[kx + 0], w.x, a0);
            a1 = fmaf(r[kx + 1], w.y, a1);
            a2 = fmaf(r[kx + 2], w.z, a2);
            a3 = fmaf(r[kx + 3], w.w, a3);
        }
    }
    float *op = out + (int64_t)b * os.s0 + (int64_t)c * os.s1 + (int64_t)y * os.s2 + x;
    f32x4 res = {a0, a1, a2, a3};
    *reinterpret_cast<f32x4 *>(op) = res;
}

// forward, generic: one thread per output element, any strides / K.
__global__ void fac_fwd_generic_f32(const float *__restrict__ in, Str4 is, const float *__restrict__ kern,
                                    Str4 ks, float *__restrict__ out, Str4 os, int64_t total, int C, int Ho,
                                    int Wo, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % Wo);
    const int y = (int)((idx / Wo) % Ho);
    const int c = (int)((idx / ((int64_t)Wo * Ho)) % C);
    const int64_t b = idx / ((int64_t)Wo * Ho * C);
    const float *ip = in + b * is.s0 + (int64_t)c * is.s1;
    const float *kp = kern + b * ks.s0 + (int64_t)c * K * K * ks.s1 + (int64_t)y * ks.s2 + (int64_t)x * ks.s3;
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx)
            acc = fmaf(ip[(int64_t)(y + ky) * is.s2 + (int64_t)(x + kx) * is.s3],
                       kp[(int64_t)(ky * K + kx) * ks.s1], acc);
    out[b * os.s0 + (int64_t)c * os.s1 + (int64_t)y * os.s2 + (int64_t)x * os.s3] = acc;
}

// ------------------------------------------------------------------------------------------------
// backward, fused row kernel (see file header).  K in {1,3,5}; innermost strides 1; Wo % 4 == 0;
// filter / grad_kernel / grad_output rows 16-byte aligned.  TPR = lanes per row (power of two <= 64).
// CLAMP (round 5): `in` / `gin` are the UNPADDED [B, C, Ho, Wo] tensors.  The padded grid still exists -- as coordinates: the
// input is read with clamped indices, and grad_input is the ADJOINT of the replicate padding applied on the fly: padded
// position (Y, X) adds into (clamp(Y - K/2), clamp(X - K/2)).  A thread owns 4 padded columns of an OUTPUT row R: interior rows
// have one padded row (R + K/2), the first / last row own the K/2 + 1 padded rows that clamp onto them and walk them in order
// (plain store for the first, read-modify-write of their own elements for the rest); the 3 padded columns of either border
// lie inside one thread's 4 and are summed in registers.  Fixed summation order, no atomics -- unlike torch's
// replication_pad2d_backward, whose atomic adds were one of the step's two sources of run-to-run noise.
template <int K, int TPR, bool H16 = false, bool CLAMP = false>
__global__ __launch_bounds__(256) void fac_bwd_rows_f32(const float *__restrict__ in, Str4 is,
                                                        const void *__restrict__ kern, Str4 ks,
                                                        const float *__restrict__ gout, Str4 gs,
                                                        float *__restrict__ gin, Str4 gis,
                                                        void *__restrict__ gkern, Str4 gks, int C, int Ho,
                                                        int Wo, float kslope, const float *__restrict__ f_slot = nullptr,
                                                        float *__restrict__ g_slot = nullptr) {
    // H16: filters and grad_kernel are fp16 planes scaled by f_slot[0] / g_slot[0]; |max| of grad_kernel recorded into g_slot
    if constexpr (H16) saturate_fp16_conversions();
    const float finv = H16 ? 1.f / f_slot[0] : 1.f, gsc = (H16 && g_slot) ? g_slot[0] : 1.f;
    [[maybe_unused]] float gk_amax = 0.f;
    // kslope: grad_kernel leaves multiplied by (kernel > 0 ? 1 : kslope) -- the derivative of the LeakyReLU that produced the
    // filters, so that the layer below receives the gradient of its PRE-activation (1.0 = plain grad_kernel, bit for bit)
    static_assert(K == 1 || K == 3 || K == 5, "carry scheme needs K-1 <= 4");
    static_assert(!CLAMP || K == 5, "the in-kernel replicate padding is written for K = 5");
    constexpr int RPW = 64 / TPR;          // rows per wave
    constexpr int ROWS = 4 * RPW;          // rows per 256-thread workgroup
    constexpr int NS = 4 + K - 1;          // private partial sums / input values per thread
    constexpr int NU = K - 1;              // partials owned by the next lane
    constexpr int R2 = K / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane / TPR, tx = lane % TPR;
    const int row = blockIdx.x * ROWS + wave * RPW + r;   // row of the padded grid (CLAMP: output row)
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    const int Hi = Ho + K - 1, Wi = Wo + K - 1;
    const int nx4 = Wo >> 2;
    const bool rowok = CLAMP ? row < Ho : row < Hi;
    const int nchunks = nx4 / TPR + 1;     // the lane right after the last loading lane writes the tail
    // padded rows this thread walks: [Ylo, Yhi]
    const int Ylo = CLAMP ? (row == 0 ? 0 : row + R2) : row;
    const int Yhi = CLAMP ? (row == Ho - 1 ? Ho - 1 + 2 * R2 : row + R2) : row;

    const float *inpl = in + (int64_t)b * is.s0 + (int64_t)c * is.s1;
    const int64_t kbase = (int64_t)b * ks.s0 + (int64_t)c * K * K * ks.s1;
    const float *gbase = gout + (int64_t)b * gs.s0 + (int64_t)c * gs.s1;
    const bool has_gk = gkern != nullptr;
    const int64_t gkbase = (int64_t)b * gks.s0 + (int64_t)c * K * K * gks.s1;
    float *ginpl = gin ? gin + (int64_t)b * gis.s0 + (int64_t)c * gis.s1 : nullptr;

    for (int yi = 0; yi < (CLAMP ? K : 1); ++yi) {
        const int Y = Ylo + yi;
        const bool yok = rowok && Y <= Yhi;
        if constexpr (CLAMP) {
            if (__builtin_amdgcn_ballot_w64(yok) == 0) break;      // (wave-uniform: the shuffles below need every lane)
        }
        const float *inrow = inpl + (int64_t)(CLAMP ? min(max(Y - R2, 0), Ho - 1) : Y) * is.s2;
        float *ginrow = ginpl ? ginpl + (int64_t)(CLAMP ? row : Y) * gis.s2 : nullptr;
        float carry[NU > 0 ? NU : 1];
#pragma unroll
        for (int m = 0; m < NU; ++m) carry[m] = 0.f;

        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int x4 = chunk * TPR + tx;
            const int x = x4 << 2;
            const bool active = yok && x4 < nx4;
            float s[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) s[j] = 0.f;
            if (active) {
                float inr[NS];
                if (has_gk) {
                    if constexpr (CLAMP) {
                        if (x >= 4 && x + NS <= Wo) {
                            // interior columns: the NS = 8 values start at x - 2, an 8-byte aligned address (rows are 16-byte aligned,
                            // Wo % 4 == 0): four 8-byte loads instead of eight clamped dword loads with an address each
                            const float2 *p2 = reinterpret_cast<const float2 *>(inrow + x - R2);
#pragma unroll
                            for (int j = 0; j < NS / 2; ++j) {
                                const float2 t = p2[j];
                                inr[2 * j] = t.x;
                                inr[2 * j + 1] = t.y;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < NS; ++j) inr[j] = inrow[min(max(x + j - R2, 0), Wo - 1)];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NS; ++j) inr[j] = inrow[x + j];   // x + j <= Wo - 4 + 3 + K - 1 < Wi
                    }
                }
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int y = Y - ky;
                    if (y < 0 || y >= Ho) continue;
                    const float4 g = *reinterpret_cast<const float4 *>(gbase + (int64_t)y * gs.s2 + x);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int t = ky * K + kx;
                        const f32x4 w = ld_filter4<H16>(kern, kbase + (int64_t)t * ks.s1 + (int64_t)y * ks.s2 + x, finv);
                        if (has_gk) {
                            const f32x4 p = {inr[kx + 0] * g.x * (w.x > 0.f ? 1.f : kslope), inr[kx + 1] * g.y * (w.y > 0.f ? 1.f : kslope),
                                             inr[kx + 2] * g.z * (w.z > 0.f ? 1.f : kslope), inr[kx + 3] * g.w * (w.w > 0.f ? 1.f : kslope)};
                            const int64_t go_ = gkbase + (int64_t)t * gks.s1 + (int64_t)y * gks.s2 + x;
                            if constexpr (H16) {
                                gk_amax = amax_acc(amax_acc(gk_amax, p[0], p[1]), p[2], p[3]);
                                st_stream4h(static_cast<_Float16 *>(gkern) + go_, p, gsc);
                            } else {
                                st_stream4(static_cast<float *>(gkern) + go_, p);
                            }
                        }
                        s[kx + 0] = fmaf(w.x, g.x, s[kx + 0]);
                        s[kx + 1] = fmaf(w.y, g.y, s[kx + 1]);
                        s[kx + 2] = fmaf(w.z, g.z, s[kx + 2]);
                        s[kx + 3] = fmaf(w.w, g.w, s[kx + 3]);
                    }
                }
            }
            // hand the K-1 upper partials to the lane that owns those columns (all lanes take part)
            float o[4] = {s[0], s[1], s[2], s[3]};
            if constexpr (NU > 0) {
#pragma unroll
                for (int m = 0; m < NU; ++m) {
                    float left = __shfl_up(s[4 + m], 1, TPR);
                    if (tx == 0) left = carry[m];
                    const float last = __shfl(s[4 + m], TPR - 1, TPR);
                    o[m] += left;
                    carry[m] = last;
                }
            }
            if constexpr (CLAMP) {
                if (ginrow != nullptr && yok && x < Wi) {
                    // padded columns x .. x + 3 -> output columns clamp(X - 2): the left border's three (X = 0, 1, 2 -> 0) sit in the
                    // thread with x == 0, the right border's three (X = Wo + 1 .. Wo + 3 -> Wo - 1) in the tail thread x == Wo
                    // (yi > 0 only in the first / last output row: this thread wrote the elements in an earlier pass and adds to them)
                    if (x == 0) {
                        const float v0 = (o[0] + o[1]) + o[2];
                        if (yi == 0) { ginrow[0] = v0; ginrow[1] = o[3]; }
                        else { ginrow[0] += v0; ginrow[1] += o[3]; }
                    } else if (x == Wo) {
                        const float v1 = (o[1] + o[2]) + o[3];
                        if (yi == 0) { ginrow[Wo - 2] = o[0]; ginrow[Wo - 1] = v1; }
                        else { ginrow[Wo - 2] += o[0]; ginrow[Wo - 1] += v1; }
                    } else {
                        float2 *g2 = reinterpret_cast<float2 *>(ginrow + x - R2);      // 8-byte aligned, like the input above
                        if (yi == 0) {
                            g2[0] = float2{o[0], o[1]};
                            g2[1] = float2{o[2], o[3]};
                        } else {
                            const float2 a = g2[0], b = g2[1];
                            g2[0] = float2{a.x + o[0], a.y + o[1]};
                            g2[1] = float2{b.x + o[2], b.y + o[3]};
                        }
                    }
                }
            } else {
                if (ginrow != nullptr && rowok) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (x + m < Wi) ginrow[x + m] = o[m];
                }
            }
        }
    }
    if constexpr (H16) {
        if (has_gk && g_slot) ScaleSlot{g_slot}.record(gk_amax);
    }
}

// ------------------------------------------------------------------------------------------------
// fac_bwd_rows_p16x8 (round 6): fac_bwd_rows_f32<5, TPR, H16, CLAMP> -- the training step's FAC backward on fp16 filter /
// grad_kernel planes and the unpadded input -- with EIGHT pixels per thread instead of four.  The two big streams of the op
// (25 filter planes read, 25 grad_kernel planes written) are fp16: four pixels are an 8-byte access per lane, and at 8 bytes
// per lane the kernel ran at 0.44 of the HBM peak where its fp32 twin, at 16 bytes per lane, runs at 0.73.  Eight pixels
// make every plane access 16 bytes again (Wo % 8 == 0).  Same arithmetic in the same order per output element: the partial
// sums of grad_input, the hand-over of the K - 1 = 4 upper partials to the next lane (carried across chunks), the replicate
// padding's adjoint folded in a fixed order, grad_kernel times the LeakyReLU derivative of the filters, |max| recorded.
typedef _Float16 f16x8_fac __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_fac __attribute__((ext_vector_type(4)));
template <int TPR>
__global__ __launch_bounds__(256) void fac_bwd_rows_p16x8(const float *__restrict__ in, Str4 is, const _Float16 *__restrict__ kern,
                                                          Str4 ks, const float *__restrict__ gout, Str4 gs,
                                                          float *__restrict__ gin, Str4 gis, _Float16 *__restrict__ gkern,
                                                          Str4 gks, int C, int Ho, int Wo, float kslope,
                                                          const float *__restrict__ f_slot, float *__restrict__ g_slot) {
    saturate_fp16_conversions();           // (no matrix instructions in this kernel)
    constexpr int K = 5, PX = 8, NS = PX + K - 1, NU = K - 1, R2 = K / 2;
    constexpr int RPW = 64 / TPR, ROWS = 4 * RPW;
    const float finv = 1.f / f_slot[0], gsc = g_slot[0];
    float gk_amax = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane / TPR, tx = lane % TPR;
    const int row = blockIdx.x * ROWS + wave * RPW + r;   // output row
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    const int Wi = Wo + K - 1;
    const int nxp = Wo / PX;
    const bool rowok = row < Ho;
    const int nchunks = nxp / TPR + 1;     // the lane right after the last loading lane writes the tail
    const int Ylo = row == 0 ? 0 : row + R2;
    const int Yhi = row == Ho - 1 ? Ho - 1 + 2 * R2 : row + R2;
    const float *inpl = in + (int64_t)b * is.s0 + (int64_t)c * is.s1;
    const int64_t kbase = (int64_t)b * ks.s0 + (int64_t)c * K * K * ks.s1;
    const float *gbase = gout + (int64_t)b * gs.s0 + (int64_t)c * gs.s1;
    const int64_t gkbase = (int64_t)b * gks.s0 + (int64_t)c * K * K * gks.s1;
    float *ginpl = gin ? gin + (int64_t)b * gis.s0 + (int64_t)c * gis.s1 : nullptr;

    for (int yi = 0; yi < K; ++yi) {
        const int Y = Ylo + yi;
        const bool yok = rowok && Y <= Yhi;
        if (__builtin_amdgcn_ballot_w64(yok) == 0) break;          // (wave-uniform: the shuffles below need every lane)
        const float *inrow = inpl + (int64_t)min(max(Y - R2, 0), Ho - 1) * is.s2;
        float *ginrow = ginpl ? ginpl + (int64_t)row * gis.s2 : nullptr;
        float carry[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) carry[m] = 0.f;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int xp = chunk * TPR + tx;
            const int x = xp * PX;
            const bool active = yok && xp < nxp;
            float s[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) s[j] = 0.f;
            if (active) {
                float inr[NS];             // padded columns x .. x + 11 = input columns clamp(x + j - 2)
                if (x >= PX && x + NS <= Wo) {
                    const float2 *p2 = reinterpret_cast<const float2 *>(inrow + x - R2);     // 8-byte aligned (x % 8 == 0, rows 16-byte aligned)
#pragma unroll
                    for (int j = 0; j < NS / 2; ++j) {
                        const float2 t = p2[j];
                        inr[2 * j] = t.x;
                        inr[2 * j + 1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NS; ++j) inr[j] = inrow[min(max(x + j - R2, 0), Wo - 1)];
                }
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int y = Y - ky;
                    if (y < 0 || y >= Ho) continue;
                    const float4 ga = *reinterpret_cast<const float4 *>(gbase + (int64_t)y * gs.s2 + x);
                    const float4 gb = *reinterpret_cast<const float4 *>(gbase + (int64_t)y * gs.s2 + x + 4);
                    const float g[PX] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                    // The kernel is paced by its vector arithmetic (1.7 G (pixel, tap) pairs per launch), so the two power-of-two
                    // scales stay OUT of the inner loop -- exactly: the filters are accumulated as stored (their scale leaves with
                    // the partial sums, below) and grad_output enters pre-multiplied by grad_kernel's scale, with the LeakyReLU
                    // slope folded in for the filters that are <= 0; |max| is taken on the scaled products and de-scaled once.
                    float gg[PX], gk[PX];
#pragma unroll
                    for (int i = 0; i < PX; ++i) gg[i] = g[i] * gsc, gk[i] = gg[i] * kslope;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int t = ky * K + kx;
                        const u32x4_fac q = __builtin_nontemporal_load(
                            reinterpret_cast<const u32x4_fac *>(kern + kbase + (int64_t)t * ks.s1 + (int64_t)y * ks.s2 + x));
                        const f16x8_fac wh = __builtin_bit_cast(f16x8_fac, q);
                        float p[PX];
#pragma unroll
                        for (int i = 0; i < PX; ++i) {
                            const float w = (float)wh[i];              // filter * f_slot scale (the sign is the filter's)
                            // (in * (grad_out * slope) where the filter is <= 0: one rounding apart, in fp32, from the fp32 kernels'
                            //  (in * grad_out) * slope -- a select between two precomputed factors and ONE product per value instead of
                            //  two products and a select: 242 -> 186 us per launch on the same box)
                            p[i] = inr[kx + i] * (w > 0.f ? gg[i] : gk[i]);
                            gk_amax = amax_acc(gk_amax, p[i]);
                            s[kx + i] = fmaf(w, g[i], s[kx + i]);
                        }
                        const u32x4_fac o = {pack_f16(p[0], p[1]), pack_f16(p[2], p[3]), pack_f16(p[4], p[5]), pack_f16(p[6], p[7])};
                        __builtin_nontemporal_store(o, reinterpret_cast<u32x4_fac *>(gkern + gkbase + (int64_t)t * gks.s1 + (int64_t)y * gks.s2 + x));
                    }
                }
            }
            // hand the K - 1 upper partials to the lane that owns those columns (all lanes take part)
            float o[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) o[i] = s[i];
#pragma unroll
            for (int m = 0; m < NU; ++m) {
                float left = __shfl_up(s[PX + m], 1, TPR);
                if (tx == 0) left = carry[m];
                const float last = __shfl(s[PX + m], TPR - 1, TPR);
                o[m] += left;
                carry[m] = last;
            }
#pragma unroll
            for (int i = 0; i < PX; ++i) o[i] *= finv;              // (the filters' power-of-two scale: exact, and it commutes with the sums)
            if (ginrow != nullptr && yok && x < Wi) {
                // padded columns x .. x + 7 -> output columns clamp(X - 2): the left border's three (X = 0, 1, 2 -> 0) sit in the thread
                // with x == 0, the right border's three (X = Wo + 1 .. Wo + 3 -> Wo - 1) in the tail thread x == Wo (which holds only
                // the four carried partials).  yi > 0 (first / last output row): this thread wrote the elements in an earlier pass.
                if (x == 0) {
                    const float v0 = (o[0] + o[1]) + o[2];
                    if (yi == 0) {
                        ginrow[0] = v0; ginrow[1] = o[3];
#pragma unroll
                        for (int i = 4; i < PX; ++i) ginrow[i - 2] = o[i];
                    } else {
                        ginrow[0] += v0; ginrow[1] += o[3];
#pragma unroll
                        for (int i = 4; i < PX; ++i) ginrow[i - 2] += o[i];
                    }
                } else if (x == Wo) {
                    const float v1 = (o[1] + o[2]) + o[3];
                    if (yi == 0) { ginrow[Wo - 2] = o[0]; ginrow[Wo - 1] = v1; }
                    else { ginrow[Wo - 2] += o[0]; ginrow[Wo - 1] += v1; }
                } else {
                    float2 *g2 = reinterpret_cast<float2 *>(ginrow + x - R2);
#pragma unroll
                    for (int i = 0; i < PX / 2; ++i) {
                        if (yi == 0) {
                            g2[i] = float2{o[2 * i], o[2 * i + 1]};
                        } else {
                            const float2 a = g2[i];
                            g2[i] = float2{a.x + o[2 * i], a.y + o[2 * i + 1]};
                        }
                    }
                }
            }
        }
    }
    ScaleSlot{g_slot}.record(gk_amax * (1.f / gsc));       // (|max| of the UNSCALED gradient, like every other writer; NaN / Inf stay what they are)
}

__global__ void fac_bwd_input_generic_f32(const float *__restrict__ kern, Str4 ks,
                                          const float *__restrict__ gout, Str4 gs, float *__restrict__ gin,
                                          Str4 gis, int64_t total, int C, int Ho, int Wo, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Hi = Ho + K - 1, Wi = Wo + K - 1;
    const int X = (int)(idx % Wi);
    const int Y = (int)((idx / Wi) % Hi);
    const int c = (int)((idx / ((int64_t)Wi * Hi)) % C);
    const int64_t b = idx / ((int64_t)Wi * Hi * C);
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx) {
            const int y = Y - ky, x = X - kx;
            if (y < 0 || y > Ho - 1 || x < 0 || x > Wo - 1) continue;
            const float w = kern[b * ks.s0 + ((int64_t)c * K * K + ky * K + kx) * ks.s1 + (int64_t)y * ks.s2 +
                                 (int64_t)x * ks.s3];
            acc = fmaf(w, gout[b * gs.s0 + (int64_t)c * gs.s1 + (int64_t)y * gs.s2 + (int64_t)x * gs.s3], acc);
        }
    gin[b * gis.s0 + (int64_t)c * gis.s1 + (int64_t)Y * gis.s2 + (int64_t)X * gis.s3] = acc;
}

__global__ void fac_bwd_kernel_generic_f32(const float *__restrict__ in, Str4 is,
                                           const float *__restrict__ gout, Str4 gs, float *__restrict__ gkern,
                                           Str4 gks, int64_t total, int C, int Ho, int Wo, int K,
                                           const float *__restrict__ kern, Str4 ks, float kslope) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % Wo);
    const int y = (int)((idx / Wo) % Ho);
    const int kx = (int)((idx / ((int64_t)Wo * Ho)) % K);
    const int ky = (int)((idx / ((int64_t)Wo * Ho * K)) % K);
    const int c = (int)((idx / ((int64_t)Wo * Ho * K * K)) % C);
    const int64_t b = idx / ((int64_t)Wo * Ho * K * K * C);
    float v = in[b * is.s0 + (int64_t)c * is.s1 + (int64_t)(y + ky) * is.s2 + (int64_t)(x + kx) * is.s3] *
              gout[b * gs.s0 + (int64_t)c * gs.s1 + (int64_t)y * gs.s2 + (int64_t)x * gs.s3];
    if (kslope != 1.f && kern[b * ks.s0 + ((int64_t)c * K * K + ky * K + kx) * ks.s1 + (int64_t)y * ks.s2 + (int64_t)x * ks.s3] <= 0.f)
        v *= kslope;
    gkern[b * gks.s0 + ((int64_t)c * K * K + ky * K + kx) * gks.s1 + (int64_t)y * gks.s2 + (int64_t)x * gks.s3] = v;
}

// rows of `p` start 16-byte aligned and the three outer strides keep that alignment
static bool vec4_ok(const void *p, const Str4 &s) {
    return p != nullptr && aligned16(p) && s.s3 == 1 && (s.s0 % 4 == 0) && (s.s1 % 4 == 0) && (s.s2 % 4 == 0);
}

int check_shapes(const int64_t *ish, const int64_t *ksh, int K, int64_t *B, int64_t *C, int64_t *Ho, int64_t *Wo) {
    if (K < 1) return fail(EBFI_ERR_ARG, "fac: kernel_size %d < 1", K);
    *B = ish[0]; *C = ish[1]; *Ho = ksh[2]; *Wo = ksh[3];
    if (ksh[0] != *B) return fail(EBFI_ERR_ARG, "fac: batch mismatch input %lld vs kernel %lld", (long long)ish[0], (long long)ksh[0]);
    if (ksh[1] != *C * K * K)
        return fail(EBFI_ERR_ARG, "fac: kernel has %lld channels, expected C*K*K = %lld", (long long)ksh[1], (long long)(*C * K * K));
    if (ish[2] - K != *Ho - 1 || ish[3] - K != *Wo - 1)
        return fail(EBFI_ERR_ARG, "fac: input %lldx%lld is not output %lldx%lld + K-1 (K=%d)", (long long)ish[2],
                    (long long)ish[3], (long long)*Ho, (long long)*Wo, K);
    if (*B < 0 || *C < 0 || *Ho < 0 || *Wo < 0 || *Ho > (1 << 24) || *Wo > (1 << 24) || *B * *C > (1LL << 31) - 1)
        return fail(EBFI_ERR_ARG, "fac: tensor extent out of range");
    return EBFI_OK;
}

template <int K, bool H16 = false, bool CLAMP = false>
void launch_fwd_tile(hipStream_t st, const float *in, Str4 is, const void *kern, Str4 ks, float *out, Str4 os,
                     int B, int C, int Ho, int Wo, const float *f_slot = nullptr) {
    const double bytes = B * C * (double)Ho * Wo * (4.0 + (H16 ? 2.0 : 4.0) * K * K + 4.0);   // in + K*K filter planes + out
    const char *name = H16 ? "fac_fwd_tile_f32/p16" : "fac_fwd_tile_f32";
    if (Wo <= 64) {
        dim3 grid((unsigned)ceil_div(Wo, 64), (unsigned)ceil_div(Ho, 16), (unsigned)(B * C));
        ProfScope ps(name, st, 2.0 * B * C * (double)Ho * Wo * K * K, bytes);
        hipLaunchKernelGGL((fac_fwd_tile_f32<K, 16, 64, H16, CLAMP>), grid, dim3(256), 0, st, in, is, kern, ks, out, os, C, Ho, Wo, f_slot);
    } else {
        dim3 grid((unsigned)ceil_div(Wo, 128), (unsigned)ceil_div(Ho, 8), (unsigned)(B * C));
        ProfScope ps(name, st, 2.0 * B * C * (double)Ho * Wo * K * K, bytes);
        hipLaunchKernelGGL((fac_fwd_tile_f32<K, 8, 128, H16, CLAMP>), grid, dim3(256), 0, st, in, is, kern, ks, out, os, C, Ho, Wo, f_slot);
    }
}

template <int K, int TPR, bool H16 = false, bool CLAMP = false>
void launch_bwd_rows_t(hipStream_t st, const float *in, Str4 is, const void *kern, Str4 ks, const float *go,
                       Str4 gs, float *gin, Str4 gis, void *gk, Str4 gks, int B, int C, int Ho, int Wo, float kslope,
                       const float *f_slot = nullptr, float *g_slot = nullptr) {
    constexpr int ROWS = 4 * (64 / TPR);
    dim3 grid((unsigned)ceil_div(CLAMP ? Ho : Ho + K - 1, ROWS), (unsigned)(B * C));
    const double px = (double)B * C * Ho * Wo;     // filters + gout + in read, grad_in + grad_kernel written
    const double e = H16 ? 2.0 : 4.0;
    ProfScope ps(H16 ? "fac_bwd_rows_f32/p16" : "fac_bwd_rows_f32", st, 4.0 * px * K * K,
                 px * (e * K * K + 4.0 + 4.0 + (gin ? 4.0 : 0.0) + (gk ? e * K * K : 0.0)));
    hipLaunchKernelGGL((fac_bwd_rows_f32<K, TPR, H16, CLAMP>), grid, dim3(256), 0, st, in, is, kern, ks, go, gs, gin, gis, gk, gks,
                       C, Ho, Wo, kslope, f_slot, g_slot);
}

template <int K, bool H16 = false, bool CLAMP = false>
void launch_bwd_rows(hipStream_t st, const float *in, Str4 is, const void *kern, Str4 ks, const float *go, Str4 gs,
                     float *gin, Str4 gis, void *gk, Str4 gks, int B, int C, int Ho, int Wo, float kslope,
                     const float *f_slot = nullptr, float *g_slot = nullptr) {
    const int nx4 = Wo / 4;
    if (nx4 <= 8) launch_bwd_rows_t<K, 8, H16, CLAMP>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, B, C, Ho, Wo, kslope, f_slot, g_slot);
    else if (nx4 <= 16) launch_bwd_rows_t<K, 16, H16, CLAMP>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, B, C, Ho, Wo, kslope, f_slot, g_slot);
    else if (nx4 <= 32) launch_bwd_rows_t<K, 32, H16, CLAMP>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, B, C, Ho, Wo, kslope, f_slot, g_slot);
    else launch_bwd_rows_t<K, 64, H16, CLAMP>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, B, C, Ho, Wo, kslope, f_slot, g_slot);
}

}  // namespace

extern "C" int ebfi_fac_forward(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                                const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                                int kernel_size, void *output, const int64_t output_shape[4],
                                const int64_t output_stride[4], int dtype, void *stream) {
    if (!input || !kernel || !output || !input_shape || !input_stride || !kernel_shape || !kernel_stride ||
        !output_shape || !output_stride)
        return fail(EBFI_ERR_ARG, "fac_forward: null argument");
    if (dtype != EBFI_F32) return fail(EBFI_ERR_UNSUPPORTED, "fac_forward: dtype %d not implemented (fp32 only)", dtype);
    int64_t B, C, Ho, Wo;
    const int K = kernel_size;
    if (int rc = check_shapes(input_shape, kernel_shape, K, &B, &C, &Ho, &Wo)) return rc;
    if (output_shape[0] != B || output_shape[1] != C || output_shape[2] != Ho || output_shape[3] != Wo)
        return fail(EBFI_ERR_ARG, "fac_forward: output shape mismatch");
    if (B * C * Ho * Wo == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *in = static_cast<const float *>(input);
    const float *kern = static_cast<const float *>(kernel);
    float *out = static_cast<float *>(output);
    const Str4 is = str4(input_stride), ks = str4(kernel_stride), os = str4(output_stride);

    // gridDim.z carries the (b, c) plane index: at most 65535 planes on the tiled path
    const bool fast = (K == 1 || K == 3 || K == 5) && is.s3 == 1 && (Wo % 4 == 0) && vec4_ok(kern, ks) &&
                      vec4_ok(out, os) && B * C <= 65535;
    if (fast) {
        if (K == 5) launch_fwd_tile<5>(st, in, is, kern, ks, out, os, (int)B, (int)C, (int)Ho, (int)Wo);
        else if (K == 3) launch_fwd_tile<3>(st, in, is, kern, ks, out, os, (int)B, (int)C, (int)Ho, (int)Wo);
        else launch_fwd_tile<1>(st, in, is, kern, ks, out, os, (int)B, (int)C, (int)Ho, (int)Wo);
        return check_launch("fac_fwd_tile_f32");
    }
    const int64_t total = B * C * Ho * Wo;
    {
        ProfScope ps("fac_fwd_generic_f32", st);
        hipLaunchKernelGGL(fac_fwd_generic_f32, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, in, is, kern, ks,
                           out, os, total, (int)C, (int)Ho, (int)Wo, K);
    }
    return check_launch("fac_fwd_generic_f32");
}

extern "C" int ebfi_fac_backward_ex(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                                    const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                                    int kernel_size, const void *grad_output, const int64_t grad_output_stride[4],
                                    void *grad_input, const int64_t grad_input_stride[4], void *grad_kernel,
                                    const int64_t grad_kernel_stride[4], float kernel_leaky_slope, int dtype, void *stream);

extern "C" int ebfi_fac_backward(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                                 const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                                 int kernel_size, const void *grad_output, const int64_t grad_output_stride[4],
                                 void *grad_input, const int64_t grad_input_stride[4], void *grad_kernel,
                                 const int64_t grad_kernel_stride[4], int dtype, void *stream) {
    return ebfi_fac_backward_ex(input, input_shape, input_stride, kernel, kernel_shape, kernel_stride, kernel_size, grad_output,
                                grad_output_stride, grad_input, grad_input_stride, grad_kernel, grad_kernel_stride, 1.f, dtype,
                                stream);
}

// kernel_leaky_slope != 1: the filters are the output of a LeakyReLU(kernel_leaky_slope) layer and grad_kernel leaves as the
// gradient of that layer's PRE-activation (multiplied by 1 where kernel > 0, by the slope elsewhere) -- the conv that
// produced the filters then needs neither its saved output nor a grad*act' side tensor in its own backward
extern "C" int ebfi_fac_backward_ex(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                                    const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                                    int kernel_size, const void *grad_output, const int64_t grad_output_stride[4],
                                    void *grad_input, const int64_t grad_input_stride[4], void *grad_kernel,
                                    const int64_t grad_kernel_stride[4], float kernel_leaky_slope, int dtype, void *stream) {
    const float kslope = kernel_leaky_slope;
    if (!input || !kernel || !grad_output || !input_shape || !input_stride || !kernel_shape || !kernel_stride ||
        !grad_output_stride)
        return fail(EBFI_ERR_ARG, "fac_backward: null argument");
    if ((grad_input && !grad_input_stride) || (grad_kernel && !grad_kernel_stride))
        return fail(EBFI_ERR_ARG, "fac_backward: gradient pointer without strides");
    if (dtype != EBFI_F32) return fail(EBFI_ERR_UNSUPPORTED, "fac_backward: dtype %d not implemented (fp32 only)", dtype);
    int64_t B, C, Ho, Wo;
    const int K = kernel_size;
    if (int rc = check_shapes(input_shape, kernel_shape, K, &B, &C, &Ho, &Wo)) return rc;
    if (!grad_input && !grad_kernel) return EBFI_OK;
    if (B * C * Ho * Wo == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *in = static_cast<const float *>(input);
    const float *kern = static_cast<const float *>(kernel);
    const float *go = static_cast<const float *>(grad_output);
    float *gin = static_cast<float *>(grad_input);
    float *gk = static_cast<float *>(grad_kernel);
    const Str4 is = str4(input_stride), ks = str4(kernel_stride), gs = str4(grad_output_stride);
    const Str4 gis = gin ? str4(grad_input_stride) : Str4{0, 0, 0, 1};
    const Str4 gks = gk ? str4(grad_kernel_stride) : Str4{0, 0, 0, 1};

    const bool fast = (K == 1 || K == 3 || K == 5) && (Wo % 4 == 0) && is.s3 == 1 && vec4_ok(kern, ks) &&
                      vec4_ok(go, gs) && (!gk || vec4_ok(gk, gks)) && (!gin || gis.s3 == 1) && B * C <= 65535;
    if (fast) {
        if (K == 5) launch_bwd_rows<5>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, (int)B, (int)C, (int)Ho, (int)Wo, kslope);
        else if (K == 3) launch_bwd_rows<3>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, (int)B, (int)C, (int)Ho, (int)Wo, kslope);
        else launch_bwd_rows<1>(st, in, is, kern, ks, go, gs, gin, gis, gk, gks, (int)B, (int)C, (int)Ho, (int)Wo, kslope);
        return check_launch("fac_bwd_rows_f32");
    }
    if (gin) {
        const int64_t total = B * C * (Ho + K - 1) * (Wo + K - 1);
        ProfScope ps("fac_bwd_input_generic_f32", st);
        hipLaunchKernelGGL(fac_bwd_input_generic_f32, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, kern, ks,
                           go, gs, gin, gis, total, (int)C, (int)Ho, (int)Wo, K);
    }
    if (int rc = check_launch("fac_bwd_input_generic_f32")) return rc;
    if (gk) {
        const int64_t total = B * C * K * K * Ho * Wo;
        ProfScope ps("fac_bwd_kernel_generic_f32", st);
        hipLaunchKernelGGL(fac_bwd_kernel_generic_f32, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, in, is,
                           go, gs, gk, gks, total, (int)C, (int)Ho, (int)Wo, K, kern, ks, kslope);
    }
    return check_launch("fac_bwd_kernel_generic_f32");
}

// ------------------------------------------------------------------------------------------------ fp16 filter storage (round 4)
// The FAC op with its two big tensors kept as fp16 PLANES scaled by the power-of-two of a scale slot (c16.hpp ScaleSlot,
// ebfi_f16_scales_finish): filters16 / grad_kernel16 [B, C*K*K, Ho, Wo] halves, everything contiguous; input_pad
// [B, C, Ho+K-1, Wo+K-1], output / grad_output [B, C, Ho, Wo], grad_input_pad like input_pad (fp32).  K = 5 (the model's),
// Wo % 4 == 0.  grad_kernel16 leaves multiplied by the LeakyReLU(kernel_leaky_slope) derivative of the filters, times
// g_slot[0], its |max| recorded into g_slot (the WRITER of an fp16 tensor records, c16.hpp).
extern "C" int ebfi_fac_forward_p16(const float *input, int input_is_unpadded, const void *filters16, const void *f_slot, float *output,
                                    int B, int C, int Ho, int Wo, int K, void *stream) {
    if (!input || !filters16 || !f_slot || !output) return fail(EBFI_ERR_ARG, "fac_forward_p16: null argument");
    if (K != 5 || Wo % 4 != 0 || B < 0 || C < 1 || Ho < 1 || Wo < 4 || (int64_t)B * C > 65535)
        return fail(EBFI_ERR_UNSUPPORTED, "fac_forward_p16: K = 5, Wo %% 4 == 0, B*C <= 65535 (K=%d Wo=%d)", K, Wo);
    if (!aligned16(filters16) || !aligned16(output)) return fail(EBFI_ERR_ARG, "fac_forward_p16: 16-byte aligned tensors");
    if (B == 0) return EBFI_OK;
    const int64_t Hi = Ho + K - 1, Wi = Wo + K - 1, HW = (int64_t)Ho * Wo;
    const Str4 ks{(int64_t)C * K * K * HW, HW, Wo, 1}, os{C * HW, HW, Wo, 1};
    if (input_is_unpadded)
        launch_fwd_tile<5, true, true>(static_cast<hipStream_t>(stream), input, os, filters16, ks, output, os, B, C, Ho, Wo,
                                       static_cast<const float *>(f_slot));
    else
        launch_fwd_tile<5, true>(static_cast<hipStream_t>(stream), input, Str4{C * Hi * Wi, Hi * Wi, Wi, 1}, filters16, ks, output, os,
                                 B, C, Ho, Wo, static_cast<const float *>(f_slot));
    return check_launch("fac_fwd_tile_f32/p16");
}

extern "C" int ebfi_fac_backward_p16(const float *input, int input_is_unpadded, const void *filters16, const void *f_slot,
                                     const float *grad_output, float *grad_input, void *grad_kernel16, void *g_slot,
                                     float kernel_leaky_slope, int B, int C, int Ho, int Wo, int K, void *stream) {
    if (!input || !filters16 || !f_slot || !grad_output) return fail(EBFI_ERR_ARG, "fac_backward_p16: null argument");
    if (grad_kernel16 && !g_slot) return fail(EBFI_ERR_ARG, "fac_backward_p16: grad_kernel16 needs its scale slot");
    if (K != 5 || Wo % 4 != 0 || B < 0 || C < 1 || Ho < 1 || Wo < 4 || (int64_t)B * C > 65535)
        return fail(EBFI_ERR_UNSUPPORTED, "fac_backward_p16: K = 5, Wo %% 4 == 0, B*C <= 65535 (K=%d Wo=%d)", K, Wo);
    if (!aligned16(filters16) || !aligned16(grad_output) || (grad_kernel16 && !aligned16(grad_kernel16)))
        return fail(EBFI_ERR_ARG, "fac_backward_p16: 16-byte aligned tensors");
    if (!grad_input && !grad_kernel16) return EBFI_OK;
    if (B == 0) return EBFI_OK;
    const int64_t Hi = Ho + K - 1, Wi = Wo + K - 1, HW = (int64_t)Ho * Wo;
    const Str4 ks{(int64_t)C * K * K * HW, HW, Wo, 1}, gs{C * HW, HW, Wo, 1};
    if (input_is_unpadded && grad_kernel16 && Wo % 8 == 0 && Wo >= 16 && dev_getenv("EBFI_FAC_BWD_X4") == nullptr) {
        // the step's configuration: eight pixels per thread, 16-byte accesses of the fp16 planes (fac_bwd_rows_p16x8)
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int nxp = Wo / 8;
        const double px = (double)B * C * Ho * Wo;
        // (label = kernel symbol / role: bench.py looks the launch's PMC traffic up by the symbol)
        ProfScope ps("fac_bwd_rows_p16x8/p16", st, 4.0 * px * K * K, px * (2.0 * K * K + 4.0 + 4.0 + (grad_input ? 4.0 : 0.0) + 2.0 * K * K));
#define EBFI_LAUNCH_FACB8(TPR_)                                                                                          \
    hipLaunchKernelGGL((fac_bwd_rows_p16x8<TPR_>), dim3((unsigned)ceil_div(Ho, 4 * (64 / TPR_)), (unsigned)(B * C)), dim3(256), 0, st, \
                       input, gs, static_cast<const _Float16 *>(filters16), ks, grad_output, gs, grad_input, gs,         \
                       static_cast<_Float16 *>(grad_kernel16), ks, C, Ho, Wo, kernel_leaky_slope,                        \
                       static_cast<const float *>(f_slot), static_cast<float *>(g_slot))
        if (nxp <= 8) EBFI_LAUNCH_FACB8(8);
        else if (nxp <= 16) EBFI_LAUNCH_FACB8(16);
        else if (nxp <= 32) EBFI_LAUNCH_FACB8(32);
        else EBFI_LAUNCH_FACB8(64);
#undef EBFI_LAUNCH_FACB8
    } else if (input_is_unpadded) {
        launch_bwd_rows<5, true, true>(static_cast<hipStream_t>(stream), input, gs, filters16, ks, grad_output, gs, grad_input, gs,
                                       grad_kernel16, ks, B, C, Ho, Wo, kernel_leaky_slope, static_cast<const float *>(f_slot),
                                       static_cast<float *>(g_slot));
    } else {
        const Str4 is{C * Hi * Wi, Hi * Wi, Wi, 1};
        launch_bwd_rows<5, true>(static_cast<hipStream_t>(stream), input, is, filters16, ks, grad_output, gs, grad_input, is,
                                 grad_kernel16, ks, B, C, Ho, Wo, kernel_leaky_slope, static_cast<const float *>(f_slot),
                                 static_cast<float *>(g_slot));
    }
    return check_launch("fac_bwd_rows_f32/p16");
}

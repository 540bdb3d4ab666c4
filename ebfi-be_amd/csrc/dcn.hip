// Modulated deformable convolution v2 (DCNv2) for gfx950, fp32.
//
// Semantics: reference models/DCNv2/src/cuda/dcn_v2_cuda.cu:20-216 (wrappers) and
// src/cuda/dcn_v2_im2col_cuda.cu:25-402 (kernels); entry points replace `_ext.dcn_v2_forward` /
// `_ext.dcn_v2_backward` (src/dcn_v2.h:9-92).
//
// The reference materialises a [B, C*kh*kw, Ho*Wo] column tensor in HBM (302 MB at B=8, 64ch,
// 128x128), runs one big matmul, and in backward loops over samples launching 3 kernels + 3
// matmuls each.  Here the column tensor never leaves the CU:
//
//   forward   workgroup = 64 output pixels x 64 output channels.  The K = C*kh*kw contraction is
//             walked in chunks of one deformable group's channels (<= 72 rows): the 4 waves sample
//             the chunk's columns (offset/mask read coalesced along pixels, 4-corner gathers from
//             the L2-resident input planes, x mask) straight into LDS next to the matching weight
//             slice, then each wave drives one 32x32 tile with exact-fp32 MFMA
//             (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, so parity with the fp32 reference
//             holds to rounding order).  bias is added last like the reference.
//   backward  (1) data kernel: per pixel tile, colgrad = W^T . grad_out for one chunk via 16x16x4
//             fp32 MFMA into LDS, then the same sampling walk produces grad_offset / grad_mask
//             (deterministic, one writer each) and scatters grad_input with fp32 atomics -- the
//             reference does the same with atomicAdd (dcn_v2_im2col_cuda.cu:249);
//             (2) weight kernel: re-samples the columns per tile and contracts them against
//             grad_out over PIXELS with MFMA, each workgroup accumulating its share of tiles in
//             registers and writing one partial [Co, C*kk] slab; (3) a tiny reduction kernel sums
//             the slabs (and the bias partials) in a fixed order => grad_weight / grad_bias are
//             bit-reproducible.
#include "common.hpp"

using namespace ebfi;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// split precision (see conv2d.hip): a value as the pair (bf16 hi << 16 | bf16 lo), lo = bf16(v - hi)
__device__ __forceinline__ unsigned split_word(float v) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const __bf16 h = (__bf16)v;
    const float hf = (float)h;
    bf16x2 p = {(__bf16)(v - hf), h};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void peel(const unsigned (&w)[8], bf16x8 &hi, bf16x8 &lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);
        l[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

constexpr int NP = 64;     // pixels per workgroup tile
constexpr int KC = 72;     // max contraction rows per chunk (one group's channels x taps)
constexpr int KCP = 80;    // KC rounded up to a multiple of 16 (MFMA m/n tiles in backward)
constexpr int WSTR = 65;   // forward: weight-slice row stride (co fastest, odd => conflict-free transposing store)
constexpr int S80 = 80;    // backward data: row stride of the [k][m] operand images (2 rows -> disjoint banks)
constexpr int S81 = 81;    // backward weight: row stride of transposed images written column-wise

struct Geom {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg;
    int Ho, Wo, HWo, kk, cpg, CB, nsub, nchunks, tiles_per_img, Kd;
};

struct Tap {
    float w1, w2, w3, w4;   // bilinear weights of (low,low) (low,high) (high,low) (high,high)
    int o1, o2, o3, o4;     // plane offsets of the four corners, -1 when outside the image
    float lh, lw;           // fractional parts (for the coordinate gradient)
    int h0, w0;             // integer low corner (may be -1)
    int q0, q1;             // offsets of the 2-pixel pair (columns c, c+1) in the low / high row, -1 = row outside
    float mlx, mly, mhx, mhy;   // which pair element is the low-column / high-column corner (0/1 masks)
    bool valid;             // -1 < h < H and -1 < w < W   (dcn_v2_im2col_cuda.cu:180)
};

__device__ __forceinline__ Tap make_tap(float h, float w, int H, int W) {
    Tap t;
    t.valid = (h > -1.f) && (w > -1.f) && (h < (float)H) && (w < (float)W);
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    t.o1 = t.o2 = t.o3 = t.o4 = -1;
    t.lh = t.lw = 0.f;
    t.h0 = t.w0 = 0;
    t.q0 = t.q1 = -1;
    t.mlx = t.mly = t.mhx = t.mhy = 0.f;
    if (t.valid) {
        const int h0 = (int)floorf(h), w0 = (int)floorf(w);
        t.h0 = h0; t.w0 = w0;
        const int h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h - (float)h0, lw = w - (float)w0;
        const float hh = 1.f - lh, hw = 1.f - lw;
        t.lh = lh; t.lw = lw;
        t.w1 = hh * hw; t.w2 = hh * lw; t.w3 = lh * hw; t.w4 = lh * lw;
        if (h0 >= 0 && w0 >= 0) t.o1 = h0 * W + w0;
        if (h0 >= 0 && w1 <= W - 1) t.o2 = h0 * W + w1;
        if (h1 <= H - 1 && w0 >= 0) t.o3 = h1 * W + w0;
        if (h1 <= H - 1 && w1 <= W - 1) t.o4 = h1 * W + w1;
        // the two corners of a row are adjacent in memory: fetch them as ONE 8-byte access at column c
        // (half the gather instructions through the texture addresser).  c is clamped so the pair stays
        // inside the row; the masks say which element plays which corner (a corner outside reads as 0).
        const int c = w0 < 0 ? 0 : (w0 > W - 2 ? W - 2 : w0);
        t.mlx = (w0 == c) ? 1.f : 0.f;          // low corner  = pair.x
        t.mly = (w0 == c + 1) ? 1.f : 0.f;      // low corner  = pair.y  (w0 == W-1)
        t.mhx = (w1 == c) ? 1.f : 0.f;          // high corner = pair.x  (w0 == -1)
        t.mhy = (w1 == c + 1) ? 1.f : 0.f;      // high corner = pair.y
        t.q0 = h0 >= 0 ? h0 * W + c : -1;
        t.q1 = h1 <= H - 1 ? h1 * W + c : -1;
    }
    return t;
}

struct __attribute__((packed, aligned(4))) Pair {
    float x, y;
};
// W >= 2 is required for the paired fetch (checked by the launchers; W == 1 never occurs on this path)
__device__ __forceinline__ void corners(const float *__restrict__ plane, const Tap &t, float &v1, float &v2,
                                        float &v3, float &v4) {
    Pair a = {0.f, 0.f}, b = {0.f, 0.f};
    if (t.q0 >= 0) a = *reinterpret_cast<const Pair *>(plane + t.q0);
    if (t.q1 >= 0) b = *reinterpret_cast<const Pair *>(plane + t.q1);
    v1 = a.x * t.mlx + a.y * t.mly;
    v2 = a.x * t.mhx + a.y * t.mhy;
    v3 = b.x * t.mlx + b.y * t.mly;
    v4 = b.x * t.mhx + b.y * t.mhy;
}

// Sample `n` consecutive channels of one (pixel, tap) into the LDS column image.  NCH > 0 unrolls exactly
// NCH channels so that all 4*NCH corner gathers are in flight together; NCH == 0 is the generic loop.
// SPLIT: store the split-precision word of the sample (bit pattern in the float slot) instead of the float.
template <bool SPLIT>
__device__ __forceinline__ float col_word(float v) {
    if constexpr (SPLIT) return __uint_as_float(split_word(v));
    else return v;
}
template <int NCH, bool SPLIT = false>
__device__ __forceinline__ void sample_cols(const float *__restrict__ plane, int64_t plane_stride, const Tap &t,
                                            float mask, float *col, int col_stride, int n) {
    if constexpr (NCH > 0) {
        float v[NCH][4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) corners(plane + c * plane_stride, t, v[c][0], v[c][1], v[c][2], v[c][3]);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            col[c * col_stride] = col_word<SPLIT>((t.w1 * v[c][0] + t.w2 * v[c][1] + t.w3 * v[c][2] + t.w4 * v[c][3]) * mask);
    } else {
        for (int c = 0; c < n; ++c, plane += plane_stride) {
            float v1, v2, v3, v4;
            corners(plane, t, v1, v2, v3, v4);
            col[c * col_stride] = col_word<SPLIT>((t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4) * mask);
        }
    }
}

struct Chunk {
    int grp, cbase, cb, KL;   // deformable group, first global channel, channels, rows = cb*kk
};
__device__ __forceinline__ Chunk get_chunk(const Geom &g, int chunk) {
    Chunk c;
    c.grp = chunk / g.nsub;
    const int c0 = (chunk - c.grp * g.nsub) * g.CB;
    c.cb = min(g.CB, g.cpg - c0);
    c.cbase = c.grp * g.cpg + c0;
    c.KL = c.cb * g.kk;
    return c;
}

// sample position of (pixel p, tap) of group grp; pad_w_eff lets the caller apply the reference's
// pad_h-for-pad_w quirk of col2im.
struct TapPos {
    float h, w, mask;
};
__device__ __forceinline__ TapPos tap_pos(const Geom &g, const float *__restrict__ off, const float *__restrict__ msk,
                                          int b, int grp, int tap, int p, int pad_w_eff) {
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int i = tap / g.kw, j = tap - i * g.kw;
    const int64_t ob = ((int64_t)(b * g.dg + grp) * 2 * g.kk + 2 * tap) * g.HWo + p;
    const float dy = off[ob], dx = off[ob + g.HWo];
    TapPos r;
    r.mask = msk[((int64_t)(b * g.dg + grp) * g.kk + tap) * g.HWo + p];
    r.h = (float)(ho * g.sh - g.ph + i * g.dh) + dy;
    r.w = (float)(wo * g.sw - pad_w_eff + j * g.dw) + dx;
    return r;
}

// forward-kernel variants of tap_pos / sample_cols: the pixel's (ho, wo) are computed once per thread instead of per item,
// and the corner pairs are fetched through a buffer descriptor of the sample (32-bit lane offsets, the channel offset in
// a scalar register, rows outside the image as an out-of-range offset that reads 0) -- the sampling walk is VALU-bound
// (9000 vector instructions per wave), not bandwidth-bound, so address arithmetic and per-load branches are what it pays for.
// roff / rmsk: descriptors of this sample's offset [dg*2*kk, HWo] and mask [dg*kk, HWo] planes; p4 = 4*p
__device__ __forceinline__ TapPos tap_pos_hw(const Geom &g, __amdgpu_buffer_rsrc_t roff, __amdgpu_buffer_rsrc_t rmsk, int grp,
                                             int tap, unsigned p4, int ho, int wo) {
    const int i = tap / g.kw, j = tap - i * g.kw;
    const unsigned plane = (unsigned)g.HWo * 4u;
    const unsigned ob = (unsigned)(grp * 2 * g.kk + 2 * tap) * plane + p4;
    const float dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob, 0, 0));
    const float dx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob + plane, 0, 0));
    TapPos r;
    r.mask = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmsk, (unsigned)(grp * g.kk + tap) * plane + p4, 0, 0));
    r.h = (float)(ho * g.sh - g.ph + i * g.dh) + dy;
    r.w = (float)(wo * g.sw - g.pw + j * g.dw) + dx;
    return r;
}

// the same with the tap given as (row i, column j) of the kernel window: the forward walk steps (tap, i, j) incrementally in
// scalar registers -- `tap / kw` and `idx / cb` per item were vector integer divisions, 40 of an item's 65 set-up instructions
__device__ __forceinline__ TapPos tap_pos_ij(const Geom &g, __amdgpu_buffer_rsrc_t roff, __amdgpu_buffer_rsrc_t rmsk, int grp,
                                             int tap, int i, int j, unsigned p4, int ho, int wo) {
    const unsigned plane = (unsigned)g.HWo * 4u;
    const unsigned ob = (unsigned)(grp * 2 * g.kk + 2 * tap) * plane + p4;
    const float dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob, 0, 0));
    const float dx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob + plane, 0, 0));
    TapPos r;
    r.mask = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmsk, (unsigned)(grp * g.kk + tap) * plane + p4, 0, 0));
    r.h = (float)(ho * g.sh - g.ph + i * g.dh) + dy;
    r.w = (float)(wo * g.sw - g.pw + j * g.dw) + dx;
    return r;
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int NCH, bool SPLIT>
__device__ __forceinline__ void sample_cols_buf(__amdgpu_buffer_rsrc_t rx, unsigned chan_byte, unsigned plane_bytes, const Tap &t,
                                                float mask, float *col, int col_stride, int n) {
    const unsigned o0 = t.q0 >= 0 ? (unsigned)t.q0 * 4u : 0x80000000u, o1 = t.q1 >= 0 ? (unsigned)t.q1 * 4u : 0x80000000u;
    auto one = [&](unsigned cb, float &s) {
        // (channel offset added per lane: as a scalar offset it would need a provably wave-uniform value)
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + cb, 0, 0), bq = __builtin_amdgcn_raw_buffer_load_b64(rx, o1 + cb, 0, 0);
        const float ax = __uint_as_float(a.x), ay = __uint_as_float(a.y), bx = __uint_as_float(bq.x), by = __uint_as_float(bq.y);
        const float v1 = ax * t.mlx + ay * t.mly, v2 = ax * t.mhx + ay * t.mhy;
        const float v3 = bx * t.mlx + by * t.mly, v4 = bx * t.mhx + by * t.mhy;
        s = (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4) * mask;
    };
    if constexpr (NCH > 0) {
        float sv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) one(chan_byte + (unsigned)c * plane_bytes, sv[c]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) col[c * col_stride] = col_word<SPLIT>(sv[c]);
    } else {
        for (int c = 0; c < n; ++c) {
            float sv;
            one(chan_byte + (unsigned)c * plane_bytes, sv);
            col[c * col_stride] = col_word<SPLIT>(sv);
        }
    }
}

// Forward sampling with pre-formed corner coefficients.  The two corners of a row are fetched as ONE 8-byte pair at column
// c = clamp(w0, 0, W-2) (make_tap); which pair element plays the low / high column corner depends only on the edge case:
//   0 <= w0 <= W-2: (x, y) = (low, high)      w0 == -1: x = high, low is outside      w0 == W-1: y = low, high is outside
// so the sample is cA * a.x + cB * a.y + cC * b.x + cD * b.y with the bilinear weights (and the modulation mask) folded
// into cA .. cD once per (pixel, tap).  Rows outside the image get an out-of-range buffer offset (reads 0); a sample outside
// (-1, H) x (-1, W) has all coefficients 0 (dcn_v2_im2col_cuda.cu:180).
struct TapCoef {
    float cA, cB, cC, cD;
    unsigned o0, o1;        // byte offsets of the pair in the low / high row inside a channel plane (0x80000000 = outside)
};
__device__ __forceinline__ TapCoef make_tap_coef(float h, float w, float mask, int H, int W) {
    TapCoef t;
    const bool valid = (h > -1.f) && (w > -1.f) && (h < (float)H) && (w < (float)W);
    h = valid ? h : 0.f;                    // (a NaN / far-off position must give an exact 0, not NaN * 0)
    w = valid ? w : 0.f;
    const float fh = floorf(h), fw = floorf(w);
    const int h0 = (int)fh, w0 = (int)fw;
    const float lh = h - fh, lw = w - fw;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const float m = valid ? mask : 0.f;
    const float w1 = hh * hw * m, w2 = hh * lw * m, w3 = lh * hw * m, w4 = lh * lw * m;
    const bool lo_edge = w0 < 0, hi_edge = w0 > W - 2;
    t.cA = lo_edge ? w2 : (hi_edge ? 0.f : w1);
    t.cB = lo_edge ? 0.f : (hi_edge ? w1 : w2);
    t.cC = lo_edge ? w4 : (hi_edge ? 0.f : w3);
    t.cD = lo_edge ? 0.f : (hi_edge ? w3 : w4);
    const int c = lo_edge ? 0 : (hi_edge ? W - 2 : w0);
    t.o0 = (valid && h0 >= 0) ? (unsigned)(h0 * W + c) * 4u : 0x80000000u;
    t.o1 = (valid && h0 + 1 <= H - 1) ? (unsigned)((h0 + 1) * W + c) * 4u : 0x80000000u;
    return t;
}
template <int NCH, bool SPLIT>
__device__ __forceinline__ void sample_coef(__amdgpu_buffer_rsrc_t rx, unsigned chan_byte, unsigned plane_bytes, const TapCoef &t,
                                            float *col, int col_stride) {
    // the channel offset is wave-uniform (the walk's run bounds are scalar): it rides in the instruction's scalar offset, the
    // lane offset stays the pair's position inside a plane (out of range = 0x80000000: still past the extent with any channel)
    u32x2 a[NCH], b[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(chan_byte + (unsigned)c * plane_bytes));
        a[c] = __builtin_amdgcn_raw_buffer_load_b64(rx, t.o0, so, 0);
        b[c] = __builtin_amdgcn_raw_buffer_load_b64(rx, t.o1, so, 0);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float s = t.cA * __uint_as_float(a[c].x) + t.cB * __uint_as_float(a[c].y) + t.cC * __uint_as_float(b[c].x) +
                        t.cD * __uint_as_float(b[c].y);
        col[c * col_stride] = col_word<SPLIT>(s);
    }
}

// ------------------------------------------------------------------------------------------------ forward
// X3: the 64 x (cb*kk) x 64 product on the bf16 matrix cores in split precision (operands as bf16 hi + lo pairs, three
// MFMAs per product, ~1e-5 of the exact kernel; see conv2d.hip).  LDS then holds the pair words and 16-row k-steps need
// the images padded to KCP rows.  The default (X3 = false) is the exact fp32 kernel the known-answer tests pin.
template <bool X3>
__global__ __launch_bounds__(256, X3 ? 3 : 4) void dcn_fwd_f32(const float *__restrict__ x, const float *__restrict__ wgt,
                                                              const float *__restrict__ bias, const float *__restrict__ off,
                                                              const float *__restrict__ msk, float *__restrict__ out, Geom g) {
    constexpr int ROWS = X3 ? KCP : KC + 2;
    __shared__ float sW[ROWS * WSTR];   // [kl][co]
    __shared__ float sCol[ROWS * NP];   // [kl][px]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform on purpose: run bounds, taps and their (i, j) stay scalar)
    // workgroup ids go round-robin over the 8 XCDs: with tile = id every XCD's 4 MB L2 saw the gathers of ALL samples (33.5 MB
    // of input at B = 8); ids congruent mod 8 now cover one contiguous eighth of the tiles -- one sample per XCD at B = 8, whose
    // 4.2 MB of planes its L2 can keep (same permutation as xcd_tile in conv2d.hip; speed only)
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) {
        const int T = gridDim.x, q = T >> 3;
        tile = (tile & 7) * q + (tile >> 3);
    }
    const int b = tile / g.tiles_per_img;
    const int p0 = (tile - b * g.tiles_per_img) * NP;
    const int co_base = blockIdx.y * 64;
    const int mt = wave >> 1, nt = wave & 1;
    const bool tile_live = co_base + mt * 32 < g.Co;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // Weight slice staging: a thread owns fixed (co, kl) elements; the global offset advances by a
    // constant per chunk, so the loads can be issued before the sampling walk and committed after it.
    constexpr int NWT = (64 * (KC + 2) + 255) / 256;
    const int klmax = ((g.CB * g.kk) + 1) & ~1;          // rows of a full chunk, padded to even
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wgt), 0, (unsigned)g.Co * (unsigned)g.Kd * 4u, 0x00020000);
    unsigned w_off[NWT];
    int w_dst[NWT];
#pragma unroll
    for (int it = 0; it < NWT; ++it) {
        const int i = tid + it * 256;
        const int co = i / klmax, kl = i - co * klmax;
        // rows co >= Co are never stored and rows kl >= KL meet zero columns: no masking beyond the slice size
        w_off[it] = (i < 64 * klmax) ? (unsigned)((co_base + co) * g.Kd + kl) * 4u : 0x80000000u;
        w_dst[it] = kl * WSTR + co;
    }
    const int px = tid & (NP - 1), p = p0 + px;
    const bool p_ok = p < g.HWo;
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const unsigned plane_bytes = (unsigned)(g.H * g.W) * 4u;
    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (int64_t)b * g.C * g.H * g.W), 0, (unsigned)g.C * plane_bytes, 0x00020000);
    const unsigned p4 = (unsigned)p * 4u;
    const __amdgpu_buffer_rsrc_t roff = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(off + (int64_t)b * g.dg * 2 * g.kk * g.HWo), 0, (unsigned)(g.dg * 2 * g.kk) * (unsigned)g.HWo * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmsk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(msk + (int64_t)b * g.dg * g.kk * g.HWo), 0, (unsigned)(g.dg * g.kk) * (unsigned)g.HWo * 4u, 0x00020000);
    if constexpr (X3) {   // weight rows past the staged slice are never written: they must read as zero pairs
        for (int i = klmax * WSTR + tid; i < ROWS * WSTR; i += 256) sW[i] = 0.f;
    }

    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        const Chunk ck = get_chunk(g, chunk);
        const int KLp = (ck.KL + 1) & ~1;
        float rwv[NWT];
        const unsigned wb = (unsigned)(ck.cbase * g.kk) * 4u;
#pragma unroll
        for (int it = 0; it < NWT; ++it)
            rwv[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, w_off[it] + wb, 0, 0));
        __syncthreads();   // previous chunk's MFMA reads are done
        // columns: a thread keeps its pixel; the chunk's cb * kk (channel, tap) samples are cut into four equal runs in
        // tap-major order, one per wave (18 each at cb = 8, kk = 9: three taps touched per wave).  Per (pixel, tap) the four
        // corner COEFFICIENTS are formed once -- bilinear weight x modulation mask x which element of the fetched pair plays
        // which corner -- so that a channel costs two 8-byte gathers and four FMAs (the walk is VALU-bound: it used to form
        // the corners with eight selects per channel and the tap geometry twice per tap; 272 -> see DESIGN section 7).
        {
            const int R = ck.cb * g.kk;
            const int i1 = (R * (wave + 1)) >> 2;
            int idx = (R * wave) >> 2;
            int tap = idx / ck.cb, c0 = idx - tap * ck.cb;            // one scalar division per chunk; stepped from here on
            int ti = tap / g.kw, tj = tap - ti * g.kw;
            TapPos tp = {0.f, 0.f, 0.f};
            if (p_ok && idx < i1) tp = tap_pos_ij(g, roff, rmsk, ck.grp, tap, ti, tj, p4, ho, wo);
            while (idx < i1) {
                const int n = min(ck.cb - c0, i1 - idx);
                int ni = ti, nj = tj + 1;
                if (nj == g.kw) { nj = 0; ++ni; }
                TapPos tp_next = {0.f, 0.f, 0.f};
                if (p_ok && idx + n < i1) tp_next = tap_pos_ij(g, roff, rmsk, ck.grp, tap + 1, ni, nj, p4, ho, wo);
                float *col = sCol + (c0 * g.kk + tap) * NP + px;
                const int cstride = g.kk * NP;
                if (p_ok) {
                    const TapCoef t = make_tap_coef(tp.h, tp.w, tp.mask, g.H, g.W);
                    unsigned cb_ = (unsigned)(ck.cbase + c0) * plane_bytes;
                    int c = 0;
                    for (; c + 4 <= n; c += 4, cb_ += 4u * plane_bytes) sample_coef<4, X3>(rxs, cb_, plane_bytes, t, col + c * cstride, cstride);
                    for (; c + 2 <= n; c += 2, cb_ += 2u * plane_bytes) sample_coef<2, X3>(rxs, cb_, plane_bytes, t, col + c * cstride, cstride);
                    for (; c < n; ++c, cb_ += plane_bytes) sample_coef<1, X3>(rxs, cb_, plane_bytes, t, col + c * cstride, cstride);
                } else {
                    for (int c = 0; c < n; ++c) col[c * cstride] = 0.f;
                }
                tp = tp_next;
                idx += n;
                c0 = 0;                                              // (a run continues at channel 0 of the next tap)
                ++tap; ti = ni; tj = nj;
            }
        }
        if constexpr (X3) {   // rows up to the next multiple of 16 take part in the last k-step
            const int KL16 = (ck.KL + 15) & ~15;
            for (int i = ck.KL * NP + tid; i < KL16 * NP; i += 256) sCol[i] = 0.f;
        } else {
            if (KLp != ck.KL && tid < NP) sCol[ck.KL * NP + tid] = 0.f;
        }
#pragma unroll
        for (int it = 0; it < NWT; ++it)
            if (tid + it * 256 < 64 * klmax) sW[w_dst[it]] = col_word<X3>(rwv[it]);
        __syncthreads();
        if (tile_live) {
            if constexpr (X3) {
                const int KL16 = (ck.KL + 15) & ~15;
                const float *ap = sW + 8 * (lane >> 5) * WSTR + mt * 32 + (lane & 31);
                const float *bp = sCol + 8 * (lane >> 5) * NP + nt * 32 + (lane & 31);
#pragma unroll
                for (int ks = 0; ks < KCP; ks += 16) {
                    if (ks < KL16) {      // wave-uniform
                        unsigned aw[8], bw[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            aw[j] = __float_as_uint(ap[(ks + j) * WSTR]);
                            bw[j] = __float_as_uint(bp[(ks + j) * NP]);
                        }
                        bf16x8 ah, al, bh, bl;
                        peel(aw, ah, al);
                        peel(bw, bh, bl);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    }
                }
            } else {
                const float *ap = sW + (lane >> 5) * WSTR + mt * 32 + (lane & 31);
                const float *bp = sCol + (lane >> 5) * NP + nt * 32 + (lane & 31);
                for (int ks = 0; ks < KLp; ks += 2)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ks * WSTR], bp[ks * NP], acc, 0, 0, 0);
            }
        }
    }
    if (tile_live) {
        const int po = p0 + nt * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < g.Co && po < g.HWo) out[(int64_t)(b * g.Co + co) * g.HWo + po] = acc[r] + bias[co];
        }
    }
}

// (development builds with -DDCN_STAMPS -DEBFI_ABLATE: per-phase s_memtime sums of one workgroup, read by tools/dcnprof.py)
#ifdef DCN_STAMPS
__device__ unsigned long long g_dcn_stamps[16];
#define DSTAMP(k)                                                       \
    do {                                                                \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();     \
        _acc[k] += _t - _t0;                                            \
        _t0 = _t;                                                       \
    } while (0)
#else
#define DSTAMP(k)
#endif

// ------------------------------------------------------------------------------------------------ forward, LDS-staged window
// dcn_fwd_win (round 4): the forward for the model-shaped configuration -- 3x3 taps, stride 1, dilation 1, padding 1, EIGHT
// channels per deformable group -- with the sampling window of a pixel tile staged in LDS.
//
// dcn_fwd_f32 above gathers every corner pair from global memory: 2.4 M divergent 8-byte gather instructions per launch at the
// benchmark size (16-32 cache lines each through the L1 / texture addresser) and ~180 instructions of per-tap set-up for every
// eight samples; the matrix product is not its limit.  Here a workgroup owns a 16 x 8 pixel tile and walks the deformable groups:
//   stage   the group's 8 input channels over the window [y0 - 1 - R, y0 + 9 + R] x [x0 - 1 - R, x0 + 17 + R] (R = 6: 23 x 31
//           positions) into LDS as [position][8 channels] (32 bytes per position; positions outside the image are zeros, so a
//           corner needs no validity test of its own), coalesced row reads, two 16-byte LDS stores per position;
//   sample  one item = (pixel, tap): the tap is wave-uniform; offset / mask loads, ONE set of bilinear coefficients (x mask),
//           four corners x 8 channels = eight 16-byte LDS reads, 32 FMAs, eight column stores.  A sample whose low corner falls
//           outside the window (|offset| > R) takes the global-gather path of dcn_fwd_f32 for that lane (same arithmetic);
//   product as in dcn_fwd_f32: [k][pixel] column image x weight slice, exact fp32 MFMA (or split precision), 64 output channels
//           x 128 pixels per workgroup = two 32 x 32 tiles per wave.
// Sample arithmetic: c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4 with c = bilinear weight * mask (fmaf chain in this order); results
// agree with dcn_fwd_f32 to fp32 rounding, not bit for bit.
constexpr int WPX = 16, WPY = 8, WNP = WPX * WPY;
constexpr int WR = 6;
constexpr int WWD = WPX + 3 + 2 * WR, WHT = WPY + 3 + 2 * WR, WPOS = WWD * WHT;      // 31 x 23 = 713 positions
template <bool X3>
constexpr int dcn_win_lds_bytes() {
    return (KC * (WSTR + WNP) + WPOS * 8) * 4;       // 78.4 KB: two workgroups per CU
}
template <bool X3>
__global__ __launch_bounds__(256, 2) void dcn_fwd_win(const float *__restrict__ x, const float *__restrict__ wgt,
                                                      const float *__restrict__ bias, const float *__restrict__ off,
                                                      const float *__restrict__ msk, float *__restrict__ out, Geom g, int tiles_x,
                                                      int tiles_y) {
    constexpr int ROWS = KC;                  // 72 rows exactly (the split-precision product's last 16-row step: see below)
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float *sW = dsm;                          // [kl][co], kl = tap * 8 + channel
    float *sCol = sW + ROWS * WSTR;           // [kl][pixel]
    // the window as TWO planes [position][4 channels]: neighbouring pixels read neighbouring positions, and 16 lanes x 16 bytes at a
    // 16-byte stride cover all banks exactly twice (the minimum for 256 bytes); with 32 bytes per position lanes i and i + 4
    // met in the same banks -- a 4-way conflict on every corner read
    float *sWin = sCol + ROWS * WNP;          // channels 0..3
    float *sWin1 = sWin + WPOS * 4;           // channels 4..7
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;                    // (one contiguous eighth of the tiles per XCD: see dcn_fwd_f32)
    if ((gridDim.x & 7) == 0) {
        const int T = gridDim.x, q = T >> 3;
        tile = (tile & 7) * q + (tile >> 3);
    }
    const int tpi = tiles_x * tiles_y;
    const int b = tile / tpi, tt = tile - b * tpi;
    const int y0 = (tt / tiles_x) * WPY, x0 = (tt % tiles_x) * WPX;
    const int oy = y0 - 1 - WR, ox = x0 - 1 - WR;
    const int co_base = blockIdx.y * 64;
    const int mt = wave >> 1, nt = wave & 1;                 // 32-row co tile, 64-pixel half (two 32-pixel tiles each)
    const bool tile_live = co_base + mt * 32 < g.Co;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;

    constexpr int KL = 72;                                   // rows of a chunk (9 taps x 8 channels)
    constexpr int NWT = (64 * KL + 255) / 256;               // 18 weight elements per thread
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wgt), 0, (unsigned)g.Co * (unsigned)g.Kd * 4u, 0x00020000);
    unsigned w_off[NWT];
    int w_dst[NWT];
#pragma unroll
    for (int it = 0; it < NWT; ++it) {
        const int i = tid + it * 256;                        // (co, r) with r = channel * 9 + tap: contiguous in memory
        const int co = i / KL, r = i - co * KL;
        w_off[it] = (unsigned)((co_base + co) * g.Kd + r) * 4u;
        w_dst[it] = ((r % 9) * 8 + r / 9) * WSTR + co;       // LDS row = tap * 8 + channel
    }
    const unsigned plane_bytes = (unsigned)(g.H * g.W) * 4u;
    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (int64_t)b * g.C * g.H * g.W), 0, (unsigned)g.C * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t roff = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(off + (int64_t)b * g.dg * 18 * g.HWo), 0, (unsigned)(g.dg * 18) * (unsigned)g.HWo * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmsk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(msk + (int64_t)b * g.dg * 9 * g.HWo), 0, (unsigned)(g.dg * 9) * (unsigned)g.HWo * 4u, 0x00020000);
    // window positions of this thread (fixed over the chunks): 713 = 2 x 256 + 201
    constexpr int NWP = (WPOS + 255) / 256;
    unsigned win_off[NWP];
#pragma unroll
    for (int k = 0; k < NWP; ++k) {
        const int pos = tid + k * 256;
        const int wy = pos / WWD, wx = pos - wy * WWD;
        const int yy = oy + wy, xx = ox + wx;
        win_off[k] = (pos < WPOS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (unsigned)(yy * g.W + xx) * 4u : 0x80000000u;
    }

    // Per chunk the wave handles the wave-items it = wave, wave + 4, .. < 18 (tap = it >> 1, pixel half = it & 1): at most 5.
    // Everything a chunk needs from global memory -- weight slice, window, and the offsets / masks of the wave's items -- is
    // requested one chunk ahead (before the previous chunk's matrix product) and only consumed after the next barrier: the first
    // version loaded offsets inside the item loop and spent 10 us per chunk waiting for them (8 waves per CU hide nothing).
    constexpr int NIT = 5;
    const int py_l = lane >> 4, px_l = lane & 15;
    float rwv[NWT], wv[NWP][8], ody[NIT], odx[NIT], omk[NIT];
    auto prefetch = [&](int grp) {
        const unsigned cbyte = (unsigned)(grp * 8) * plane_bytes;
        const unsigned wb = (unsigned)(grp * 8 * 9) * 4u;
#pragma unroll
        for (int it = 0; it < NWT; ++it) rwv[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, w_off[it] + wb, 0, 0));
#pragma unroll
        for (int k = 0; k < NWP; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                wv[k][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, win_off[k], (unsigned)__builtin_amdgcn_readfirstlane((int)(cbyte + (unsigned)c * plane_bytes)), 0));
        const unsigned plane = (unsigned)g.HWo * 4u;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = wave + 4 * j;
            const int tap = it >> 1, px = (it & 1) * 64 + lane;
            const int yo = y0 + (px >> 4), xo = x0 + (px & 15);
            const bool ok = it < 18 && yo < g.Ho && xo < g.Wo;
            const unsigned p4 = ok ? (unsigned)(yo * g.Wo + xo) * 4u : 0x80000000u;
            const unsigned ob = ok ? (unsigned)(grp * 18 + 2 * tap) * plane + p4 : 0x80000000u;
            ody[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob, 0, 0));
            odx[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ok ? ob + plane : ob, 0, 0));
            omk[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmsk, ok ? (unsigned)(grp * 9 + tap) * plane + p4 : ob, 0, 0));
        }
    };
    (void)py_l; (void)px_l;
#ifdef DCN_STAMPS
    unsigned long long _acc[16] = {0};
    unsigned long long _t0 = __builtin_amdgcn_s_memtime();
#endif
    prefetch(0);
    DSTAMP(0);
    for (int grp = 0; grp < g.dg; ++grp) {
        const unsigned cbyte = (unsigned)(grp * 8) * plane_bytes;
        __syncthreads();                                     // the previous chunk's MFMA reads are done
        DSTAMP(1);
#pragma unroll
        for (int it = 0; it < NWT; ++it) sW[w_dst[it]] = col_word<X3>(rwv[it]);
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
            const int pos = tid + k * 256;
            if (pos < WPOS) {
                *reinterpret_cast<f32x4 *>(sWin + pos * 4) = f32x4{wv[k][0], wv[k][1], wv[k][2], wv[k][3]};
                *reinterpret_cast<f32x4 *>(sWin1 + pos * 4) = f32x4{wv[k][4], wv[k][5], wv[k][6], wv[k][7]};
            }
        }
        float cdy[NIT], cdx[NIT], cmk[NIT];                  // this chunk's offsets leave the prefetch registers
#pragma unroll
        for (int j = 0; j < NIT; ++j) cdy[j] = ody[j], cdx[j] = odx[j], cmk[j] = omk[j];
        DSTAMP(2);
        __syncthreads();                                     // window and weight slice are in LDS
        DSTAMP(3);
        if (grp + 1 < g.dg) prefetch(grp + 1);               // (in flight during the sampling and the matrix product below)
        DSTAMP(5);
        // ---- sampling
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = wave + 4 * j;
            if (it >= 18) break;                             // (wave-uniform)
            const int tap = it >> 1, px = (it & 1) * 64 + lane;
            const int ti = tap / 3, tj = tap - ti * 3;
            const int py = px >> 4, pxx = px & 15;
            const int yo = y0 + py, xo = x0 + pxx;
            const bool p_ok = yo < g.Ho && xo < g.Wo;
            float *col = sCol + (tap * 8) * WNP + px;
            float sv[8];
            // straight-line common path (the compiler overlaps the LDS reads of one item with the arithmetic of its neighbours):
            // a pixel outside the output, an excluded sample and a sample outside the window all get zero coefficients and
            // read position 0; only the last case -- rare -- then takes the divergent global-gather branch
            const float dy = cdy[j], dx = cdx[j], mk = cmk[j];
            float h = (float)(yo - 1 + ti) + dy, w = (float)(xo - 1 + tj) + dx;
            const bool valid = p_ok && (h > -1.f) && (w > -1.f) && (h < (float)g.H) && (w < (float)g.W);   // dcn_v2_im2col_cuda.cu:180
            h = valid ? h : 0.f;
            w = valid ? w : 0.f;
            const float fh = floorf(h), fw = floorf(w);
            const int h0 = (int)fh, w0 = (int)fw;
            const float lh = h - fh, lw = w - fw, hh = 1.f - lh, hw = 1.f - lw;
            const int rh = h0 - oy, rw_ = w0 - ox;
            const bool inwin = valid && rh >= 0 && rh < WHT - 1 && rw_ >= 0 && rw_ < WWD - 1;
            const float m = inwin ? mk : 0.f;
            const float c1 = hh * hw * m, c2 = hh * lw * m, c3 = lh * hw * m, c4 = lh * lw * m;
            {
                const int idx = inwin ? rh * WWD + rw_ : 0;
                const f32x4 *q = reinterpret_cast<const f32x4 *>(sWin) + idx;
                const f32x4 *q1 = reinterpret_cast<const f32x4 *>(sWin1) + idx;
                const f32x4 a0 = q[0], b0 = q[1], c0 = q[WWD], d0 = q[WWD + 1];               // (h0, w0), (h0, w0 + 1), row h0 + 1
                const f32x4 a1 = q1[0], b1 = q1[1], c1v = q1[WWD], d1 = q1[WWD + 1];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sv[c] = fmaf(c4, d0[c], fmaf(c3, c0[c], fmaf(c2, b0[c], c1 * a0[c])));
                    sv[4 + c] = fmaf(c4, d1[c], fmaf(c3, c1v[c], fmaf(c2, b1[c], c1 * a1[c])));
                }
            }
            if (valid && !inwin) {
                // outside the staged window: the corners from global memory (rows / columns outside the image read as 0)
                const float e1 = hh * hw * mk, e2 = hh * lw * mk, e3 = lh * hw * mk, e4 = lh * lw * mk;
                const bool r0 = h0 >= 0, r1 = h0 + 1 <= g.H - 1, k0 = w0 >= 0, k1 = w0 + 1 <= g.W - 1;
                const unsigned o00 = (r0 && k0) ? (unsigned)(h0 * g.W + w0) * 4u : 0x80000000u;
                const unsigned o01 = (r0 && k1) ? (unsigned)(h0 * g.W + w0 + 1) * 4u : 0x80000000u;
                const unsigned o10 = (r1 && k0) ? (unsigned)((h0 + 1) * g.W + w0) * 4u : 0x80000000u;
                const unsigned o11 = (r1 && k1) ? (unsigned)((h0 + 1) * g.W + w0 + 1) * 4u : 0x80000000u;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned cb_ = cbyte + (unsigned)c * plane_bytes;
                    const float v1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o00 == 0x80000000u ? o00 : o00 + cb_, 0, 0));
                    const float v2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o01 == 0x80000000u ? o01 : o01 + cb_, 0, 0));
                    const float v3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o10 == 0x80000000u ? o10 : o10 + cb_, 0, 0));
                    const float v4 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o11 == 0x80000000u ? o11 : o11 + cb_, 0, 0));
                    sv[c] = fmaf(e4, v4, fmaf(e3, v3, fmaf(e2, v2, e1 * v1)));
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) col[c * WNP] = col_word<X3>(sv[c]);
        }
        DSTAMP(4);
        __syncthreads();
        DSTAMP(6);
        if (tile_live) {
            if constexpr (X3) {
                const float *ap = sW + 8 * (lane >> 5) * WSTR + mt * 32 + (lane & 31);
#pragma unroll
                for (int ks = 0; ks < KCP; ks += 16) {
                    // (the last step covers rows 64..79: its upper lane half would read rows 72..79, which do not exist -- zero
                    // pairs instead of 8 more LDS rows per image, which is what lets two workgroups share a CU)
                    const bool live = ks + 8 * (lane >> 5) < KL;
                    unsigned aw[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) aw[j] = live ? __float_as_uint(ap[(ks + j) * WSTR]) : 0u;
                    bf16x8 ah, al;
                    peel(aw, ah, al);
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const float *bp = sCol + 8 * (lane >> 5) * WNP + nt * 64 + t2 * 32 + (lane & 31);
                        unsigned bw[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) bw[j] = live ? __float_as_uint(bp[(ks + j) * WNP]) : 0u;
                        bf16x8 bh, bl;
                        peel(bw, bh, bl);
                        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t2], 0, 0, 0);
                        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t2], 0, 0, 0);
                        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t2], 0, 0, 0);
                    }
                }
            } else {
                const float *ap = sW + (lane >> 5) * WSTR + mt * 32 + (lane & 31);
                const float *bp = sCol + (lane >> 5) * WNP + nt * 64 + (lane & 31);
#pragma unroll 4
                for (int ks = 0; ks < KL; ks += 2) {
                    const float a = ap[ks * WSTR];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[ks * WNP], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[ks * WNP + 32], acc[1], 0, 0, 0);
                }
            }
        }
#ifdef DCN_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]));
#endif
        DSTAMP(7);
    }
#ifdef DCN_STAMPS
    if (blockIdx.x == 37 && blockIdx.y == 0 && threadIdx.x == 0)
        for (int k = 0; k < 16; ++k) g_dcn_stamps[k] = _acc[k];
#endif
    if (tile_live) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int px = nt * 64 + t2 * 32 + (lane & 31);
            const int yo = y0 + (px >> 4), xo = x0 + (px & 15);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.Co && yo < g.Ho && xo < g.Wo) out[(int64_t)(b * g.Co + co) * g.HWo + yo * g.Wo + xo] = acc[t2][r] + bias[co];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: data
// grad_offset, grad_mask (plain stores, one owner per element) and grad_input.
//
// Workgroup = 16x16 output pixels of one sample, walked as 4 bands of 4x16 = 64 pixels.  Per
// deformable-group chunk: colgrad = W^T . grad_out per band (16x16x4 fp32 MFMA, grad_out fragments
// loaded straight into registers), the sampling walk, and the grad_input scatter -- which lands in an
// LDS box covering the tile's footprint +- R pixels, flushed once per chunk with CONTIGUOUS fp32 global atomics.
// The box accumulates in 32-bit FIXED POINT with integer LDS atomics: `ds_add_f32` measured ~25x slower than `ds_add_u32`
// on gfx950 (the kernel ran 1.72 ms with float adds, 0.58 ms with integer ones).  The scale is a power of two chosen per
// (tile, chunk) from an a-priori bound of any cell's sum -- |colgrad| <= max_px ||gout[:,px]|| * max_k ||W[:,k]||
// (Cauchy-Schwarz), times max |mask| of the tile, times the largest sum of bilinear weights x |mask| / max|mask| any
// cell receives (counted exactly, rounded up, by a positions-only pass over the chunk's samples) -- so no sum can
// overflow, one unit is <= 2^-30 of that bound (~1e-8 of the largest possible contribution: on a par with fp32
// rounding) and the box sums do not depend on the order of arrival.  The reference issues 4 global atomics per (pixel, tap, channel)
// (dcn_v2_im2col_cuda.cu:236-250); here that is ~3.5 per (pixel, channel), row-contiguous, which is what
// the memory-side atomic units like.  Samples that fall outside the box (|offset| > R) go to global
// memory directly; shapes whose footprint does not fit the box run with the box disabled.
constexpr int BT = 16;                 // tile edge (output pixels)
constexpr int BOX_CH = 8;              // channels the LDS box holds (one chunk of the model-shaped configs)
constexpr int BOX_CELLS = 1024;        // cells per channel

struct BoxGeom {
    int use, R, BH, BW;                // box = tile footprint in input space +- R, BH x BW cells per channel
};

// two maxima over the 256 threads with one barrier pair; `red` = 8 floats of LDS
__device__ __forceinline__ void wg_max256_2(float &a, float &b, float *red) {
    for (int o = 32; o > 0; o >>= 1) {
        a = fmaxf(a, __shfl_xor(a, o, 64));
        b = fmaxf(b, __shfl_xor(b, o, 64));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = a;
        red[4 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    b = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
}

// wn2[k] = sum_co W[co][k]^2: with the tile's largest |grad_out| column norm it bounds every column-gradient value
// (Cauchy-Schwarz), which fixes the fixed-point scale of the LDS box before anything is scattered into it
// (round 6: 64 columns per workgroup, four thread rows each summing every 4th output channel with eight loads in flight, fixed-order
//  combine through LDS -- the first form walked the Co rows of a column in one thread, 64 dependent loads: 27 us for 147 KB)
__global__ __launch_bounds__(256) void dcn_colnorm2_kernel(const float *__restrict__ wgt, float *__restrict__ wn2, int Co, int Kd) {
    __shared__ float part[4][64];
    const int kk = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kk;
    float s0 = 0.f, s1 = 0.f;
    if (k < Kd) {
        int co = q;
        for (; co + 28 < Co; co += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = wgt[(int64_t)(co + 4 * j) * Kd + k];
#pragma unroll
            for (int j = 0; j < 8; j += 2) { s0 = fmaf(v[j], v[j], s0); s1 = fmaf(v[j + 1], v[j + 1], s1); }
        }
        for (; co < Co; co += 4) {
            const float v = wgt[(int64_t)co * Kd + k];
            s0 = fmaf(v, v, s0);
        }
    }
    part[q][kk] = s0 + s1;
    __syncthreads();
    if (q == 0 && k < Kd) wn2[k] = (part[0][kk] + part[1][kk]) + (part[2][kk] + part[3][kk]);
}

// max over the 256 threads of a workgroup (all threads get it); `red` = 4 floats of LDS, free before and after
__device__ __forceinline__ float wg_max256(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}


__global__ __launch_bounds__(256, 2) void dcn_bwd_data_f32(const float *__restrict__ x, const float *__restrict__ wgt,
                                                        const float *__restrict__ off, const float *__restrict__ msk,
                                                        const float *__restrict__ gout, float *__restrict__ gx,
                                                        float *__restrict__ goff, float *__restrict__ gmsk,
                                                        const float *__restrict__ wn2, Geom g, BoxGeom bx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sWt = smem;                      // weight slice    [co][kl]   64 x S80
    float *sCG = sWt + 64 * S80;            // column gradient [kl][px]   KCP x NP
    float *sBox = sCG + KCP * NP;           // grad_input box  [ch][BH][BW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + BT - 1) / BT, tiles_y = (g.Ho + BT - 1) / BT;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * BT, x0 = tx * BT;
    const int nco = (g.Co + 63) / 64;
    const int pad_w_quirk = g.ph;      // col2im is launched with (pad_h, pad_h): dcn_v2_im2col_cuda.cu:368
    const int by0 = y0 * g.sh - g.ph - bx.R, bx0 = x0 * g.sw - pad_w_quirk - bx.R;   // box origin (input coords)
    const unsigned go_bytes = (unsigned)g.Co * (unsigned)g.HWo * 4u;
    const __amdgpu_buffer_rsrc_t rgo =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gout + (int64_t)b * g.Co * g.HWo), 0, go_bytes, 0x00020000);
    // the sampling walk fetches through descriptors like the forward kernel (32-bit lane offsets, no per-load branches)
    const unsigned plane_bytes = (unsigned)(g.H * g.W) * 4u;
    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (int64_t)b * g.C * g.H * g.W), 0, (unsigned)g.C * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t roff = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(off + (int64_t)b * g.dg * 2 * g.kk * g.HWo), 0, (unsigned)(g.dg * 2 * g.kk) * (unsigned)g.HWo * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmsk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(msk + (int64_t)b * g.dg * g.kk * g.HWo), 0, (unsigned)(g.dg * g.kk) * (unsigned)g.HWo * 4u, 0x00020000);

#ifdef DCN_STAMPS
    unsigned long long _acc[16] = {0};
    unsigned long long _t0 = __builtin_amdgcn_s_memtime();
#endif
    // largest squared column norm of grad_out over the tile's pixels (thread = pixel)
    float gmax2 = 0.f;
    const int t_ho = y0 + (tid >> 4), t_wo = x0 + (tid & 15);
    const bool t_ok = t_ho < g.Ho && t_wo < g.Wo;
    if (bx.use) {
        float s2 = 0.f;
        const unsigned pb = t_ok ? (unsigned)(t_ho * g.Wo + t_wo) * 4u : 0x80000000u;
        for (int co = 0; co < g.Co; ++co) {
            const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rgo, pb + (unsigned)co * (unsigned)g.HWo * 4u, 0, 0));
            s2 += v * v;
        }
        // (fmaxf drops NaN operands: a NaN / Inf column must poison the bound explicitly, or the integer box below would
        // launder it into a finite sum)
        const int bad = __syncthreads_or(!(s2 <= 3.0e38f));
        gmax2 = wg_max256(s2, sCG);
        if (bad) gmax2 = __builtin_nanf("");
    }
    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        const Chunk ck = get_chunk(g, chunk);
        bool box_on = bx.use && ck.cb <= BOX_CH;
        __syncthreads();                   // previous chunk's flush has read the box
        float fx_scale = 1.f, fx_inv = 1.f;
        if (box_on) {
            for (int i = tid; i < ck.cb * bx.BH * bx.BW; i += 256) sBox[i] = 0.f;
            // fixed-point scale of this chunk's box: bound of any cell sum -> the power of two that maps it below 2^30
            float w2 = tid < ck.KL ? wn2[ck.cbase * g.kk + tid] : 0.f;
            for (int k = tid + 256; k < ck.KL; k += 256) w2 = fmaxf(w2, wn2[ck.cbase * g.kk + k]);
            float mm = 0.f;
            const unsigned mb0 = t_ok ? (unsigned)(t_ho * g.Wo + t_wo) * 4u : 0x80000000u;
            for (int tap = 0; tap < g.kk; ++tap)
                mm = fmaxf(mm, fabsf(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                   rmsk, mb0 + (unsigned)(ck.grp * g.kk + tap) * (unsigned)g.HWo * 4u, 0, 0))));
            mm = wg_max256(mm, sCG);
            // the largest sum of (bilinear weight * |mask| / max|mask|) any cell of this chunk's box will receive, in
            // 1/1024 units rounded up: the same walk over (pixel, tap) as the scatter below, positions only
            int *cnt = reinterpret_cast<int *>(sCG);
            const int cells = bx.BH * bx.BW;
            __syncthreads();
            for (int i = tid; i < cells; i += 256) cnt[i] = 0;
            __syncthreads();
            const float mnorm = mm > 0.f ? 1024.f / mm : 0.f;
            {   // thread = pixel (BT*BT == 256 threads); its taps in batches of 9 with all 27 loads of a batch in flight
                const unsigned plane = (unsigned)g.HWo * 4u;
                const unsigned p4 = t_ok ? (unsigned)(t_ho * g.Wo + t_wo) * 4u : 0x80000000u;
                for (int tap0 = 0; tap0 < g.kk; tap0 += 9) {
                    float dy[9], dx[9], mk[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int tap = tap0 + j;
                        const unsigned sel = tap < g.kk ? p4 : 0x80000000u;
                        const unsigned ob = (unsigned)(ck.grp * 2 * g.kk + 2 * tap) * plane + sel;
                        dy[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob, 0, 0));
                        dx[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob + plane, 0, 0));
                        mk[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmsk, (unsigned)(ck.grp * g.kk + tap) * plane + sel, 0, 0));
                    }
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int tap = tap0 + j;
                        if (tap >= g.kk || !t_ok) continue;
                        const int ki = tap / g.kw, kj = tap - ki * g.kw;
                        // grad_input positions (the pad_h-for-pad_w quirk of col2im applies to them)
                        const Tap tq = make_tap((float)(t_ho * g.sh - g.ph + ki * g.dh) + dy[j],
                                                (float)(t_wo * g.sw - pad_w_quirk + kj * g.dw) + dx[j], g.H, g.W);
                        if (!tq.valid) continue;
                        const int r0 = tq.h0 - by0, c0 = tq.w0 - bx0;
                        if (r0 < 0 || r0 + 1 >= bx.BH || c0 < 0 || c0 + 1 >= bx.BW) continue;
                        const float am = fabsf(mk[j]) * mnorm;
                        int *cc = cnt + r0 * bx.BW + c0;
                        atomicAdd(cc, (int)ceilf(tq.w1 * am));
                        atomicAdd(cc + 1, (int)ceilf(tq.w2 * am));
                        atomicAdd(cc + bx.BW, (int)ceilf(tq.w3 * am));
                        atomicAdd(cc + bx.BW + 1, (int)ceilf(tq.w4 * am));
                    }
                }
            }
            __syncthreads();
            int cmax = 0;
            for (int i = tid; i < cells; i += 256) cmax = max(cmax, cnt[i]);
            float load = (float)cmax;                                                  // exact: counts < 2^24
            wg_max256_2(load, w2, sCG + cells);
            load *= 1.f / 1024.f;
            const float bound = sqrtf(gmax2) * sqrtf(w2) * mm * load * 1.001f;
            if (bound > 0.f && bound < 3.0e38f) {
                int e;
                (void)frexpf(bound, &e);                       // bound = m * 2^e, 0.5 <= m < 1
                e = min(max(30 - e, -120), 120);
                fx_scale = ldexpf(1.f, e);
                fx_inv = ldexpf(1.f, -e);
            } else if (!(bound == 0.f)) {
                // NaN / Inf in grad_output, the weights or the masks (or a bound beyond fp32): the integer box would turn them
                // into finite garbage (__float2int_rn maps NaN to 0 and saturates Inf).  This chunk scatters with plain float
                // global atomics instead, which propagate non-finite values like the reference's col2im does (round-2 advisory).
                box_on = false;
            }
        }
        const bool first_sub = (chunk % g.nsub) == 0;   // first channel sub-block of this group
        DSTAMP(0);

        for (int band = 0; band < 4; ++band) {
            // this lane's pixel for the MFMA B operand: n-tile `wave` = row band*4 + wave of the tile
            const int q_ho = y0 + band * 4 + wave, q_wo = x0 + (lane & 15);
            const bool q_ok = q_ho < g.Ho && q_wo < g.Wo;
            f32x4 acc[KCP / 16];
#pragma unroll
            for (int m = 0; m < KCP / 16; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int cob = 0; cob < nco; ++cob) {
                // grad_out fragments straight to registers: B[k = co][j = px]
                float bv[16];
                const unsigned gbase = q_ok ? (unsigned)((cob * 64 + (lane >> 4)) * g.HWo + q_ho * g.Wo + q_wo) * 4u
                                            : 0x80000000u;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)   // rows co >= Co fall outside the descriptor and read 0
                    bv[ks] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rgo, gbase + (unsigned)(4 * ks) * (unsigned)g.HWo * 4u, 0, 0));
                if (band == 0 || nco > 1) {
                    __syncthreads();       // previous readers of sWt are done
                    {   // the slice as ONE batch of loads (a load per loop iteration is a chain of dependent round trips)
                        constexpr int NWS = (64 * KCP + 255) / 256;
                        const __amdgpu_buffer_rsrc_t rwg = __builtin_amdgcn_make_buffer_rsrc(
                            const_cast<float *>(wgt), 0, (unsigned)g.Co * (unsigned)g.Kd * 4u, 0x00020000);
                        float wv[NWS];
#pragma unroll
                        for (int it = 0; it < NWS; ++it) {
                            const int i = tid + it * 256;
                            const int co = i / KCP, kl = i - co * KCP;
                            const bool ok = i < 64 * KCP && kl < ck.KL && cob * 64 + co < g.Co;
                            wv[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                rwg, ok ? (unsigned)((cob * 64 + co) * g.Kd + ck.cbase * g.kk + kl) * 4u : 0x80000000u, 0, 0));
                        }
#pragma unroll
                        for (int it = 0; it < NWS; ++it) {
                            const int i = tid + it * 256;
                            const int co = i / KCP, kl = i - co * KCP;
                            if (i < 64 * KCP) sWt[co * S80 + kl] = wv[it];
                        }
                    }
                    __syncthreads();
                }
                DSTAMP(1);
                // colgrad[kl][px] += sum_co W[co][kl] * gout[co][px]
                const float *ap = sWt + (lane >> 4) * S80 + (lane & 15);
                // all KCP/16 row tiles unconditionally (rows >= KL of the slice are zero): a per-MFMA `if (m < mtiles)`
                // made the compiler shuffle the accumulators between register files around every instruction
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                    for (int m = 0; m < KCP / 16; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks * 4 * S80 + m * 16], bv[ks], acc[m], 0, 0, 0);
                }
            }
            DSTAMP(2);
            __syncthreads();               // previous band's sampling has read sCG
#pragma unroll
            for (int m = 0; m < KCP / 16; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sCG[(m * 16 + (lane >> 4) * 4 + r) * NP + wave * 16 + (lane & 15)] = acc[m][r];
            }
            __syncthreads();
            DSTAMP(3);

            for (int it = tid; it < g.kk * NP; it += 256) {
                DSTAMP(7);
                const int px = it & (NP - 1), tap = it / NP;
                const int ho = y0 + band * 4 + (px >> 4), wo = x0 + (px & 15);
                if (ho >= g.Ho || wo >= g.Wo) continue;
                const int p = ho * g.Wo + wo;
                const TapPos tp = tap_pos_hw(g, roff, rmsk, ck.grp, tap, (unsigned)p * 4u, ho, wo);
                const Tap t = make_tap(tp.h, tp.w, g.H, g.W);
#ifdef DCN_STAMPS
                asm volatile("" ::"v"(t.w1));
#endif
                DSTAMP(8);
                // grad_input positions follow the quirk; identical to `t` whenever pad_h == pad_w
                Tap tq = t;
                if (pad_w_quirk != g.pw) {
                    const TapPos tpq = tap_pos(g, off, msk, b, ck.grp, tap, p, pad_w_quirk);
                    tq = make_tap(tpq.h, tpq.w, g.H, g.W);
                }
                // box cell of the low corner, or -1 when any of the four corners would leave the box
                int cell = -1;
                if (box_on && tq.valid) {
                    const int r0 = tq.h0 - by0, c0 = tq.w0 - bx0;
                    if (r0 >= 0 && r0 + 1 < bx.BH && c0 >= 0 && c0 + 1 < bx.BW) cell = r0 * bx.BW + c0;
                }
                float val_h = 0.f, val_w = 0.f, mval = 0.f;
                float *gplane = gx + (int64_t)(b * g.C + ck.cbase) * g.H * g.W;
                float *bplane = sBox;
                const unsigned o0 = t.q0 >= 0 ? (unsigned)t.q0 * 4u : 0x80000000u, o1 = t.q1 >= 0 ? (unsigned)t.q1 * 4u : 0x80000000u;
                // the corner pairs of all channels of the chunk are fetched as ONE batch (a loop with a fetch per iteration
                // is a chain of dependent memory round trips: the walk was bound by exactly that)
                constexpr int CMAX = 8;                          // channels per batch (a 3x3 chunk has exactly 8)
                const unsigned v0off = t.valid ? o0 : 0x80000000u, v1off = t.valid ? o1 : 0x80000000u;
                for (int cg0 = 0; cg0 < ck.cb; cg0 += CMAX) {
                u32x2 qa[CMAX], qb[CMAX];
                const unsigned chan0 = (unsigned)(ck.cbase + cg0) * plane_bytes;
#pragma unroll
                for (int cj = 0; cj < CMAX; ++cj) {
                    const unsigned cb_ = cg0 + cj < ck.cb ? chan0 + (unsigned)cj * plane_bytes : 0x80000000u;   // out of range: reads 0
                    qa[cj] = __builtin_amdgcn_raw_buffer_load_b64(rxs, v0off + cb_, 0, 0);
                    qb[cj] = __builtin_amdgcn_raw_buffer_load_b64(rxs, v1off + cb_, 0, 0);
                }
#pragma unroll
                for (int cj = 0; cj < CMAX; ++cj, gplane += (int64_t)g.H * g.W, bplane += bx.BH * bx.BW) {
                    const int cl = cg0 + cj;
                    if (cl >= ck.cb) break;
                    const float cg = sCG[(cl * g.kk + tap) * NP + px];
#ifdef DCN_STAMPS
                    if (cj == 0) { asm volatile("" ::"v"(qa[0].x), "v"(qb[0].x)); DSTAMP(9); }
#endif
                    if (t.valid) {
                        const u32x2 pa = qa[cj], pb = qb[cj];
                        const float ax = __uint_as_float(pa.x), ay = __uint_as_float(pa.y);
                        const float bxv = __uint_as_float(pb.x), byv = __uint_as_float(pb.y);
                        const float v1 = ax * t.mlx + ay * t.mly, v2 = ax * t.mhx + ay * t.mhy;
                        const float v3 = bxv * t.mlx + byv * t.mly, v4 = bxv * t.mhx + byv * t.mhy;
                        mval += cg * (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4);
                        // d(sample)/dh and d(sample)/dw  (dmcn_get_coordinate_weight_cuda, :82-123)
                        const float wh = -(1.f - t.lw) * v1 - t.lw * v2 + (1.f - t.lw) * v3 + t.lw * v4;
                        const float ww = -(1.f - t.lh) * v1 + (1.f - t.lh) * v2 - t.lh * v3 + t.lh * v4;
                        val_h += wh * cg * tp.mask;
                        val_w += ww * cg * tp.mask;
                    }
                    if (tq.valid) {
                        const float top = cg * tp.mask;
                        if (cell >= 0) {   // whole 2x2 footprint inside the LDS box; cells outside the image are dropped by the flush
                            int *icell = reinterpret_cast<int *>(bplane) + cell;
                            const float tops = top * fx_scale;
                            atomicAdd(icell, __float2int_rn(tq.w1 * tops));
                            atomicAdd(icell + 1, __float2int_rn(tq.w2 * tops));
                            atomicAdd(icell + bx.BW, __float2int_rn(tq.w3 * tops));
                            atomicAdd(icell + bx.BW + 1, __float2int_rn(tq.w4 * tops));
                        } else {
                            if (tq.o1 >= 0) atomicAdd(gplane + tq.o1, tq.w1 * top);
                            if (tq.o2 >= 0) atomicAdd(gplane + tq.o2, tq.w2 * top);
                            if (tq.o3 >= 0) atomicAdd(gplane + tq.o3, tq.w3 * top);
                            if (tq.o4 >= 0) atomicAdd(gplane + tq.o4, tq.w4 * top);
                        }
                    }
                }
                }   // channel batches
                DSTAMP(10);
                const int64_t ob = ((int64_t)(b * g.dg + ck.grp) * 2 * g.kk + 2 * tap) * g.HWo + p;
                const int64_t mb = ((int64_t)(b * g.dg + ck.grp) * g.kk + tap) * g.HWo + p;
                if (first_sub) {
                    goff[ob] = val_h;
                    goff[ob + g.HWo] = val_w;
                    gmsk[mb] = mval;
                } else {   // same thread owns this element in every sub-block: plain read-modify-write
                    goff[ob] += val_h;
                    goff[ob + g.HWo] += val_w;
                    gmsk[mb] += mval;
                }
                DSTAMP(11);
            }
            DSTAMP(4);
        }
        // ---- flush the box: row-contiguous global atomics, untouched cells skipped
        __syncthreads();
        DSTAMP(5);
        if (box_on) {
            const int cells = bx.BH * bx.BW;
            for (int i = tid; i < ck.cb * cells; i += 256) {
                const int q = reinterpret_cast<const int *>(sBox)[i];
                if (q == 0) continue;
                const float v = (float)q * fx_inv;
                const int cl = i / cells, rem = i - cl * cells;
                const int yy = by0 + rem / bx.BW, xx = bx0 + rem % bx.BW;
                if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W)
                    atomicAdd(gx + ((int64_t)(b * g.C + ck.cbase + cl) * g.H + yy) * g.W + xx, v);
            }
        }
        DSTAMP(6);
    }
#ifdef DCN_STAMPS
    if (blockIdx.x == 37 && threadIdx.x == 0)
        for (int k = 0; k < 16; ++k) g_dcn_stamps[k] = _acc[k];
#endif
}

// ------------------------------------------------------------------------------------------------ backward: weight
// slab[wg][co*Kd + k] = sum over this workgroup's pixel tiles of gout[co][p] * col[k][p];
// slab[wg][Co*Kd + co] = sum of gout[co][p].
__global__ __launch_bounds__(256) void dcn_bwd_weight_f32(const float *__restrict__ x, const float *__restrict__ off,
                                                          const float *__restrict__ msk,
                                                          const float *__restrict__ gout, float *__restrict__ slab,
                                                          Geom g, int total_tiles) {
    __shared__ float sGt[NP * S81];    // grad_out tile transposed [px][co]
    __shared__ float sCt[NP * S81];    // columns transposed       [px][kl]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64;
    const int64_t slab_stride = (int64_t)g.Co * g.Kd + g.Co;
    float *my = slab + (int64_t)blockIdx.x * slab_stride;
    float bpart[16];                   // bias partials: pixel (tid & 63) of channels (tid >> 6) + 4k
#pragma unroll
    for (int k = 0; k < 16; ++k) bpart[k] = 0.f;

    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        const Chunk ck = get_chunk(g, chunk);
        const int ntiles = (ck.KL + 15) >> 4;
        f32x4 acc[KCP / 16];
#pragma unroll
        for (int n = 0; n < KCP / 16; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
            const int b = tile / g.tiles_per_img;
            const int p0 = (tile - b * g.tiles_per_img) * NP;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = tid + 256 * k;
                const int co = i / NP, px = i - co * NP;      // co = (tid >> 6) + 4k, px = tid & 63
                float v = 0.f;
                if (co_base + co < g.Co && p0 + px < g.HWo)
                    v = gout[(int64_t)(b * g.Co + co_base + co) * g.HWo + p0 + px];
                sGt[px * S81 + co] = v;
                if (chunk == 0) bpart[k] += v;
            }
            // zero the kl padding columns [KL, ntiles*16) once per tile (cheap) so MFMA reads zeros
            for (int i = tid; i < NP * 16; i += 256) {
                const int px = i >> 4, kl = (ntiles - 1) * 16 + (i & 15);
                if (kl >= ck.KL) sCt[px * S81 + kl] = 0.f;
            }
            for (int it = tid; it < g.kk * NP; it += 256) {
                const int px = it & (NP - 1), tap = it / NP;
                const int p = p0 + px;
                if (p < g.HWo) {
                    const TapPos tp = tap_pos(g, off, msk, b, ck.grp, tap, p, g.pw);
                    const Tap t = make_tap(tp.h, tp.w, g.H, g.W);
                    // corner pairs of 8 channels per batch through the sample's descriptor (no dependent round trips)
                    const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(x + (int64_t)b * g.C * g.H * g.W), 0, (unsigned)g.C * (unsigned)(g.H * g.W) * 4u, 0x00020000);
                    const unsigned plane_bytes = (unsigned)(g.H * g.W) * 4u;
                    const unsigned o0 = t.q0 >= 0 ? (unsigned)t.q0 * 4u : 0x80000000u, o1 = t.q1 >= 0 ? (unsigned)t.q1 * 4u : 0x80000000u;
                    for (int cg0 = 0; cg0 < ck.cb; cg0 += 8) {
                        u32x2 qa[8], qb[8];
                        const unsigned chan0 = (unsigned)(ck.cbase + cg0) * plane_bytes;
#pragma unroll
                        for (int cj = 0; cj < 8; ++cj) {
                            const unsigned cb_ = cg0 + cj < ck.cb ? chan0 + (unsigned)cj * plane_bytes : 0x80000000u;
                            qa[cj] = __builtin_amdgcn_raw_buffer_load_b64(rxs, o0 + cb_, 0, 0);
                            qb[cj] = __builtin_amdgcn_raw_buffer_load_b64(rxs, o1 + cb_, 0, 0);
                        }
#pragma unroll
                        for (int cj = 0; cj < 8; ++cj) {
                            const int cl = cg0 + cj;
                            if (cl >= ck.cb) break;
                            const float ax = __uint_as_float(qa[cj].x), ay = __uint_as_float(qa[cj].y);
                            const float bxv = __uint_as_float(qb[cj].x), byv = __uint_as_float(qb[cj].y);
                            const float v1 = ax * t.mlx + ay * t.mly, v2 = ax * t.mhx + ay * t.mhy;
                            const float v3 = bxv * t.mlx + byv * t.mly, v4 = bxv * t.mhx + byv * t.mhy;
                            const float val = (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4);
                            sCt[px * S81 + cl * g.kk + tap] = val * tp.mask;
                        }
                    }
                } else {
                    for (int cl = 0; cl < ck.cb; ++cl) sCt[px * S81 + cl * g.kk + tap] = 0.f;
                }
            }
            __syncthreads();
            // acc[co][kl] += sum_px gout[co][px] * col[kl][px];  wave owns co m-tile `wave`
            const float *ap = sGt + (lane >> 4) * S81 + wave * 16 + (lane & 15);
            const float *bp = sCt + (lane >> 4) * S81 + (lane & 15);
            for (int ks = 0; ks < NP; ks += 4) {
                const float av = ap[ks * S81];
                // every column tile unconditionally (tiles >= ntiles are never stored): a per-MFMA condition makes the
                // compiler shuffle the accumulators between register files around every instruction
#pragma unroll
                for (int n = 0; n < KCP / 16; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[ks * S81 + n * 16], acc[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < KCP / 16; ++n)
            if (n < ntiles) {
                const int kl = n * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co_base + wave * 16 + (lane >> 4) * 4 + r;
                    if (kl < ck.KL && co < g.Co) my[(int64_t)co * g.Kd + (int64_t)ck.cbase * g.kk + kl] = acc[n][r];
                }
            }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float v = bpart[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        const int co = co_base + wave + 4 * k;
        if (lane == 0 && co < g.Co) my[(int64_t)g.Co * g.Kd + co] = v;
    }
}

// dcn_bwd_weight_win (round 4): the weight gradient for the model-shaped configuration with the sampling of dcn_fwd_win --
// 16 x 8 pixel tiles, the deformable group's window staged in LDS, items = (pixel, tap) with all 8 channels.  One 512-thread
// workgroup per CU (93 KB of LDS: grad_out tile [co][px], column image [kl][px] with kl = tap * 8 + channel, window planes), so
// nothing but its own prefetch hides a load: the grad_out tile, the window and the wave's offsets / masks of the NEXT
// (group, tile) step are requested before the matrix product of the current one.  Product: acc[co][kl] += sum_px
// gout[co][px] * col[kl][px] with 16x16x4 fp32 MFMA; wave w owns co rows 16 (w & 3) .. +15 and the kl tiles {0,1,2} (w < 4) or
// {3,4} (w >= 4).  Slab layout and reduction as dcn_bwd_weight_f32; 256 workgroups = half its slab traffic.
constexpr int GSTR = WNP + 4;          // row stride of both [row][pixel] images: a 16x16x4 operand read (lane = (row l & 15, pixel l >> 4))
                                       // lands on bank 4 * row + pixel: each bank exactly twice, the minimum for 64 lanes
constexpr int dcn_wwin_lds_bytes() { return ((64 + KC) * GSTR + WPOS * 8) * 4; }
__global__ __launch_bounds__(512) void dcn_bwd_weight_win(const float *__restrict__ x, const float *__restrict__ off,
                                                          const float *__restrict__ msk, const float *__restrict__ gout,
                                                          float *__restrict__ slab, Geom g, int tiles_x, int tiles_y, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float *sG = dsm;                       // [co][pixel]
    float *sCol = sG + 64 * GSTR;          // [kl][pixel]
    float *sWin = sCol + KC * GSTR;        // channels 0..3 per position
    float *sWin1 = sWin + WPOS * 4;        // channels 4..7
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co_base = blockIdx.y * 64;
    const int64_t slab_stride = (int64_t)g.Co * g.Kd + g.Co;
    float *my = slab + (int64_t)blockIdx.x * slab_stride;
    const int tpi = tiles_x * tiles_y;
    const unsigned plane_bytes = (unsigned)(g.H * g.W) * 4u;
    const int mrow = wave & 3, nlo = wave < 4 ? 0 : 3, ncnt = wave < 4 ? 3 : 2;
    float bpart[16];                       // bias partials: channel tid / 128 + 4 k, pixel tid & 127
#pragma unroll
    for (int k = 0; k < 16; ++k) bpart[k] = 0.f;
    constexpr int NIT = 3, NWP = 2;        // wave-items per wave (18 over 8 waves), window positions per thread (713 over 512)
    float pg[16], wv[NWP][8], ody[NIT], odx[NIT], omk[NIT];
    auto prefetch = [&](int grp, int tile) {
        const int b = tile / tpi, tt = tile - b * tpi;
        const int y0 = (tt / tiles_x) * WPY, x0 = (tt % tiles_x) * WPX;
        const int oy = y0 - 1 - WR, ox = x0 - 1 - WR;
        const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (int64_t)b * g.C * g.H * g.W), 0, (unsigned)g.C * plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rgo = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(gout + (int64_t)b * g.Co * g.HWo), 0, (unsigned)g.Co * (unsigned)g.HWo * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t roff = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(off + (int64_t)b * g.dg * 18 * g.HWo), 0, (unsigned)(g.dg * 18) * (unsigned)g.HWo * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rmsk = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(msk + (int64_t)b * g.dg * 9 * g.HWo), 0, (unsigned)(g.dg * 9) * (unsigned)g.HWo * 4u, 0x00020000);
        const unsigned plane = (unsigned)g.HWo * 4u;
        {
            const int px = tid & (WNP - 1);
            const int yo = y0 + (px >> 4), xo = x0 + (px & 15);
            const bool ok = yo < g.Ho && xo < g.Wo;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = co_base + (tid >> 7) + 4 * k;
                const unsigned o = (ok && co < g.Co) ? (unsigned)co * plane + (unsigned)(yo * g.Wo + xo) * 4u : 0x80000000u;
                pg[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rgo, o, 0, 0));
            }
        }
        const unsigned cbyte = (unsigned)(grp * 8) * plane_bytes;
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
            const int pos = tid + k * 512;
            const int wy = pos / WWD, wx = pos - wy * WWD;
            const int yy = oy + wy, xx = ox + wx;
            const unsigned wo = (pos < WPOS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (unsigned)(yy * g.W + xx) * 4u : 0x80000000u;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                wv[k][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, wo, (unsigned)__builtin_amdgcn_readfirstlane((int)(cbyte + (unsigned)c * plane_bytes)), 0));
        }
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = wave + 8 * j;
            const int tap = it >> 1, px = (it & 1) * 64 + lane;
            const int yo = y0 + (px >> 4), xo = x0 + (px & 15);
            const bool ok = it < 18 && yo < g.Ho && xo < g.Wo;
            const unsigned p4 = ok ? (unsigned)(yo * g.Wo + xo) * 4u : 0x80000000u;
            const unsigned ob = ok ? (unsigned)(grp * 18 + 2 * tap) * plane + p4 : 0x80000000u;
            ody[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ob, 0, 0));
            odx[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(roff, ok ? ob + plane : ob, 0, 0));
            omk[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmsk, ok ? (unsigned)(grp * 9 + tap) * plane + p4 : ob, 0, 0));
        }
    };
    const bool any_tile = (int)blockIdx.x < total_tiles;
#ifdef DCN_STAMPS
    unsigned long long _acc[16] = {0};
    unsigned long long _t0 = __builtin_amdgcn_s_memtime();
#endif
    if (any_tile) prefetch(0, blockIdx.x);
    DSTAMP(0);
    for (int grp = 0; grp < g.dg; ++grp) {
        f32x4 acc[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
            const int b = tile / tpi, tt = tile - b * tpi;
            const int y0 = (tt / tiles_x) * WPY, x0 = (tt % tiles_x) * WPX;
            const int oy = y0 - 1 - WR, ox = x0 - 1 - WR;
            __syncthreads();                                 // the previous step's MFMA reads are done
            DSTAMP(1);
            {
                const int px = tid & (WNP - 1);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    sG[((tid >> 7) + 4 * k) * GSTR + px] = pg[k];
                    if (grp == 0) bpart[k] += pg[k];
                }
            }
#pragma unroll
            for (int k = 0; k < NWP; ++k) {
                const int pos = tid + k * 512;
                if (pos < WPOS) {
                    *reinterpret_cast<f32x4 *>(sWin + pos * 4) = f32x4{wv[k][0], wv[k][1], wv[k][2], wv[k][3]};
                    *reinterpret_cast<f32x4 *>(sWin1 + pos * 4) = f32x4{wv[k][4], wv[k][5], wv[k][6], wv[k][7]};
                }
            }
            float cdy[NIT], cdx[NIT], cmk[NIT];
#pragma unroll
            for (int j = 0; j < NIT; ++j) cdy[j] = ody[j], cdx[j] = odx[j], cmk[j] = omk[j];
            DSTAMP(2);
            __syncthreads();                                 // grad_out tile and window are in LDS
            DSTAMP(3);
            {   // the next step's data: next tile of this group, else the first tile of the next group
                int ntile = tile + (int)gridDim.x, ngrp = grp;
                if (ntile >= total_tiles) { ntile = blockIdx.x; ++ngrp; }
                if (ngrp < g.dg) prefetch(ngrp, ntile);
            }
            DSTAMP(4);
            const unsigned cbyte = (unsigned)(grp * 8) * plane_bytes;
            const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(x + (int64_t)b * g.C * g.H * g.W), 0, (unsigned)g.C * plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int it = wave + 8 * j;
                if (it >= 18) break;                         // (wave-uniform)
                const int tap = it >> 1, px = (it & 1) * 64 + lane;
                const int ti = tap / 3, tj = tap - ti * 3;
                const int yo = y0 + (px >> 4), xo = x0 + (px & 15);
                const bool p_ok = yo < g.Ho && xo < g.Wo;
                float *col = sCol + (tap * 8) * GSTR + px;
                float sv[8];
                const float dy = cdy[j], dx = cdx[j], mk = cmk[j];
                float h = (float)(yo - 1 + ti) + dy, w = (float)(xo - 1 + tj) + dx;
                const bool valid = p_ok && (h > -1.f) && (w > -1.f) && (h < (float)g.H) && (w < (float)g.W);
                h = valid ? h : 0.f;
                w = valid ? w : 0.f;
                const float fh = floorf(h), fw = floorf(w);
                const int h0 = (int)fh, w0 = (int)fw;
                const float lh = h - fh, lw = w - fw, hh = 1.f - lh, hw = 1.f - lw;
                const int rh = h0 - oy, rw_ = w0 - ox;
                const bool inwin = valid && rh >= 0 && rh < WHT - 1 && rw_ >= 0 && rw_ < WWD - 1;
                const float m = inwin ? mk : 0.f;
                const float c1 = hh * hw * m, c2 = hh * lw * m, c3 = lh * hw * m, c4 = lh * lw * m;
                {
                    const int idx = inwin ? rh * WWD + rw_ : 0;
                    const f32x4 *q = reinterpret_cast<const f32x4 *>(sWin) + idx;
                    const f32x4 *q1 = reinterpret_cast<const f32x4 *>(sWin1) + idx;
                    const f32x4 a0 = q[0], b0 = q[1], c0 = q[WWD], d0 = q[WWD + 1];
                    const f32x4 a1 = q1[0], b1 = q1[1], c1v = q1[WWD], d1 = q1[WWD + 1];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sv[c] = fmaf(c4, d0[c], fmaf(c3, c0[c], fmaf(c2, b0[c], c1 * a0[c])));
                        sv[4 + c] = fmaf(c4, d1[c], fmaf(c3, c1v[c], fmaf(c2, b1[c], c1 * a1[c])));
                    }
                }
                if (valid && !inwin) {
                    const float e1 = hh * hw * mk, e2 = hh * lw * mk, e3 = lh * hw * mk, e4 = lh * lw * mk;
                    const bool r0 = h0 >= 0, r1 = h0 + 1 <= g.H - 1, k0 = w0 >= 0, k1 = w0 + 1 <= g.W - 1;
                    const unsigned o00 = (r0 && k0) ? (unsigned)(h0 * g.W + w0) * 4u : 0x80000000u;
                    const unsigned o01 = (r0 && k1) ? (unsigned)(h0 * g.W + w0 + 1) * 4u : 0x80000000u;
                    const unsigned o10 = (r1 && k0) ? (unsigned)((h0 + 1) * g.W + w0) * 4u : 0x80000000u;
                    const unsigned o11 = (r1 && k1) ? (unsigned)((h0 + 1) * g.W + w0 + 1) * 4u : 0x80000000u;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const unsigned cb_ = cbyte + (unsigned)c * plane_bytes;
                        const float v1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o00 == 0x80000000u ? o00 : o00 + cb_, 0, 0));
                        const float v2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o01 == 0x80000000u ? o01 : o01 + cb_, 0, 0));
                        const float v3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o10 == 0x80000000u ? o10 : o10 + cb_, 0, 0));
                        const float v4 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxs, o11 == 0x80000000u ? o11 : o11 + cb_, 0, 0));
                        sv[c] = fmaf(e4, v4, fmaf(e3, v3, fmaf(e2, v2, e1 * v1)));
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) col[c * GSTR] = sv[c];
            }
            DSTAMP(5);
            __syncthreads();
            DSTAMP(6);
            // acc[co][kl] += sum_px gout[co][px] * col[kl][px]
            // operands of 8 k-steps (32 pixels) per batch, the next batch's LDS reads issued before this batch's 24 MFMAs: the
            // plain loop waited for its reads in front of every MFMA group (stamps: 10 000 cycles per step for 3 000 of MFMA issue)
            const float *ap = sG + (mrow * 16 + (lane & 15)) * GSTR + (lane >> 4);
            const float *bp0 = sCol + ((nlo + 0) * 16 + (lane & 15)) * GSTR + (lane >> 4);
            const float *bp1 = sCol + ((nlo + 1) * 16 + (lane & 15)) * GSTR + (lane >> 4);
            const int kl2 = (nlo + 2) * 16 + (lane & 15);
            const bool live2 = ncnt > 2 && kl2 < KC;
            const float *bp2 = sCol + (live2 ? kl2 : 0) * GSTR + (lane >> 4);
            const bool live1 = (nlo + 1) * 16 + (lane & 15) < KC;            // (tile 4 = rows 64..79: rows >= 72 do not exist)
            struct Ops {
                float a[8], b0[8], b1[8], b2[8];
            };
            auto load_ops = [&](int k0) {
                Ops o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o.a[j] = ap[k0 + 4 * j];
                    o.b0[j] = bp0[k0 + 4 * j];
                    o.b1[j] = live1 ? bp1[k0 + 4 * j] : 0.f;
                    o.b2[j] = live2 ? bp2[k0 + 4 * j] : 0.f;
                }
                return o;
            };
            auto multiply = [&](const Ops &o) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[j], o.b0[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[j], o.b1[j], acc[1], 0, 0, 0);
                    if (ncnt > 2) acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[j], o.b2[j], acc[2], 0, 0, 0);   // (wave-uniform)
                }
            };
            static_assert(WNP == 128, "four batches of 32 pixels");
            Ops o0 = load_ops(0), o1;
            o1 = load_ops(32);
            __builtin_amdgcn_sched_barrier(0);
            multiply(o0);
            __builtin_amdgcn_sched_barrier(0);
            o0 = load_ops(64);
            __builtin_amdgcn_sched_barrier(0);
            multiply(o1);
            __builtin_amdgcn_sched_barrier(0);
            o1 = load_ops(96);
            __builtin_amdgcn_sched_barrier(0);
            multiply(o0);
            __builtin_amdgcn_sched_barrier(0);
            multiply(o1);
#ifdef DCN_STAMPS
            asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
#endif
            DSTAMP(7);
        }
#ifdef DCN_STAMPS
        if (grp == g.dg - 1 && blockIdx.x == 37 && blockIdx.y == 0 && threadIdx.x == 0)
            for (int k = 0; k < 16; ++k) g_dcn_stamps[8 + (k & 7)] = _acc[k & 7];
#endif
#pragma unroll
        for (int n = 0; n < 3; ++n)
            if (n < ncnt) {
                const int kl = (nlo + n) * 16 + (lane & 15);          // LDS row = tap * 8 + channel
                const int tap = kl >> 3, c = kl & 7;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co_base + mrow * 16 + (lane >> 4) * 4 + r;
                    if (kl < KC && co < g.Co) my[(int64_t)co * g.Kd + (int64_t)(grp * 8 + c) * 9 + tap] = acc[n][r];
                }
            }
    }
    // bias partials: channel (tid >> 7) + 4 k summed over the 128 threads (two waves) that hold its pixels
    __syncthreads();
    float *red = sG;                                          // [16][8 waves]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float v = bpart[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[k * 8 + wave] = v;
    }
    __syncthreads();
    if (tid < 64) {
        const int k = tid >> 2, q = tid & 3;                  // channel q + 4 k: waves 2 q, 2 q + 1
        const int co = co_base + q + 4 * k;
        if (co < g.Co) my[(int64_t)g.Co * g.Kd + co] = red[k * 8 + 2 * q] + red[k * 8 + 2 * q + 1];
    }
}

// 64 consecutive elements per workgroup, 4 thread rows each summing every 4th slab, fixed-order combine
__global__ __launch_bounds__(256) void dcn_bwd_reduce_f32(const float *__restrict__ slab, int nslabs, int64_t n_weight,
                                                          int64_t n_total, float *__restrict__ gw,
                                                          float *__restrict__ gb) {
    __shared__ float part[4][64];
    const int jj = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + jj;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < n_total) {
        const float *p = slab + j;
        int k = kq;
        for (; k + 12 < nslabs; k += 16) {
            s0 += p[(int64_t)k * n_total];
            s1 += p[(int64_t)(k + 4) * n_total];
            s2 += p[(int64_t)(k + 8) * n_total];
            s3 += p[(int64_t)(k + 12) * n_total];
        }
        for (; k < nslabs; k += 4) s0 += p[(int64_t)k * n_total];
    }
    part[kq][jj] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (kq == 0 && j < n_total) {
        const float s = (part[0][jj] + part[1][jj]) + (part[2][jj] + part[3][jj]);
        if (j < n_weight) gw[j] = s;
        else gb[j - n_weight] = s;
    }
}

int make_geom(Geom &g, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
              int dw, int dg) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 ||
        dh <= 0 || dw <= 0 || dg <= 0)
        return fail(EBFI_ERR_ARG, "dcn: non-positive dimension");
    if (C % dg != 0) return fail(EBFI_ERR_ARG, "dcn: channels %d not divisible by deformable_group %d", C, dg);
    if (kh * kw > KC) return fail(EBFI_ERR_UNSUPPORTED, "dcn: kernel %dx%d has more than %d taps", kh, kw, KC);
    if (W < 2) return fail(EBFI_ERR_UNSUPPORTED, "dcn: input width %d < 2 (corner pairs are fetched as 8-byte accesses)", W);
    g = Geom{B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    if (g.Ho <= 0 || g.Wo <= 0) return fail(EBFI_ERR_ARG, "dcn: empty output %dx%d", g.Ho, g.Wo);
    g.HWo = g.Ho * g.Wo;
    g.kk = kh * kw;
    g.cpg = C / dg;
    g.CB = KC / g.kk;
    if (g.CB > g.cpg) g.CB = g.cpg;
    g.nsub = (g.cpg + g.CB - 1) / g.CB;
    g.nchunks = dg * g.nsub;
    g.tiles_per_img = (g.HWo + NP - 1) / NP;
    g.Kd = C * g.kk;
    if ((int64_t)B * C * H * W > (1LL << 31) - 1 || (int64_t)B * g.tiles_per_img > (1LL << 31) - 1)
        return fail(EBFI_ERR_ARG, "dcn: tensor too large for 32-bit plane indexing");
    return EBFI_OK;
}

int weight_grid(const Geom &g) {
    const int64_t tiles = (int64_t)g.B * g.tiles_per_img;
    return (int)(tiles < 512 ? tiles : 512);
}

}  // namespace

extern "C" int ebfi_dcn_forward(const void *input, const void *weight, const void *bias, const void *offset,
                                const void *mask, void *output, int B, int C, int H, int W, int Co, int kh, int kw,
                                int sh, int sw, int ph, int pw, int dh, int dw, int deformable_group, int dtype,
                                void *stream) {
    if (!input || !weight || !bias || !offset || !mask || !output) return fail(EBFI_ERR_ARG, "dcn_forward: null argument");
    if (dtype != EBFI_F32 && dtype != EBFI_F32_BF16X3MMA)
        return fail(EBFI_ERR_UNSUPPORTED, "dcn_forward: dtype %d not implemented (fp32 tensors; exact or bf16x3 matrix operands)", dtype);
    Geom g;
    if (int rc = make_geom(g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the model-shaped configuration (3x3, stride 1, dilation 1, padding 1, 8 channels per deformable group): sampling window in LDS
    if (kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && g.cpg == 8 &&
        !dev_getenv("EBFI_DCN_NO_WINDOW")) {
        const int tiles_x = (g.Wo + WPX - 1) / WPX, tiles_y = (g.Ho + WPY - 1) / WPY;
        const int64_t tiles = (int64_t)B * tiles_x * tiles_y;
        if (tiles <= 2147483647LL) {
            const dim3 wgrid((unsigned)tiles, (unsigned)((Co + 63) / 64));
            const double P = (double)B * g.HWo;
            const bool x3 = dtype == EBFI_F32_BF16X3MMA;
            ProfScope ps(x3 ? "dcn_fwd_bf16x3" : "dcn_fwd_f32", st, 2.0 * P * C * g.kk * (4 + Co),
                         4.0 * (P * (C + 3.0 * deformable_group * g.kk + Co) + (double)Co * C * g.kk));
            if (x3) {
                if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&dcn_fwd_win<true>), dcn_win_lds_bytes<true>())) return rc;
                hipLaunchKernelGGL(dcn_fwd_win<true>, wgrid, dim3(256), dcn_win_lds_bytes<true>(), st, static_cast<const float *>(input),
                                   static_cast<const float *>(weight), static_cast<const float *>(bias), static_cast<const float *>(offset),
                                   static_cast<const float *>(mask), static_cast<float *>(output), g, tiles_x, tiles_y);
            } else {
                if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&dcn_fwd_win<false>), dcn_win_lds_bytes<false>())) return rc;
                hipLaunchKernelGGL(dcn_fwd_win<false>, wgrid, dim3(256), dcn_win_lds_bytes<false>(), st, static_cast<const float *>(input),
                                   static_cast<const float *>(weight), static_cast<const float *>(bias), static_cast<const float *>(offset),
                                   static_cast<const float *>(mask), static_cast<float *>(output), g, tiles_x, tiles_y);
            }
            return check_launch("dcn_fwd_win");
        }
    }
    dim3 grid((unsigned)(B * g.tiles_per_img), (unsigned)((Co + 63) / 64));
    if (dtype == EBFI_F32_BF16X3MMA) {
        const double P = (double)B * g.HWo;
        ProfScope ps("dcn_fwd_bf16x3", st, 2.0 * P * C * g.kk * (4 + Co),
                     4.0 * (P * (C + 3.0 * deformable_group * g.kk + Co) + (double)Co * C * g.kk));
        hipLaunchKernelGGL(dcn_fwd_f32<true>, grid, dim3(256), 0, st, static_cast<const float *>(input),
                           static_cast<const float *>(weight), static_cast<const float *>(bias),
                           static_cast<const float *>(offset), static_cast<const float *>(mask),
                           static_cast<float *>(output), g);
        return check_launch("dcn_fwd_bf16x3");
    }
    {
        const double P = (double)B * g.HWo;
        ProfScope ps("dcn_fwd_f32", st, 2.0 * P * C * g.kk * (4 + Co),
                     4.0 * (P * (C + 3.0 * deformable_group * g.kk + Co) + (double)Co * C * g.kk));
        hipLaunchKernelGGL(dcn_fwd_f32<false>, grid, dim3(256), 0, st, static_cast<const float *>(input),
                           static_cast<const float *>(weight), static_cast<const float *>(bias),
                           static_cast<const float *>(offset), static_cast<const float *>(mask),
                           static_cast<float *>(output), g);
    }
    return check_launch("dcn_fwd_f32");
}

extern "C" size_t ebfi_dcn_backward_workspace(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                                              int ph, int pw, int dh, int dw, int deformable_group, int dtype) {
    (void)dtype;
    Geom g;
    if (make_geom(g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group) != EBFI_OK) return 0;
    return (size_t)weight_grid(g) * ((size_t)g.Co * g.Kd + g.Co) * sizeof(float);
}

extern "C" int ebfi_dcn_backward(const void *input, const void *weight, const void *bias, const void *offset,
                                 const void *mask, const void *grad_output, void *grad_input, void *grad_offset,
                                 void *grad_mask, void *grad_weight, void *grad_bias, int B, int C, int H, int W,
                                 int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                 int deformable_group, void *workspace, size_t workspace_bytes, int dtype,
                                 void *stream) {
    (void)bias;
    if (!input || !weight || !offset || !mask || !grad_output || !grad_input || !grad_offset || !grad_mask ||
        !grad_weight || !grad_bias)
        return fail(EBFI_ERR_ARG, "dcn_backward: null argument");
    if (dtype != EBFI_F32) return fail(EBFI_ERR_UNSUPPORTED, "dcn_backward: dtype %d not implemented (fp32 only)", dtype);
    Geom g;
    if (int rc = make_geom(g, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group)) return rc;
    const size_t need = ebfi_dcn_backward_workspace(B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group, dtype);
    if (!workspace || workspace_bytes < need)
        return fail(EBFI_ERR_WORKSPACE, "dcn_backward: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *x = static_cast<const float *>(input), *wgt = static_cast<const float *>(weight);
    const float *off = static_cast<const float *>(offset), *msk = static_cast<const float *>(mask);
    const float *go = static_cast<const float *>(grad_output);
    if (hipMemsetAsync(grad_input, 0, (size_t)B * C * H * W * sizeof(float), st) != hipSuccess)
        return fail(EBFI_ERR_LAUNCH, "dcn_backward: memset of grad_input failed");
    if (B == 0) {
        (void)hipMemsetAsync(grad_weight, 0, (size_t)Co * g.Kd * sizeof(float), st);
        (void)hipMemsetAsync(grad_bias, 0, (size_t)Co * sizeof(float), st);
        return EBFI_OK;
    }
    {
        // LDS box: tile footprint in input space (stride, dilation) + the largest halo R <= 8 that fits
        BoxGeom bx{0, 0, 0, 0};
        const int span_y = (BT - 1) * sh + (kh - 1) * dh + 2, span_x = (BT - 1) * sw + (kw - 1) * dw + 2;
        for (int R = 8; R >= 0; --R)
            if ((span_y + 2 * R) * (span_x + 2 * R) <= BOX_CELLS) {
                bx = BoxGeom{1, R, span_y + 2 * R, span_x + 2 * R};
                break;
            }
        const size_t lds = (size_t)(64 * S80 + KCP * NP + BOX_CH * BOX_CELLS) * sizeof(float);
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&dcn_bwd_data_f32), (int)lds)) return rc;
        const int64_t tiles = (int64_t)B * ceil_div(g.Ho, BT) * ceil_div(g.Wo, BT);
        // squared column norms of the weight for the box scale; they live at the head of the workspace, which the
        // weight-gradient kernel only starts to fill after the data kernel (same stream)
        float *wn2 = static_cast<float *>(workspace);
        {
            ProfScope ps("dcn_colnorm2", st);
            hipLaunchKernelGGL(dcn_colnorm2_kernel, dim3((unsigned)ceil_div(g.Kd, 64)), dim3(256), 0, st, wgt, wn2, Co, g.Kd);
        }
        ProfScope ps("dcn_bwd_data_f32", st);
        hipLaunchKernelGGL(dcn_bwd_data_f32, dim3((unsigned)tiles), dim3(256), lds, st, x, wgt, off, msk, go,
                           static_cast<float *>(grad_input), static_cast<float *>(grad_offset),
                           static_cast<float *>(grad_mask), wn2, g, bx);
    }
    if (int rc = check_launch("dcn_bwd_data_f32")) return rc;
    int nwg = weight_grid(g);
    const bool win = kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && g.cpg == 8 &&
                     !dev_getenv("EBFI_DCN_NO_WINDOW");
    if (win) {
        // the model-shaped configuration: sampling window in LDS (dcn_bwd_weight_win), one workgroup per CU
        const int tiles_x = (g.Wo + WPX - 1) / WPX, tiles_y = (g.Ho + WPY - 1) / WPY;
        const int64_t tiles = (int64_t)B * tiles_x * tiles_y;
        if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "dcn_backward: too many tiles");
        const int cap = nwg < 256 ? nwg : 256;                // (never more slabs than the workspace of weight_grid(g) holds)
        nwg = (int)(tiles < cap ? tiles : cap);
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&dcn_bwd_weight_win), dcn_wwin_lds_bytes())) return rc;
        ProfScope ps("dcn_bwd_weight_f32", st);
        hipLaunchKernelGGL(dcn_bwd_weight_win, dim3((unsigned)nwg, (unsigned)((Co + 63) / 64)), dim3(512), dcn_wwin_lds_bytes(), st, x, off,
                           msk, go, static_cast<float *>(workspace), g, tiles_x, tiles_y, (int)tiles);
    } else {
        ProfScope ps("dcn_bwd_weight_f32", st);
        hipLaunchKernelGGL(dcn_bwd_weight_f32, dim3((unsigned)nwg, (unsigned)((Co + 63) / 64)), dim3(256), 0, st, x, off,
                           msk, go, static_cast<float *>(workspace), g, B * g.tiles_per_img);
    }
    if (int rc = check_launch("dcn_bwd_weight_f32")) return rc;
    const int64_t n_weight = (int64_t)Co * g.Kd, n_total = n_weight + Co;
    {
        ProfScope ps("dcn_bwd_reduce_f32", st);
        hipLaunchKernelGGL(dcn_bwd_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st,
                           static_cast<const float *>(workspace), nwg, n_weight, n_total,
                           static_cast<float *>(grad_weight), static_cast<float *>(grad_bias));
    }
    return check_launch("dcn_bwd_reduce_f32");
}

#ifdef DCN_STAMPS
extern "C" int ebfi_dcn_debug_stamps(unsigned long long *host_out, int reset) {
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dcn_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dcn_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

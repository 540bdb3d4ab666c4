// 2-D convolution (NCHW, fp32) as LDS-tiled implicit GEMM on the fp32 matrix cores of gfx950.
//
// Replaces what the reference gets from cuDNN through nn.Conv2d inside ConvLayer
// (models/model_misc/submodules.py:159-200): conv + bias + activation, and its two gradients.
// ~90 % of the model's FLOPs are 3x3 stride-1 convs with 64/128 input channels
// (ResidualControl 12 x 5 convs, KernelConv 128->1600, Reconstruction).
//
//   forward / data-gradient  `conv_fwd_f32<KS,S,MT>`:  GEMM  M = out channels, N = pixels,
//       K = in channels x taps.  A 256-thread workgroup owns MT*32 output channels x (4 rows x 64
//       cols) output pixels.  Per chunk of 8 input channels it stages the (4S-S+KS) x (64S-S+KS)
//       input halo tile [ci][y][x] and the weight slice [tap][ci][co] in LDS ONCE; the im2col
//       matrix is never formed: for tap (ky,kx) the B operand of lane `px` is simply
//       tile[ci][S*y+ky][S*px+kx] (consecutive lanes -> consecutive banks), k runs over channel
//       pairs.  v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain) keeps the 1e-3 parity
//       bar with lots of margin; each wave reuses its A/B registers over a 2 x MT tile block.
//       Epilogue fuses bias + LeakyReLU / Sigmoid.  With `transposed` the weight slice is read as
//       W[k][m][KK-1-tap]: the same kernel is the stride-1 data gradient, and `dact` folds the
//       activation derivative (from the saved output) into the staging of grad_output.
//   weight gradient `conv_wgrad_f32<KS,S>`:  GEMM  M = out channels, N = (ci,tap), K = pixels.
//       Each workgroup walks its share of 2x32-pixel tiles, stages grad_out [co][px] (times the
//       activation derivative) and the input halo tile [ci][y][x] with odd strides (conflict-free
//       operand fetch), accumulates a 64 x (64*KK) block in registers and writes ONE partial slab;
//       `conv_wgrad_reduce_f32` sums slabs in fixed order => deterministic grad_weight / grad_bias.
#include <type_traits>

#include "common.hpp"
#include "c16.hpp"

using namespace ebfi;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SIGMOID = 2 };

struct ConvGeom {
    int B, Cin, H, W, Cout, Ho, Wo, pad;
    int groups = 1;    // grouped convolution (split-precision kernels only): Cin = input channels PER GROUP, Cout = all output
                       // channels; output channel co reads input channels [grp*Cin, (grp+1)*Cin), grp = co / (Cout/groups)
    // round 6: where the fp32 output goes (store_out_tile; the matrix-core kernels of launch_fwd_bf16 and conv_fwd_f16_ws):
    //   0  out[B][Cout][Ho][Wo]
    //   1  THROUGH PixelShuffle(2): out[B][Cout/4][2Ho][2Wo], channel 4c + 2py + px of pixel (y, x) -> (c, 2y + py, 2x + px)
    //      (Cout % 4 == 0) -- the reconstruction head's 64 -> 256 convolution (model_singleframe.py:257-260) hands the next
    //      layer its input without the PixelShuffle copy
    //   2  through its INVERSE: out[B][4Cout][Ho/2][Wo/2], channel c of pixel (Y, X) -> (4c + 2(Y&1) + (X&1), Y/2, X/2)
    //      (Ho, Wo even) -- the data gradient of the layer that READ a shuffled tensor hands the gradient back in the layout of
    //      the convolution that wrote it.  Addend / mask / residual operands stay in the kernel's natural layout.
    int store = 0;
};

// Optional extras of the split-precision forward / data-gradient epilogue: out = act(acc + bias + addend) * act'(mask_y)
// (addend, mask_y shaped like the output; NULL = absent).  They let a chain of layers hand PRE-activation gradients from
// layer to layer: the data gradient of layer k adds the other gradient paths into its input and applies the derivative of
// the activation that produced that input, so no layer's weight / data gradient has to re-read a saved activation.
struct EpiExtra {
    const float *addend;
    const float *mask_y;
    int mask_act;
    float mask_slope;
    // round 4: the output additionally (or, with a NULL fp32 output, only) as a scaled fp16 image in the c16 layout (c16.hpp)
    // for the backward kernels that will stage it: out16 [B][Cout/16][Ho][Wo][16], multiplied by slot16[0]; |max| of the
    // unscaled values is recorded into slot16 (one atomic per wave and launch).  Cout must be a multiple of 16.
    _Float16 *out16 = nullptr;
    float *slot16 = nullptr;
    // planar16 != 0: out16 is a PLANAR fp16 tensor [B][Cout][Ho][Wo] instead (the filters of the FAC op, whose kernels read
    // planes: csrc/fac.hip); any Cout
    int planar16 = 0;
    // mask16 (instead of mask_y): the mask tensor as a c16 IMAGE (only the signs are read) -- inside the backward chain of
    // ResidualControl the activations whose derivative a data gradient applies already exist as images (the operands of the
    // weight gradients): half the bytes of the fp32 tensor.  (A positive value below the fp16 denormal range of its scale
    // reads as 0 and takes the slope: 1e-8 of the tensor's |max|.)
    const _Float16 *mask16 = nullptr;
    // round 5 (XM bit 2): the tail of a ResidualControl round folded into the epilogue of its grouped second-layer convolution
    // (model_singleframe.py:130-133): with a = act(acc + bias) the kernel writes out = a * post_scale[b][co] +
    // post_res[b][co % post_resC][pixel] (the exposure- / time-scaled residual: `cat(s_ex * a0 + x, s_t * a1 + x)`), out16 = the
    // image of THAT, and pre16 = the image of `a` itself (scaled by pre_slot[0], |max| recorded there): what the fused backward
    // stage reads for the scale gradients and the LeakyReLU mask.  No fp32 `a`, no separate scale / residual / concat launch.
    const float *post_scale = nullptr;
    const float *post_res = nullptr;
    int post_resC = 0;
    _Float16 *pre16 = nullptr;
    float *pre_slot = nullptr;
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}
// derivative of the activation expressed through its OUTPUT y (what autograd saved)
__device__ __forceinline__ float act_grad(float y, int act, float slope) {
    if (act == ACT_LEAKY) return y > 0.f ? 1.f : slope;
    if (act == ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// Activation derivative with the activation kind as a COMPILE-TIME constant: a runtime `if (dact)` around the
// extra load makes hipcc branch around every unrolled load and wait for each one separately (dependent memory
// round trips instead of one batch) -- that alone cost the weight-gradient kernel more than half of its time.
template <int DACT>
__device__ __forceinline__ float act_grad_c(float y, float slope) {
    if constexpr (DACT == ACT_LEAKY) return y > 0.f ? 1.f : slope;
    else if constexpr (DACT == ACT_SIGMOID) return y * (1.f - y);
    else return 1.f;
}

constexpr int TY = 4, TX = 64;

// Staging loads go through buffer descriptors: one 32-bit offset VGPR per load instead of a 64-bit
// address pair, and the hardware range check returns 0 for out-of-range offsets (zero padding and
// channel tails for free).  SENT is added to running chunk offsets and must stay out of range without
// wrapping: the launchers require every per-sample tensor to be smaller than 2 GiB.
constexpr unsigned SENT = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, bytes, 0x00020000);
}
// s_waitcnt vmcnt(N) with everything else left alone (gfx9 encoding: vmcnt = bits 15:14 | 3:0, expcnt 6:4, lgkmcnt 11:8).  Given
// explicitly where a pipeline of register stages knows exactly which loads it needs: the wait-count pass honours an existing
// s_waitcnt, so its own, more conservative guesses (a full drain around guarded stores or a re-used destination register)
// become redundant.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(((N >> 4) << 14) | (0x7 << 4) | (0xf << 8) | (N & 15));
}
// A staging load's offset as a BRANCH-FREE select.  Written as `ok ? off : SENT`, hipcc may sink the offset arithmetic into a
// branch and duplicate the load into both arms; the arm of the masked lanes then opens with s_waitcnt vmcnt(0) (it re-uses the
// first destination register as the address temporary while the other arm's load is pending): every prefetch of a ragged
// producer wave drained ALL the loads in flight (read off the ISA, round 5).  The empty asm pins `off` as computed for every lane.
__device__ __forceinline__ unsigned sel_off(bool ok, unsigned off) {
    asm volatile("" : "+v"(off));
    return ok ? off : 0x80000000u;
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 0);   // out-of-range offsets: store dropped
}

// Epilogue shared by the forward / data-gradient kernels: acc[m][n] is the 32x32 block (co tile m, x half n) of output row
// `yo`; bias + activation, then NCHW stores through a buffer descriptor (lanes = consecutive x).  Everything conditional is
// resolved once per tile: invalid pixels get an out-of-range offset (the hardware drops the store), channels >= Cout fall
// past the descriptor's extent, the bias is fetched as one batch (a missing bias reads zeros from an empty descriptor) and the
// activation is selected outside the element loop.  The previous per-element `if` chain compiled to ~70 instructions and a
// dependent bias load per element (4500 instructions per thread), several microseconds per workgroup.
// XM: which epilogue extras are compiled in (so that a launch pays registers only for what it can use): bit 0 = addend / mask,
// bit 1 = the fp16 side output (image or planes), bit 2 = the ResidualControl tail, bit 3 = the shuffled output layouts of
// ConvGeom::store (a kernel built without it ignores g.store: its launcher refuses).  0 = the plain epilogue; `true` from older call sites means bit 0.
template <int MT, int XM = 0>
__device__ __forceinline__ void store_out_tile(float *__restrict__ out, const float *__restrict__ bias, const f32x16 (&acc)[MT][2],
                                               const ConvGeom &g, int b, int co_base, int yo, int x0, int lane, int act,
                                               float slope, EpiExtra ex = EpiExtra{nullptr, nullptr, 0, 0.f}, float oscale = 1.f,
                                               float *amax16 = nullptr, float *amax_pre = nullptr) {
    // oscale: the accumulators are oscale-times too small (operands were scaled by powers of two: conv2d_f16.inc.hpp); 1 otherwise
    // amax16 (with ex.out16): running |max| of what this wave wrote into the fp16 image (recorded by the caller at the end)
    const int HWo = g.Ho * g.Wo;
    const unsigned plane = (unsigned)HWo * 4u;
    constexpr bool EXTRA = (XM & 7) != 0, XAM = (XM & 1) != 0, X16 = (XM & 2) != 0, XRC = (XM & 4) != 0;
    constexpr bool XST = (XM & 8) != 0;                             // the shuffled output layouts (ConvGeom::store) are compiled in
    if constexpr (EXTRA) {
        // Everything below is derived from `lane` and loop-invariant over the caller's tile walk: left alone, the compiler
        // computes the per-lane offsets and descriptors once at kernel start, keeps them live across the MFMA main loop (where
        // the 168-register forward kernel has none to spare) and spills 13-19 of them to scratch.  Laundering the lane id
        // through an opaque asm pins the address arithmetic inside the epilogue: no scratch (and a kernel that needs scratch
        // growing mid-stream is what broke the first image-writing builds on non-default streams and in graph replays).
        asm volatile("" : "+v"(lane));
    }
    const bool has32 = out != nullptr;
    const float *anyp = bias ? bias : (out ? out : reinterpret_cast<const float *>(ex.slot16));     // base of the empty descriptors
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(has32 ? out + (int64_t)b * g.Cout * HWo : anyp, has32 ? (unsigned)g.Cout * plane : 0u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(bias ? bias : anyp, bias ? (unsigned)g.Cout * 4u : 0u);
    // the extras: descriptors over the same sample of tensors shaped like the output (absent: empty descriptor, reads 0)
    const bool has_a = XAM && ex.addend != nullptr, has_m = XAM && ex.mask_y != nullptr, has16 = X16 && ex.out16 != nullptr;
    const bool has_m16 = XAM && ex.mask16 != nullptr;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(has_a ? ex.addend + (int64_t)b * g.Cout * HWo : anyp, has_a ? (unsigned)g.Cout * plane : 0u);
    const __amdgpu_buffer_rsrc_t rm = make_rsrc(has_m ? ex.mask_y + (int64_t)b * g.Cout * HWo : anyp, has_m ? (unsigned)g.Cout * plane : 0u);
    const int cb16 = g.Cout >> 4;                                  // 16-channel blocks of the fp16 image
    const bool planar = X16 && ex.planar16 != 0;
    const __amdgpu_buffer_rsrc_t rm16 = __builtin_amdgcn_make_buffer_rsrc(
        has_m16 ? const_cast<_Float16 *>(ex.mask16) + (int64_t)b * cb16 * HWo * 16 : const_cast<_Float16 *>(reinterpret_cast<const _Float16 *>(anyp)), 0,
        has_m16 ? (unsigned)cb16 * (unsigned)HWo * 32u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r16 = __builtin_amdgcn_make_buffer_rsrc(
        has16 ? ex.out16 + (planar ? (int64_t)b * g.Cout * HWo : (int64_t)b * cb16 * HWo * 16)
              : const_cast<_Float16 *>(reinterpret_cast<const _Float16 *>(anyp)), 0,
        has16 ? (planar ? (unsigned)g.Cout * (unsigned)HWo * 2u : (unsigned)cb16 * (unsigned)HWo * 32u) : 0u, 0x00020000);
    const float s16 = has16 ? ex.slot16[0] : 1.f;
    // XRC: post-activation scale + residual, image of the activation output (see EpiExtra)
    const bool has_rc = XRC && ex.post_scale != nullptr;
    const __amdgpu_buffer_rsrc_t rps = make_rsrc(has_rc ? ex.post_scale + (int64_t)b * g.Cout : anyp, has_rc ? (unsigned)g.Cout * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(has_rc ? ex.post_res + (int64_t)b * ex.post_resC * HWo : anyp,
                                                 has_rc ? (unsigned)ex.post_resC * plane : 0u);
    const bool has_p16 = has_rc && ex.pre16 != nullptr;        // (inference: scale + residual only, no image of `a`)
    const __amdgpu_buffer_rsrc_t rp16 = __builtin_amdgcn_make_buffer_rsrc(
        has_p16 ? ex.pre16 + (int64_t)b * cb16 * HWo * 16 : const_cast<_Float16 *>(reinterpret_cast<const _Float16 *>(anyp)), 0,
        has_p16 ? (unsigned)cb16 * (unsigned)HWo * 32u : 0u, 0x00020000);
    const float sp16 = has_p16 ? ex.pre_slot[0] : 1.f;
    const int res_shift = has_rc ? co_base - co_base % ex.post_resC : 0;       // (channel blocks never straddle post_resC: both multiples of 64)
    float am = 0.f, amp = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    // ---- EXTRA path: the tile is walked in blocks of 8 values = (row tile m, x half n, 16-channel block j); the per-pixel
    // operands of a block (addend, mask, residual) are REQUESTED ONE BLOCK AHEAD into a second small register set.  Before
    // round 5 every block loaded, waited and stored in turn -- sixteen dependent memory round trips per output row and wave
    // (s_waitcnt vmcnt(7) .. vmcnt(0) sixteen times in the ISA): ~19 us of the 55 us of an image-writing data gradient.
    struct BlockIn {
        float add[8];                      // addend
        float my[8];                       // mask_y (fp32 form)
        unsigned m16[4];                   // mask16: the two 8-byte halves of this lane's piece pair
        float res[8];                      // post_res (XRC)
    };
    // g.store != 0 (ConvGeom): byte offsets of this lane's first channel (co_base + 32m + 4h) in the shuffled / unshuffled output
    auto shuffled_base = [&](int m) { return (unsigned)(((co_base + m * 32) >> 2) + h) * 4u * plane; };    // + pixel (2yo, 2xo) of the 2Wo-wide plane
    auto unshuffled_base = [&](int m, bool px_ok, int xo) {
        return px_ok ? (unsigned)(4 * (co_base + m * 32 + 4 * h) + 2 * (yo & 1) + (xo & 1)) * (plane >> 2) +
                           (unsigned)((yo >> 1) * (g.Wo >> 1) + (xo >> 1)) * 4u
                     : SENT;
    };
    auto blk_geom = [&](int m, int n, int j, bool &px_ok, unsigned &base, int &cblk, int &xo) {
        xo = x0 + n * 32 + l31;
        px_ok = yo < g.Ho && xo < g.Wo;
        base = (px_ok && co_base + m * 32 + 4 * h < g.Cout) ? (unsigned)(co_base + m * 32 + 4 * h) * plane + (unsigned)(yo * g.Wo + xo) * 4u : SENT;
        cblk = ((co_base + m * 32) >> 4) + j;
    };
    auto row_off = [](int j, int i) { return (unsigned)((i & 3) + 8 * (2 * j + (i >> 2))); };   // row of value i within the tile, minus 4h
    [[maybe_unused]] auto issue = [&](int m, int n, int j, BlockIn &in) {
        bool px_ok; unsigned base; int cblk, xo;
        blk_geom(m, n, j, px_ok, base, cblk, xo);
        if constexpr (XAM) {
            if (has_a) {
#pragma unroll
                for (int i = 0; i < 8; ++i) in.add[i] = buf_ld(ra, base + row_off(j, i) * plane);
            }
            if (has_m) {
#pragma unroll
                for (int i = 0; i < 8; ++i) in.my[i] = buf_ld(rm, base + row_off(j, i) * plane);
            }
            if (has_m16) {                 // this lane's 4 channels of either half-piece
                typedef unsigned u32x2_m __attribute__((ext_vector_type(2)));
                const unsigned om = (px_ok && cblk < cb16) ? ((unsigned)cblk * (unsigned)HWo * 2u + (unsigned)(yo * 2 * g.Wo + xo)) * 16u + 8u * (unsigned)h : SENT;
                const u32x2_m lo = __builtin_amdgcn_raw_buffer_load_b64(rm16, om, 0, 0);
                const u32x2_m hi = __builtin_amdgcn_raw_buffer_load_b64(rm16, om + (unsigned)g.Wo * 16u, 0, 0);
                in.m16[0] = lo.x; in.m16[1] = lo.y; in.m16[2] = hi.x; in.m16[3] = hi.y;
            }
        }
        if constexpr (XRC) {
            if (has_rc) {
                const unsigned rbase = base == SENT ? SENT : base - (unsigned)res_shift * plane;
#pragma unroll
                for (int i = 0; i < 8; ++i) in.res[i] = buf_ld(rrs, rbase + row_off(j, i) * plane);
            }
        }
    };
    [[maybe_unused]] auto finish = [&](int m, int n, int j, const BlockIn &in, auto actf) {
        bool px_ok; unsigned base; int cblk, xo;
        blk_geom(m, n, j, px_ok, base, cblk, xo);
        float u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = acc[m][n][8 * j + i] * oscale;
        [[maybe_unused]] float psc[8];     // XRC: the per-channel scales, requested together with the bias (one round trip, not two)
        if constexpr (XRC) {
            if (has_rc) {
#pragma unroll
                for (int i = 0; i < 8; ++i) psc[i] = buf_ld(rps, (unsigned)(co_base + m * 32 + 4 * h) * 4u + row_off(j, i) * 4u);
            }
        }
        if (bias != nullptr) {             // (per-channel: the 32 pixel lanes of a half read one address -- served from L1)
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = buf_ld(rb, (unsigned)(co_base + m * 32 + 4 * h) * 4u + row_off(j, i) * 4u);
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] += t[i];
        }
        if constexpr (XAM) {
            if (has_a) {
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] += in.add[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = actf(u[i]);
        if constexpr (XAM) {
            if (has_m) {                   // LeakyReLU masks only (the launchers refuse anything else with extras)
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] *= in.my[i] > 0.f ? 1.f : ex.mask_slope;
            }
            if (has_m16) {                 // the same signs from the image
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    // positive = sign bit clear and not zero (fp16 bits; NaN cannot occur in a saturated image)
                    const unsigned hbits = (in.m16[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                    u[i] *= (hbits != 0u && hbits < 0x8000u) ? 1.f : ex.mask_slope;
                }
            }
        }
        if constexpr (XRC) {
            if (has_rc) {
                if (has_p16) {
                    const unsigned o16 = (px_ok && cblk < cb16) ? ((unsigned)cblk * (unsigned)HWo * 2u + (unsigned)((yo * 2 + h) * g.Wo + xo)) * 16u : SENT;
#pragma unroll
                    for (int i = 0; i < 8; ++i) amp = amax_acc(amp, u[i]);
                    const u32x4_c16 q = c16_gather_halves(pack_f16(u[0] * sp16, u[1] * sp16), pack_f16(u[2] * sp16, u[3] * sp16),
                                                          pack_f16(u[4] * sp16, u[5] * sp16), pack_f16(u[6] * sp16, u[7] * sp16));
                    __builtin_amdgcn_raw_buffer_store_b128(q, rp16, o16, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] = u[i] * psc[i] + in.res[i];
            }
        }
        if (has32) {
            if (XST && g.store == 1) {     // through the pixel shuffle: this lane's 4 consecutive channels are one 2x2 output block
                const unsigned b1 = px_ok ? shuffled_base(m) + (unsigned)(2 * yo * 2 * g.Wo + 2 * xo) * 4u : SENT;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        typedef float f32x2_st __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x2_st __attribute__((ext_vector_type(2)));
                        const f32x2_st v{u[4 * q + 2 * py], u[4 * q + 2 * py + 1]};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_st, v), ro,
                                                              b1 + (unsigned)(2 * (2 * j + q)) * 4u * plane + (unsigned)(py * 2 * g.Wo) * 4u, 0, 0);
                    }
                }
            } else {
                const unsigned sb = (XST && g.store == 2) ? unshuffled_base(m, px_ok, xo) : base;
#pragma unroll
                for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[i]), ro, sb + row_off(j, i) * plane, 0, 0);
            }
        }
        if constexpr (X16) {
            if (has16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) am = amax_acc(am, u[i]);
                if (planar) {
                    // planes: one 2-byte store per value, lanes = consecutive pixels (64-byte runs per channel)
                    const unsigned pb = (px_ok && co_base + m * 32 + 4 * h < g.Cout)
                                            ? (unsigned)(co_base + m * 32 + 4 * h) * (unsigned)HWo * 2u + (unsigned)(yo * g.Wo + xo) * 2u : SENT;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const _Float16 hv = (_Float16)(u[i] * s16);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), r16, pb + row_off(j, i) * (unsigned)HWo * 2u, 0, 0);
                    }
                } else {
                    const u32x4_c16 q = c16_gather_halves(pack_f16(u[0] * s16, u[1] * s16), pack_f16(u[2] * s16, u[3] * s16),
                                                          pack_f16(u[4] * s16, u[5] * s16), pack_f16(u[6] * s16, u[7] * s16));
                    // lower lanes: channels 0..7 of 32 consecutive pixels = one 512-byte run, upper lanes: channels 8..15
                    const unsigned o16 = (px_ok && cblk < cb16)
                                             ? ((unsigned)cblk * (unsigned)HWo * 2u + (unsigned)((yo * 2 + h) * g.Wo + xo)) * 16u : SENT;
                    __builtin_amdgcn_raw_buffer_store_b128(q, r16, o16, 0, 0);
                }
            }
        }
    };
    auto emit = [&](auto actf) {
        if constexpr (EXTRA) {
            constexpr int NBLK = MT * 4;   // block id = (m * 2 + n) * 2 + j
            BlockIn ia, ib;
            issue(0, 0, 0, ia);
#pragma unroll
            for (int k = 0; k < NBLK; ++k) {
                const int m = k >> 2, n = (k >> 1) & 1, j = k & 1;
                // the NEXT block's operands are requested before this block waits for its own ...
                if (k + 1 < NBLK) issue((k + 1) >> 2, ((k + 1) >> 1) & 1, (k + 1) & 1, (k & 1) ? ia : ib);
                finish(m, n, j, (k & 1) ? ib : ia, actf);
                // ... and nothing further ahead: eight unrolled copies of the body with all their loads batched up front cost 13-19
                // spilled registers in the 168-register forward kernel
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float bvm[16];             // (per 32-row tile: both tiles' biases up front cost 16 more live registers)
#pragma unroll
                for (int r = 0; r < 16; ++r) bvm[r] = buf_ld(rb, (unsigned)(co_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 4u);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int xo = x0 + n * 32 + l31;
                    const bool px_ok = yo < g.Ho && xo < g.Wo;
                    if (XST && g.store == 1) {    // through the pixel shuffle (see `finish`): 8 eight-byte stores instead of 16 dwords
                        const unsigned b1 = px_ok ? shuffled_base(m) + (unsigned)(2 * yo * 2 * g.Wo + 2 * xo) * 4u : SENT;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
#pragma unroll
                            for (int py = 0; py < 2; ++py) {
                                typedef float f32x2_st __attribute__((ext_vector_type(2)));
                                typedef unsigned u32x2_st __attribute__((ext_vector_type(2)));
                                const int r = 4 * k + 2 * py;
                                const f32x2_st v{actf(acc[m][n][r] * oscale + bvm[r]), actf(acc[m][n][r + 1] * oscale + bvm[r + 1])};
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_st, v), ro,
                                                                      b1 + (unsigned)(2 * k) * 4u * plane + (unsigned)(py * 2 * g.Wo) * 4u, 0, 0);
                            }
                        }
                        continue;
                    }
                    const unsigned base = (XST && g.store == 2) ? unshuffled_base(m, px_ok, xo)
                                          : (px_ok && co_base + m * 32 + 4 * h < g.Cout)
                                              ? (unsigned)(co_base + m * 32 + 4 * h) * plane + (unsigned)(yo * g.Wo + xo) * 4u : SENT;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(actf(acc[m][n][r] * oscale + bvm[r])), ro,
                                                              base + (unsigned)((r & 3) + 8 * (r >> 2)) * plane, 0, 0);
                }
            }
        }
    };
    if constexpr (EXTRA) {
        // ONE instance of the (large, fully unrolled) body: no activation = LeakyReLU with slope 1; the sigmoid never meets
        // the extras (launchers check).  Three instances per call and two calls per tile made the kernel 26 000 instructions
        // long -- the epilogue ran out of the instruction cache and cost more than the bytes it saved.
        const float sl = act == ACT_LEAKY ? slope : 1.f;
        emit([sl](float v) { return v > 0.f ? v : v * sl; });
    } else {
        if (act == ACT_LEAKY) emit([slope](float v) { return v > 0.f ? v : v * slope; });
        else if (act == ACT_SIGMOID) emit([](float v) { return 1.f / (1.f + __expf(-v)); });
        else emit([](float v) { return v; });
    }
    if (amax16) *amax16 = amax_acc(*amax16, am);
    if constexpr (XRC) {
        if (amax_pre) *amax_pre = amax_acc(*amax_pre, amp);
    }
}

// ------------------------------------------------------------------------------------------------
// KernelConv -> FAC fused epilogue (reference model_singleframe.py:159-163 + KernelConv2D.py:82-87 +
// KernelConv2D_kernel.cu:25-53): the conv's output rows are the per-pixel 5x5 FILTERS of the filter-adaptive convolution
// that follows it.  With the weight rows laid out one FAC channel per 32-row matrix tile (25 taps + 7 zero rows, see
// ebfi_amd.weightbank kind "facrows"), a consumer wave holds, per output pixel (lane & 31), all 25 filters of channel
// `m` split over its two lane halves (row = (r&3) + 8(r>>2) + 4h): it applies LeakyReLU(acc + bias), multiplies each by
// the replicate-clamped neighbour of the feature map `ev` (fp32, fetched through the L1: the 12 x 68 window of a tile is
// re-read 25 times), adds the two halves by one cross-lane exchange and stores ONE value.  The [B,1600,h,w] filter tensor
// (839 MB at B=8 256x256, 11.8 GB at B=8 720x1280) is never written.  Inference only: the training step needs the filters
// again in the backward pass (recomputing them costs a second 128 -> 1600 convolution, 1.0 ms against the 0.3 ms of the
// store + the FAC launch), so training keeps the unfused pair.
struct FacEpi {
    const float *ev;       // [B, C, H, W]: the feature map the filters are applied to (unpadded; clamped here = ReplicationPad2d(2))
    int C;                 // FAC channels (= weight rows / 32)
};

template <int MT>
__device__ __forceinline__ void fac_epilogue_tile(float *__restrict__ out, const float *__restrict__ bias, const f32x16 (&acc)[MT][2],
                                                  const ConvGeom &g, const FacEpi &fac, int b, int co_base, int yo, int x0, int lane,
                                                  float slope, float oscale = 1.f) {
    // oscale: the accumulators are oscale-times too small (fp16 operands scaled by powers of two, conv2d_f16.inc.hpp); 1 otherwise
    constexpr int K = 5, R = 2;
    const int HW = g.Ho * g.Wo;                       // same-padded 3x3: output size = input size = the size of ev
    const int h = lane >> 5, l31 = lane & 31;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(bias, (unsigned)g.Cout * 4u);
    float part[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (co_base >> 5) + m;
        const bool c_ok = c < fac.C;
        const __amdgpu_buffer_rsrc_t rev = make_rsrc(fac.ev + ((int64_t)b * fac.C + (c_ok ? c : 0)) * HW, c_ok ? (unsigned)HW * 4u : 0u);
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = buf_ld(rb, (unsigned)(co_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 4u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int xo = x0 + n * 32 + l31;
            unsigned rowoff[K], coloff[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                rowoff[i] = (unsigned)(min(max(yo + i - R, 0), g.Ho - 1) * g.Wo) * 4u;
                coloff[i] = (unsigned)min(max(xo + i - R, 0), g.Wo - 1) * 4u;
            }
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // tap of register r: t0 for the lower lane half, t0 + 4 for the upper one; rows >= 25 are the zero rows of the
                // tile (their filter value is exactly 0): any valid address will do
                constexpr int dummy = 0;
                const int t0 = (r & 3) + 8 * (r >> 2), t1 = t0 + 4;
                const unsigned o0 = t0 < K * K ? rowoff[t0 / K] + coloff[t0 % K] : rowoff[dummy] + coloff[dummy];
                const unsigned o1 = t1 < K * K ? rowoff[t1 / K] + coloff[t1 % K] : rowoff[dummy] + coloff[dummy];
                e[r] = buf_ld(rev, h ? o1 : o0);
            }
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float f = acc[m][n][r] * oscale + bv[r];
                f = f > 0.f ? f : f * slope;
                sum = fmaf(f, e[r], sum);
            }
            part[m][n] = sum;
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (co_base >> 5) + m;
        // the two lane halves hold disjoint tap subsets of the same pixel: exchange and add, then the lower half stores
        // x-half 0 and the upper half x-half 1 (one coalesced 256-byte store per wave and channel)
        const float p0 = part[m][0] + __shfl_xor(part[m][0], 32, 64);
        const float p1 = part[m][1] + __shfl_xor(part[m][1], 32, 64);
        const int xo = x0 + h * 32 + l31;
        if (c < fac.C && yo < g.Ho && xo < g.Wo) out[((int64_t)b * fac.C + c) * HW + (int64_t)yo * g.Wo + xo] = h ? p1 : p0;
    }
}

// ------------------------------------------------------------------------------------------------
// forward (TR = false) and stride-1 data gradient (TR = true)
//   x    [B,Cin,H,W]   (for dgrad: grad_output [B,Cout_fwd,..])      dact_y: optional, same shape as x
//   w    forward: [Cout,Cin,KS,KS];  TR: the FORWARD conv's weight [Cin,Cout,KS,KS] (its Cout = our Cin)
//   out  [B,Cout,Ho,Wo]
// Staging is organised so that everything address-related is computed once per workgroup: a thread owns
// fixed tile positions (its image offsets and validity never change) and walks the chunk's channels, and
// fixed weight-slice elements whose global offset just advances by a constant per chunk.  The loads of
// chunk c+1 are issued into registers before the MFMA block of chunk c and committed to LDS after it.
template <int KS, int S, int MT, int CK, bool TR, int DACT>
__global__ __launch_bounds__(256) void conv_fwd_f32(const float *__restrict__ x, const float *__restrict__ dact_y,
                                                    const float *__restrict__ w, const float *__restrict__ bias,
                                                    float *__restrict__ out, ConvGeom g, int act, float slope,
                                                    float dslope) {
    constexpr int KK = KS * KS;
    constexpr int IH = S * (TY - 1) + KS, IW = S * (TX - 1) + KS;
    constexpr int PS = IH * IW;            // plane stride of the staged input tile
    constexpr int COS = 32 * MT;           // output channels of this block
    // LDS weight slice [ci][tap][co]; the row stride is chosen so that the transposing commit is (nearly)
    // conflict-free: forward walks rows (+1 bank per lane), the data gradient walks taps backwards and co.
    constexpr int COSP = TR ? COS + 4 : COS + 1;
    constexpr int NPOS = (PS + 255) / 256; // tile positions owned by a thread
    constexpr int WEL = KK * CK * COS;     // weight-slice elements
    constexpr int NW = (WEL + 255) / 256;
    __shared__ float sIn[CK * PS];
    __shared__ float sW[KK * CK * COSP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + TX - 1) / TX, tiles_y = (g.Ho + TY - 1) / TY;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TY, x0 = tx * TX;
    const int co_base = blockIdx.y * COS;
    const int iy0 = S * y0 - g.pad, ix0 = S * x0 - g.pad;
    const int HW = g.H * g.W;
    const unsigned plane_bytes = (unsigned)HW * 4u, x_bytes = (unsigned)g.Cin * plane_bytes;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)b * g.Cin * HW, x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(DACT ? dact_y + (int64_t)b * g.Cin * HW : x, DACT ? x_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rwt = make_rsrc(w, (unsigned)g.Cout * (unsigned)g.Cin * KK * 4u);

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    unsigned in_off[NPOS];                 // byte offset of the owned position inside a channel plane
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
        const int pos = tid + q * 256;
        const int r = pos / IW, c = pos - r * IW;
        const int yy = iy0 + r, xx = ix0 + c;
        in_off[q] = (pos < PS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (unsigned)(yy * g.W + xx) * 4u : SENT;
    }
    unsigned w_off[NW];                    // byte offset of the owned weight elements for chunk 0
    int w_dst[NW];                         // their place in the LDS slice [ci][tap][co]
#pragma unroll
    for (int it = 0; it < NW; ++it) {
        const int i = tid + it * 256;
        int off, dst;
        if (!TR) {   // W[co][ci][tap]: for fixed co the (ci,tap) run is contiguous in memory
            const int co = i / (CK * KK), rem = i - co * (CK * KK);
            const int ci = rem / KK, tap = rem - ci * KK;
            off = ((co_base + co) * g.Cin + ci) * KK + tap;
            dst = (ci * KK + tap) * COSP + co;
        } else {     // data gradient: k = forward out-channel, m = forward in-channel, taps flipped
            const int ci = i / (COS * KK), rem = i - ci * (COS * KK);
            const int co = rem / KK, tap = rem - co * KK;
            off = (ci * g.Cout + co_base + co) * KK + tap;
            dst = (ci * KK + KK - 1 - tap) * COSP + co;
        }
        w_off[it] = i < WEL ? (unsigned)off * 4u : SENT;
        w_dst[it] = dst;
    }
    // Rows co >= Cout / channels >= Cin of the slice may alias other (finite) weights or read 0 past the
    // end: those rows are never stored, those channels meet zero-filled input, so no masking is needed.
    const unsigned w_chunk = (unsigned)(TR ? CK * g.Cout * KK : CK * KK) * 4u;

    float rin[NPOS * CK], rw[NW];
    auto prefetch = [&](int chunk) {
        const unsigned cb = (unsigned)chunk * (unsigned)CK * plane_bytes;
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
#pragma unroll
            for (int ci = 0; ci < CK; ++ci) {
                const unsigned o = in_off[q] + cb + (unsigned)ci * plane_bytes;
                float v = buf_ld(rx, o);
                if constexpr (DACT != 0) v *= act_grad_c<DACT>(buf_ld(ry, o), dslope);
                rin[q * CK + ci] = v;
            }
        const unsigned wb = (unsigned)chunk * w_chunk;
#pragma unroll
        for (int it = 0; it < NW; ++it) rw[it] = buf_ld(rwt, w_off[it] + wb);
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
            if (tid + q * 256 < PS) {
#pragma unroll
                for (int ci = 0; ci < CK; ++ci) sIn[ci * PS + tid + q * 256] = rin[q * CK + ci];
            }
#pragma unroll
        for (int it = 0; it < NW; ++it)
            if (tid + it * 256 < WEL) sW[w_dst[it]] = rw[it];
    };

    const int nchunks = (g.Cin + CK - 1) / CK;
    prefetch(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();          // previous chunk's operand reads are done
        commit();
        __syncthreads();
        prefetch(chunk + 1);      // past the last chunk every offset is out of range: reads 0, never committed
        // ---- MFMA: wave owns output row `wave` of the tile (2 x-halves) x MT co tiles
        const float *bp = sIn + (lane >> 5) * PS + (S * wave) * IW + S * (lane & 31);
        const float *ap = sW + (lane >> 5) * (KK * COSP) + (lane & 31);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int cp = 0; cp < CK; cp += 2) {
                float a[MT], bv[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = ap[(cp * KK + tap) * COSP + m * 32];
#pragma unroll
                for (int n = 0; n < 2; ++n) bv[n] = bp[cp * PS + ky * IW + kx + n * 32 * S];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: bias + activation, NCHW store (lanes = consecutive x)
    store_out_tile<MT>(out, bias, acc, g, b, co_base, y0 + wave, x0, lane, act, slope);
}

// ------------------------------------------------------------------------------------------------
// bf16 matrix-core variant of the forward / data-gradient kernel (fp32 tensors in HBM, bf16 MFMA
// operands, fp32 accumulation): v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate, which turns
// the 3x3 convs from matrix-bound into HBM-bound.  Same tiling and software pipeline as conv_fwd_f32;
// what changes is the operand images: MFMA wants 8 consecutive k (= channels) per lane, so LDS holds
// channel-minor images [position][16 ch] and [tap][co][16 ch] with a 48-byte row pitch (16-byte aligned
// for ds_read_b128 and 3*16 B => the 16 lanes of a read group hit 16 distinct 16-B slots).  A thread still
// owns fixed tile positions: it loads their 16 channels (coalesced along x across threads), converts with
// v_cvt_pk_bf16_f32 and writes two 16-byte pieces.  Weights are pre-packed once per call to bf16
// [tap][co][ci16] (conv_pack_w_bf16) so their staging is plain 16-byte copies.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CKB = 16;                // channels per chunk
constexpr int PITCH = 24;              // bf16 elements per LDS row (16 data + 8 pad = 48 bytes)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// W fp32 -> packed bf16.  forward: Wp[tap][co][ci] = W[co][ci][tap];  TR: Wp[tap][m][k] = W[k][m][KK-1-tap]
// (rows = our output channels M, cols = our contraction channels K padded to K16 with zeros)
// split != 0: a second image with the rounding remainders follows, wp[total + idx] = bf16(w - float(bf16(w)))
__global__ void conv_pack_w_bf16(const float *__restrict__ w, __bf16 *__restrict__ wp, int M, int K, int K16, int KK,
                                 int transposed, int split) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)KK * M * K16;
    if (idx >= total) return;
    const int k = (int)(idx % K16);
    const int m = (int)((idx / K16) % M);
    const int tap = (int)(idx / ((int64_t)K16 * M));
    float v = 0.f;
    if (k < K) v = transposed ? w[((int64_t)k * M + m) * KK + (KK - 1 - tap)] : w[((int64_t)m * K + k) * KK + tap];
    const __bf16 h = (__bf16)v;
    wp[idx] = h;
    if (split) wp[total + idx] = (__bf16)(v - (float)h);
}

template <int KS, int MT, int DACT>
__global__ __launch_bounds__(256) void conv_fwd_bf16(const float *__restrict__ x, const float *__restrict__ dact_y,
                                                     const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                                                     float *__restrict__ out, ConvGeom g, int K16, int act, float slope,
                                                     float dslope) {
    constexpr int S = 1;
    constexpr int KK = KS * KS;
    constexpr int IH = S * (TY - 1) + KS, IW = S * (TX - 1) + KS;
    constexpr int PS = IH * IW;            // positions of the staged input tile
    constexpr int COS = 32 * MT;
    constexpr int NPOS = (PS + 255) / 256;
    constexpr int WPIECES = KK * COS * 2;  // 16-byte pieces of the weight slice
    constexpr int NWB = (WPIECES + 255) / 256;
    __shared__ __attribute__((aligned(16))) __bf16 sIn[PS * PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 sW[KK * COS * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + TX - 1) / TX, tiles_y = (g.Ho + TY - 1) / TY;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TY, x0 = tx * TX;
    const int co_base = blockIdx.y * COS;
    const int iy0 = S * y0 - g.pad, ix0 = S * x0 - g.pad;
    const int HW = g.H * g.W;
    const unsigned plane_bytes = (unsigned)HW * 4u, x_bytes = (unsigned)g.Cin * plane_bytes;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)b * g.Cin * HW, x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(DACT ? dact_y + (int64_t)b * g.Cin * HW : x, DACT ? x_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(wp), 0, (unsigned)KK * (unsigned)g.Cout * (unsigned)K16 * 2u, 0x00020000);

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    unsigned in_off[NPOS];
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
        const int pos = tid + q * 256;
        const int r = pos / IW, c = pos - r * IW;
        const int yy = iy0 + r, xx = ix0 + c;
        in_off[q] = (pos < PS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (unsigned)(yy * g.W + xx) * 4u : SENT;
    }
    unsigned w_off[NWB];                   // byte offset of the owned 16-byte weight pieces for chunk 0
    int w_dst[NWB];                        // their element offset in the LDS slice
#pragma unroll
    for (int it = 0; it < NWB; ++it) {
        const int i = tid + it * 256;
        const int row = i >> 1, half = i & 1;          // row = tap*COS + co
        const int tap = row / COS, co = row - tap * COS;
        // rows co >= Cout alias other (finite) rows or fall past the end: never stored
        w_off[it] = i < WPIECES ? (unsigned)(((tap * g.Cout + co_base + co) * K16 + half * 8) * 2) : SENT;
        w_dst[it] = row * PITCH + half * 8;
    }

    float rin[NPOS * CKB];
    u32x4 rw[NWB];
    auto prefetch = [&](int chunk) {
        const unsigned cb = (unsigned)chunk * (unsigned)CKB * plane_bytes;
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
#pragma unroll
            for (int ci = 0; ci < CKB; ++ci) {
                const unsigned o = in_off[q] + cb + (unsigned)ci * plane_bytes;
                float v = buf_ld(rx, o);
                if constexpr (DACT != 0) v *= act_grad_c<DACT>(buf_ld(ry, o), dslope);
                rin[q * CKB + ci] = v;
            }
        const unsigned wb = (unsigned)chunk * (unsigned)(CKB * 2);
#pragma unroll
        for (int it = 0; it < NWB; ++it) rw[it] = __builtin_amdgcn_raw_buffer_load_b128(rwt, w_off[it] + wb, 0, 0);
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
            if (tid + q * 256 < PS) {
                u32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = pack_bf16(rin[q * CKB + 2 * j], rin[q * CKB + 2 * j + 1]);
                    hi[j] = pack_bf16(rin[q * CKB + 8 + 2 * j], rin[q * CKB + 8 + 2 * j + 1]);
                }
                u32x4 *dst = reinterpret_cast<u32x4 *>(sIn + (tid + q * 256) * PITCH);
                dst[0] = lo;
                dst[1] = hi;
            }
#pragma unroll
        for (int it = 0; it < NWB; ++it)
            if (tid + it * 256 < WPIECES) *reinterpret_cast<u32x4 *>(sW + w_dst[it]) = rw[it];
    };

    const int nchunks = K16 / CKB;
    prefetch(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();
        commit();
        __syncthreads();
        prefetch(chunk + 1);      // past the last chunk every offset is out of range: reads 0, never committed
        // lane: column (pixel / out channel) lane&31, k-half lane>>5 (channels 8h..8h+7 of the chunk)
        const __bf16 *bp = sIn + ((S * wave) * IW + S * (lane & 31)) * PITCH + (lane >> 5) * 8;
        const __bf16 *ap = sW + (lane & 31) * PITCH + (lane >> 5) * 8;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            bf16x8 a[MT], bv[2];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8 *>(ap + (tap * COS + m * 32) * PITCH);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bv[n] = *reinterpret_cast<const bf16x8 *>(bp + (ky * IW + kx + n * 32 * S) * PITCH);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }
    store_out_tile<MT>(out, bias, acc, g, b, co_base, y0 + wave, x0, lane, act, slope);
}

// ------------------------------------------------------------------------------------------------
// Split-precision variant ("bf16x3"): every fp32 operand is carried as hi = bf16(v) and lo = bf16(v - hi), and a
// product is accumulated as lo*hi + hi*lo + hi*hi on the bf16 matrix cores (fp32 accumulation).  The dropped lo*lo
// term and the rounding of lo are ~2^-17 relative: results agree with the exact-fp32 kernels to ~1e-5, at 3/16 of
// their MFMA time.  Both images cost the LDS bytes of fp32, so the tile is 8 rows x 64 px for 512 threads (one weight
// slice per CU instead of two): one workgroup of 8 waves per CU.
constexpr int TYB = 8, NTB = 512;

// Workgroup ids are dealt round-robin over the 8 XCDs, each with its own L2 (MI355X_MICROARCH.md, workgroup dispatch; used for
// speed only).  Neighbouring pixel tiles share halo rows and partial cache lines: with tile = workgroup id they sat on different
// XCDs and every L2 fetched its own copy from the fabric (rocprofv3 FETCH_SIZE, calibrated on tools/bwprobe.hip: 1.75-2.3x the
// algorithmic bytes).  xcd_tile() permutes the tile index so that the ids congruent mod 8 -- one XCD -- cover one CONTIGUOUS
// eighth of the tile sequence (at B = 8: one image each).  Bijective on [0, T) for every T; callers apply it only when the
// grid's x extent is a multiple of 8 (otherwise id % 8 is not the XCD of the workgroup).
__device__ __forceinline__ int xcd_tile(int t, int T) {
    const int q = T >> 3, r = T & 7, x = t & 7, j = t >> 3;
    return x * q + min(x, r) + j;
}

// In-kernel phase stamps: compiled in by tools/kbench.hip only (-DEBFI_KBENCH); the product build has no stamp code.
#ifdef EBFI_KBENCH
// stamps go to a small LDS area behind the operand buffers (a global store per stamp would sit in the wave's vmcnt queue and
// perturb exactly the waits being measured) and are copied out once at the end of the kernel
__device__ unsigned long long *g_kb_stamps = nullptr;      // [workgroup][2 waves][KB_NSTAMP]
constexpr int KB_NSTAMP = 32;
#define KB_LDS_BYTES (2 * KB_NSTAMP * 8)
#define KB_STAMP(i)                                                                                                     \
    do {                                                                                                                \
        if (threadIdx.x == 0 || threadIdx.x == 256)                                                                     \
            reinterpret_cast<unsigned long long *>(smd + KB_LDS_OFF)[(threadIdx.x >> 8) * KB_NSTAMP + (i)] =            \
                __builtin_amdgcn_s_memtime();                                                                           \
    } while (0)
#define KB_FLUSH()                                                                                                      \
    do {                                                                                                                \
        __syncthreads();                                                                                                \
        if (g_kb_stamps && threadIdx.x < 2 * KB_NSTAMP)                                                                 \
            g_kb_stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * KB_NSTAMP + threadIdx.x] =                  \
                reinterpret_cast<unsigned long long *>(smd + KB_LDS_OFF)[threadIdx.x];                                  \
    } while (0)
// (for kernels whose wave roles leave through different exits: every stamping thread copies its own row, no barrier)
#define KB_FLUSH_SELF()                                                                                                 \
    do {                                                                                                                \
        if (g_kb_stamps && (threadIdx.x == 0 || threadIdx.x == 256))                                                    \
            for (int i_ = 0; i_ < KB_NSTAMP; ++i_)                                                                      \
                g_kb_stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (threadIdx.x >> 8)) * KB_NSTAMP + i_] = \
                    reinterpret_cast<unsigned long long *>(smd + KB_LDS_OFF)[(threadIdx.x >> 8) * KB_NSTAMP + i_];      \
    } while (0)
#define KB_CLEAR_SELF()                                                                                                 \
    do {                                                                                                                \
        if (threadIdx.x == 0 || threadIdx.x == 256)                                                                     \
            for (int i_ = 0; i_ < KB_NSTAMP; ++i_)                                                                      \
                reinterpret_cast<unsigned long long *>(smd + KB_LDS_OFF)[(threadIdx.x >> 8) * KB_NSTAMP + i_] = 0ull;   \
    } while (0)
#else
#define KB_LDS_BYTES 0
#define KB_STAMP(i) do { } while (0)
#define KB_FLUSH() do { } while (0)
#define KB_FLUSH_SELF() do { } while (0)
#define KB_CLEAR_SELF() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// conv_fwd_bf16x3_db: forward / data gradient of the split-precision mode, double-buffered.  With one 512-thread
// workgroup per CU nothing else can cover the commit phase, so the operand images of chunk c+1 are written into a second
// LDS buffer BETWEEN the MFMAs of chunk c (one barrier per chunk; measured 1-4 % over the single-buffer form).  Two buffers only fit without padding: rows are 16 channels = 32 bytes, and a row's two 16-byte
// halves are swapped when bit 3 of the row index is set.  ds_read_b128 is served in groups of 16 lanes
// ({0-3,12-15,20-27}, ...): rows r..r+27 of such a group then hit 16 distinct 16-byte slots of the 256-byte LDS row,
// the same conflict-free property the 48-byte pitch bought, in 2/3 of the space (158 KB for both buffers).
// VEC = 4: the input tile is fetched as 16-byte quads of 4 consecutive pixels (a thread owns one quad position x 8 channels
// = 8 loads instead of 32); needs W % 4 == 0 and padding KS/2.  Measured (tools/kbench, 64->64 at 128x128): equal to the
// dword form without the activation derivative (39.1 vs 38.4 us), 16 % faster with it (46.8 vs 55.9 us: half the registers
// held across the matrix block) -- so the launcher picks it for DACT != 0 only.  What the measurements of round 2 say about
// this kernel (ablation builds of tools/kbench): its time is prologue 6.8 us (first input tile: 5-6 us from a cold start on
// all 256 CUs at once) + 4 x 4.9 us per 16-channel chunk (matrix pipe 70 % busy; 3.45 us at 100 %) + 4.3 us of output
// stores, i.e. fill and drain of the one tile a CU gets are a third of it; the memory side alone (no MFMA) takes 23 us, the
// matrix side alone 29.6 us, the separate weight-packing launch another 5 us (now done once per step, ebfi_amd.weightbank).
template <int KS, int MT, int DACT, int VEC>
__global__ __launch_bounds__(NTB) void conv_fwd_bf16x3_db(const float *__restrict__ x, const float *__restrict__ dact_y,
                                                          const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                                                          float *__restrict__ out, ConvGeom g, int K16, int act, float slope,
                                                          float dslope, EpiExtra epi, int tiles_total) {
    constexpr bool QLD = VEC == 4;                     // input tile fetched as 16-byte quads
    constexpr int KK = KS * KS;
    constexpr int IH = TYB - 1 + KS, IW = TX - 1 + KS;
    constexpr int PS = IH * IW;            // positions of the staged input tile
    constexpr int COS = 32 * MT;
    constexpr int NPOS = (PS + NTB - 1) / NTB;
    constexpr int WPIECES = KK * COS * 2;  // 16-byte pieces of ONE weight image (hi or lo)
    constexpr int NWB = (2 * WPIECES + NTB - 1) / NTB;
    constexpr int INB = PS * 32, WB = KK * COS * 32;   // bytes of one input / weight image
    constexpr int BUFB = 2 * INB + 2 * WB;             // one buffer: input hi | input lo | weight hi | weight lo
    extern __shared__ __attribute__((aligned(16))) char smd[];
    [[maybe_unused]] constexpr int KB_LDS_OFF = 2 * BUFB;
#ifdef EBFI_KBENCH
    if (threadIdx.x < 2 * KB_NSTAMP) reinterpret_cast<unsigned long long *>(smd + KB_LDS_OFF)[threadIdx.x] = 0ull;
    __syncthreads();
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + TX - 1) / TX, tiles_y = (g.Ho + TYB - 1) / TYB;
    const bool xcd_map = (gridDim.x & 7) == 0;             // (see xcd_tile: neighbouring tiles on one XCD)
    int t = xcd_map ? xcd_tile(blockIdx.x, tiles_total) : blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TYB, x0 = tx * TX;
    const int co_base = blockIdx.y * COS;
    const int iy0 = y0 - g.pad, ix0 = x0 - g.pad;
    const int HW = g.H * g.W;
    const unsigned plane_bytes = (unsigned)HW * 4u, x_bytes = (unsigned)g.Cin * plane_bytes;
    // grouped convolution: this block's output channels read the input channels of their group only
    const int64_t in_base = ((int64_t)b * g.groups + co_base / (g.Cout / g.groups)) * g.Cin * HW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + in_base, x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(DACT ? dact_y + in_base : x, DACT ? x_bytes : 0u);
    const unsigned img_bytes = (unsigned)KK * (unsigned)g.Cout * (unsigned)K16 * 2u;   // one packed weight image
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(wp), 0, 2u * img_bytes, 0x00020000);

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    unsigned in_off[NPOS];
    int in_dst[NPOS];                      // byte offset of the row's half 0 (half 1 sits at the other 16 bytes of the row)
    // VEC: quad (row qr, quad qq) of channel half qh; its pixel j sits in tile column 4*qq + j - SH (SH aligns the quads
    // to 16 bytes in global memory: the tile starts `pad` pixels left of a multiple of 64)
    constexpr int SH = (4 - (KS / 2) % 4) % 4;
    constexpr int NQ = (IW + SH + 3) / 4;  // quads per tile row
    constexpr int NITEM = IH * NQ * 2;
    static_assert(!QLD || NITEM <= NTB, "one quad item per thread");
    const int qh = tid & 1, qr = (tid >> 1) / NQ, qq = (tid >> 1) - qr * NQ;
    unsigned q_off = SENT;
    if constexpr (QLD) {
        const int yy = iy0 + qr, xq = ix0 - SH + 4 * qq;
        if (tid < NITEM && yy >= 0 && yy < g.H && xq >= 0 && xq + 3 < g.W)
            q_off = (unsigned)(yy * g.W + xq) * 4u + (unsigned)(8 * qh) * plane_bytes;
    }
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
        const int pos = tid + q * NTB;
        const int r = pos / IW, c = pos - r * IW;
        const int yy = iy0 + r, xx = ix0 + c;
        in_off[q] = (pos < PS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (unsigned)(yy * g.W + xx) * 4u : SENT;
        in_dst[q] = pos * 32 + (((pos >> 3) & 1) << 4);
    }
    unsigned w_off[NWB];                   // byte offset of the owned 16-byte weight pieces for chunk 0 (hi image, then lo)
    int w_dst[NWB];                        // their byte offset inside a buffer
#pragma unroll
    for (int it = 0; it < NWB; ++it) {
        const int i = tid + it * NTB;
        const int sel = i >= WPIECES ? 1 : 0, j = i - sel * WPIECES;
        const int row = j >> 1, half = j & 1;          // row = tap*COS + co
        const int tap = row / COS, co = row - tap * COS;
        w_off[it] = i < 2 * WPIECES ? (unsigned)sel * img_bytes + (unsigned)(((tap * g.Cout + co_base + co) * K16 + half * 8) * 2)
                                    : SENT;
        w_dst[it] = 2 * INB + sel * WB + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
    }
    // reader side: lane = (column lane&31, channel half lane>>5)
    const int hsel = lane >> 5, l31 = lane & 31;
    const int a_lane = 2 * INB + l31 * 32 + ((hsel ^ ((l31 >> 3) & 1)) << 4);   // + (tap*COS + m*32)*32: multiples of 1 KB keep bit 3
    const int pbase = wave * IW + l31;
    unsigned fbits = 0;                    // bit tap: half-swap of this lane's row for that tap (x halves n = 0, 1 agree)
#pragma unroll
    for (int tap = 0; tap < KK; ++tap)
        fbits |= (unsigned)((((pbase + (tap / KS) * IW + (tap % KS)) >> 3) & 1) ^ hsel) << tap;

    float rin[!QLD ? NPOS * CKB : 1];
    float ryv[(!QLD && DACT != 0) ? NPOS * CKB : 1];
    u32x4 rq[QLD ? 8 : 1], rqy[(QLD && DACT != 0) ? 8 : 1];     // VEC: 4 pixels of channels 8*qh + k
    u32x4 rw[NWB];
    // The vector-memory path takes one wave instruction every ~16 cycles whatever its width (address processing of 4 lanes
    // per cycle): the 37+ loads of a chunk issued back to back stall all eight waves for ~5000 cycles in VMEM issue while the
    // matrix cores idle (in-kernel stamps, tools/kbench).  The loads of chunk c+1 are therefore issued in NLG groups BETWEEN
    // the taps of chunk c (load group g in front of the MFMAs of tap g) and committed after the last tap.
    constexpr int NIN = QLD ? 8 : NPOS * CKB;          // input loads per thread and chunk
    constexpr int NLOAD = NIN + NWB;                        // (activation-derivative loads ride along)
    constexpr int NLG = KK >= 9 ? 6 : 1;                    // groups; the last taps carry none so that the data has landed
    constexpr int LPG = (NLOAD + NLG - 1) / NLG;
    auto prefetch_group = [&](int chunk, int grp) {
        const unsigned cb = (unsigned)chunk * (unsigned)CKB * plane_bytes;
        const unsigned wb = (unsigned)chunk * (unsigned)(CKB * 2);
#pragma unroll
        for (int i = 0; i < LPG; ++i) {
            const int l = grp * LPG + i;
            if (l < NIN) {
                if constexpr (QLD) {
                    const unsigned o = q_off + cb + (unsigned)l * plane_bytes;
                    rq[l] = __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0);
                    if constexpr (DACT != 0) rqy[l] = __builtin_amdgcn_raw_buffer_load_b128(ry, o, 0, 0);
                } else {
                    const int q = l / CKB, ci = l - q * CKB;
                    const unsigned o = in_off[q] + cb + (unsigned)ci * plane_bytes;
                    rin[l] = buf_ld(rx, o);
                    if constexpr (DACT != 0) ryv[l] = buf_ld(ry, o);
                }
            } else if (l < NLOAD) {
                rw[l - NIN] = __builtin_amdgcn_raw_buffer_load_b128(rwt, w_off[l - NIN] + wb, 0, 0);
            }
        }
    };
    auto prefetch = [&](int chunk) {
#pragma unroll
        for (int grp = 0; grp < NLG; ++grp) prefetch_group(chunk, grp);
    };
    auto commit = [&](int buf) {
        char *base = smd + buf * BUFB;
        if constexpr (QLD) {
            if (tid < NITEM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * qq + j - SH;
                    if (c < 0 || c >= IW) continue;
                    u32x4 hv, lv;
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        float v0 = __uint_as_float(rq[k][j]), v1 = __uint_as_float(rq[k + 1][j]);
                        if constexpr (DACT != 0) {
                            v0 *= act_grad_c<DACT>(__uint_as_float(rqy[k][j]), dslope);
                            v1 *= act_grad_c<DACT>(__uint_as_float(rqy[k + 1][j]), dslope);
                        }
                        const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
                        hv[k >> 1] = pack_bf16((float)a0, (float)a1);
                        lv[k >> 1] = pack_bf16(v0 - (float)a0, v1 - (float)a1);
                    }
                    const int pos = qr * IW + c;
                    const int d = pos * 32 + ((qh ^ ((pos >> 3) & 1)) << 4);
                    *reinterpret_cast<u32x4 *>(base + d) = hv;
                    *reinterpret_cast<u32x4 *>(base + INB + d) = lv;
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
            if (tid + q * NTB < PS) {
                u32x4 h0, h1, l0, l1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v0 = rin[q * CKB + 2 * j], v1 = rin[q * CKB + 2 * j + 1];
                    if constexpr (DACT != 0) {
                        v0 *= act_grad_c<DACT>(ryv[q * CKB + 2 * j], dslope);
                        v1 *= act_grad_c<DACT>(ryv[q * CKB + 2 * j + 1], dslope);
                    }
                    const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
                    const unsigned hp = pack_bf16((float)a0, (float)a1);
                    const unsigned lp = pack_bf16(v0 - (float)a0, v1 - (float)a1);
                    if (j < 4) { h0[j] = hp; l0[j] = lp; } else { h1[j - 4] = hp; l1[j - 4] = lp; }
                }
                const int d0 = in_dst[q], d1 = in_dst[q] ^ 16;
                *reinterpret_cast<u32x4 *>(base + d0) = h0;
                *reinterpret_cast<u32x4 *>(base + d1) = h1;
                *reinterpret_cast<u32x4 *>(base + INB + d0) = l0;
                *reinterpret_cast<u32x4 *>(base + INB + d1) = l1;
            }
        }
#pragma unroll
        for (int it = 0; it < NWB; ++it)
            if (tid + it * NTB < 2 * WPIECES) *reinterpret_cast<u32x4 *>(base + w_dst[it]) = rw[it];
    };

    const int nchunks = K16 / CKB;
    // operand fragments of tap t+1 are fetched from LDS before the MFMAs of tap t are issued (two register sets)
    bf16x8 ah[2][MT], al[2][MT], bh[2][2], bl[2][2];
    auto tap_read = [&](const char *base, int tap, int set) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const char *bp = base + pbase * 32 + (int)(((fbits >> tap) & 1u) << 4) + (ky * IW + kx) * 32;
        const char *ap = base + a_lane + tap * COS * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[set][m] = *reinterpret_cast<const bf16x8 *>(ap + m * 1024);
            al[set][m] = *reinterpret_cast<const bf16x8 *>(ap + m * 1024 + WB);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bh[set][n] = *reinterpret_cast<const bf16x8 *>(bp + n * 1024);
            bl[set][n] = *reinterpret_cast<const bf16x8 *>(bp + n * 1024 + INB);
        }
    };
    auto tap_mfma = [&](int set) {
#ifdef KB_NO_MFMA                         // ablation (harness only): keep the operand reads alive, issue no matrix work
#pragma unroll
        for (int m = 0; m < MT; ++m) { asm volatile("" ::"v"(ah[set][m]), "v"(al[set][m])); }
#pragma unroll
        for (int n = 0; n < 2; ++n) { asm volatile("" ::"v"(bh[set][n]), "v"(bl[set][n])); }
        return;
#endif
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][m], bh[set][n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][m], bl[set][n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][m], bh[set][n], acc[m][n], 0, 0, 0);
            }
    };
    // P15: the commit of chunk c+1 hidden behind the matrix work of chunk c (3x3, dword loads, no activation derivative).  In
    // the plain form a chunk costs 4.9 us against 3.45 us of matrix work (in-kernel stamps): ~0.9 us of it is the commit
    // (conversion + LDS writes, with the load latency in front) and the barrier skew it causes, during which no wave of the
    // CU has matrix work.  Here the inputs of chunk c+1 are loaded in front of taps 0..3 of chunk c and converted / written
    // to the other LDS buffer in four slices behind taps 5..8 (each slice waits only for ITS loads, issued >= 4 taps
    // earlier); only the five weight pieces are committed at the chunk's end.  (A second register set for a two-chunk
    // distance does not fit: 256 VGPRs + 123 spills, 3.5x slower -- measured.)
    constexpr bool P15 = !QLD && DACT == 0 && KS == 3 && NPOS == 2;
    if constexpr (P15) {
        // PERSISTENT over pixel tiles: this workgroup walks tiles blockIdx.x, + gridDim.x, ... of its output-channel
        // block.  The chunk pipeline runs across the tile boundary: during the LAST chunk of a tile the loads / commit
        // slices are those of the NEXT tile's first chunk, so a new tile starts its matrix work right after the stores of
        // the finished one have been issued -- the 6.8 us prologue and most of the 4.3 us store drain of a tile (a third of
        // a 4-chunk tile) are paid once per workgroup instead of once per tile.
        const int grp = co_base / (g.Cout / g.groups);
        auto tile_coords = [&](int tt, int &tb, int &ty0, int &tx0) {
            int u = xcd_map ? xcd_tile(tt, tiles_total) : tt;      // (neighbouring tiles on one XCD: shared halo lines hit its L2)
            const int txi = u % tiles_x; u /= tiles_x;
            const int tyi = u % tiles_y;
            tb = u / tiles_y; ty0 = tyi * TYB; tx0 = txi * TX;
        };
        auto tile_offsets = [&](int ty0, int tx0, bool live, unsigned (&o)[NPOS]) {
#pragma unroll
            for (int q = 0; q < NPOS; ++q) {
                const int pos = tid + q * NTB;
                const int r = pos / IW, c = pos - r * IW;
                const int yy = ty0 - g.pad + r, xx = tx0 - g.pad + c;
                o[q] = (live && pos < PS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (unsigned)(yy * g.W + xx) * 4u : SENT;
            }
        };
        auto load_inputs = [&](const float *src, unsigned nbytes, const unsigned (&offs)[NPOS], int chunk, int l0, int l1) {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(src, nbytes);
            const unsigned cb = (unsigned)chunk * (unsigned)CKB * plane_bytes;
#pragma unroll
            for (int l = l0; l < l1; ++l) {
                const int q = l / CKB, ci = l - q * CKB;
                rin[l] = buf_ld(r, offs[q] + cb + (unsigned)ci * plane_bytes);
            }
        };
        auto load_weights = [&](int chunk, int i0, int i1) {
            const unsigned wb = (unsigned)chunk * (unsigned)(CKB * 2);
#pragma unroll
            for (int it = i0; it < i1; ++it)
                if (it < NWB) rw[it] = __builtin_amdgcn_raw_buffer_load_b128(rwt, w_off[it] + wb, 0, 0);
        };
        // slice sl in 0..3: position sl >> 1, channel half sl & 1 -> the hi and the lo 16-byte piece of that half
        // (loads 8*sl .. 8*sl+7 in issue order)
        auto commit_slice = [&](int buf, int sl) {
            const int q = sl >> 1, hf = sl & 1;
            if (tid + q * NTB < PS) {
                u32x4 hv, lv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = rin[q * CKB + 8 * hf + 2 * j], v1 = rin[q * CKB + 8 * hf + 2 * j + 1];
                    const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
                    hv[j] = pack_bf16((float)a0, (float)a1);
                    lv[j] = pack_bf16(v0 - (float)a0, v1 - (float)a1);
                }
                char *dst = smd + buf * BUFB + (hf ? (in_dst[q] ^ 16) : in_dst[q]);
                *reinterpret_cast<u32x4 *>(dst) = hv;
                *reinterpret_cast<u32x4 *>(dst + INB) = lv;
            }
        };
        auto commit_weights = [&](int buf) {
            char *base = smd + buf * BUFB;
#pragma unroll
            for (int it = 0; it < NWB; ++it)
                if (tid + it * NTB < 2 * WPIECES) *reinterpret_cast<u32x4 *>(base + w_dst[it]) = rw[it];
        };
        int tcur = blockIdx.x, cb_ = b, cy0 = y0, cx0 = x0;
        const float *xcur = x + in_base;
        KB_STAMP(0);
        load_inputs(xcur, x_bytes, in_off, 0, 0, NPOS * CKB);
        load_weights(0, 0, NWB);
        KB_STAMP(1);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) commit_slice(0, sl);
        commit_weights(0);
        __syncthreads();
        KB_STAMP(2);
        int item = 0;
        for (;;) {
            const int tn = tcur + (int)gridDim.x;
            const bool nlive = tn < tiles_total;
            int nb_, ny0, nx0;
            tile_coords(nlive ? tn : 0, nb_, ny0, nx0);
            unsigned noff[NPOS];
            tile_offsets(ny0, nx0, nlive, noff);
            const float *xnext = x + ((int64_t)nb_ * g.groups + grp) * g.Cin * HW;
            for (int chunk = 0; chunk < nchunks; ++chunk, ++item) {
                const char *base = smd + (item & 1) * BUFB;
                const int nb = (item & 1) ^ 1;
                // what is staged during this chunk: the tile's next chunk, or the first chunk of the next tile
                const bool last = chunk == nchunks - 1;
                const float *lsrc = last ? xnext : xcur;
                const unsigned lbytes = (last && !nlive) ? 0u : x_bytes;      // past the last tile: empty descriptor, reads 0
                const int lchunk = last ? 0 : chunk + 1;
                unsigned loff[NPOS];
#pragma unroll
                for (int q = 0; q < NPOS; ++q) loff[q] = last ? noff[q] : in_off[q];
                if (item < 6) KB_STAMP(3 + 4 * item);
                tap_read(base, 0, 0);
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) {
                    if (tap + 1 < KK) tap_read(base, tap + 1, (tap + 1) & 1);
#ifndef KB_NO_LOADS
                    if (tap < 4) load_inputs(lsrc, lbytes, loff, lchunk, 8 * tap, 8 * tap + 8);
                    if (tap == 4) load_weights(lchunk, 0, NWB);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    tap_mfma(tap & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap >= 5) commit_slice(nb, tap - 5);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (item < 6) KB_STAMP(5 + 4 * item);
                commit_weights(nb);
                __syncthreads();
                if (item < 6) KB_STAMP(6 + 4 * item);
            }
            // the finished tile leaves (stores are asynchronous), the accumulators restart for the next one
            if (epi.addend != nullptr || epi.mask_y != nullptr)
                store_out_tile<MT, true>(out, bias, acc, g, cb_, co_base, cy0 + wave, cx0, lane, act, slope, epi);
            else
                store_out_tile<MT, false>(out, bias, acc, g, cb_, co_base, cy0 + wave, cx0, lane, act, slope);
            if (!nlive) break;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
            tcur = tn; cb_ = nb_; cy0 = ny0; cx0 = nx0; xcur = xnext;
#pragma unroll
            for (int q = 0; q < NPOS; ++q) in_off[q] = noff[q];
        }
#ifdef EBFI_KBENCH
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        KB_STAMP(31);
        KB_FLUSH();
        return;
    } else {
    KB_STAMP(0);
    prefetch(0);
    KB_STAMP(1);
    commit(0);
    __syncthreads();
    KB_STAMP(2);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char *base = smd + (chunk & 1) * BUFB;
        if (chunk < 6) KB_STAMP(3 + 4 * chunk);
        tap_read(base, 0, 0);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            if (tap + 1 < KK) tap_read(base, tap + 1, (tap + 1) & 1);
#ifndef KB_NO_LOADS
            if (tap < NLG) prefetch_group(chunk + 1, tap);     // past the last chunk every offset is out of range: reads 0
#endif
            __builtin_amdgcn_sched_barrier(0);
            tap_mfma(tap & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (chunk < 6) KB_STAMP(4 + 4 * chunk);
        commit((chunk & 1) ^ 1);  // unconditional (zeros after the last chunk); that buffer was last read before the previous barrier
        if (chunk < 6) KB_STAMP(5 + 4 * chunk);
        __syncthreads();
        if (chunk < 6) KB_STAMP(6 + 4 * chunk);
    }
    }
    if (epi.addend != nullptr || epi.mask_y != nullptr)
        store_out_tile<MT, true>(out, bias, acc, g, b, co_base, y0 + wave, x0, lane, act, slope, epi);
    else
        store_out_tile<MT, false>(out, bias, acc, g, b, co_base, y0 + wave, x0, lane, act, slope);
#ifdef EBFI_KBENCH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KB_STAMP(31);
    KB_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// weight gradient: slab[split][co][ci*KK + tap] partial sums; slab[split][Cout*Cin*KK + co] bias partials
constexpr int WTY = 2;                 // output rows of a contraction tile
constexpr int GSLOTS = 64, GS = GSLOTS + 1;   // grad_out image: 2 rows x 32 slots per channel, odd row stride

template <int KS, int S, int WTXO = 0>
struct WCfg {
    // output columns per tile: the staged input row must fit 32 (or 64) lanes; 3x3 stride-1 may pick 26/28/30
    // (WTXO) so that ceil(Wo / WTX) * WTX wastes as few columns as possible (128 -> 26: 1.5 % instead of 15 %)
    static constexpr int WTX = WTXO ? WTXO : ((KS == 1) ? 32 : (KS == 3) ? 30 : (S == 1 ? 26 : 28));
    static constexpr int IH = S * (WTY - 1) + KS, IW = S * (WTX - 1) + KS;
    static constexpr int IWP = IW <= 32 ? 32 : 64;              // lanes per staged input row
    static constexpr int TROWS = 256 / IWP;                     // thread rows walking (row, channel)
    static constexpr int CIB = KS == 7 ? 8 : (S == 2 ? 32 : 64);   // input channels per workgroup
    // LDS image strides with IWS == KS and PS == KS*KS (mod 32): column n = ci*KK + ky*KS + kx of the GEMM then
    // sits in bank n mod 32, so the 32 lanes of a B-operand fetch never collide
    static constexpr int IWS = IW + ((KS - IW) % 32 + 32) % 32;
    static constexpr int PS = IH * IWS + ((KS * KS - IH * IWS) % 32 + 32) % 32;
    static constexpr int NI = IH * CIB / TROWS;                 // input elements per thread per tile
    static constexpr int NTW = (CIB * KS * KS + 63) / 64;       // n-tiles (of 32 columns) per wave
    static_assert(IW <= 64 && WTX % 2 == 0 && WTX <= 32 && CIB % TROWS == 0, "tile configuration");
};

template <int KS, int S, int WTXO, int DACT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32(const float *__restrict__ x, const float *__restrict__ gout,
                                                      const float *__restrict__ yact, float *__restrict__ slab,
                                                      float *__restrict__ gpre_out, ConvGeom g, float dslope,
                                                      int total_tiles, int need_bias) {
    using C = WCfg<KS, S, WTXO>;
    constexpr int KK = KS * KS, WTX = C::WTX, IH = C::IH, IW = C::IW, IWP = C::IWP, TROWS = C::TROWS;
    constexpr int CIB = C::CIB, PS = C::PS, IWS = C::IWS, NI = C::NI, NTW = C::NTW;
    constexpr int NG = 64 / 4;             // grad_out channels per thread per tile (4 thread rows of 64 slots)
    constexpr int CPR = CIB / TROWS;       // channel steps per input row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sG = smem;                      // [64 co][GS]
    float *sIn = smem + 64 * GS;           // [CIB ci][PS] + one all-zero plane

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64, ci_base = blockIdx.z * CIB;
    const int ci_cnt = min(CIB, g.Cin - ci_base);
    const int ncols = ci_cnt * KK;         // valid (ci,tap) columns of this block
    const int mt = wave & 1, nh = wave >> 1;   // wave: co tile mt, n-tiles nh, nh+2, nh+4, ...
    const int tiles_x = (g.Wo + WTX - 1) / WTX, tiles_y = (g.Ho + WTY - 1) / WTY;
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;

    f32x16 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // per-lane LDS offset of column n = (nh + 2*q)*32 + (lane&31): ci*PS + ky*IWS + kx
    int boff[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = (nh + 2 * q) * 32 + (lane & 31);
        const int ci = n / KK, tap = n - ci * KK;
        const int ky = tap / KS, kx = tap - ky * KS;
        boff[q] = (n < ncols) ? ci * PS + ky * IWS + kx : CIB * PS;   // unused columns read the all-zero plane
    }
    for (int i = tid; i < PS; i += 256) sIn[CIB * PS + i] = 0.f;   // zero plane behind the channel planes

    // thread-fixed staging coordinates
    const int gslot = tid & 63, gpy = gslot >> 5, gpx = gslot & 31, gco = tid >> 6;      // grad_out: slot, channel row
    const int icol = tid & (IWP - 1), irow = tid / IWP;                                   // input: column, thread row
    const unsigned go_bytes = (unsigned)g.Cout * (unsigned)HWo * 4u, x_bytes = (unsigned)g.Cin * (unsigned)HW * 4u;

    float rg[NG], ri[NI];
    float bsum = 0.f;                      // bias: channel (tid & 63), pixel-slot quarter (tid >> 6), all tiles
    // grad_out * act'(y): the y values are fetched in NYG small groups spread over the MFMA block of the running
    // tile, each multiplied into rg a few k-steps later.  Loading all of them up front next to rg/ri does not fit the
    // register budget (the compiler then waits for them right away, i.e. for the whole prefetch, before the MFMAs).
    constexpr int NYG = 4, YG = NG / NYG;
    unsigned g0_next = SENT;
    __amdgpu_buffer_rsrc_t rya_next = make_rsrc(gout, 0u);
    float ry[YG];
    auto dact_load = [&](int grp) {
#pragma unroll
        for (int j = 0; j < YG; ++j)
            ry[j] = buf_ld(rya_next, g0_next + (unsigned)(4 * (grp * YG + j)) * (unsigned)HWo * 4u);
    };
    auto dact_apply = [&](int grp) {
#pragma unroll
        for (int j = 0; j < YG; ++j) rg[grp * YG + j] *= act_grad_c<DACT>(ry[j], dslope);
    };
    auto prefetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;                   // b >= B past the last tile: descriptors below get 0 records
        const int y0 = ty * WTY, x0 = tx * WTX;
        const int iy0 = S * y0 - g.pad, ix0 = S * x0 - g.pad;
        const bool live = tile < total_tiles;
        const __amdgpu_buffer_rsrc_t rgo = make_rsrc(gout + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rya = make_rsrc((DACT ? yact : gout) + (int64_t)(live ? b : 0) * g.Cout * HWo,
                                                     (live && DACT) ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rxi = make_rsrc(x + (int64_t)(live ? b : 0) * g.Cin * HW, live ? x_bytes : 0u);
        // grad_out: this thread's pixel slot for channels gco, gco+4, ...
        const int gy = y0 + gpy, gx = x0 + gpx;
        const unsigned g0 = (gpx < WTX && gy < g.Ho && gx < g.Wo)
                                ? (unsigned)((co_base + gco) * HWo + gy * g.Wo + gx) * 4u : SENT;
        g0_next = g0;
        rya_next = rya;
#pragma unroll
        for (int it = 0; it < NG; ++it)   // channels >= Cout fall out of range; the act' factor is applied later (dact_group)
            rg[it] = buf_ld(rgo, g0 + (unsigned)(4 * it) * (unsigned)HWo * 4u);
        // input halo tile: this thread's column, rows r = 0..IH-1, channels irow, irow+TROWS, ...
        const int xx = ix0 + icol;
        const bool col_ok = icol < IW && xx >= 0 && xx < g.W;
#pragma unroll
        for (int r = 0; r < IH; ++r) {
            const int yy = iy0 + r;
            const unsigned base = (col_ok && yy >= 0 && yy < g.H)
                                      ? (unsigned)((ci_base + irow) * HW + yy * g.W + xx) * 4u : SENT;
#pragma unroll
            for (int k = 0; k < CPR; ++k)
                ri[r * CPR + k] = buf_ld(rxi, base + (unsigned)(k * TROWS) * (unsigned)HW * 4u);
        }
    };
    auto commit = [&](int tile) {
#pragma unroll
        for (int it = 0; it < NG; ++it) sG[(gco + 4 * it) * GS + gslot] = rg[it];
        if (gpre_out != nullptr && blockIdx.z == 0) {   // side output: grad_out * act'(y), consumed by the data gradient
            int t = tile;
            const int tx = t % tiles_x; t /= tiles_x;
            const int ty = t % tiles_y;
            const int b = t / tiles_y;
            const int gy = ty * WTY + gpy, gx = tx * WTX + gpx;
            // descriptor over this sample's [Cout, Ho, Wo]: channels >= Cout and masked pixels are dropped by the hardware
            const __amdgpu_buffer_rsrc_t rgp = make_rsrc(gpre_out + (int64_t)b * g.Cout * HWo, go_bytes);
            const unsigned o0 = (gpx < WTX && gy < g.Ho && gx < g.Wo) ? (unsigned)((co_base + gco) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
            for (int it = 0; it < NG; ++it) buf_st(rgp, o0 + (unsigned)(4 * it) * (unsigned)HWo * 4u, rg[it]);
        }
        if (icol < IW) {
#pragma unroll
            for (int r = 0; r < IH; ++r)
#pragma unroll
                for (int k = 0; k < CPR; ++k) sIn[(irow + k * TROWS) * PS + r * IWS + icol] = ri[r * CPR + k];
        }
    };

    prefetch(blockIdx.x);
    if constexpr (DACT != 0) {             // first tile: no MFMA block to hide behind yet
#pragma unroll
        for (int grp = 0; grp < NYG; ++grp) {
            dact_load(grp);
            dact_apply(grp);
        }
    }
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        __syncthreads();
        commit(tile);
        __syncthreads();
        prefetch(tile + gridDim.x);      // past the end: zero-record descriptors, nothing is read
        if (need_bias && blockIdx.z == 0) {   // all 256 threads: 16 of the 64 pixel slots of one channel each
            const float *gp = sG + (tid & 63) * GS + (tid >> 6) * 16;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int p = 0; p < 16; p += 2) { s0 += gp[p]; s1 += gp[p + 1]; }
            bsum += s0 + s1;
        }
        // k = output pixel; lane half h takes pixel 2*ks + h (same row: WTX is even)
        const float *ap = sG + (mt * 32 + (lane & 31)) * GS + (lane >> 5);
        const int bh = S * (lane >> 5);
        auto ksteps = [&](int k0, int k1) {
#pragma unroll
            for (int ks = k0; ks < k1; ++ks) {
                const int py = (2 * ks) / WTX, px = (2 * ks) % WTX;
                const float a = ap[py * 32 + px];
                const int poff = (S * py) * IWS + S * px + bh;
#pragma unroll
                for (int q = 0; q < NTW; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sIn[boff[q] + poff], acc[q], 0, 0, 0);
                }
            }
        };
        constexpr int KSTEPS = WTY * WTX / 2;
        if constexpr (DACT != 0) {
            constexpr int SEG = KSTEPS / (NYG + 1);       // k-steps between a group's loads and its multiply
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < NYG; ++grp) {
                dact_load(grp);
                __builtin_amdgcn_sched_barrier(0);
                ksteps(grp * SEG, (grp + 1) * SEG);
                __builtin_amdgcn_sched_barrier(0);
                dact_apply(grp);
                __builtin_amdgcn_sched_barrier(0);
            }
            ksteps(NYG * SEG, KSTEPS);
        } else {
            ksteps(0, KSTEPS);
        }
    }
    // ---- write this workgroup's partial slab
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    float *my = slab + (int64_t)blockIdx.x * (wsz + g.Cout);
    const __amdgpu_buffer_rsrc_t rsl = make_rsrc(my, (unsigned)wsz * 4u);   // rows co >= Cout fall past the slab: dropped
    const unsigned co_row = (unsigned)(g.Cin * KK) * 4u;
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = (nh + 2 * q) * 32 + (lane & 31);
        const unsigned o0 = n < ncols ? (unsigned)(((co_base + mt * 32 + 4 * (lane >> 5)) * g.Cin + ci_base) * KK + n) * 4u : SENT;
#pragma unroll
        for (int r = 0; r < 16; ++r) buf_st(rsl, o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * co_row, acc[q][r]);
    }
    if (need_bias && blockIdx.z == 0) {    // combine the four slot quarters in a fixed order
        __syncthreads();
        sG[tid] = bsum;
        __syncthreads();
        if (tid < 64 && co_base + tid < g.Cout)
            my[wsz + co_base + tid] = (sG[tid] + sG[tid + 64]) + (sG[tid + 128] + sG[tid + 192]);
    }
}

// ------------------------------------------------------------------------------------------------
// Split-precision weight gradient (KS in {1,3}, stride 1).  Same GEMM view and the same LDS images as
// conv_wgrad_f32 -- grad_out [64 co][2 rows x 32 slots], input halo tile [ci][row][col] with the conflict-free
// strides of WCfg -- but every 32-bit LDS word holds the PAIR (bf16 hi << 16 | bf16 lo) of its fp32 value, split
// once at commit time.  A v_mfma_f32_32x32x16_bf16 lane needs 8 consecutive k = 8 consecutive pixel slots: it reads
// the 8 words (any alignment: they are dwords, so the kx-shifted windows need no extra copies), and two v_perm_b32
// per word pair peel the hi and the lo vector apart.  Products accumulate as lo*hi + hi*lo + hi*hi in fp32.
// The matrix work per tile drops to 3/16 of the fp32 kernel's, so the tile loop is paced by staging: 512 threads
// (8 waves: 2 co tiles x 4 column groups) share one 121 KB double-buffered image per CU -- tile t+1 is committed
// into the other buffer right after the MFMAs of tile t (one barrier per tile) while its global loads, issued before
// those MFMAs, are in flight.  Slab layout and the reduce kernel are those of the fp32 path.
constexpr int WXT = 512;

__device__ __forceinline__ unsigned split_word(float v) {
    const __bf16 h = (__bf16)v;
    const float hf = (float)h;
    return pack_bf16(v - hf, hf);          // low half: remainder, high half: leading 8 significant bits
}
__device__ __forceinline__ float join_word(unsigned w) { return __uint_as_float(w & 0xffff0000u) + __uint_as_float(w << 16); }

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

template <int KS> struct X3CIB { static constexpr int value = KS == 7 ? 16 : 64; };   // input channels per workgroup

// WX = threads per workgroup, CIBT = input channels per workgroup (0: the default of X3CIB).  <512, 64>: one workgroup
// of 8 waves per CU (121 KB of LDS).  <256, 32>: 78 KB, TWO workgroups per CU -- each keeps its own tile in flight and
// they run out of phase, so one's staging loads / commit overlap the other's matrix block (the tile loop of the
// one-per-CU form is paced by load latency: see the note at `Stage`); the grad_out tile is then fetched once per
// 32-channel block instead of once per 64.
template <int KS, int WTXO, int DACT, int WX, int CIBT>
__global__ __launch_bounds__(WX) void conv_wgrad_x3(const float *__restrict__ x, const float *__restrict__ gout,
                                                     const float *__restrict__ yact, float *__restrict__ slab,
                                                     float *__restrict__ gpre_out, ConvGeom g, float dslope, int total_tiles,
                                                     int need_bias) {
    using C = WCfg<KS, 1, WTXO>;
    constexpr int KK = KS * KS, WTX = C::WTX, IH = C::IH, IW = C::IW;
    constexpr int IWP = 32;                // lanes per staged input row; a 32-px tile has IW = 34: columns 32, 33 go separately
    constexpr int EXC = IW > IWP ? IW - IWP : 0;
    constexpr int CIB = CIBT ? CIBT : X3CIB<KS>::value, PS = C::PS, IWS = C::IWS;   // 7x7: 16 input channels (784 columns) per workgroup
    constexpr int WXT = WX, NW64 = WX / 64, NQ = WX / 128;   // thread rows of 64 slots; column groups of the waves
    constexpr int TROWS = WXT / IWP;       // thread rows walking (row, channel) of the input tile
    constexpr int CPR = CIB / TROWS;       // channel steps per input row
    constexpr int NI = IH * CPR;           // input elements per thread per tile
    constexpr int NG = 64 / (WXT / 64);    // grad_out channels per thread per tile (8 thread rows of 64 slots)
    constexpr int NTW = (CIB * KK + 32 * NQ - 1) / (32 * NQ);   // n-tiles (of 32 columns) per wave: NQ column groups
    constexpr int BUF = 64 * GS + (CIB + 1) * PS; // words per buffer: grad_out image, channel planes, zero plane
    constexpr int NEX = (EXC * IH * CIB + WXT - 1) / WXT;   // extra-column elements per thread
    static_assert(CIB % TROWS == 0 && WTX <= 32 && IWS >= IW, "tile configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned smw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64, ci_base = blockIdx.z * CIB;   // ci_base: inside the block's group
    const int grp = co_base / (g.Cout / g.groups);                     // grouped convolution: input channels of this group
    const int ci_cnt = min(CIB, g.Cin - ci_base);
    const int ncols = ci_cnt * KK;
    const int mt = wave & 1, nq = wave >> 1;   // wave: co tile mt, n-tiles nq, nq+NQ, nq+2*NQ, ...
    const int tiles_x = (g.Wo + WTX - 1) / WTX, tiles_y = (g.Ho + WTY - 1) / WTY;
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;

    f32x16 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // per-lane word offset of column n = (nq + 4q)*32 + (lane&31) inside a buffer, at k-half (lane>>5)
    int boff[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = (nq + NQ * q) * 32 + (lane & 31);
        const int ci = n / KK, tap = n - ci * KK;
        const int ky = tap / KS, kx = tap - ky * KS;
        boff[q] = 64 * GS + ((n < ncols) ? ci * PS + ky * IWS + kx : CIB * PS) + 8 * (lane >> 5);
    }
    const int aoff = (mt * 32 + (lane & 31)) * GS + 8 * (lane >> 5);
    const bool last_live = (nq + NQ * (NTW - 1)) * 32 < ncols;   // e.g. 18 n-tiles over 4 column groups: 5, 5, 4, 4
    // both buffers start as zero words (= +0.0 pairs): pad slots / pad columns / the zero plane are never written later
    for (int i = tid; i < 2 * BUF; i += WXT) smw[i] = 0u;

    // thread-fixed staging coordinates
    const int gslot = tid & 63, gpy = gslot >> 5, gpx = gslot & 31, gco = tid >> 6;   // grad_out: slot, channel row (0..7)
    const int icol = tid & (IWP - 1), irow = tid / IWP;                              // input: column, thread row (0..15)
    const unsigned go_bytes = (unsigned)g.Cout * (unsigned)HWo * 4u, x_bytes = (unsigned)g.Cin * (unsigned)HW * 4u;

    // registers of one tile in flight between its global loads and its commit to LDS
    struct Stage {
        float rg[NG], ry[DACT != 0 ? NG : 1], ri[NI];
        float rex[NEX > 0 ? NEX : 1];      // 32-px tiles: the IW - 32 extra halo columns, element e = tid + i*WXT ->
                                           // (channel e % CIB, row (e / CIB) / EXC, column 32 + (e / CIB) % EXC)
    };
    // The tile loop is paced by the LATENCY of its staging loads (ablations in tools/kbench at 64->64, 128x128, B=8: without
    // any matrix work or LDS operand read the kernel still takes 43 of its 46 us; without the loads 37; with neither 27 --
    // one tile = 51 KB per CU in flight against ~3 us of load latency under load).  PF2 keeps TWO tiles in flight in
    // registers; measured SLOWER (69 vs 53 us: the second register set pushes the kernel to 256 VGPRs + 15 spills), as the
    // round-1 attempt was -- left selectable for the day the matrix block is restructured to need fewer registers.
    constexpr bool PF2 = false;
    Stage sa, sb;
    float bacc[NG];                        // bias: this thread's slot of channels gco + 8*it, summed over its tiles
#pragma unroll
    for (int it = 0; it < NG; ++it) bacc[it] = 0.f;
    auto prefetch = [&](int tile, Stage &s) {
        float (&rg)[NG] = s.rg;
        float (&ri)[NI] = s.ri;
        auto &ry = s.ry;
        auto &rex = s.rex;
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int y0 = ty * WTY, x0 = tx * WTX;
        const int iy0 = y0 - g.pad, ix0 = x0 - g.pad;
        const bool live = tile < total_tiles;
        const __amdgpu_buffer_rsrc_t rgo = make_rsrc(gout + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rya = make_rsrc((DACT ? yact : gout) + (int64_t)(live ? b : 0) * g.Cout * HWo,
                                                     (live && DACT) ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rxi = make_rsrc(x + ((int64_t)(live ? b : 0) * g.groups + grp) * g.Cin * HW, live ? x_bytes : 0u);
        const int gy = y0 + gpy, gx = x0 + gpx;
        const unsigned g0 = (gpx < WTX && gy < g.Ho && gx < g.Wo) ? (unsigned)((co_base + gco) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const unsigned o = g0 + (unsigned)(NW64 * it) * (unsigned)HWo * 4u;   // channels >= Cout fall out of range
            rg[it] = buf_ld(rgo, o);
            if constexpr (DACT != 0) ry[it] = buf_ld(rya, o);
        }
        const int xx = ix0 + icol;
        const bool col_ok = icol < IW && xx >= 0 && xx < g.W;
#pragma unroll
        for (int r = 0; r < IH; ++r) {
            const int yy = iy0 + r;
            const unsigned base = (col_ok && yy >= 0 && yy < g.H) ? (unsigned)((ci_base + irow) * HW + yy * g.W + xx) * 4u : SENT;
#pragma unroll
            for (int k = 0; k < CPR; ++k) ri[r * CPR + k] = buf_ld(rxi, base + (unsigned)(k * TROWS) * (unsigned)HW * 4u);
        }
        if constexpr (EXC > 0) {
#pragma unroll
            for (int i = 0; i < NEX; ++i) {
                const int e = tid + i * WXT, rc = e / CIB;
                const int yy = iy0 + rc / EXC, xe = ix0 + IWP + rc % EXC;
                const bool ok = e < EXC * IH * CIB && yy >= 0 && yy < g.H && xe >= 0 && xe < g.W;
                rex[i] = buf_ld(rxi, ok ? (unsigned)((ci_base + (e & (CIB - 1))) * HW + yy * g.W + xe) * 4u : SENT);
            }
        }
    };
    auto commit = [&](int tile, int buf, Stage &s) {
        float (&rg)[NG] = s.rg;
        float (&ri)[NI] = s.ri;
        auto &ry = s.ry;
        auto &rex = s.rex;
        unsigned *sG = smw + buf * BUF, *sIn = sG + 64 * GS;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            if constexpr (DACT != 0) rg[it] *= act_grad_c<DACT>(ry[it], dslope);
            bacc[it] += rg[it];
            sG[(gco + NW64 * it) * GS + gslot] = split_word(rg[it]);
        }
        if (gpre_out != nullptr && blockIdx.z == 0) {   // side output: grad_out * act'(y), consumed by the data gradient
            int t = tile;
            const int tx = t % tiles_x; t /= tiles_x;
            const int ty = t % tiles_y;
            const int b = t / tiles_y;
            const int gy = ty * WTY + gpy, gx = tx * WTX + gpx;
            const bool live = tile < total_tiles;
            // descriptor over this sample's [Cout, Ho, Wo]: channels >= Cout and masked pixels are dropped by the hardware
            const __amdgpu_buffer_rsrc_t rgp = make_rsrc(gpre_out + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
            const unsigned o0 = (gpx < WTX && gy < g.Ho && gx < g.Wo) ? (unsigned)((co_base + gco) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
            for (int it = 0; it < NG; ++it) buf_st(rgp, o0 + (unsigned)(NW64 * it) * (unsigned)HWo * 4u, rg[it]);
        }
        if (icol < IW) {
#pragma unroll
            for (int r = 0; r < IH; ++r)
#pragma unroll
                for (int k = 0; k < CPR; ++k) sIn[(irow + k * TROWS) * PS + r * IWS + icol] = split_word(ri[r * CPR + k]);
        }
        if constexpr (EXC > 0) {
#pragma unroll
            for (int i = 0; i < NEX; ++i) {
                const int e = tid + i * WXT, rc = e / CIB;
                if (e < EXC * IH * CIB) sIn[(e & (CIB - 1)) * PS + (rc / EXC) * IWS + IWP + rc % EXC] = split_word(rex[i]);
            }
        }
    };

    __syncthreads();                       // zero fill done
    auto matrix_block = [&](int cur) {
        const unsigned *sA = smw + cur * BUF + aoff;
        const unsigned *sB = smw + cur * BUF;
#ifdef KB_NO_LDSREAD
        if (g.pad != 12345) { } else
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {   // 16 pixel slots per step: row kk>>1, slots (kk&1)*16 + 8h .. +7
            const int row = kk >> 1, px0 = (kk & 1) * 16;
            unsigned aw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) aw[j] = sA[row * 32 + px0 + j];
            bf16x8 ah, al;
            peel(aw, ah, al);
#pragma unroll
            for (int q = 0; q < NTW; ++q) {
                if (q == NTW - 1 && !last_live) continue;   // wave-uniform: this wave's last n-tile lies past the columns
                unsigned bw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bw[j] = sB[boff[q] + row * IWS + px0 + j];
                bf16x8 bh, bl;
                peel(bw, bh, bl);
#ifdef KB_NO_MFMA        // ablation (tools/kbench only): operand reads and peels stay alive, no matrix work
                asm volatile("" ::"v"(al), "v"(ah), "v"(bl), "v"(bh));
#else
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[q], 0, 0, 0);
#endif
            }
        }
    };
    const int G = gridDim.x;
    prefetch(blockIdx.x, sa);
    commit(blockIdx.x, 0, sa);
    __syncthreads();
    int cur = 0;
    if constexpr (PF2) {
        // two tiles in flight: while tile t is multiplied, the loads of t+G (issued one iteration ago) and t+2G are
        // outstanding; t+G is committed after the matrix block and its registers take the loads of t+3G
        prefetch(blockIdx.x + G, sa);
        prefetch(blockIdx.x + 2 * G, sb);
        for (int tile = blockIdx.x; tile < total_tiles; tile += 2 * G) {
            __builtin_amdgcn_sched_barrier(0);
            matrix_block(cur);
            __builtin_amdgcn_sched_barrier(0);
            commit(tile + G, cur ^ 1, sa);          // the other buffer was last read before the previous barrier
#ifndef KB_NO_LOADS
            prefetch(tile + 3 * G, sa);             // past the end: zero-record descriptors, nothing is read
#endif
            __syncthreads();
            cur ^= 1;
            if (tile + G >= total_tiles) break;
            __builtin_amdgcn_sched_barrier(0);
            matrix_block(cur);
            __builtin_amdgcn_sched_barrier(0);
            commit(tile + 2 * G, cur ^ 1, sb);
#ifndef KB_NO_LOADS
            prefetch(tile + 4 * G, sb);
#endif
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (int tile = blockIdx.x; tile < total_tiles; tile += G) {
#ifndef KB_NO_LOADS
            prefetch(tile + G, sa);            // past the end: zero-record descriptors, nothing is read
#endif
            __builtin_amdgcn_sched_barrier(0); // keep the loads above the matrix block
            matrix_block(cur);
            __builtin_amdgcn_sched_barrier(0);
            commit(tile + G, cur ^ 1, sa);     // the other buffer was last read before the previous barrier
            __syncthreads();
            cur ^= 1;
        }
    }
    // ---- write this workgroup's partial slab
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    float *my = slab + (int64_t)blockIdx.x * (wsz + g.Cout);
    const __amdgpu_buffer_rsrc_t rsl = make_rsrc(my, (unsigned)wsz * 4u);   // rows co >= Cout fall past the slab: dropped
    const unsigned co_row = (unsigned)(g.Cin * KK) * 4u;
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = (nq + NQ * q) * 32 + (lane & 31);
        const unsigned o0 = n < ncols ? (unsigned)(((co_base + mt * 32 + 4 * (lane >> 5)) * g.Cin + ci_base) * KK + n) * 4u : SENT;
#ifdef KB_NO_SLAB
        if (g.pad == 12345)
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) buf_st(rsl, o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * co_row, acc[q][r]);
    }
    if (need_bias && blockIdx.z == 0) {    // lanes of a wave hold the 64 slots of channels gco + 8*it: fixed-order butterfly
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            float v = bacc[it];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0 && co_base + gco + NW64 * it < g.Cout) my[wsz + co_base + gco + NW64 * it] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_x3_ws: the 3x3 split-precision weight gradient with SPECIALISED waves.  In conv_wgrad_x3 every wave stages,
// commits and multiplies in turn, and the ablations (tools/kbench) showed the four phases of a tile adding up: at 64 -> 128
// 38 us of commit / slab / reduce + 14 of loads + 10 of LDS operand reads + 21 of matrix work, nothing overlapping, a second
// resident workgroup changing little.  Here waves 4..7 (256 threads) are PRODUCERS: they keep two tiles of global loads in
// flight (they hold no accumulators, so the second register set that spilled in conv_wgrad_x3 fits), split and write the
// next tile's images; waves 0..3, one per SIMD, are CONSUMERS: each owns BOTH 32-row output tiles of its n-tiles (every
// B-operand read serves two MFMA triples) and does nothing but operand reads, peels and MFMAs.  One workgroup barrier per
// tile hands the double-buffered image from one side to the other; the two roles run separate loops with the same number
// of barriers, so the register allocation is the maximum of the two paths, not their sum.
template <int DACT>
__global__ __launch_bounds__(512) void conv_wgrad_x3_ws(const float *__restrict__ x, const float *__restrict__ gout,
                                                        const float *__restrict__ yact, float *__restrict__ slab,
                                                        float *__restrict__ gpre_out, ConvGeom g, float dslope, int total_tiles,
                                                        int need_bias) {
    using C = WCfg<3, 1, 32>;
    constexpr int KS = 3, KK = 9, WTX = C::WTX, IH = C::IH, IW = C::IW;
    constexpr int IWP = 32, EXC = IW - IWP, CIB = 64, PS = C::PS, IWS = C::IWS;
    constexpr int PT = 256, NW64 = PT / 64;                // producer threads; thread rows of 64 grad_out slots
    constexpr int TROWS = PT / IWP, CPR = CIB / TROWS, NI = IH * CPR, NG = 64 / NW64;
    constexpr int NEX = (EXC * IH * CIB + PT - 1) / PT;
    constexpr int NQ = 4, NTW = (CIB * KK + 32 * NQ - 1) / (32 * NQ);   // consumer waves = column groups; n-tiles per wave
    constexpr int BUF = 64 * GS + (CIB + 1) * PS;
    static_assert(WTX == 32 && EXC == 2 && CIB % TROWS == 0, "tile configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned smw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64, ci_base = blockIdx.z * CIB;
    const int grp = co_base / (g.Cout / g.groups);
    const int ci_cnt = min(CIB, g.Cin - ci_base);
    const int ncols = ci_cnt * KK;
    const int tiles_x = (g.Wo + WTX - 1) / WTX, tiles_y = (g.Ho + WTY - 1) / WTY;
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    const int G = gridDim.x;
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    float *my = slab + (int64_t)blockIdx.x * (wsz + g.Cout);

    for (int i = tid; i < 2 * BUF; i += 512) smw[i] = 0u;   // pad slots, pad columns and the zero plane stay zero
    __syncthreads();

    if (wave < NQ) {
        // ------------------------------------------------------------------ consumers
        // 18 n-tiles x 2 output tiles = 36 blocks, 9 per wave: n-tiles nq, nq+4, nq+8, nq+12 with BOTH output tiles, plus output
        // tile (nq & 1) of n-tile 16 + (nq >> 1)  (5 / 5 / 4 / 4 whole n-tiles left two waves waiting a fifth of the time)
        const int nq = wave;
        constexpr int NTF = NTW - 1;       // n-tiles held with both output tiles
        static_assert(NTW == 5, "block distribution assumes 18 n-tiles over 4 waves");
        const int xm = nq & 1, xt = 16 + (nq >> 1);
        f32x16 acc[2][NTF], accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NTF; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        auto col_off = [&](int n) {
            const int ci = n / KK, tap = n - ci * KK;
            const int ky = tap / KS, kx = tap - ky * KS;
            return 64 * GS + ((n < ncols) ? ci * PS + ky * IWS + kx : CIB * PS) + 8 * (lane >> 5);
        };
        int boff[NTW];
#pragma unroll
        for (int q = 0; q < NTF; ++q) boff[q] = col_off((nq + NQ * q) * 32 + (lane & 31));
        boff[NTF] = col_off(xt * 32 + (lane & 31));
        const int aoff = (lane & 31) * GS + 8 * (lane >> 5);
        __syncthreads();                   // (A) the first tile is committed
        int cur = 0;
        for (int tile = blockIdx.x; tile < total_tiles; tile += G) {
            const unsigned *sA = smw + cur * BUF + aoff;
            const unsigned *sB = smw + cur * BUF;
#ifdef WS_NO_CONSUME
            if (g.pad != 12345) { } else
#endif
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int row = kk >> 1, px0 = (kk & 1) * 16;
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    unsigned aw[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) aw[j] = sA[m * 32 * GS + row * 32 + px0 + j];
                    peel(aw, ah[m], al[m]);
                }
                // the operand words of n-tile q+1 are requested before the MFMAs of n-tile q are issued: with one consumer
                // wave per SIMD nothing else covers the LDS latency
                unsigned bw[2][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bw[0][j] = sB[boff[0] + row * IWS + px0 + j];
#pragma unroll
                for (int q = 0; q < NTW; ++q) {
#ifndef WS_NO_LDSREAD
                    if (q + 1 < NTW) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bw[(q + 1) & 1][j] = sB[boff[q + 1] + row * IWS + px0 + j];
                    }
#else
                    if (q + 1 < NTW) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bw[(q + 1) & 1][j] = bw[q & 1][j] + (unsigned)boff[q + 1];
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 bh, bl;
                    peel(bw[q & 1], bh, bl);
                    if (q < NTF) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
#ifndef ABL_ONE_MFMA
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m][q], 0, 0, 0);
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m][q], 0, 0, 0);
#endif
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m][q], 0, 0, 0);
                        }
                    } else {               // the ninth block: one output tile of the extra n-tile
                        const bf16x8 xh = xm ? ah[1] : ah[0], xl = xm ? al[1] : al[0];
#ifndef ABL_ONE_MFMA
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, accx, 0, 0, 0);
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, accx, 0, 0, 0);
#endif
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, accx, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();               // (B) this image has been read, the other one is complete
            cur ^= 1;
        }
        const __amdgpu_buffer_rsrc_t rsl = make_rsrc(my, (unsigned)wsz * 4u);   // rows co >= Cout fall past the slab: dropped
        const unsigned co_row = (unsigned)(g.Cin * KK) * 4u;
        auto store_block = [&](const f32x16 &a, int m, int ntile) {
            const int n = ntile * 32 + (lane & 31);
            const unsigned o0 = n < ncols ? (unsigned)(((co_base + m * 32 + 4 * (lane >> 5)) * g.Cin + ci_base) * KK + n) * 4u : SENT;
#pragma unroll
            for (int r = 0; r < 16; ++r) buf_st(rsl, o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * co_row, a[r]);
        };
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < NTF; ++q) store_block(acc[m][q], m, nq + NQ * q);
        store_block(accx, xm, xt);
        return;
    }
    // ---------------------------------------------------------------------- producers
    const int ptid = tid - 64 * NQ;
    const int gslot = ptid & 63, gpy = gslot >> 5, gpx = gslot & 31, gco = ptid >> 6;
    const int icol = ptid & (IWP - 1), irow = ptid / IWP;
    const unsigned go_bytes = (unsigned)g.Cout * (unsigned)HWo * 4u, x_bytes = (unsigned)g.Cin * (unsigned)HW * 4u;
    struct Stage {
        float rg[NG], ry[DACT != 0 ? NG : 1], ri[NI], rex[NEX];
    };
    Stage sa, sb;
    float bacc[NG];
#pragma unroll
    for (int it = 0; it < NG; ++it) bacc[it] = 0.f;
    auto prefetch = [&](int tile, Stage &s) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int y0 = ty * WTY, x0 = tx * WTX;
        const int iy0 = y0 - g.pad, ix0 = x0 - g.pad;
        const bool live = tile < total_tiles;
        const __amdgpu_buffer_rsrc_t rgo = make_rsrc(gout + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rya = make_rsrc((DACT ? yact : gout) + (int64_t)(live ? b : 0) * g.Cout * HWo,
                                                     (live && DACT) ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rxi = make_rsrc(x + ((int64_t)(live ? b : 0) * g.groups + grp) * g.Cin * HW, live ? x_bytes : 0u);
        const int gy = y0 + gpy, gx = x0 + gpx;
        const unsigned g0 = (gpx < WTX && gy < g.Ho && gx < g.Wo) ? (unsigned)((co_base + gco) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const unsigned o = g0 + (unsigned)(NW64 * it) * (unsigned)HWo * 4u;
            s.rg[it] = buf_ld(rgo, o);
            if constexpr (DACT != 0) s.ry[it] = buf_ld(rya, o);
        }
        const int xx = ix0 + icol;
        const bool col_ok = icol < IW && xx >= 0 && xx < g.W;
#pragma unroll
        for (int r = 0; r < IH; ++r) {
            const int yy = iy0 + r;
            const unsigned base = (col_ok && yy >= 0 && yy < g.H) ? (unsigned)((ci_base + irow) * HW + yy * g.W + xx) * 4u : SENT;
#pragma unroll
            for (int k = 0; k < CPR; ++k) s.ri[r * CPR + k] = buf_ld(rxi, base + (unsigned)(k * TROWS) * (unsigned)HW * 4u);
        }
#pragma unroll
        for (int i = 0; i < NEX; ++i) {
            const int e = ptid + i * PT, rc = e / CIB;
            const int yy = iy0 + rc / EXC, xe = ix0 + IWP + rc % EXC;
            const bool ok = e < EXC * IH * CIB && yy >= 0 && yy < g.H && xe >= 0 && xe < g.W;
            s.rex[i] = buf_ld(rxi, ok ? (unsigned)((ci_base + (e & (CIB - 1))) * HW + yy * g.W + xe) * 4u : SENT);
        }
    };
    auto commit = [&](int tile, int buf, Stage &s) {
        unsigned *sG = smw + buf * BUF, *sIn = sG + 64 * GS;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            if constexpr (DACT != 0) s.rg[it] *= act_grad_c<DACT>(s.ry[it], dslope);
            bacc[it] += s.rg[it];
            sG[(gco + NW64 * it) * GS + gslot] = split_word(s.rg[it]);
        }
        if (gpre_out != nullptr && blockIdx.z == 0) {
            int t = tile;
            const int tx = t % tiles_x; t /= tiles_x;
            const int ty = t % tiles_y;
            const int b = t / tiles_y;
            const int gy = ty * WTY + gpy, gx = tx * WTX + gpx;
            const bool live = tile < total_tiles;
            const __amdgpu_buffer_rsrc_t rgp = make_rsrc(gpre_out + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
            const unsigned o0 = (gpx < WTX && gy < g.Ho && gx < g.Wo) ? (unsigned)((co_base + gco) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
            for (int it = 0; it < NG; ++it) buf_st(rgp, o0 + (unsigned)(NW64 * it) * (unsigned)HWo * 4u, s.rg[it]);
        }
        if (icol < IW) {
#pragma unroll
            for (int r = 0; r < IH; ++r)
#pragma unroll
                for (int k = 0; k < CPR; ++k) sIn[(irow + k * TROWS) * PS + r * IWS + icol] = split_word(s.ri[r * CPR + k]);
        }
#pragma unroll
        for (int i = 0; i < NEX; ++i) {
            const int e = ptid + i * PT, rc = e / CIB;
            if (e < EXC * IH * CIB) sIn[(e & (CIB - 1)) * PS + (rc / EXC) * IWS + IWP + rc % EXC] = split_word(s.rex[i]);
        }
    };
    prefetch(blockIdx.x, sa);
    commit(blockIdx.x, 0, sa);
    prefetch(blockIdx.x + G, sa);          // two tiles of loads in flight from here on
    prefetch(blockIdx.x + 2 * G, sb);
    __syncthreads();                       // (A)
    int cur = 0;
    for (int tile = blockIdx.x; tile < total_tiles; tile += 2 * G) {
#ifndef WS_NO_PRODUCE
        commit(tile + G, cur ^ 1, sa);     // while the consumers multiply tile `tile` from image `cur`
        prefetch(tile + 3 * G, sa);        // (past the end: zero-record descriptors, nothing is read)
#endif
        __syncthreads();                   // (B)
        cur ^= 1;
        if (tile + G >= total_tiles) break;
#ifndef WS_NO_PRODUCE
        commit(tile + 2 * G, cur ^ 1, sb);
        prefetch(tile + 4 * G, sb);
#endif
        __syncthreads();                   // (B)
        cur ^= 1;
    }
    if (need_bias && blockIdx.z == 0) {    // lanes of a wave hold the 64 slots of channels gco + 4*it: fixed-order butterfly
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            float v = bacc[it];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0 && co_base + gco + NW64 * it < g.Cout) my[wsz + co_base + gco + NW64 * it] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 matrix-core weight gradient (KS in {1,3}, stride 1): M = out channels, N = (tap, ci), K = pixels.
// MFMA wants 8 consecutive k = 8 consecutive PIXELS per lane.  NCHW gives exactly that for grad_out
// ([co][2 rows x 32 slots], pitch 72 = 9*16 B).  For the input operand the 8 pixels start at x + kx, which
// is only 16-byte aligned for kx = 0 -- so the halo tile is stored KS times, shifted by kx
// (copy_kx[ci][r][x] = in[ci][r][x+kx]); every B fragment is then one aligned ds_read_b128 and the 32
// lanes of an n-tile (32 consecutive channels of one tap) step by an odd multiple of 16 B: conflict-free.
// Staging keeps fp32 -> bf16 conversion in registers: a thread owns a pixel PAIR, so copies 0 and 2 are
// plain packed 32-bit stores and copy 1 takes its second half from the next lane (one shuffle).
template <int KS>
struct WCfgB {
    static constexpr int WTX = WCfg<KS, 1>::WTX, IH = WTY - 1 + KS, IW = WTX - 1 + KS;   // IW == 32
    static constexpr int CIB = 32, ROWP = 40;                       // channels per workgroup; row pitch (elements)
    static constexpr int PLANE = IH * ROWP + ((IH * 5) % 2 == 0 ? 8 : 0);   // odd multiple of 16 bytes
    static constexpr int GP = 72;                                   // grad_out pitch (elements)
    static constexpr int NI = IH * CIB / 16;                        // input pixel pairs per thread per tile
    static constexpr int NTAP = (KS * KS + 1) / 2;                  // taps per wave (two wave groups split the taps)
    static constexpr int LDS_ELEMS = 64 * GP + KS * CIB * PLANE + ROWP;
    static_assert(IW == 32, "staging assumes 16 pixel pairs per row");
};

template <int KS, int DACT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16(const float *__restrict__ x, const float *__restrict__ gout,
                                                         const float *__restrict__ yact, float *__restrict__ slab,
                                                         ConvGeom g, float dslope, int total_tiles, int need_bias) {
    using C = WCfgB<KS>;
    constexpr int KK = KS * KS, WTX = C::WTX, CIB = C::CIB, ROWP = C::ROWP, PLANE = C::PLANE, GP = C::GP;
    constexpr int NI = C::NI, NTAP = C::NTAP;
    extern __shared__ __attribute__((aligned(16))) __bf16 smemb[];
    __bf16 *sG = smemb;                    // [64 co][GP]
    __bf16 *sB = smemb + 64 * GP;          // [KS copies][CIB][PLANE]
    unsigned *sG32 = reinterpret_cast<unsigned *>(sG), *sB32 = reinterpret_cast<unsigned *>(sB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64, ci_base = blockIdx.z * CIB;
    const int ci_cnt = min(CIB, g.Cin - ci_base);
    const int mt = wave & 1, th = wave >> 1;   // wave: co tile mt, taps th*NTAP .. (n-tile = one tap x 32 channels)
    const int tiles_x = (g.Wo + WTX - 1) / WTX, tiles_y = (g.Ho + WTY - 1) / WTY;
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;

    f32x16 acc[NTAP];
#pragma unroll
    for (int q = 0; q < NTAP; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // pad columns of the shifted copies are never written by the staging: zero them once (they meet zero
    // grad_out slots, but must be finite)
    for (int i = tid; i < (KS * CIB * PLANE + ROWP) / 2; i += 256) sB32[i] = 0u;

    // thread-fixed staging coordinates: grad_out pixel pair gp of channel rows grow + 8*it; input pixel pair pc
    // of (row, channel) pairs walked by thread row trow
    const int gp = tid & 31, grow = tid >> 5, gpy = gp >> 4, gpx = 2 * (gp & 15);
    const int pc = tid & 15, trow = tid >> 4;
    const unsigned go_bytes = (unsigned)g.Cout * (unsigned)HWo * 4u, x_bytes = (unsigned)g.Cin * (unsigned)HW * 4u;

    float rg[16], ri[2 * NI];
    float bsum = 0.f;                      // bias: channel (tid & 63), slot quarter (tid >> 6)
    auto prefetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int y0 = ty * WTY, x0 = tx * WTX;
        const int iy0 = y0 - g.pad, ix0 = x0 - g.pad;
        const bool live = tile < total_tiles;
        const __amdgpu_buffer_rsrc_t rgo = make_rsrc(gout + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rya = make_rsrc((DACT ? yact : gout) + (int64_t)(live ? b : 0) * g.Cout * HWo,
                                                     (live && DACT) ? go_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rxi = make_rsrc(x + (int64_t)(live ? b : 0) * g.Cin * HW, live ? x_bytes : 0u);
        const int gy = y0 + gpy, gx0 = x0 + gpx;
        const bool row_ok = gpx < WTX && gy < g.Ho;
        const unsigned base = (unsigned)((co_base + grow) * HWo + gy * g.Wo + gx0) * 4u;
        const unsigned g0 = (row_ok && gx0 < g.Wo) ? base : SENT, g1 = (row_ok && gx0 + 1 < g.Wo) ? base + 4u : SENT;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const unsigned step = (unsigned)(8 * it) * (unsigned)HWo * 4u;   // channels >= Cout fall out of range
            float v0 = buf_ld(rgo, g0 + step), v1 = buf_ld(rgo, g1 + step);
            if constexpr (DACT != 0) {
                v0 *= act_grad_c<DACT>(buf_ld(rya, g0 + step), dslope);
                v1 *= act_grad_c<DACT>(buf_ld(rya, g1 + step), dslope);
            }
            rg[2 * it] = v0;
            rg[2 * it + 1] = v1;
        }
        const int xx = ix0 + 2 * pc;
        const bool c0_ok = xx >= 0 && xx < g.W, c1_ok = xx + 1 >= 0 && xx + 1 < g.W;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int r = it / (CIB / 16), ci = trow + 16 * (it % (CIB / 16));
            const int yy = iy0 + r;
            const bool r_ok = yy >= 0 && yy < g.H;
            const unsigned o = (unsigned)((ci_base + ci) * HW + yy * g.W + xx) * 4u;   // channels >= Cin: out of range
            ri[2 * it] = buf_ld(rxi, (r_ok && c0_ok) ? o : SENT);
            ri[2 * it + 1] = buf_ld(rxi, (r_ok && c1_ok) ? o + 4u : SENT);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) sG32[((grow + 8 * it) * GP) / 2 + gp] = pack_bf16(rg[2 * it], rg[2 * it + 1]);
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int r = it / (CIB / 16), ci = trow + 16 * (it % (CIB / 16));
            const float v0 = ri[2 * it], v1 = ri[2 * it + 1];
            const int e = (ci * PLANE + r * ROWP) / 2 + pc;           // 32-bit index of pixel pair pc in copy 0
            sB32[e] = pack_bf16(v0, v1);
            if constexpr (KS == 3) {
                const float nx = __shfl_down(v0, 1, 16);               // first pixel of the next pair (same row)
                sB32[(CIB * PLANE) / 2 + e] = pack_bf16(v1, pc < 15 ? nx : 0.f);          // copy 1: in[x+1]
                if (pc > 0) sB32[(2 * CIB * PLANE) / 2 + e - 1] = pack_bf16(v0, v1);       // copy 2: in[x+2]
            }
        }
    };

    prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        __syncthreads();
        commit();
        __syncthreads();
        prefetch(tile + gridDim.x);
        if (need_bias && blockIdx.z == 0) {   // 16 of the 64 pixel slots of one channel per thread (bf16-rounded values)
            const bf16x8 *gq = reinterpret_cast<const bf16x8 *>(sG + (tid & 63) * GP + (tid >> 6) * 16);
            const bf16x8 u = gq[0], v = gq[1];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)u[j] + (float)v[j];
            bsum += s;
        }
        // k-step kk = 16 pixel slots: row kk>>1, columns (kk&1)*16 + 8h .. +7
        const __bf16 *ap = sG + (mt * 32 + (lane & 31)) * GP + (lane >> 5) * 8;
        const __bf16 *bp = sB + (lane & 31) * PLANE + (lane >> 5) * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ap + kk * 16);
#pragma unroll
            for (int j = 0; j < NTAP; ++j) {
                const int q = th * NTAP + j;               // wave-uniform
                if (q < KK) {
                    const int ky = q / KS, kx = q - ky * KS;
                    const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(bp + kx * CIB * PLANE + ((kk >> 1) + ky) * ROWP +
                                                                        (kk & 1) * 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial slab: column (tap q, channel lane&31)
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    float *my = slab + (int64_t)blockIdx.x * (wsz + g.Cout);
    const int ci = lane & 31;
    if (ci < ci_cnt) {
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            const int q = th * NTAP + j;
            if (q >= KK) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.Cout) my[((int64_t)co * KK + q) * g.Cin + ci_base + ci] = acc[j][r];   // [co][tap][ci]: lanes contiguous
            }
        }
    }
    if (need_bias && blockIdx.z == 0) {    // combine the four slot quarters in a fixed order
        __syncthreads();
        float *red = reinterpret_cast<float *>(smemb);
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64 && co_base + tid < g.Cout)
            my[wsz + co_base + tid] = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
    }
}

// Table-driven packing of MANY weights in one launch: entry e of `table` names the fp32 source element of packed bf16
// element e (bits 0..29: index into `src`; bit 30: this element is the rounding remainder lo = bf16(v - float(bf16(v)))
// instead of hi = bf16(v); negative: structural zero).  The host lays the entries out as the conv kernels expect them
// ([hi image | lo image] per weight, forward and transposed layouts, channel padding, folded 3-D kernels, concatenated
// convolutions): ebfi_amd/weightbank.py.
__global__ __launch_bounds__(256) void pack_table_bf16_kernel(const float *__restrict__ src, const int32_t *__restrict__ table,
                                                              int64_t n, __bf16 *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int32_t t = table[e];
    float v = 0.f;
    if (t >= 0) v = src[t & 0x3fffffff];
    const __bf16 h = (__bf16)v;
    out[e] = (t >= 0 && (t & 0x40000000)) ? (__bf16)(v - (float)h) : h;
}

// 64 consecutive elements per workgroup, 4 thread rows each summing every 4th slab (4 loads in
// flight), then a fixed-order combine through LDS: deterministic, and short dependent chains.
// perm_cin > 0: the slabs hold the weight part as [co][tap][ci] (bf16 kernel); gw is always [co][ci][tap].
__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ slab, int nslabs, int64_t n_weight, int64_t n_total,
                                                  float *__restrict__ gw, float *__restrict__ gb, int perm_cin, int perm_kk) {
    __shared__ float part[4][64];
    const int jj = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + jj;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < n_total) {
        const float *p = slab + j;
        int k = kq;
        for (; k + 28 < nslabs; k += 32) {   // eight loads in flight per thread (fixed order of the partial sums)
            const float a0 = p[(int64_t)k * n_total], a1 = p[(int64_t)(k + 4) * n_total];
            const float a2 = p[(int64_t)(k + 8) * n_total], a3 = p[(int64_t)(k + 12) * n_total];
            const float a4 = p[(int64_t)(k + 16) * n_total], a5 = p[(int64_t)(k + 20) * n_total];
            const float a6 = p[(int64_t)(k + 24) * n_total], a7 = p[(int64_t)(k + 28) * n_total];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
            s0 += a4; s1 += a5; s2 += a6; s3 += a7;
        }
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
        if (j < n_weight) {
            int64_t o = j;
            if (perm_cin > 0) {
                const int ci = (int)(j % perm_cin);
                const int q = (int)((j / perm_cin) % perm_kk);
                const int64_t co = j / ((int64_t)perm_cin * perm_kk);
                o = (co * perm_cin + ci) * perm_kk + q;
            }
            gw[o] = s;
        } else if (gb) gb[j - n_weight] = s;
    }
}
__global__ __launch_bounds__(256) void conv_wgrad_reduce_f32(const float *__restrict__ slab, int nslabs,
                                                             int64_t n_weight, int64_t n_total,
                                                             float *__restrict__ gw, float *__restrict__ gb,
                                                             int perm_cin, int perm_kk) {
    wgrad_reduce_body(slab, nslabs, n_weight, n_total, gw, gb, perm_cin, perm_kk);
}
// MANY slabs of a SMALL gradient (the thin layers of conv2d_shift.inc.hpp: 500-700 slabs of ~2000 values): 16 consecutive elements per
// workgroup and sixteen thread rows, each summing every 16th slab with eight loads in flight -- a quarter of the dependent round
// trips of the 64 x 4 form above, on four times as many workgroups (that form ran 10-12 us on 28-37 workgroups).  Fixed order.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_tall(const float *__restrict__ slab, int nslabs, int64_t n_weight, int64_t n_total,
                                                              float *__restrict__ gw, float *__restrict__ gb) {
    __shared__ float part[16][16];
    const int jj = threadIdx.x & 15, kq = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + jj;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < n_total) {
        const float *p = slab + j;
        int k = kq;
        for (; k + 112 < nslabs; k += 128) {
            float a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = p[(int64_t)(k + 16 * i) * n_total];
            s0 += a[0]; s1 += a[1]; s2 += a[2]; s3 += a[3];
            s0 += a[4]; s1 += a[5]; s2 += a[6]; s3 += a[7];
        }
        for (; k < nslabs; k += 16) s0 += p[(int64_t)k * n_total];
    }
    part[kq][jj] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (kq == 0 && j < n_total) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += part[i][jj];
        if (j < n_weight) gw[j] = s;
        else if (gb) gb[j - n_weight] = s;
    }
}
// the reductions of a batched weight-gradient launch (conv_wgrad_f16_tr_batch): blockIdx.y = layer
struct ReduceItem {
    const float *slab;
    float *gw, *gb;
    int64_t n_weight, n_total;
    int perm_cin;
};
struct ReduceBatch {
    ReduceItem item[4];
};
__global__ __launch_bounds__(256) void conv_wgrad_reduce_batch(ReduceBatch rb, int nslabs) {
    ReduceItem it = rb.item[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if ((int)blockIdx.y == k) it = rb.item[k];
    if ((int64_t)blockIdx.x * 64 >= it.n_total) return;      // (whole workgroup: layers of different sizes share the grid)
    wgrad_reduce_body(it.slab, nslabs, it.n_weight, it.n_total, it.gw, it.gb, it.perm_cin, 9);
}

// algorithmic HBM bytes of one launch (every operand once, results once; workspaces and re-reads not counted)
double conv_bytes_fwd(const ConvGeom &g, int kk, bool dact) {
    return 4.0 * ((double)g.B * g.groups * g.Cin * g.H * g.W * (dact ? 2 : 1) + (double)g.B * g.Cout * g.Ho * g.Wo + (double)g.Cout * g.Cin * kk);
}
double conv_bytes_wgrad(const ConvGeom &g, int kk, bool dact, bool side_out) {
    return 4.0 * ((double)g.B * g.groups * g.Cin * g.H * g.W + (double)g.B * g.Cout * g.Ho * g.Wo * (1 + (dact ? 1 : 0) + (side_out ? 1 : 0)) +
                  (double)g.Cout * g.Cin * kk);
}

int make_geom(ConvGeom &g, int B, int Cin, int H, int W, int Cout, int ks, int stride, int pad) {
    if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return fail(EBFI_ERR_ARG, "conv2d: non-positive dimension");
    if (ks != 1 && ks != 3 && ks != 7) return fail(EBFI_ERR_UNSUPPORTED, "conv2d: kernel size %d (1, 3 and 7 implemented)", ks);
    if (stride != 1 && stride != 2) return fail(EBFI_ERR_UNSUPPORTED, "conv2d: stride %d (1 and 2 implemented)", stride);
    if (pad < 0 || pad > ks) return fail(EBFI_ERR_ARG, "conv2d: padding %d out of range", pad);
    g = ConvGeom{B, Cin, H, W, Cout, (H + 2 * pad - ks) / stride + 1, (W + 2 * pad - ks) / stride + 1, pad};
    if (g.Ho <= 0 || g.Wo <= 0) return fail(EBFI_ERR_ARG, "conv2d: empty output");
    // 32-bit buffer offsets with a 2^31 sentinel: one sample (plus one staged chunk) must stay below 2 GiB
    const int64_t lim = (1LL << 31) - (1LL << 26);
    if ((int64_t)(Cin + 64) * H * W * 4 >= lim || (int64_t)(Cout + 64) * g.Ho * g.Wo * 4 >= lim ||
        (int64_t)(Cout + 64) * (Cin + 64) * ks * ks * 4 >= lim)
        return fail(EBFI_ERR_ARG, "conv2d: one sample (or the weight) exceeds the 2 GiB reach of 32-bit buffer offsets");
    return EBFI_OK;
}

template <int KS, int S, bool TR, int DACT>
int launch_fwd_d(hipStream_t st, const float *x, const float *dact_y, const float *w, const float *bias, float *out,
                 const ConvGeom &g, int act, float slope, float dslope) {
    constexpr int CK = KS == 7 ? 2 : 8;      // input channels staged per chunk (LDS budget of the weight slice)
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, TY) * ceil_div(g.Wo, TX);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d: too many tiles");
    const char *name = TR ? "conv_fwd_f32/dgrad" : "conv_fwd_f32/fwd";   // one kernel, two roles
    const double flops = 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * KS * KS;   // dense, un-padded
    if (g.Cout <= 32) {
        dim3 grid((unsigned)tiles, (unsigned)ceil_div(g.Cout, 32));
        ProfScope ps(name, st, flops, conv_bytes_fwd(g, KS * KS, DACT != 0));
        hipLaunchKernelGGL((conv_fwd_f32<KS, S, 1, CK, TR, DACT>), grid, dim3(256), 0, st, x, dact_y, w, bias, out, g, act,
                           slope, dslope);
    } else {
        dim3 grid((unsigned)tiles, (unsigned)ceil_div(g.Cout, 64));
        ProfScope ps(name, st, flops, conv_bytes_fwd(g, KS * KS, DACT != 0));
        hipLaunchKernelGGL((conv_fwd_f32<KS, S, 2, CK, TR, DACT>), grid, dim3(256), 0, st, x, dact_y, w, bias, out, g, act,
                           slope, dslope);
    }
    return check_launch(name);
}

template <int KS, int S, bool TR>
int launch_fwd(hipStream_t st, const float *x, const float *dact_y, const float *w, const float *bias, float *out,
               const ConvGeom &g, int act, float slope, int dact, float dslope) {
    if constexpr (TR) {
        if (dact == ACT_LEAKY) return launch_fwd_d<KS, S, TR, ACT_LEAKY>(st, x, dact_y, w, bias, out, g, act, slope, dslope);
        if (dact == ACT_SIGMOID) return launch_fwd_d<KS, S, TR, ACT_SIGMOID>(st, x, dact_y, w, bias, out, g, act, slope, dslope);
    }
    return launch_fwd_d<KS, S, TR, ACT_NONE>(st, x, dact_y, w, bias, out, g, act, slope, dslope);
}

// tile width of the fp32 3x3 stride-1 weight-gradient kernel: fewest wasted columns among 26 / 28 / 30
int pick_wtx(int Wo) {
    int best = 30;
    int64_t best_cols = ceil_div(Wo, 30) * 30;
    for (int w : {28, 26}) {
        const int64_t cols = ceil_div(Wo, w) * w;
        if (cols < best_cols) { best = w; best_cols = cols; }
    }
    return best;
}

int wgrad_wtx_rt(const ConvGeom &g, int ks, int stride) {
    if (ks == 3 && stride == 1) return pick_wtx(g.Wo);
    if (ks == 3) return WCfg<3, 2>::WTX;
    if (ks == 1) return WCfg<1, 1>::WTX;
    return stride == 1 ? WCfg<7, 1>::WTX : WCfg<7, 2>::WTX;
}

int64_t wgrad_tiles_rt(const ConvGeom &g, int ks, int stride) {
    return (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, wgrad_wtx_rt(g, ks, stride));
}

int wgrad_cib_rt(int ks, int stride) { return ks == 7 ? 8 : (stride == 2 ? 32 : 64); }

int wgrad_splits(const ConvGeom &g, int ks, int stride, bool bf16mma = false) {
    const int64_t tiles = wgrad_tiles_rt(g, ks, stride);
    const int64_t blocks = ceil_div(g.Cout, 64) * ceil_div(g.Cin, bf16mma ? 32 : wgrad_cib_rt(ks, stride));
    // The fp32 kernel is resident at 2 workgroups per CU (registers and LDS): 512 slots.  Fill ONE round of them as
    // completely as possible -- e.g. 50 (co, ci) blocks x 10 splits = 500 -- rather than 1050 workgroups, whose third,
    // almost empty round costs a full workgroup duration.  More than 512 (co, ci) blocks: no pixel split needed.
    int64_t s = bf16mma ? ceil_div(1024, blocks) : (blocks <= 512 ? 512 / blocks : 1);
    if (s > tiles) s = tiles;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}

template <int KS, int S, int WTXO, int DACT>
int launch_wgrad_d(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                   const ConvGeom &g, float dslope, int nsplit, int need_bias) {
    using C = WCfg<KS, S, WTXO>;
    const size_t lds = (size_t)(64 * GS + (C::CIB + 1) * C::PS + 64) * sizeof(float);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_f32<KS, S, WTXO, DACT>), (int)lds)) return rc;
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, C::WTX);
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, C::CIB));
    ProfScope ps("conv_wgrad_f32", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * KS * KS,
                 conv_bytes_wgrad(g, KS * KS, DACT != 0, gpre_out != nullptr));
    hipLaunchKernelGGL((conv_wgrad_f32<KS, S, WTXO, DACT>), grid, dim3(256), lds, st, x, gout, yact, slab, gpre_out, g, dslope,
                       (int)tiles, need_bias);
    return check_launch("conv_wgrad_f32");
}

template <int KS, int S, int WTXO>
int launch_wgrad_t(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                   const ConvGeom &g, int dact, float dslope, int nsplit, int need_bias) {
    if (dact == ACT_LEAKY) return launch_wgrad_d<KS, S, WTXO, ACT_LEAKY>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
    if (dact == ACT_SIGMOID) return launch_wgrad_d<KS, S, WTXO, ACT_SIGMOID>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
    return launch_wgrad_d<KS, S, WTXO, ACT_NONE>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
}

template <int KS, int S>
int launch_wgrad(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                 const ConvGeom &g, int dact, float dslope, int nsplit, int need_bias) {
    if (KS == 3 && S == 1) {
        const int w = pick_wtx(g.Wo);
        if (w == 26) return launch_wgrad_t<KS, S, 26>(st, x, gout, yact, slab, gpre_out, g, dact, dslope, nsplit, need_bias);
        if (w == 28) return launch_wgrad_t<KS, S, 28>(st, x, gout, yact, slab, gpre_out, g, dact, dslope, nsplit, need_bias);
    }
    return launch_wgrad_t<KS, S, 0>(st, x, gout, yact, slab, gpre_out, g, dact, dslope, nsplit, need_bias);
}

// The split-precision kernel uses full 32-px tiles (two extra halo columns staged separately): a tile costs its 64 k-slots
// whatever its width, so the widest tile = the fewest tiles is always the cheapest.
int pick_wtx_x3(int) { return 32; }

// Workgroup shape of the split-precision weight gradient: 256 threads x 32 input channels at two workgroups per CU for
// the 3x3 layers with multiples of 32 input channels and few output-channel blocks, 512 threads x 64 (16 for 7x7) channels
// otherwise.  Measured (tools/kbench, 128x128, B=8): 64->64 54.5 vs 58.1 us, 128->64 87.1 vs 89.5, 64->128 85.9 vs 86.8,
// 128->1600 1765 vs 1671 (the grad_out tile is fetched once per 32-channel block: with 25 output blocks that costs more
// than the second resident workgroup hides).
// wave-specialised form (conv_wgrad_x3_ws): every 3x3 layer; EBFI_WGRAD_WS=0 restores the uniform-wave kernels (A/B runs)
bool wgrad_x3_ws(const ConvGeom &g, int ks) {   // 64-channel input blocks: layers with other channel counts keep the 32-channel form
    const char *e = dev_getenv("EBFI_WGRAD_WS");
    return ks == 3 && g.Cin % 64 == 0 && !(e && e[0] == '0');
}
bool wgrad_x3_small_wg(const ConvGeom &g, int ks) {
    const bool off = dev_getenv("EBFI_WGRAD_BIGWG") != nullptr;            // development switch (A/B runs)
    return !off && !wgrad_x3_ws(g, ks) && ks == 3 && g.Cin % 32 == 0 && g.Cout <= 256;
}

int wgrad_x3_splits(const ConvGeom &g, int ks) {
    const int wtx = ks == 1 ? wgrad_wtx_rt(g, ks, 1) : pick_wtx_x3(g.Wo);
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, wtx);
    const bool small = wgrad_x3_small_wg(g, ks);
    const int64_t blocks = ceil_div(g.Cout, 64) * ceil_div(g.Cin, small ? 32 : (ks == 7 ? 16 : 64));
    const int64_t slots = small ? 512 : 256;           // resident workgroups of the chip: fill one round of them
    int64_t s = blocks <= slots ? slots / blocks : 1;
    if (s > tiles) s = tiles;
    return (int)(s < 1 ? 1 : s);
}

template <int KS, int WTXO, int DACT, int WX, int CIBT>
int launch_wgrad_x3_w(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                      const ConvGeom &g, float dslope, int nsplit, int need_bias) {
    using C = WCfg<KS, 1, WTXO>;
    constexpr int CIB = CIBT ? CIBT : X3CIB<KS>::value;
    const size_t lds = (size_t)2 * (64 * GS + (CIB + 1) * C::PS) * sizeof(unsigned);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_x3<KS, WTXO, DACT, WX, CIBT>), (int)lds)) return rc;
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, C::WTX);
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, CIB));
    ProfScope ps("conv_wgrad_x3", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * KS * KS,
                 conv_bytes_wgrad(g, KS * KS, DACT != 0, gpre_out != nullptr));
    hipLaunchKernelGGL((conv_wgrad_x3<KS, WTXO, DACT, WX, CIBT>), grid, dim3(WX), lds, st, x, gout, yact, slab, gpre_out, g, dslope,
                       (int)tiles, need_bias);
    return check_launch("conv_wgrad_x3");
}

template <int DACT>
int launch_wgrad_x3_ws(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                       const ConvGeom &g, float dslope, int nsplit, int need_bias) {
    using C = WCfg<3, 1, 32>;
    const size_t lds = (size_t)2 * (64 * GS + 65 * C::PS) * sizeof(unsigned);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_x3_ws<DACT>), (int)lds)) return rc;
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, C::WTX);
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, 64));
    ProfScope ps("conv_wgrad_x3_ws", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * 9,
                 conv_bytes_wgrad(g, 9, DACT != 0, gpre_out != nullptr));
    hipLaunchKernelGGL((conv_wgrad_x3_ws<DACT>), grid, dim3(512), lds, st, x, gout, yact, slab, gpre_out, g, dslope, (int)tiles, need_bias);
    return check_launch("conv_wgrad_x3_ws");
}

template <int KS, int WTXO, int DACT>
int launch_wgrad_x3_d(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                      const ConvGeom &g, float dslope, int nsplit, int need_bias) {
    if constexpr (KS == 3) {
        if (wgrad_x3_ws(g, KS)) return launch_wgrad_x3_ws<DACT>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
        if (wgrad_x3_small_wg(g, KS))
            return launch_wgrad_x3_w<KS, WTXO, DACT, 256, 32>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
    }
    return launch_wgrad_x3_w<KS, WTXO, DACT, WXT, 0>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
}

template <int KS, int WTXO>
int launch_wgrad_x3_t(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                      const ConvGeom &g, int dact, float dslope, int nsplit, int need_bias) {
    if (dact == ACT_LEAKY) return launch_wgrad_x3_d<KS, WTXO, ACT_LEAKY>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
    if (dact == ACT_SIGMOID) return launch_wgrad_x3_d<KS, WTXO, ACT_SIGMOID>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
    return launch_wgrad_x3_d<KS, WTXO, ACT_NONE>(st, x, gout, yact, slab, gpre_out, g, dslope, nsplit, need_bias);
}

int launch_wgrad_x3(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, float *gpre_out,
                    const ConvGeom &g, int ks, int dact, float dslope, int nsplit, int need_bias) {
    if (ks == 1) return launch_wgrad_x3_t<1, 0>(st, x, gout, yact, slab, gpre_out, g, dact, dslope, nsplit, need_bias);
    if (ks == 7) return launch_wgrad_x3_t<7, 32>(st, x, gout, yact, slab, gpre_out, g, dact, dslope, nsplit, need_bias);
    return launch_wgrad_x3_t<3, 32>(st, x, gout, yact, slab, gpre_out, g, dact, dslope, nsplit, need_bias);
}

template <int KS, int DACT>
int launch_wgrad_bf16_d(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, const ConvGeom &g,
                        float dslope, int nsplit, int need_bias) {
    using C = WCfgB<KS>;
    const size_t lds = (size_t)C::LDS_ELEMS * 2;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_bf16<KS, DACT>), (int)lds)) return rc;
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, C::WTX);
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, C::CIB));
    ProfScope ps("conv_wgrad_bf16", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * KS * KS,
                 conv_bytes_wgrad(g, KS * KS, DACT != 0, false));
    hipLaunchKernelGGL((conv_wgrad_bf16<KS, DACT>), grid, dim3(256), lds, st, x, gout, yact, slab, g, dslope, (int)tiles,
                       need_bias);
    return check_launch("conv_wgrad_bf16");
}

template <int KS>
int launch_wgrad_bf16(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, const ConvGeom &g,
                      int dact, float dslope, int nsplit, int need_bias) {
    if (dact == ACT_LEAKY) return launch_wgrad_bf16_d<KS, ACT_LEAKY>(st, x, gout, yact, slab, g, dslope, nsplit, need_bias);
    if (dact == ACT_SIGMOID) return launch_wgrad_bf16_d<KS, ACT_SIGMOID>(st, x, gout, yact, slab, g, dslope, nsplit, need_bias);
    return launch_wgrad_bf16_d<KS, ACT_NONE>(st, x, gout, yact, slab, g, dslope, nsplit, need_bias);
}

// ------------------------------------------------------------------------------------------------
// conv_fwd_bf16x3_ws: the 3x3 split-precision forward / data gradient (64 output channels per workgroup, no folded activation
// derivative: the bulk of a training step) with SPECIALISED waves, after the weight gradient (conv_wgrad_x3_ws).  768 threads:
// waves 8..11 are producers -- they fetch the next chunk's input tile as 16-byte quads and its weight pieces one chunk ahead,
// convert and write both images; waves 0..7, two per SIMD, are consumers -- one output row each, operand fragments of tap
// t+1 read before the MFMAs of tap t (168 registers per wave: three waves per SIMD).  Same LDS images, persistent walk over
// pixel tiles and epilogue as conv_fwd_bf16x3_db; one workgroup barrier per 16-channel chunk hands a buffer over.
// Measured (B=8, 128x128): 128 -> 1600 forward / its data gradient 1.04 ms (uniform waves 1.24 / 1.18); all eligible layers
// of the step -0.42 ms.  A first form with FOUR consumer waves (two rows each, single-buffered operands, 254 registers) ran
// its consumers ALONE at 84-96 % of the real-clock matrix peak but lost it again beside the producers (1.10-1.12 ms).
constexpr int NTWS = 768;              // conv_fwd_bf16x3_ws: 8 consumer waves + 4 producer waves
template <int XM, bool FAC = false>
__global__ __launch_bounds__(NTWS) void conv_fwd_bf16x3_ws(const float *__restrict__ x, const __bf16 *__restrict__ wp,
                                                           const float *__restrict__ bias, float *__restrict__ out, ConvGeom g, int K16,
                                                           int act, float slope, EpiExtra epi, int tiles_total, FacEpi fac) {
    constexpr bool EXTRA = (XM & 7) != 0;                      // XM: epilogue extras compiled in (store_out_tile; bit 3 = output layouts)
    constexpr int KS = 3, KK = 9, MT = 2;
    constexpr int IH = TYB - 1 + KS, IW = TX - 1 + KS, PS = IH * IW, COS = 32 * MT;
    constexpr int NCW = 8, PT = 256;                           // consumer waves (one output row each); producer threads
    constexpr int WPIECES = KK * COS * 2, NWB = (2 * WPIECES + PT - 1) / PT;
    constexpr int INB = PS * 32, WB = KK * COS * 32, BUFB = 2 * INB + 2 * WB;
    extern __shared__ __attribute__((aligned(16))) char smd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + TX - 1) / TX, tiles_y = (g.Ho + TYB - 1) / TYB;
    const int co_base = blockIdx.y * COS;
    const int grp = co_base / (g.Cout / g.groups);
    const int HW = g.H * g.W;
    const unsigned plane_bytes = (unsigned)HW * 4u, x_bytes = (unsigned)g.Cin * plane_bytes;
    const int nchunks = K16 / CKB;
    const int G = gridDim.x;
    int ntiles_mine = 0;
    for (int t = blockIdx.x; t < tiles_total; t += G) ++ntiles_mine;
    const int nitems = ntiles_mine * nchunks;
    const bool xcd_map = (gridDim.x & 7) == 0;
    auto tile_coords = [&](int tt, int &tb, int &ty0, int &tx0) {
        int u = xcd_map ? xcd_tile(tt, tiles_total) : tt;          // (neighbouring tiles on one XCD: shared halo lines hit its L2)
        const int txi = u % tiles_x; u /= tiles_x;
        const int tyi = u % tiles_y;
        tb = u / tiles_y; ty0 = tyi * TYB; tx0 = txi * TX;
    };

    if (wave < NCW) {
        // static priority for the matrix-issuing waves: VALU issue is arbitrated by priority, then age (MI355X_MICROARCH.md, "two
        // waves per SIMD"); with the consumers above the converting producer wave of their SIMD the kernel measured 0.5-0.7 % faster,
        // with the producers above the consumers 2 % slower (same box, round 5)
        __builtin_amdgcn_s_setprio(1);
        // ------------------------------------------------------------------ consumers: output row `wave`, two per SIMD
        f32x16 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
        const int hsel = lane >> 5, l31 = lane & 31;
        const int a_lane = 2 * INB + l31 * 32 + ((hsel ^ ((l31 >> 3) & 1)) << 4);
        const int pbase = wave * IW + l31;
        unsigned fbits = 0;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap)
            fbits |= (unsigned)((((pbase + (tap / KS) * IW + (tap % KS)) >> 3) & 1) ^ hsel) << tap;
        bf16x8 ah[2][MT], al[2][MT], bh[2][2], bl[2][2];
        [[maybe_unused]] float amax16 = 0.f, amax_pre = 0.f;
        auto tap_read = [&](const char *base, int tap, int set) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const char *bp = base + pbase * 32 + (int)(((fbits >> tap) & 1u) << 4) + (ky * IW + kx) * 32;
            const char *ap = base + a_lane + tap * COS * 32;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[set][m] = *reinterpret_cast<const bf16x8 *>(ap + m * 1024);
#ifndef ABL_ONE_MFMA
                al[set][m] = *reinterpret_cast<const bf16x8 *>(ap + m * 1024 + WB);
#endif
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[set][n] = *reinterpret_cast<const bf16x8 *>(bp + n * 1024);
#ifndef ABL_ONE_MFMA
                bl[set][n] = *reinterpret_cast<const bf16x8 *>(bp + n * 1024 + INB);
#endif
            }
        };
        auto tap_mfma = [&](int set) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
#ifndef ABL_ONE_MFMA     // (timing ablation: what a single-product operand format would leave of this kernel; results are wrong)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][m], bh[set][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][m], bl[set][n], acc[m][n], 0, 0, 0);
#endif
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][m], bh[set][n], acc[m][n], 0, 0, 0);
                }
        };
        __syncthreads();                   // (A) the first chunk is committed
        int item = 0, tcur = blockIdx.x;
        for (int ti = 0; ti < ntiles_mine; ++ti, tcur += G) {
            for (int chunk = 0; chunk < nchunks; ++chunk, ++item) {
                const char *base = smd + (item & 1) * BUFB;
#ifdef WSF_NO_CONSUME
                if (g.pad != 12345) { } else
#endif
                {
                    tap_read(base, 0, 0);
#pragma unroll
                    for (int tap = 0; tap < KK; ++tap) {
                        if (tap + 1 < KK) tap_read(base, tap + 1, (tap + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                        tap_mfma(tap & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();           // (B) this buffer has been read, the other one is complete
            }
            int cb_, cy0, cx0;
            tile_coords(tcur, cb_, cy0, cx0);
            if constexpr (FAC) {
                fac_epilogue_tile<MT>(out, bias, acc, g, fac, cb_, co_base, cy0 + wave, cx0, lane, slope);
            } else {
                // the fp16 side images of the epilogue saturate instead of overflowing (MODE.FP16_OVFL) -- set for the epilogue ONLY:
                // under that bit the matrix cores drop non-finite operands (c16.hpp, round 6), and the loop above must pass them on
                if constexpr ((XM & 6) != 0) saturate_fp16_conversions(true);
                store_out_tile<MT, XM>(out, bias, acc, g, cb_, co_base, cy0 + wave, cx0, lane, act, slope, epi, 1.f, &amax16, &amax_pre);
                if constexpr ((XM & 6) != 0) saturate_fp16_conversions(false);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
        }
        if constexpr (EXTRA) {
            if (epi.out16 != nullptr) ScaleSlot{epi.slot16}.record(amax16);      // |max| of the fp16 side image this wave wrote
            if constexpr ((XM & 4) != 0) {
                if (epi.pre16 != nullptr) ScaleSlot{epi.pre_slot}.record(amax_pre);
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- producers
    // The input tile is fetched as 16-byte quads of 4 consecutive pixels x 8 channels (W % 4 == 0, same padding, aligned
    // tensors: the launcher checks): 16 + 9 vector-memory instructions per thread and chunk, one chunk ahead of the consumers.
    const int ptid = tid - 64 * NCW;
    const unsigned img_bytes = (unsigned)KK * (unsigned)g.Cout * (unsigned)K16 * 2u;
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(wp), 0, 2u * img_bytes, 0x00020000);
    constexpr int SH = (4 - (KS / 2) % 4) % 4;             // tile column of a quad's first pixel: 4 qq - SH
    constexpr int NQ = (IW + SH + 3) / 4;                  // quads per tile row
    constexpr int NITEM = IH * NQ * 2;                     // (row, quad, channel half)
    constexpr int NIT = (NITEM + PT - 1) / PT;             // items per producer thread
    int it_qh[NIT], it_qr[NIT], it_qq[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int id = ptid + k * PT;
        it_qh[k] = id & 1;
        it_qr[k] = (id >> 1) / NQ;
        it_qq[k] = (id >> 1) - it_qr[k] * NQ;
    }
    unsigned w_off[NWB];
    int w_dst[NWB];
#pragma unroll
    for (int it = 0; it < NWB; ++it) {
        const int i = ptid + it * PT;
        const int sel = i >= WPIECES ? 1 : 0, j = i - sel * WPIECES;
        const int row = j >> 1, half = j & 1;
        const int tap = row / COS, co = row - tap * COS;
        w_off[it] = i < 2 * WPIECES ? (unsigned)sel * img_bytes + (unsigned)(((tap * g.Cout + co_base + co) * K16 + half * 8) * 2) : SENT;
        w_dst[it] = 2 * INB + sel * WB + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
    }
    u32x4 rq[NIT][8];                      // 4 pixels of channels 8*qh + k
    u32x4 rw[NWB];
    int pf_tile = blockIdx.x, pf_chunk = 0;
    unsigned pf_off[NIT];
    const float *pf_src = x;
    unsigned pf_bytes = 0u;
    auto pf_setup = [&]() {
        const bool live = pf_tile < tiles_total;
        int tb, ty0, tx0;
        tile_coords(live ? pf_tile : 0, tb, ty0, tx0);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int yy = ty0 - g.pad + it_qr[k], xq = tx0 - g.pad - SH + 4 * it_qq[k];
            const bool ok = live && ptid + k * PT < NITEM && yy >= 0 && yy < g.H && xq >= 0 && xq + 3 < g.W;
            pf_off[k] = ok ? (unsigned)(yy * g.W + xq) * 4u + (unsigned)(8 * it_qh[k]) * plane_bytes : SENT;
        }
        pf_src = x + ((int64_t)tb * g.groups + grp) * g.Cin * HW;
        pf_bytes = live ? x_bytes : 0u;
    };
    auto prefetch = [&]() {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(pf_src, pf_bytes);
        const unsigned cb = (unsigned)pf_chunk * (unsigned)CKB * plane_bytes;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) rq[k][c] = __builtin_amdgcn_raw_buffer_load_b128(r, pf_off[k] + cb + (unsigned)c * plane_bytes, 0, 0);
        const unsigned wb = (unsigned)pf_chunk * (unsigned)(CKB * 2);
#pragma unroll
        for (int it = 0; it < NWB; ++it) {
#ifdef ABL_ONE_MFMA
            if (ptid + it * PT >= WPIECES) continue;
#endif
            rw[it] = __builtin_amdgcn_raw_buffer_load_b128(rwt, w_off[it] + wb, 0, 0);
        }
        if (++pf_chunk == nchunks) {
            pf_chunk = 0;
            pf_tile += G;
            pf_setup();
        }
    };
    auto commit = [&](int buf) {
        char *base = smd + buf * BUFB;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (ptid + k * PT < NITEM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * it_qq[k] + j - SH;
                    if (c < 0 || c >= IW) continue;
                    u32x4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float v0 = __uint_as_float(rq[k][e][j]), v1 = __uint_as_float(rq[k][e + 1][j]);
                        const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
                        hv[e >> 1] = pack_bf16((float)a0, (float)a1);
                        lv[e >> 1] = pack_bf16(v0 - (float)a0, v1 - (float)a1);
                    }
                    const int pos = it_qr[k] * IW + c;
                    const int d = pos * 32 + ((it_qh[k] ^ ((pos >> 3) & 1)) << 4);
                    *reinterpret_cast<u32x4 *>(base + d) = hv;
#ifndef ABL_ONE_MFMA
                    *reinterpret_cast<u32x4 *>(base + INB + d) = lv;
#endif
                }
            }
#pragma unroll
        for (int it = 0; it < NWB; ++it) {
#ifdef ABL_ONE_MFMA
            if (ptid + it * PT >= WPIECES) continue;
#endif
            if (ptid + it * PT < 2 * WPIECES) *reinterpret_cast<u32x4 *>(base + w_dst[it]) = rw[it];
        }
    };
    pf_setup();
    prefetch();
    commit(0);                             // item 0
    prefetch();                            // item 1 in flight
    __syncthreads();                       // (A)
    for (int item = 0; item < nitems; ++item) {
#ifndef WSF_NO_PRODUCE
        commit((item + 1) & 1);            // item + 1, while the consumers multiply item
        prefetch();                        // item + 2
#endif
        __syncthreads();                   // (B)
    }
}

#include "conv2d_f16.inc.hpp"
#include "conv2d_thin.inc.hpp"
#include "conv2d_shift.inc.hpp"

size_t bf16_pack_bytes(int M, int K, int ks) { return (size_t)ks * ks * M * (size_t)((K + 15) / 16 * 16) * 2; }

// shared by forward (TR = 0) and data gradient (TR = 1); g is the geometry of the conv actually run.
// x3 != 0: split-precision kernel (hi/lo operand images, three MFMAs per product).
template <int KS>
int launch_fwd_bf16(hipStream_t st, const float *x, const float *dact_y, const float *w, const float *bias, float *out,
                    const ConvGeom &g, int transposed, int act, float slope, int dact, float dslope, void *workspace,
                    size_t ws_bytes, int x3 = 0, EpiExtra epi = EpiExtra{nullptr, nullptr, 0, 0.f}) {
    const int K16 = (g.Cin + 15) / 16 * 16;
    const size_t need = bf16_pack_bytes(g.Cout, g.Cin, KS) * (x3 ? 2 : 1);
    if (!workspace || ws_bytes < need) return fail(EBFI_ERR_WORKSPACE, "conv2d bf16: workspace %zu bytes < required %zu", ws_bytes, need);
    __bf16 *wp = static_cast<__bf16 *>(workspace);
    const int64_t total = (int64_t)KS * KS * g.Cout * K16;
    if (w != nullptr) {   // w == nullptr: `workspace` already holds the packed images (ebfi_conv2d_pack_bf16x3 / ebfi_pack_table_bf16)
        ProfScope ps("conv_pack_w_bf16", st);
        hipLaunchKernelGGL(conv_pack_w_bf16, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, w, wp, g.Cout, g.Cin, K16,
                           KS * KS, transposed, x3);
        if (int rc = check_launch("conv_pack_w_bf16")) return rc;
    }
    const int ty = x3 ? TYB : TY;
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, ty) * ceil_div(g.Wo, TX);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d: too many tiles");
    const char *name = x3 ? (transposed ? "conv_fwd_bf16x3_db/dgrad" : "conv_fwd_bf16x3_db/fwd")
                          : (transposed ? "conv_fwd_bf16/dgrad" : "conv_fwd_bf16/fwd");   // label = kernel symbol / role
    const double flops = 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * KS * KS;
    // 32 output channels per workgroup when 64 would leave more than half of the CUs without one (small feature maps of the
    // detail branch): twice the workgroups, each with half the matrix work per staged chunk
    // (a requested fp16 side image pins the 64-channel wave-specialised form: its epilogue is the one that writes it)
    const bool few = x3 && epi.out16 == nullptr && epi.post_scale == nullptr && g.store == 0 && tiles * ceil_div(g.Cout, 64) <= 128 &&
                     dev_getenv("EBFI_CONV_NO_MT1") == nullptr;
    const int mt = (g.Cout <= 32 || few) ? 1 : 2;
    dim3 grid((unsigned)tiles, (unsigned)ceil_div(g.Cout, 32 * mt));
    if (g.store != 0 && !x3) return fail(EBFI_ERR_UNSUPPORTED, "conv2d: shuffled output layouts are written by the split-precision kernels only");
    if (x3) {
        constexpr int PSX = (TYB - 1 + KS) * (TX - 1 + KS);
        const size_t lds = (size_t)2 * (2 * PSX * 32 + 2 * KS * KS * 32 * mt * 32) + KB_LDS_BYTES;   // two buffers of unpadded hi/lo images
        // 16-byte input quads: rows must keep quads aligned, same-padding only; used where they pay (see the kernel's header)
        const bool vec4 = g.W % 4 == 0 && g.pad == KS / 2 && aligned16(x) && (!dact_y || aligned16(dact_y));
        const char *ws_env = dev_getenv("EBFI_CONV_WS");
        const bool extra = epi.addend != nullptr || epi.mask_y != nullptr || epi.out16 != nullptr || epi.post_scale != nullptr;
        // wave-specialised form (conv_fwd_bf16x3_ws): every 3x3 layer the quad-staging producers can serve (64-channel blocks, no
        // folded activation derivative); EBFI_CONV_WS=0 / 1 = never / only the long layers (development switch, A/B runs)
        const bool ws_long = ceil_div(g.Cout, 64) >= 8 || K16 >= 512;
        const bool ws_extra_ok = !extra || dev_getenv("EBFI_CONV_WS_NOEXTRA") == nullptr;
        const bool use_ws = KS == 3 && mt == 2 && vec4 && dact == ACT_NONE && ws_extra_ok && !(ws_env && ws_env[0] == '0') &&
                            (ws_long || !(ws_env && ws_env[0] == '1'));
        if (epi.out16 != nullptr && !use_ws)
            return fail(EBFI_ERR_UNSUPPORTED, "conv2d: the fp16 side image is written by the wave-specialised 3x3 kernel only "
                        "(W %% 4 == 0, same padding, more than 32 output channels, 16-byte aligned input)");
        if (epi.post_scale != nullptr && !use_ws)
            return fail(EBFI_ERR_UNSUPPORTED, "conv2d: the ResidualControl epilogue is written by the wave-specialised 3x3 kernel only");
        if (g.store != 0 && (!use_ws || extra))
            return fail(EBFI_ERR_UNSUPPORTED, "conv2d: a shuffled output layout is written by the wave-specialised 3x3 kernel only (W %% 4 == 0, "
                        "same padding, more than 32 output channels, 16-byte aligned input), without epilogue extras");
        if (use_ws) name = transposed ? "conv_fwd_bf16x3_ws/dgrad" : (epi.post_scale ? "conv_fwd_bf16x3_ws/fwd_rc" :
                                                                      (epi.out16 ? "conv_fwd_bf16x3_ws/fwd_img" :
                                                                       (g.store ? "conv_fwd_bf16x3_ws/fwd_shuffle" : "conv_fwd_bf16x3_ws/fwd")));
        ProfScope ps(name, st, flops, conv_bytes_fwd(g, KS * KS, dact != 0));
#define EBFI_LAUNCH_X3V(MT_, DA_, VEC_)                                                                                   \
    do {                                                                                                                 \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_bf16x3_db<KS, MT_, DA_, VEC_>), 160 * 1024)) \
            return rc_;                                                                                                  \
        hipLaunchKernelGGL((conv_fwd_bf16x3_db<KS, MT_, DA_, VEC_>), (VEC_ == 1 && DA_ == 0 && KS == 3) ? pgrid : grid,    \
                           dim3(NTB), lds, st, x, dact_y, wp, bias, out, g, K16, act, slope, dslope, epi, (int)tiles);   \
    } while (0)
#define EBFI_LAUNCH_X3(MT_, DA_)                                                                                          \
    do {                                                                                                                 \
        if (vec == 4) EBFI_LAUNCH_X3V(MT_, DA_, 4);                                                                      \
        else EBFI_LAUNCH_X3V(MT_, DA_, 1);                                                                               \
    } while (0)
        const char *vec_env = dev_getenv("EBFI_CONV_VEC");     // development switch (tools/kbench): 1 = dword loads, 4 = quad loads
        const int vec = !vec4 ? 1 : (vec_env ? atoi(vec_env) : (dact != ACT_NONE ? 4 : 1));
        // the persistent form (3x3, dword loads, no folded derivative): one round of workgroups, each walking
        // ceil(tiles / gx) pixel tiles of its output-channel block
        const int64_t co_blocks = ceil_div(g.Cout, 32 * mt);
        int64_t gx = 256 / co_blocks;
        if (gx < 1) gx = 1;
        // (measured, round 4: rounding gx down to a multiple of 8 so that xcd_tile applies to the 128 -> 1600 layer -- 200 instead
        // of 250 workgroups -- is time-neutral: 1.23-1.27 ms against 1.15-1.34 ms; what the L2s save the idle CUs give back)
        if (gx > tiles || dev_getenv("EBFI_CONV_NOPERSIST")) gx = tiles;
        const dim3 pgrid((unsigned)gx, (unsigned)co_blocks);
        if constexpr (KS == 3) {
            if (use_ws) {
#define EBFI_LAUNCH_X3WS(XM_)                                                                                              \
    do {                                                                                                                   \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_bf16x3_ws<XM_>), 160 * 1024)) return rc_; \
        hipLaunchKernelGGL((conv_fwd_bf16x3_ws<XM_>), pgrid, dim3(NTWS), lds, st, x, wp, bias, out, g, K16, act, slope, epi,  \
                           (int)tiles, FacEpi{nullptr, 0});                                                               \
    } while (0)
                const bool xam = epi.addend != nullptr || epi.mask_y != nullptr, x16 = epi.out16 != nullptr;
                if (epi.post_scale != nullptr) {
                    if (xam) return fail(EBFI_ERR_UNSUPPORTED, "conv2d: the ResidualControl epilogue comes without addend / mask");
                    if (x16) EBFI_LAUNCH_X3WS(6);      // training: + the images of `a` and of the result
                    else EBFI_LAUNCH_X3WS(4);          // inference: scale + residual only
                }
                else if (g.store != 0) EBFI_LAUNCH_X3WS(8);        // output through PixelShuffle(2) / its inverse (ConvGeom::store)
                else if (xam && x16) EBFI_LAUNCH_X3WS(3);
                else if (x16) EBFI_LAUNCH_X3WS(2);
                else if (xam) EBFI_LAUNCH_X3WS(1);
                else EBFI_LAUNCH_X3WS(0);
#undef EBFI_LAUNCH_X3WS
                return check_launch(name);
            }
        }
        if (mt == 1) {
            if (dact == ACT_LEAKY) EBFI_LAUNCH_X3(1, ACT_LEAKY);
            else if (dact == ACT_SIGMOID) EBFI_LAUNCH_X3(1, ACT_SIGMOID);
            else EBFI_LAUNCH_X3(1, ACT_NONE);
        } else {
            if (dact == ACT_LEAKY) EBFI_LAUNCH_X3(2, ACT_LEAKY);
            else if (dact == ACT_SIGMOID) EBFI_LAUNCH_X3(2, ACT_SIGMOID);
            else EBFI_LAUNCH_X3(2, ACT_NONE);
        }
#undef EBFI_LAUNCH_X3V
#undef EBFI_LAUNCH_X3
        return check_launch(name);
    }
    {
        ProfScope ps(name, st, flops, conv_bytes_fwd(g, KS * KS, dact != 0));
#define EBFI_LAUNCH_BF16(MT_, DA_)                                                                                     \
    hipLaunchKernelGGL((conv_fwd_bf16<KS, MT_, DA_>), grid, dim3(256), 0, st, x, dact_y, wp, bias, out, g, K16, act, slope, \
                       dslope)
        if (mt == 1) {
            if (dact == ACT_LEAKY) EBFI_LAUNCH_BF16(1, ACT_LEAKY);
            else if (dact == ACT_SIGMOID) EBFI_LAUNCH_BF16(1, ACT_SIGMOID);
            else EBFI_LAUNCH_BF16(1, ACT_NONE);
        } else {
            if (dact == ACT_LEAKY) EBFI_LAUNCH_BF16(2, ACT_LEAKY);
            else if (dact == ACT_SIGMOID) EBFI_LAUNCH_BF16(2, ACT_SIGMOID);
            else EBFI_LAUNCH_BF16(2, ACT_NONE);
        }
#undef EBFI_LAUNCH_BF16
    }
    return check_launch(name);
}

}  // namespace

extern "C" size_t ebfi_conv2d_bf16_workspace(int Cin, int Cout, int ksize) {
    const int m = Cin > Cout ? Cin : Cout;
    return 2 * bf16_pack_bytes(m, m, ksize);   // forward or transposed (data-gradient) packing, hi and lo images
}

namespace {
// ------------------------------------------------------------------------------------------------
// conv7_x3: 7x7, stride 1, at most 16 output rows, split precision -- the detail branch's output conv (16 -> 3 on the
// reflection-padded map, models/Ours/model_singleframe.py:207) forward, and the data gradients of the two 7x7 layers
// (16 <- 3; the 6-channel stem input from 64 zero-inserted gradient channels).  The fp32 matrix-core kernel spent
// 195 / 233 us per launch on 32-row tiles with 3..16 live rows; the bf16x3 instructions do the same padded tile 16x
// faster per product.  One 512-thread workgroup per 8 x 64 output tile; per 16-channel chunk the 14 x 70 input tile is
// staged as the usual hi / lo `[pos][16 ch]` images (half-swapped rows), the weights are read as fp32 (forward:
// W[m][k][tap]; tr: W[k][m][48 - tap], i.e. the transposed, flipped filter of the data gradient), split in registers and
// stored as `[tap][16 rows][16 ch]` images -- 16 rows, the upper half of the 32-row matrix tile re-reads them and is
// never stored.  Single-buffered (113 KB): a chunk is load -> commit -> 49 taps x 6 MFMA per wave.
constexpr int C7_IH = TYB + 6, C7_IW = TX + 6, C7_PS = C7_IH * C7_IW;
constexpr int C7_INB = C7_PS * 32, C7_ROWS = 16, C7_WB = 49 * C7_ROWS * 32;
constexpr int C7_LDS = 2 * C7_INB + 2 * C7_WB;
constexpr int C7_NPOS = (C7_PS + NTB - 1) / NTB;
constexpr int C7_PIECES = 49 * C7_ROWS * 2;            // 16-byte pieces of one weight image
constexpr int C7_NWP = (C7_PIECES + NTB - 1) / NTB;

__global__ __launch_bounds__(NTB) void conv7_x3_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias, float *__restrict__ out, int K, int H,
                                                       int W, int M, int Ho, int Wo, int pad, int tr, int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) char smd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wo + TX - 1) / TX, tiles_y = (Ho + TYB - 1) / TYB;
    int t = (gridDim.x & 7) == 0 ? xcd_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // (halo lines of neighbouring tiles in one L2)
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TYB, x0 = tx * TX;
    const int iy0 = y0 - pad, ix0 = x0 - pad;
    const unsigned plane_bytes = (unsigned)(H * W) * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)b * K * H * W, (unsigned)K * plane_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (unsigned)(M * K * 49) * 4u);

    unsigned in_off[C7_NPOS];
    int in_dst[C7_NPOS];
#pragma unroll
    for (int q = 0; q < C7_NPOS; ++q) {
        const int pos = tid + q * NTB;
        const int r = pos / C7_IW, c = pos - r * C7_IW;
        const int yy = iy0 + r, xx = ix0 + c;
        in_off[q] = (pos < C7_PS && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)(yy * W + xx) * 4u : SENT;
        in_dst[q] = pos * 32 + (((pos >> 3) & 1) << 4);
    }
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int hsel = lane >> 5, l31 = lane & 31;

    const int nchunks = (K + CKB - 1) / CKB;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk > 0) __syncthreads();                    // the taps of the previous chunk have read the images
        float rin[C7_NPOS * CKB];
        const unsigned cb = (unsigned)(chunk * CKB) * plane_bytes;
#pragma unroll
        for (int q = 0; q < C7_NPOS; ++q)
#pragma unroll
            for (int ci = 0; ci < CKB; ++ci) rin[q * CKB + ci] = buf_ld(rx, in_off[q] + cb + (unsigned)ci * plane_bytes);
        float wv[C7_NWP][8];
#pragma unroll
        for (int it = 0; it < C7_NWP; ++it) {
            const int i = tid + it * NTB;
            const int row = i >> 1, half = i & 1;          // row = tap * 16 + m
            const int tap = row >> 4, m = row & 15;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = chunk * CKB + half * 8 + j;
                const bool ok = i < C7_PIECES && m < M && k < K;
                const int e = tr ? (k * M + m) * 49 + 48 - tap : (m * K + k) * 49 + tap;
                wv[it][j] = buf_ld(rw, ok ? (unsigned)e * 4u : SENT);
            }
        }
#pragma unroll
        for (int q = 0; q < C7_NPOS; ++q)
            if (tid + q * NTB < C7_PS) {
                u32x4 h0, h1, l0, l1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v0 = rin[q * CKB + 2 * j], v1 = rin[q * CKB + 2 * j + 1];
                    const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
                    const unsigned hp = pack_bf16((float)a0, (float)a1);
                    const unsigned lp = pack_bf16(v0 - (float)a0, v1 - (float)a1);
                    if (j < 4) { h0[j] = hp; l0[j] = lp; } else { h1[j - 4] = hp; l1[j - 4] = lp; }
                }
                const int d0 = in_dst[q], d1 = in_dst[q] ^ 16;
                *reinterpret_cast<u32x4 *>(smd + d0) = h0;
                *reinterpret_cast<u32x4 *>(smd + d1) = h1;
                *reinterpret_cast<u32x4 *>(smd + C7_INB + d0) = l0;
                *reinterpret_cast<u32x4 *>(smd + C7_INB + d1) = l1;
            }
#pragma unroll
        for (int it = 0; it < C7_NWP; ++it) {
            const int i = tid + it * NTB;
            if (i < C7_PIECES) {
                const int row = i >> 1, half = i & 1;
                u32x4 hv, lv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = wv[it][2 * j], v1 = wv[it][2 * j + 1];
                    const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
                    hv[j] = pack_bf16((float)a0, (float)a1);
                    lv[j] = pack_bf16(v0 - (float)a0, v1 - (float)a1);
                }
                const int d = 2 * C7_INB + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
                *reinterpret_cast<u32x4 *>(smd + d) = hv;
                *reinterpret_cast<u32x4 *>(smd + C7_WB + d) = lv;
            }
        }
        __syncthreads();
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int tap = ky * 7 + kx;
                const int ra = tap * C7_ROWS + (l31 & 15);
                const char *ap = smd + 2 * C7_INB + ra * 32 + ((hsel ^ ((ra >> 3) & 1)) << 4);
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ap), al = *reinterpret_cast<const bf16x8 *>(ap + C7_WB);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int pos = (wave + ky) * C7_IW + kx + l31 + n * 32;
                    const char *bp = smd + pos * 32 + ((hsel ^ ((pos >> 3) & 1)) << 4);
                    const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(bp), bl = *reinterpret_cast<const bf16x8 *>(bp + C7_INB);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
                }
            }
        }
    }
    const int yo = y0 + wave;
    if (yo >= Ho) return;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int xo = x0 + n * 32 + l31;
        if (xo >= Wo) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hsel;
            if (m < M) out[(((int64_t)b * M + m) * Ho + yo) * Wo + xo] = act_apply(acc[n][r] + (bias ? bias[m] : 0.f), act, slope);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv7s2_dgrad_x3: data gradient of a 7x7 STRIDE-2 convolution with at most 16 input channels (the detail branch's stem,
// 6 <- 64 folded channels), by output parity instead of zero insertion:
//     gx[ci][2u+py][2v+px] = sum_co sum_{ky = (py+pad) mod 2, +2, ..} sum_{kx likewise} W[co][ci][ky][kx] g[co][u + (py+pad-ky)/2][v + (px+pad-kx)/2]
// -- each output pixel takes the 9..16 taps of its parity class, the 49 taps are spread over the four classes, and the
// input tile is the gradient itself (11 x 67 positions per 8 x 64 tile of (u, v)), not a 4x larger zero-inserted map:
// a quarter of the matrix work and staging of conv7_x3 on the zero-inserted tensor, no fill / strided copy in front.
// Same images and half-swap as conv7_x3; acc[py][px][n]; the two px classes of a row pair up into 8-byte stores.
constexpr int C7S_IH = TYB + 3, C7S_IW = TX + 3, C7S_PS = C7S_IH * C7S_IW;
constexpr int C7S_INB = C7S_PS * 32;
constexpr int C7S_LDS = 2 * C7S_INB + 2 * C7_WB;
constexpr int C7S_NPOS = (C7S_PS + NTB - 1) / NTB;

template <int pad>                         // compile-time: the parity class of a tap selects its accumulator
__global__ __launch_bounds__(NTB) void conv7s2_dgrad_x3_kernel(const float *__restrict__ g, const float *__restrict__ w,
                                                               float *__restrict__ gx, int K, int Hg, int Wg, int M, int H, int W) {
    static_assert(pad == 3, "the 11 x 67 tile covers the row / column offsets -1 .. 2 of pad 3");
    extern __shared__ __attribute__((aligned(16))) char smd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Hu = (H + 1) / 2, Wu = (W + 1) / 2;          // grid of (u, v): output pixels (2u + py, 2v + px)
    const int tiles_x = (Wu + TX - 1) / TX, tiles_y = (Hu + TYB - 1) / TYB;
    int t = (gridDim.x & 7) == 0 ? xcd_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // (halo lines of neighbouring tiles in one L2)
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int u0 = ty * TYB, v0 = tx * TX;
    const unsigned plane_bytes = (unsigned)(Hg * Wg) * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g + (int64_t)b * K * Hg * Wg, (unsigned)K * plane_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (unsigned)(M * K * 49) * 4u);

    unsigned in_off[C7S_NPOS];
    int in_dst[C7S_NPOS];
#pragma unroll
    for (int q = 0; q < C7S_NPOS; ++q) {
        const int pos = tid + q * NTB;
        const int r = pos / C7S_IW, c = pos - r * C7S_IW;
        const int yy = u0 - 1 + r, xx = v0 - 1 + c;
        in_off[q] = (pos < C7S_PS && yy >= 0 && yy < Hg && xx >= 0 && xx < Wg) ? (unsigned)(yy * Wg + xx) * 4u : SENT;
        in_dst[q] = pos * 32 + (((pos >> 3) & 1) << 4);
    }
    f32x16 acc[2][2][2];                   // [py][px][n]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i >> 2][(i >> 1) & 1][i & 1][r] = 0.f;
    const int hsel = lane >> 5, l31 = lane & 31;

    const int nchunks = (K + CKB - 1) / CKB;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk > 0) __syncthreads();
        float rin[C7S_NPOS * CKB];
        const unsigned cb = (unsigned)(chunk * CKB) * plane_bytes;
#pragma unroll
        for (int q = 0; q < C7S_NPOS; ++q)
#pragma unroll
            for (int ci = 0; ci < CKB; ++ci) rin[q * CKB + ci] = buf_ld(rx, in_off[q] + cb + (unsigned)ci * plane_bytes);
        // (inputs are committed before the weight pieces are fetched, the pieces in two halves: with the eight accumulator
        // tiles of the four parity classes the kernel has no registers for all staging values at once)
#pragma unroll
        for (int q = 0; q < C7S_NPOS; ++q)
            if (tid + q * NTB < C7S_PS) {
                u32x4 h0, h1, l0, l1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v0f = rin[q * CKB + 2 * j], v1f = rin[q * CKB + 2 * j + 1];
                    const __bf16 a0 = (__bf16)v0f, a1 = (__bf16)v1f;
                    const unsigned hp = pack_bf16((float)a0, (float)a1);
                    const unsigned lp = pack_bf16(v0f - (float)a0, v1f - (float)a1);
                    if (j < 4) { h0[j] = hp; l0[j] = lp; } else { h1[j - 4] = hp; l1[j - 4] = lp; }
                }
                const int d0 = in_dst[q], d1 = in_dst[q] ^ 16;
                *reinterpret_cast<u32x4 *>(smd + d0) = h0;
                *reinterpret_cast<u32x4 *>(smd + d1) = h1;
                *reinterpret_cast<u32x4 *>(smd + C7S_INB + d0) = l0;
                *reinterpret_cast<u32x4 *>(smd + C7S_INB + d1) = l1;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it0 = 0; it0 < C7_NWP; it0 += 2) {
            float wv[2][8];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = tid + (it0 + ii) * NTB;
                const int row = i >> 1, half = i & 1;          // row = tap * 16 + m
                const int tap = row >> 4, m = row & 15;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kc = chunk * CKB + half * 8 + j;
                    const bool ok = it0 + ii < C7_NWP && i < C7_PIECES && m < M && kc < K;
                    wv[ii][j] = buf_ld(rw, ok ? (unsigned)((kc * M + m) * 49 + tap) * 4u : SENT);   // W[co = kc][ci = m][tap]
                }
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = tid + (it0 + ii) * NTB;
                if (it0 + ii < C7_NWP && i < C7_PIECES) {
                    const int row = i >> 1, half = i & 1;
                    u32x4 hv, lv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v0f = wv[ii][2 * j], v1f = wv[ii][2 * j + 1];
                        const __bf16 a0 = (__bf16)v0f, a1 = (__bf16)v1f;
                        hv[j] = pack_bf16((float)a0, (float)a1);
                        lv[j] = pack_bf16(v0f - (float)a0, v1f - (float)a1);
                    }
                    const int d = 2 * C7S_INB + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
                    *reinterpret_cast<u32x4 *>(smd + d) = hv;
                    *reinterpret_cast<u32x4 *>(smd + C7_WB + d) = lv;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int py = (ky + pad) & 1, dy = (py + pad - ky) / 2;           // exact: py + pad - ky is even
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int px = (kx + pad) & 1, dx = (px + pad - kx) / 2;
                const int ra = (ky * 7 + kx) * C7_ROWS + (l31 & 15);
                const char *ap = smd + 2 * C7S_INB + ra * 32 + ((hsel ^ ((ra >> 3) & 1)) << 4);
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ap), al = *reinterpret_cast<const bf16x8 *>(ap + C7_WB);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int pos = (wave + dy + 1) * C7S_IW + l31 + n * 32 + dx + 1;
                    const char *bp = smd + pos * 32 + ((hsel ^ ((pos >> 3) & 1)) << 4);
                    const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(bp), bl = *reinterpret_cast<const bf16x8 *>(bp + C7S_INB);
                    f32x16 &a = acc[py][px][n];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a, 0, 0, 0);
                }
                if ((kx & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // operands of at most two taps in registers (128 accumulators)
            }
        }
    }
    const int u = u0 + wave;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const int Y = 2 * u + py;
        if (u >= Hu || Y >= H) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int v = v0 + n * 32 + l31, X = 2 * v;
            if (v >= Wu) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * hsel;
                if (m >= M) continue;
                float *o = gx + (((int64_t)b * M + m) * H + Y) * W + X;
                if (X + 1 < W && (W & 1) == 0) *reinterpret_cast<float2 *>(o) = make_float2(acc[py][0][n][r], acc[py][1][n][r]);
                else {
                    o[0] = acc[py][0][n][r];
                    if (X + 1 < W) o[1] = acc[py][1][n][r];
                }
            }
        }
    }
}

// g: [B, K = Cout, Hg, Wg] (already multiplied by act') -> gx: [B, M = Cin, H, W];  w: the layer's [Cout][Cin][7][7]
int launch_conv7s2_dgrad_x3(hipStream_t st, const float *g, const float *w, float *gx, int B, int K, int Hg, int Wg, int M, int H,
                            int W, int pad) {
    if (M > C7_ROWS) return fail(EBFI_ERR_UNSUPPORTED, "conv7s2_dgrad_x3: %d input channels (at most %d)", M, C7_ROWS);
    if (pad != 3) return fail(EBFI_ERR_UNSUPPORTED, "conv7s2_dgrad_x3: pad %d (3 only)", pad);
    if ((int64_t)K * Hg * Wg >= (int64_t)1 << 29) return fail(EBFI_ERR_UNSUPPORTED, "conv7s2_dgrad_x3: a sample of 2 GiB or more");
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv7s2_dgrad_x3_kernel<3>), C7S_LDS)) return rc;
    const int64_t tiles = (int64_t)B * ceil_div((H + 1) / 2, TYB) * ceil_div((W + 1) / 2, TX);
    {
        ProfScope ps("conv7_x3/dgrad_s2", st, 2.0 * B * (double)Hg * Wg * M * K * 49, 4.0 * B * ((double)K * Hg * Wg + (double)M * H * W));
        hipLaunchKernelGGL(conv7s2_dgrad_x3_kernel<3>, dim3((unsigned)tiles), dim3(NTB), C7S_LDS, st, g, w, gx, K, Hg, Wg, M, H, W);
    }
    return check_launch("conv7s2_dgrad_x3");
}

// x: [B, K, H, W] -> out: [B, M, Ho, Wo];  w: forward [M][K][7][7], tr [K][M][7][7]
int launch_conv7_x3(hipStream_t st, const float *x, const float *w, const float *bias, float *out, int B, int K, int H, int W,
                    int M, int Ho, int Wo, int pad, int tr, int act, float slope, const char *role) {
    if (M > C7_ROWS) return fail(EBFI_ERR_UNSUPPORTED, "conv7_x3: %d output channels (at most %d)", M, C7_ROWS);
    if ((int64_t)K * H * W >= (int64_t)1 << 29) return fail(EBFI_ERR_UNSUPPORTED, "conv7_x3: a sample of 2 GiB or more");
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv7_x3_kernel), C7_LDS)) return rc;
    const int64_t tiles = (int64_t)B * ceil_div(Ho, TYB) * ceil_div(Wo, TX);
    {
        ProfScope ps(role, st, 2.0 * B * (double)Ho * Wo * M * K * 49, 4.0 * B * ((double)K * H * W + (double)M * Ho * Wo));
        hipLaunchKernelGGL(conv7_x3_kernel, dim3((unsigned)tiles), dim3(NTB), C7_LDS, st, x, w, bias, out, K, H, W, M, Ho, Wo, pad,
                           tr, act, slope);
    }
    return check_launch("conv7_x3");
}

int conv_forward_bf16_impl(const char *who, int x3, const void *input, const void *weight, const void *bias, void *output, int B,
                           int Cin, int H, int W, int Cout, int ksize, int stride, int pad, int act, float slope,
                           void *workspace, size_t workspace_bytes, void *stream) {
    if (!input || !output || (!weight && !(x3 && workspace))) return fail(EBFI_ERR_ARG, "%s: null argument", who);
    if (act < 0 || act > 2) return fail(EBFI_ERR_ARG, "%s: unknown activation %d", who, act);
    const bool k7 = x3 && ksize == 7 && stride == 1 && weight;      // few-output-channel 7x7 (reads the fp32 weight itself)
    if (!k7 && (stride != 1 || (ksize != 1 && ksize != 3)))
        return fail(EBFI_ERR_UNSUPPORTED, "%s: k=%d stride=%d (k in {1,3}, stride 1; split precision also k=7 with <= 16 output channels)",
                    who, ksize, stride);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin, H, W, Cout, ksize, stride, pad)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *x = static_cast<const float *>(input), *w = static_cast<const float *>(weight);
    const float *bs = static_cast<const float *>(bias);
    float *o = static_cast<float *>(output);
    if (k7) return launch_conv7_x3(st, x, w, bs, o, B, Cin, H, W, Cout, g.Ho, g.Wo, pad, 0, act, slope, "conv7_x3/fwd");
    if (ksize == 3) return launch_fwd_bf16<3>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f, workspace, workspace_bytes, x3);
    return launch_fwd_bf16<1>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f, workspace, workspace_bytes, x3);
}

int conv_backward_data_bf16_impl(const char *who, int x3, const void *grad_output, const void *saved_output, const void *weight,
                                 void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize, int stride, int pad,
                                 int act, float slope, void *workspace, size_t workspace_bytes, void *stream) {
    if (!grad_output || !grad_input || (!weight && !(x3 && workspace))) return fail(EBFI_ERR_ARG, "%s: null argument", who);
    if (act != ACT_NONE && !saved_output) return fail(EBFI_ERR_ARG, "%s: activation needs saved_output", who);
    const bool k7 = x3 && ksize == 7 && stride == 1 && weight && act == ACT_NONE;
    if (stride != 1 || pad > ksize - 1 || (!k7 && ksize != 1 && ksize != 3))
        return fail(EBFI_ERR_UNSUPPORTED, "%s: k=%d stride=%d pad=%d", who, ksize, stride, pad);
    ConvGeom f;
    if (int rc = make_geom(f, B, Cin, H, W, Cout, ksize, stride, pad)) return rc;
    if (B == 0) return EBFI_OK;
    // the detail branch's output convolution (16 <- 3 channels on a full-resolution map): taps on the contraction axis of the thin
    // gradient (conv2d_shift.inc.hpp) instead of 49 taps of a 29/32-empty tile
    if (k7 && Cout == 3 && Cin == 16 && (int64_t)B * H * W >= 64 * 1024 && dev_getenv("EBFI_NO_SHIFT_WGRAD") == nullptr)
        return launch_conv7_thin_dgrad(static_cast<hipStream_t>(stream), static_cast<const float *>(grad_output), nullptr,
                                       static_cast<const float *>(weight), static_cast<float *>(grad_input), B, H, W, f.Ho, f.Wo, pad,
                                       ACT_NONE, 0.f);
    if (k7)     // transposed, flipped filter on grad_output: [B, Cout, Ho, Wo] -> [B, Cin, H, W]
        return launch_conv7_x3(static_cast<hipStream_t>(stream), static_cast<const float *>(grad_output),
                               static_cast<const float *>(weight), nullptr, static_cast<float *>(grad_input), B, Cout, f.Ho, f.Wo,
                               Cin, H, W, ksize - 1 - pad, 1, ACT_NONE, 0.f, "conv7_x3/dgrad");
    ConvGeom g{B, Cout, f.Ho, f.Wo, Cin, H, W, ksize - 1 - pad};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *go = static_cast<const float *>(grad_output), *yo = static_cast<const float *>(saved_output);
    const float *w = static_cast<const float *>(weight);
    float *gi = static_cast<float *>(grad_input);
    if (ksize == 3)
        return launch_fwd_bf16<3>(st, go, yo, w, nullptr, gi, g, 1, ACT_NONE, 0.f, act, slope, workspace, workspace_bytes, x3);
    return launch_fwd_bf16<1>(st, go, yo, w, nullptr, gi, g, 1, ACT_NONE, 0.f, act, slope, workspace, workspace_bytes, x3);
}
}  // namespace

extern "C" int ebfi_pack_table_bf16(const float *src, const int32_t *table, int64_t n, void *out, void *stream) {
    if (!src || !table || !out || n < 0) return fail(EBFI_ERR_ARG, "pack_table_bf16: null argument");
    if (n == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps("pack_table_bf16", st, 0.0, 10.0 * (double)n);
    hipLaunchKernelGGL(pack_table_bf16_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, src, table, n,
                       static_cast<__bf16 *>(out));
    return check_launch("pack_table_bf16");
}

extern "C" size_t ebfi_conv2d_packed_bytes(int Cin, int Cout, int ksize, int transposed) {
    return transposed ? 2 * bf16_pack_bytes(Cin, Cout, ksize) : 2 * bf16_pack_bytes(Cout, Cin, ksize);
}

extern "C" int ebfi_conv2d_pack_bf16x3(const void *weight, int Cin, int Cout, int ksize, int transposed, void *packed,
                                       size_t packed_bytes, void *stream) {
    if (!weight || !packed) return fail(EBFI_ERR_ARG, "conv2d_pack_bf16x3: null argument");
    if (Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) return fail(EBFI_ERR_ARG, "conv2d_pack_bf16x3: Cin %d Cout %d k %d", Cin, Cout, ksize);
    const int M = transposed ? Cin : Cout, K = transposed ? Cout : Cin, K16 = (K + 15) / 16 * 16;
    if (packed_bytes < ebfi_conv2d_packed_bytes(Cin, Cout, ksize, transposed))
        return fail(EBFI_ERR_WORKSPACE, "conv2d_pack_bf16x3: %zu bytes < required %zu", packed_bytes,
                    ebfi_conv2d_packed_bytes(Cin, Cout, ksize, transposed));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)ksize * ksize * M * K16;
    ProfScope ps("conv_pack_w_bf16", st);
    hipLaunchKernelGGL(conv_pack_w_bf16, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, static_cast<const float *>(weight),
                       static_cast<__bf16 *>(packed), M, K, K16, ksize * ksize, transposed, 1);
    return check_launch("conv_pack_w_bf16");
}

// fp32 tensors, bf16 matrix-core operands, fp32 accumulation; ksize in {1,3}, stride 1.
extern "C" int ebfi_conv2d_forward_bf16mma(const void *input, const void *weight, const void *bias, void *output, int B,
                                           int Cin, int H, int W, int Cout, int ksize, int stride, int pad, int act,
                                           float slope, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_forward_bf16_impl("conv2d_forward_bf16mma", 0, input, weight, bias, output, B, Cin, H, W, Cout, ksize, stride, pad,
                                  act, slope, workspace, workspace_bytes, stream);
}

extern "C" int ebfi_conv2d_backward_data_bf16mma(const void *grad_output, const void *saved_output, const void *weight,
                                                 void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize,
                                                 int stride, int pad, int act, float slope, void *workspace,
                                                 size_t workspace_bytes, void *stream) {
    return conv_backward_data_bf16_impl("conv2d_backward_data_bf16mma", 0, grad_output, saved_output, weight, grad_input, B, Cin,
                                        H, W, Cout, ksize, stride, pad, act, slope, workspace, workspace_bytes, stream);
}

// Split-precision ("bf16x3") counterparts: operands as bf16 hi + lo pairs, three MFMAs per product, ~1e-5 of fp32.
extern "C" int ebfi_conv2d_forward_bf16x3(const void *input, const void *weight, const void *bias, void *output, int B,
                                          int Cin, int H, int W, int Cout, int ksize, int stride, int pad, int act,
                                          float slope, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_forward_bf16_impl("conv2d_forward_bf16x3", 1, input, weight, bias, output, B, Cin, H, W, Cout, ksize, stride, pad,
                                  act, slope, workspace, workspace_bytes, stream);
}

extern "C" int ebfi_conv2d_backward_data_bf16x3(const void *grad_output, const void *saved_output, const void *weight,
                                                void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize,
                                                int stride, int pad, int act, float slope, void *workspace,
                                                size_t workspace_bytes, void *stream) {
    return conv_backward_data_bf16_impl("conv2d_backward_data_bf16x3", 1, grad_output, saved_output, weight, grad_input, B, Cin,
                                        H, W, Cout, ksize, stride, pad, act, slope, workspace, workspace_bytes, stream);
}

// grad_input[B,Cin,H,W] of a 7x7 STRIDE-2 convolution with Cin <= 16 from grad_preact[B,Cout,Ho,Wo] (= grad_output times the
// activation derivative), split precision, by output parity (no zero-inserted tensor).
extern "C" int ebfi_conv2d_backward_data_s2_bf16x3(const void *grad_preact, const void *weight, void *grad_input, int B, int Cin,
                                                   int H, int W, int Cout, int ksize, int pad, void *stream) {
    if (!grad_preact || !weight || !grad_input) return fail(EBFI_ERR_ARG, "conv2d_backward_data_s2_bf16x3: null argument");
    if (ksize != 7) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_data_s2_bf16x3: k=%d (7 only)", ksize);
    ConvGeom f;
    if (int rc = make_geom(f, B, Cin, H, W, Cout, ksize, 2, pad)) return rc;
    if (B == 0) return EBFI_OK;
    return launch_conv7s2_dgrad_x3(static_cast<hipStream_t>(stream), static_cast<const float *>(grad_preact),
                                   static_cast<const float *>(weight), static_cast<float *>(grad_input), B, Cout, f.Ho, f.Wo, Cin, H, W,
                                   pad);
}

extern "C" int ebfi_conv2d_forward(const void *input, const void *weight, const void *bias, void *output, int B, int Cin,
                                   int H, int W, int Cout, int ksize, int stride, int pad, int act, float slope,
                                   int dtype, void *stream) {
    if (!input || !weight || !output) return fail(EBFI_ERR_ARG, "conv2d_forward: null argument");
    if (dtype != EBFI_F32) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_forward: dtype %d not implemented (fp32 only)", dtype);
    if (act < 0 || act > 2) return fail(EBFI_ERR_ARG, "conv2d_forward: unknown activation %d", act);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin, H, W, Cout, ksize, stride, pad)) return rc;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *x = static_cast<const float *>(input), *w = static_cast<const float *>(weight);
    const float *bs = static_cast<const float *>(bias);
    float *o = static_cast<float *>(output);
    if (ksize == 3 && stride == 1) return launch_fwd<3, 1, false>(st, x, nullptr, w, bs, o, g, act, slope, 0, 0.f);
    if (ksize == 3 && stride == 2) return launch_fwd<3, 2, false>(st, x, nullptr, w, bs, o, g, act, slope, 0, 0.f);
    if (ksize == 1 && stride == 1) return launch_fwd<1, 1, false>(st, x, nullptr, w, bs, o, g, act, slope, 0, 0.f);
    if (ksize == 7 && stride == 1) return launch_fwd<7, 1, false>(st, x, nullptr, w, bs, o, g, act, slope, 0, 0.f);
    if (ksize == 7 && stride == 2) return launch_fwd<7, 2, false>(st, x, nullptr, w, bs, o, g, act, slope, 0, 0.f);
    return fail(EBFI_ERR_UNSUPPORTED, "conv2d_forward: k=%d stride=%d not implemented", ksize, stride);
}

// grad_input[B,Cin,H,W] = conv^T(grad_output * act'(saved_output)); stride 1 only (any pad <= k-1).
extern "C" int ebfi_conv2d_backward_data(const void *grad_output, const void *saved_output, const void *weight,
                                         void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize,
                                         int stride, int pad, int act, float slope, int dtype, void *stream) {
    if (!grad_output || !weight || !grad_input) return fail(EBFI_ERR_ARG, "conv2d_backward_data: null argument");
    if (dtype != EBFI_F32) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_data: dtype %d not implemented", dtype);
    if (act != ACT_NONE && !saved_output) return fail(EBFI_ERR_ARG, "conv2d_backward_data: activation needs saved_output");
    if (stride != 1 || pad > ksize - 1)
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_data: stride 1 and pad <= k-1 only (got s=%d p=%d k=%d); strided layers "
                    "zero-insert grad_output on the host side", stride, pad, ksize);
    ConvGeom f;   // geometry of the forward conv, to validate
    if (int rc = make_geom(f, B, Cin, H, W, Cout, ksize, stride, pad)) return rc;
    if (B == 0) return EBFI_OK;
    // the data gradient is a forward conv over grad_output: channels Cout -> Cin, same spatial size
    ConvGeom g{B, Cout, f.Ho, f.Wo, Cin, H, W, ksize - 1 - pad};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *go = static_cast<const float *>(grad_output), *yo = static_cast<const float *>(saved_output);
    const float *w = static_cast<const float *>(weight);
    float *gi = static_cast<float *>(grad_input);
    if (ksize == 3) return launch_fwd<3, 1, true>(st, go, yo, w, nullptr, gi, g, ACT_NONE, 0.f, act, slope);
    if (ksize == 7) return launch_fwd<7, 1, true>(st, go, yo, w, nullptr, gi, g, ACT_NONE, 0.f, act, slope);
    return launch_fwd<1, 1, true>(st, go, yo, w, nullptr, gi, g, ACT_NONE, 0.f, act, slope);
}

extern "C" size_t ebfi_conv2d_backward_weight_workspace(int B, int Cin, int H, int W, int Cout, int ksize, int stride,
                                                        int pad, int dtype) {
    (void)dtype;
    ConvGeom g;
    if (make_geom(g, B, Cin, H, W, Cout, ksize, stride, pad) != EBFI_OK) return 0;
    int slabs = wgrad_splits(g, ksize, stride);
    if (thin_wgrad_geometry(g, ksize, stride).kind != 0 && thin_wgrad_slabs(g) > slabs)
        slabs = thin_wgrad_slabs(g);            // (the thin-layer kernels write one slab per sample and row band: conv2d_thin.inc.hpp)
    if (const ShiftPlan sp = shift_wgrad_geometry(g, ksize, stride); sp.kind != 0 && shift_wgrad_slabs(g, sp) > slabs)
        slabs = shift_wgrad_slabs(g, sp);       // (one per sample, 16-row band and 64-column segment: conv2d_shift.inc.hpp)
    return (size_t)slabs * ((size_t)Cout * Cin * ksize * ksize + Cout) * sizeof(float);
}

// grad_weight[Cout,Cin,k,k] (and grad_bias[Cout] when non-NULL), both fully overwritten, deterministic.
extern "C" int ebfi_conv2d_backward_weight_ex(const void *input, const void *grad_output, const void *saved_output,
                                              void *grad_weight, void *grad_bias, void *grad_preact_out, int B, int Cin,
                                              int H, int W, int Cout, int ksize, int stride, int pad, int act, float slope,
                                              void *workspace, size_t workspace_bytes, int dtype, void *stream);

extern "C" int ebfi_conv2d_backward_weight(const void *input, const void *grad_output, const void *saved_output,
                                           void *grad_weight, void *grad_bias, int B, int Cin, int H, int W, int Cout,
                                           int ksize, int stride, int pad, int act, float slope, void *workspace,
                                           size_t workspace_bytes, int dtype, void *stream) {
    return ebfi_conv2d_backward_weight_ex(input, grad_output, saved_output, grad_weight, grad_bias, nullptr, B, Cin, H, W,
                                          Cout, ksize, stride, pad, act, slope, workspace, workspace_bytes, dtype, stream);
}

// Same, plus an optional side output grad_preact_out[B,Cout,Ho,Wo] = grad_output * act'(saved_output) (fp32 kernels
// only): lets the caller run the data gradient on it without re-reading the saved activation.
extern "C" int ebfi_conv2d_backward_weight_ex(const void *input, const void *grad_output, const void *saved_output,
                                              void *grad_weight, void *grad_bias, void *grad_preact_out, int B, int Cin,
                                              int H, int W, int Cout, int ksize, int stride, int pad, int act, float slope,
                                              void *workspace, size_t workspace_bytes, int dtype, void *stream) {
    if (!input || !grad_output || !grad_weight) return fail(EBFI_ERR_ARG, "conv2d_backward_weight: null argument");
    const bool bf16mma = dtype == EBFI_F32_BF16MMA && stride == 1 && (ksize == 1 || ksize == 3);
    const bool x3 = dtype == EBFI_F32_BF16X3MMA && stride == 1 && (ksize == 1 || ksize == 3 || ksize == 7);   // else: exact fp32 kernel
    if (dtype != EBFI_F32 && dtype != EBFI_F32_BF16MMA && dtype != EBFI_F32_BF16X3MMA)
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight: dtype %d not implemented", dtype);
    if (act != ACT_NONE && !saved_output) return fail(EBFI_ERR_ARG, "conv2d_backward_weight: activation needs saved_output");
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin, H, W, Cout, ksize, stride, pad)) return rc;
    const size_t need = ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, ksize, stride, pad, dtype);
    if (!workspace || workspace_bytes < need)
        return fail(EBFI_ERR_WORKSPACE, "conv2d_backward_weight: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_weight = (int64_t)Cout * Cin * ksize * ksize, n_total = n_weight + Cout;
    if (B == 0) {
        (void)hipMemsetAsync(grad_weight, 0, (size_t)n_weight * sizeof(float), st);
        if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), st);
        return EBFI_OK;
    }
    const float *x = static_cast<const float *>(input), *go = static_cast<const float *>(grad_output);
    const float *yo = static_cast<const float *>(saved_output);
    float *slab = static_cast<float *>(workspace);
    // layers with <= 6 channels on one side: direct fp32 kernels that stream the thick tensor once (conv2d_thin.inc.hpp), whatever
    // operand format was asked for (they are exact)
    // ... unless split precision was asked for: then the matrix cores with the taps on the row axis of the thin side (conv2d_shift.inc.hpp)
    if (x3) {
        if (const ShiftPlan sp = shift_wgrad_plan(g, ksize, stride, input, grad_output, saved_output, grad_preact_out); sp.kind != 0)
            return launch_wgrad_shift(st, sp, x, go, yo, static_cast<float *>(grad_preact_out), slab, g, act, slope,
                                      static_cast<float *>(grad_weight), static_cast<float *>(grad_bias));
    }
    if (const ThinPlan tp = thin_wgrad_plan(g, ksize, stride, input, grad_output, saved_output, grad_preact_out); tp.kind != 0)
        return launch_wgrad_thin(st, tp, x, go, yo, static_cast<float *>(grad_preact_out), slab, g, ksize, act, slope,
                                 static_cast<float *>(grad_weight), static_cast<float *>(grad_bias));
    int nsplit = wgrad_splits(g, ksize, stride);
    if (x3) nsplit = wgrad_x3_splits(g, ksize);       // never more than the fp32 count the workspace is sized for
    if (bf16mma) {   // half-size channel blocks and short tiles: fewer, longer-lived workgroups (workspace is sized for the fp32 count)
        const int64_t blocks = ceil_div(Cout, 64) * ceil_div(Cin, 32);
        const int64_t want = ceil_div(768, blocks);
        if (want < nsplit) nsplit = (int)(want < 1 ? 1 : want);
    }
    const int need_bias = grad_bias != nullptr;
    float *gpre = static_cast<float *>(grad_preact_out);
    if (gpre && bf16mma) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_ex: grad_preact_out with bf16 operands");
    int rc;
    if (x3) rc = launch_wgrad_x3(st, x, go, yo, slab, gpre, g, ksize, act, slope, nsplit, need_bias);
    else if (bf16mma && ksize == 3) rc = launch_wgrad_bf16<3>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else if (bf16mma) rc = launch_wgrad_bf16<1>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else if (ksize == 3 && stride == 1) rc = launch_wgrad<3, 1>(st, x, go, yo, slab, gpre, g, act, slope, nsplit, need_bias);
    else if (ksize == 3 && stride == 2) rc = launch_wgrad<3, 2>(st, x, go, yo, slab, gpre, g, act, slope, nsplit, need_bias);
    else if (ksize == 1 && stride == 1) rc = launch_wgrad<1, 1>(st, x, go, yo, slab, gpre, g, act, slope, nsplit, need_bias);
    else if (ksize == 7 && stride == 1) rc = launch_wgrad<7, 1>(st, x, go, yo, slab, gpre, g, act, slope, nsplit, need_bias);
    else if (ksize == 7 && stride == 2) rc = launch_wgrad<7, 2>(st, x, go, yo, slab, gpre, g, act, slope, nsplit, need_bias);
    else return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight: k=%d stride=%d not implemented", ksize, stride);
    if (rc) return rc;
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st, slab, nsplit,
                           n_weight, n_total, static_cast<float *>(grad_weight), static_cast<float *>(grad_bias),
                           bf16mma ? Cin : 0, ksize * ksize);
    }
    return check_launch("conv_wgrad_reduce_f32");
}

// The thin-layer forward (conv2d_thin.inc.hpp): a 3x3 stride-1 same-padded layer with <= 3 output channels from the fp32 weight
// itself (the (co, tap)-row form needs its own operand layout, not a bank image).  Returns EBFI_ERR_UNSUPPORTED -- without
// recording an error -- for every other shape: the caller then takes its usual entry point.
extern "C" int ebfi_conv2d_thin_forward(const void *input, const void *weight, const void *bias, void *output, int B, int Cin, int H,
                                        int W, int Cout, int ksize, int stride, int pad, int act, float slope, void *stream) {
    if (!input || !weight || !output) return fail(EBFI_ERR_ARG, "conv2d_thin_forward: null argument");
    if (ksize != 3 || stride != 1 || pad != 1 || Cout > 3 || Cin > 64) return EBFI_ERR_UNSUPPORTED;
    if (act < 0 || act > 2) return fail(EBFI_ERR_ARG, "conv2d_thin_forward: activation %d", act);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin, H, W, Cout, ksize, stride, pad)) return rc;
    if (!thin_out_fwd_ok(g, ksize, stride)) return EBFI_ERR_UNSUPPORTED;
    return launch_thin_out_fwd(static_cast<hipStream_t>(stream), static_cast<const float *>(input), static_cast<const float *>(weight),
                               static_cast<const float *>(bias), static_cast<float *>(output), g, act, slope);
}

// ------------------------------------------------------------------------------------------------
// Building blocks of hand-scheduled layer chains (ebfi_amd/rc_fused.py): a split-precision convolution on weights that
// are ALREADY packed (weight bank), optionally grouped, with the epilogue extras of EpiExtra; and the matching grouped
// weight gradient on pre-activation gradients.  The data gradient of a layer is this same forward convolution over its
// gradient with the transposed images, so one entry point serves both directions.
extern "C" int ebfi_conv2d_packed_x3_c16(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                                         int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                         float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                                         void *out16, void *slot16, int out16_planar, void *stream);
extern "C" int ebfi_conv2d_packed_x3(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                                     int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                     float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                                     void *stream) {
    return ebfi_conv2d_packed_x3_c16(input, packed, packed_bytes, bias, output, B, Cin_per_group, H, W, Cout, ksize, pad, groups, act,
                                     slope, addend, mask_y, mask_act, mask_slope, nullptr, nullptr, 0, stream);
}

// Same, plus the output as a scaled fp16 image in the c16 layout (c16.hpp; out16 [B][Cout/16][Ho][Wo][16], scale and |max|
// record in slot16): what the fp16 weight gradient of the NEXT layer stages.  3x3 layers on the wave-specialised kernel only
// (quad-aligned rows, 64-channel blocks); anything else is refused rather than silently skipping the side image.
namespace {
// out_layout (ConvGeom::store): 0 = NCHW, 1 = through PixelShuffle(2), 2 = through its inverse -- fp32 output only
int check_out_layout(const char *who, int out_layout, int Cout, int Ho, int Wo, const void *output, const void *addend, const void *out16) {
    if (out_layout == 0) return EBFI_OK;
    if (out_layout < 0 || out_layout > 2) return fail(EBFI_ERR_ARG, "%s: output layout %d (0 plain, 1 pixel-shuffled, 2 unshuffled)", who, out_layout);
    if (!output || addend || out16) return fail(EBFI_ERR_UNSUPPORTED, "%s: a shuffled output is the fp32 tensor alone (no addend, no fp16 side output)", who);
    if (out_layout == 1 && Cout % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "%s: pixel-shuffled output needs Cout %% 4 == 0 (Cout = %d)", who, Cout);
    if (out_layout == 2 && ((Ho | Wo) & 1)) return fail(EBFI_ERR_UNSUPPORTED, "%s: unshuffled output needs even sizes (%d x %d)", who, Ho, Wo);
    if (((int64_t)Cout + 128) * Ho * Wo * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "%s: one output sample exceeds the 2 GiB reach of 32-bit buffer offsets", who);
    return EBFI_OK;
}

int packed_x3_impl(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                   int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                   float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                   void *out16, void *slot16, int out16_planar, int out_layout, void *stream);
}  // namespace

extern "C" int ebfi_conv2d_packed_x3_c16(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                                         int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                         float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                                         void *out16, void *slot16, int out16_planar, void *stream) {
    return packed_x3_impl(input, packed, packed_bytes, bias, output, B, Cin_per_group, H, W, Cout, ksize, pad, groups, act, slope, addend,
                          mask_y, mask_act, mask_slope, out16, slot16, out16_planar, 0, stream);
}

// The same convolution with its fp32 output stored THROUGH PixelShuffle(2) (out_layout 1: output [B, Cout/4, 2H', 2W']) or through
// its inverse (2: [B, 4Cout, H'/2, W'/2]) -- ConvGeom::store.  Round 6: the reconstruction head (model_singleframe.py:257-260:
// conv 64 -> 256, PixelShuffle(2), LeakyReLU) writes the shuffled tensor itself.
extern "C" int ebfi_conv2d_packed_x3_shuffled(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                                              int B, int Cin, int H, int W, int Cout, int act, float slope, int out_layout, void *stream) {
    return packed_x3_impl(input, packed, packed_bytes, bias, output, B, Cin, H, W, Cout, 3, 1, 1, act, slope, nullptr, nullptr, 0, 0.f,
                          nullptr, nullptr, 0, out_layout, stream);
}

namespace {
int packed_x3_impl(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                   int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                   float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                   void *out16, void *slot16, int out16_planar, int out_layout, void *stream) {
    // out16_planar != 0: out16 is a planar fp16 tensor [B, Cout, H, W] (the FAC op's filters) instead of a c16 image; `output`
    // may be NULL when out16 is given (the fp16 tensor alone)
    if (!input || !packed || (!output && !out16)) return fail(EBFI_ERR_ARG, "conv2d_packed_x3: null argument");
    if ((out16 != nullptr) != (slot16 != nullptr)) return fail(EBFI_ERR_ARG, "conv2d_packed_x3: out16 and slot16 come together");
    if (out16 && ((!out16_planar && Cout % 16 != 0) || ksize != 3 || !aligned16(out16)))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_x3: the fp16 side image needs a 3x3 layer with Cout %% 16 == 0 (Cout = %d)", Cout);
    if (act < 0 || act > 2 || mask_act < 0 || mask_act > 2) return fail(EBFI_ERR_ARG, "conv2d_packed_x3: unknown activation");
    if (ksize != 1 && ksize != 3) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_x3: k=%d", ksize);
    if (groups < 1 || Cout % groups != 0 || (groups > 1 && (Cout / groups) % 64 != 0))
        return fail(EBFI_ERR_ARG, "conv2d_packed_x3: %d output channels in %d groups (groups need multiples of 64 channels)", Cout, groups);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin_per_group, H, W, Cout, ksize, 1, pad)) return rc;
    if ((int64_t)groups * (Cin_per_group + 64) * H * W * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "conv2d_packed_x3: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    g.groups = groups;
    if (int rc = check_out_layout("conv2d_packed_x3", out_layout, Cout, g.Ho, g.Wo, output, addend, out16)) return rc;
    g.store = out_layout;
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((addend || mask_y || out16) && (act == ACT_SIGMOID || mask_act == ACT_SIGMOID))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_x3: the epilogue extras take LeakyReLU / no activation only");
    const EpiExtra epi{static_cast<const float *>(addend), mask_act == ACT_LEAKY ? static_cast<const float *>(mask_y) : nullptr, mask_act,
                       mask_slope, static_cast<_Float16 *>(out16), static_cast<float *>(slot16), out16 ? out16_planar : 0};
    const float *x = static_cast<const float *>(input), *bs = static_cast<const float *>(bias);
    float *o = static_cast<float *>(output);
    void *ws = const_cast<void *>(packed);
    if (ksize == 3) return launch_fwd_bf16<3>(st, x, nullptr, nullptr, bs, o, g, 0, act, slope, 0, 0.f, ws, packed_bytes, 1, epi);
    return launch_fwd_bf16<1>(st, x, nullptr, nullptr, bs, o, g, 0, act, slope, 0, 0.f, ws, packed_bytes, 1, epi);
}
}  // namespace

// The grouped second-layer convolution of a ResidualControl round with the round's tail in its epilogue (EpiExtra, XM bit 2;
// reference model_singleframe.py:127-133): a = LeakyReLU(conv3x3(input) + bias); pre16 = image(a); output = a * post_scale[b, co]
// + post_res[b, co % res_channels]; out16 = image(output).  3x3, same padding, 64-channel blocks, W % 4 == 0.
extern "C" int ebfi_conv2d_packed_x3_rc(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                                        int B, int Cin_per_group, int H, int W, int Cout, int groups, float slope,
                                        const void *post_scale, const void *post_res, int res_channels, void *pre16, void *pre_slot,
                                        void *out16, void *slot16, void *stream) {
    if (!input || !packed || !post_scale || !post_res || !output) return fail(EBFI_ERR_ARG, "conv2d_packed_x3_rc: null argument");
    const bool images = pre16 || pre_slot || out16 || slot16;       // all four (training) or none (inference)
    if (images && (!pre16 || !pre_slot || !out16 || !slot16))
        return fail(EBFI_ERR_ARG, "conv2d_packed_x3_rc: the two images come with their slots, or none of the four is given");
    if (groups < 1 || Cout % groups != 0 || (Cout / groups) % 64 != 0 || res_channels < 64 || res_channels % 64 != 0 || Cout % res_channels != 0)
        return fail(EBFI_ERR_ARG, "conv2d_packed_x3_rc: %d output channels in %d groups, residual of %d channels (multiples of 64)", Cout,
                    groups, res_channels);
    if (images && (!aligned16(pre16) || !aligned16(out16))) return fail(EBFI_ERR_ARG, "conv2d_packed_x3_rc: 16-byte aligned images");
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin_per_group, H, W, Cout, 3, 1, 1)) return rc;
    if ((int64_t)groups * (Cin_per_group + 64) * H * W * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "conv2d_packed_x3_rc: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    g.groups = groups;
    if (B == 0) return EBFI_OK;
    EpiExtra epi{nullptr, nullptr, 0, 0.f, static_cast<_Float16 *>(out16), static_cast<float *>(slot16), 0};
    epi.post_scale = static_cast<const float *>(post_scale);
    epi.post_res = static_cast<const float *>(post_res);
    epi.post_resC = res_channels;
    epi.pre16 = static_cast<_Float16 *>(pre16);
    epi.pre_slot = static_cast<float *>(pre_slot);
    return launch_fwd_bf16<3>(static_cast<hipStream_t>(stream), static_cast<const float *>(input), nullptr, nullptr,
                              static_cast<const float *>(bias), static_cast<float *>(output), g, 0, ACT_LEAKY, slope, 0, 0.f,
                              const_cast<void *>(packed), packed_bytes, 1, epi);
}

// KernelConv (3x3, Cin -> C*25 filters, LeakyReLU) fused with the FAC that consumes the filters: see fac_epilogue_tile.
// `packed`: split-precision forward images of the weight with rows re-tiled to 32 per FAC channel ([C*32, Cin, 3, 3], rows
// 25..31 of every tile zero), `bias32` the bias in the same row layout ([C*32], zeros in the pad rows).
extern "C" int ebfi_kernelconv_fac_fused_x3(const void *input, const void *packed, size_t packed_bytes, const void *bias32,
                                            const void *feat, void *output, int B, int Cin, int H, int W, int C, int fac_ksize,
                                            float slope, void *stream) {
    if (!input || !packed || !bias32 || !feat || !output) return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_x3: null argument");
    if (fac_ksize != 5) return fail(EBFI_ERR_UNSUPPORTED, "kernelconv_fac_fused_x3: FAC kernel size %d (5 is built)", fac_ksize);
    if (C < 1) return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_x3: %d channels", C);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin, H, W, C * 32, 3, 1, 1)) return rc;
    if ((int64_t)(Cin + 64) * H * W * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_x3: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    const int K16 = (Cin + 15) / 16 * 16;
    const size_t need = 2 * bf16_pack_bytes(g.Cout, g.Cin, 3);
    if (packed_bytes < need) return fail(EBFI_ERR_WORKSPACE, "kernelconv_fac_fused_x3: packed images %zu bytes < required %zu", packed_bytes, need);
    // the producers of the wave-specialised kernel stage 16-byte quads: rows must keep quads aligned
    if (W % 4 != 0 || !aligned16(input))
        return fail(EBFI_ERR_UNSUPPORTED, "kernelconv_fac_fused_x3: needs W %% 4 == 0 and a 16-byte aligned input (W = %d)", W);
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t tiles = (int64_t)B * ceil_div(g.Ho, TYB) * ceil_div(g.Wo, TX);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_x3: too many tiles");
    constexpr int PSX = (TYB - 1 + 3) * (TX - 1 + 3);
    const size_t lds = (size_t)2 * (2 * PSX * 32 + 2 * 9 * 32 * 2 * 32) + KB_LDS_BYTES;
    const int64_t co_blocks = ceil_div(g.Cout, 64);
    int64_t gx = 256 / co_blocks;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    const double flops = 2.0 * B * g.Ho * g.Wo * (double)(C * 25) * Cin * 9 + 2.0 * B * g.Ho * g.Wo * (double)C * 25;
    const double bytes = 4.0 * B * (double)g.Ho * g.Wo * (Cin + 2.0 * C);      // conv input + feature map + output: no filter tensor
    ProfScope ps("conv_fwd_bf16x3_ws/kernelconv_fac", st, flops, bytes);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_bf16x3_ws<0, true>), 160 * 1024)) return rc;
    hipLaunchKernelGGL((conv_fwd_bf16x3_ws<0, true>), dim3((unsigned)gx, (unsigned)co_blocks), dim3(NTWS), lds, st,
                       static_cast<const float *>(input), static_cast<const __bf16 *>(packed), static_cast<const float *>(bias32),
                       static_cast<float *>(output), g, K16, ACT_LEAKY, slope, EpiExtra{nullptr, nullptr, 0, 0.f}, (int)tiles,
                       FacEpi{static_cast<const float *>(feat), C});
    return check_launch("conv_fwd_bf16x3_ws/kernelconv_fac");
}

// The same fused pair on fp16 OPERANDS (round 6; inference): conv_fwd_f16_ws<.., FAC> -- one matrix-core product per tap instead
// of three.  `input`: fp32 NCHW -- the producers multiply by in_slot's power of two while converting (the caller sets that scale
// from the tensor itself right before the launch: ebfi_amd.fac) and record |max| there -- or its c16 image; `packed16` is the "facrows" fp16
// image [tap][C * 32][K16] of the layer's weight, scaled by w_slot[0] (ebfi_pack_table_f16).  The training step runs this
// layer on fp16 operands since round 4 (Sharp / Final within 4e-4 of the fp32 result against the path's 1e-3 bar).
extern "C" int ebfi_kernelconv_fac_fused_f16(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                                             const void *bias32, const void *feat, void *output, int B, int Cin, int H, int W, int C,
                                             int fac_ksize, float slope, void *in_slot, const void *w_slot, void *stream) {
    if (!input || !packed16 || !bias32 || !feat || !output || !in_slot || !w_slot)
        return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_f16: null argument");
    if (fac_ksize != 5) return fail(EBFI_ERR_UNSUPPORTED, "kernelconv_fac_fused_f16: FAC kernel size %d (5 is built)", fac_ksize);
    if (C < 1) return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_f16: %d channels", C);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin, H, W, C * 32, 3, 1, 1)) return rc;
    if ((int64_t)(Cin + 64) * H * W * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_f16: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    const int K16 = (Cin + 15) / 16 * 16;
    const size_t need = (size_t)9 * g.Cout * K16 * 2;
    if (packed_bytes < need) return fail(EBFI_ERR_WORKSPACE, "kernelconv_fac_fused_f16: packed image %zu bytes < required %zu", packed_bytes, need);
    if (W % 4 != 0 || !aligned16(input))
        return fail(EBFI_ERR_UNSUPPORTED, "kernelconv_fac_fused_f16: needs W %% 4 == 0 and a 16-byte aligned input (W = %d)", W);
    // input_is_c16: the input as the scaled fp16 image of the tensor (c16.hpp; written by ebfi_to_c16 with in_slot's scale): the
    // producers copy 16-byte pieces -- half the bytes per staged chunk, no conversion.  It pays here because the layer has 25
    // output-channel blocks and every one of them stages the whole input again.
    if (input_is_c16 != 0 && input_is_c16 != 1) return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_f16: input storage %d", input_is_c16);
    if (input_is_c16 && Cin % 16 != 0) return fail(EBFI_ERR_UNSUPPORTED, "kernelconv_fac_fused_f16: an fp16 input image needs Cin %% 16 == 0");
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t tiles = (int64_t)B * ceil_div(g.Ho, TYB) * ceil_div(g.Wo, TX);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "kernelconv_fac_fused_f16: too many tiles");
    constexpr int PSX = (TYB - 1 + 3) * (TX - 1 + 3);
    const size_t lds = (size_t)2 * (PSX * 32 + 9 * 64 * 32) + KB_LDS_BYTES;
    const int64_t co_blocks = ceil_div(g.Cout, 64);
    int64_t gx = 256 / co_blocks;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    const double flops = 2.0 * B * g.Ho * g.Wo * (double)(C * 25) * Cin * 9 + 2.0 * B * g.Ho * g.Wo * (double)C * 25;
    // conv input (in the element size it is stored in) + feature map + output: no filter tensor
    const double bytes = B * (double)g.Ho * g.Wo * ((input_is_c16 ? 2.0 : 4.0) * Cin + 8.0 * C);
    ProfScope ps(input_is_c16 ? "conv_fwd_f16_ws/kernelconv_fac_img" : "conv_fwd_f16_ws/kernelconv_fac", st, flops, bytes);
#define EBFI_LAUNCH_FACF16(IN_)                                                                                          \
    do {                                                                                                                 \
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_f16_ws<false, IN_, false, true>), 160 * 1024)) return rc; \
        hipLaunchKernelGGL((conv_fwd_f16_ws<false, IN_, false, true>), dim3((unsigned)gx, (unsigned)co_blocks), dim3(NTF16), lds, st, \
                           static_cast<const float *>(input), static_cast<const _Float16 *>(packed16),                   \
                           static_cast<const float *>(bias32), static_cast<float *>(output), g, K16, ACT_LEAKY, slope,   \
                           EpiExtra{nullptr, nullptr, 0, 0.f}, (int)tiles, ScaleSlot{static_cast<float *>(in_slot)},    \
                           static_cast<const float *>(w_slot), FacEpi{static_cast<const float *>(feat), C});             \
    } while (0)
    if (input_is_c16) EBFI_LAUNCH_FACF16(true);
    else EBFI_LAUNCH_FACF16(false);
#undef EBFI_LAUNCH_FACF16
    return check_launch("conv_fwd_f16_ws/kernelconv_fac");
}

// ------------------------------------------------------------------------------------------------
// fp16 single-product forms for the BACKWARD pass of the training step (conv2d_f16.inc.hpp): the data gradient as a
// convolution of the (pre-activation) gradient with packed fp16 TRANSPOSED weight images, and the weight gradient, both
// with power-of-two operand scales kept in device slots {scale, running |max|}.
extern "C" int ebfi_conv2d_packed_f16_c16(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                                          const void *bias, void *output, int B, int Cin_per_group, int H, int W, int Cout, int ksize,
                                          int pad, int groups, int act, float slope, const void *addend, const void *mask_y,
                                          int mask_act, float mask_slope, void *in_slot, const void *w_slot, void *out16,
                                          void *slot16, int out16_planar, int mask_is_c16, void *stream);
extern "C" int ebfi_conv2d_packed_f16(const void *input, const void *packed16, size_t packed_bytes, const void *bias, void *output,
                                      int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                      float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                                      void *in_slot, const void *w_slot, void *stream) {
    if (!output) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: null argument");
    return ebfi_conv2d_packed_f16_c16(input, 0, packed16, packed_bytes, bias, output, B, Cin_per_group, H, W, Cout, ksize, pad, groups,
                                      act, slope, addend, mask_y, mask_act, mask_slope, in_slot, w_slot, nullptr, nullptr, 0, 0, stream);
}

// Same with fp16 operand STORAGE (round 4, c16.hpp): input_is_c16 != 0: `input` is the scaled fp16 image
// [B][groups*Cin/16][H][W][16] of the tensor, written by its producer with in_slot's scale (Cin_per_group % 16 == 0);
// out16 / slot16: the output as such an image for the next backward kernel -- in addition to `output`, or alone (output NULL);
// out16_planar != 0: as PLANAR fp16 [B][Cout][H][W] instead (the FAC filters; `output` must then be NULL).
// With a site's FORWARD fp16 weight image this is the fp16-operand forward convolution (bias + LeakyReLU in the epilogue).
// mask_is_c16 != 0: mask_y is the c16 IMAGE of the mask tensor (Cout % 16 == 0; only its signs are read).
namespace {
int packed_f16_impl(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                    const void *bias, void *output, int B, int Cin_per_group, int H, int W, int Cout, int ksize,
                    int pad, int groups, int act, float slope, const void *addend, const void *mask_y,
                    int mask_act, float mask_slope, void *in_slot, const void *w_slot, void *out16,
                    void *slot16, int out16_planar, int mask_is_c16, int out_layout, void *stream);
}  // namespace

extern "C" int ebfi_conv2d_packed_f16_c16(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                                          const void *bias, void *output, int B, int Cin_per_group, int H, int W, int Cout, int ksize,
                                          int pad, int groups, int act, float slope, const void *addend, const void *mask_y,
                                          int mask_act, float mask_slope, void *in_slot, const void *w_slot, void *out16,
                                          void *slot16, int out16_planar, int mask_is_c16, void *stream) {
    return packed_f16_impl(input, input_is_c16, packed16, packed_bytes, bias, output, B, Cin_per_group, H, W, Cout, ksize, pad, groups, act,
                           slope, addend, mask_y, mask_act, mask_slope, in_slot, w_slot, out16, slot16, out16_planar, mask_is_c16, 0, stream);
}

// The fp16-operand convolution with its fp32 output stored through PixelShuffle(2) or its inverse (ConvGeom::store; see
// ebfi_conv2d_packed_x3_shuffled).  With a site's TRANSPOSED image, out_layout 2 and mask_y = the layer's own (shuffled) input this
// is the data gradient of the layer behind a PixelShuffle + LeakyReLU, handed back as the pre-activation gradient of the
// convolution in front of the shuffle, in that convolution's layout: no pixel-unshuffle copy, no mask pass.
extern "C" int ebfi_conv2d_packed_f16_shuffled(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                                               const void *bias, void *output, int B, int Cin, int H, int W, int Cout, int act, float slope,
                                               const void *mask_y, int mask_act, float mask_slope, void *in_slot, const void *w_slot,
                                               int out_layout, void *stream) {
    return packed_f16_impl(input, input_is_c16, packed16, packed_bytes, bias, output, B, Cin, H, W, Cout, 3, 1, 1, act, slope, nullptr, mask_y,
                           mask_act, mask_slope, in_slot, w_slot, nullptr, nullptr, 0, 0, out_layout, stream);
}

namespace {
int packed_f16_impl(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                    const void *bias, void *output, int B, int Cin_per_group, int H, int W, int Cout, int ksize,
                    int pad, int groups, int act, float slope, const void *addend, const void *mask_y,
                    int mask_act, float mask_slope, void *in_slot, const void *w_slot, void *out16,
                    void *slot16, int out16_planar, int mask_is_c16, int out_layout, void *stream) {
    if (!input || !packed16 || (!output && !out16)) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: null argument");
    if ((out16 != nullptr) != (slot16 != nullptr)) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: out16 and slot16 come together");
    if (out16 && !out16_planar && (Cout % 16 != 0 || !aligned16(out16)))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: fp16 output image needs Cout %% 16 == 0");
    if (mask_is_c16 && mask_y && (Cout % 16 != 0 || !aligned16(mask_y)))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: an fp16 mask image needs Cout %% 16 == 0");
    if (out16_planar && (!out16 || output || W % 4 != 0 || !aligned16(out16)))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: planar fp16 output comes alone (output NULL), W %% 4 == 0");
    // input_is_c16: 0 = fp32 NCHW, 1 = c16 image, 2 = planar fp16 [B, groups*Cin, H, W] (scaled by in_slot like an image)
    if (input_is_c16 < 0 || input_is_c16 > 2) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: input storage %d", input_is_c16);
    if (input_is_c16 == 1 && (Cin_per_group % 16 != 0 || !in_slot))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: an fp16 input image needs Cin %% 16 == 0 and its scale slot");
    if (input_is_c16 == 2 && !in_slot) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: a planar fp16 input needs its scale slot");
    if (act < 0 || act > 2 || mask_act < 0 || mask_act > 2) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: unknown activation");
    if (ksize != 3 || pad != 1) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: 3x3 same-padded convolutions only (k=%d pad=%d)", ksize, pad);
    if (groups < 1 || Cout % groups != 0 || (groups > 1 && (Cout / groups) % 64 != 0))
        return fail(EBFI_ERR_ARG, "conv2d_packed_f16: %d output channels in %d groups (groups need multiples of 64 channels)", Cout, groups);
    if ((input_is_c16 != 1 && W % 4 != 0) || !aligned16(input))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: needs W %% 4 == 0 and a 16-byte aligned input (W = %d)", W);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin_per_group, H, W, Cout, ksize, 1, pad)) return rc;
    if ((int64_t)groups * (Cin_per_group + 64) * H * W * 4 >= (1LL << 31) - (1LL << 26) || (int64_t)(Cout + 64) * H * W * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "conv2d_packed_f16: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    g.groups = groups;
    if (int rc = check_out_layout("conv2d_packed_f16", out_layout, Cout, g.Ho, g.Wo, output, addend, out16)) return rc;
    g.store = out_layout;
    const int K16 = (g.Cin + 15) / 16 * 16;
    const size_t need = (size_t)9 * g.Cout * K16 * 2;
    if (packed_bytes < need) return fail(EBFI_ERR_WORKSPACE, "conv2d_packed_f16: packed image %zu bytes < required %zu", packed_bytes, need);
    if (B == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((addend || mask_y || out16) && (act == ACT_SIGMOID || mask_act == ACT_SIGMOID))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: the epilogue extras take LeakyReLU / no activation only");
    const bool m16 = mask_is_c16 != 0 && mask_y != nullptr && mask_act == ACT_LEAKY;
    const EpiExtra epi{static_cast<const float *>(addend), (mask_act == ACT_LEAKY && !m16) ? static_cast<const float *>(mask_y) : nullptr,
                       mask_act, mask_slope, static_cast<_Float16 *>(out16), static_cast<float *>(slot16), out16_planar ? 1 : 0,
                       m16 ? static_cast<const _Float16 *>(mask_y) : nullptr};
    const int64_t tiles = (int64_t)B * ceil_div(g.Ho, TYB) * ceil_div(g.Wo, TX);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_packed_f16: too many tiles");
    constexpr int PSX = (TYB - 1 + 3) * (TX - 1 + 3);
    const size_t lds = (size_t)2 * (PSX * 32 + 9 * 64 * 32) + KB_LDS_BYTES;
    const int64_t co_blocks = ceil_div(g.Cout, 64);
    int64_t gx = 256 / co_blocks;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    const dim3 grid((unsigned)gx, (unsigned)co_blocks);
    const bool extra = epi.addend != nullptr || epi.mask_y != nullptr || epi.mask16 != nullptr || epi.out16 != nullptr;
    const double flops = 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * 9;
    // algorithmic bytes: input + output in the element size they are stored in (fp16 images: 2 bytes) + the packed weights
    const double px = (double)g.B * g.Ho * g.Wo;
    const double io_bytes = px * g.groups * g.Cin * (input_is_c16 ? 2.0 : 4.0) + px * g.Cout * ((output ? 4.0 : 0.0) + (out16 ? 2.0 : 0.0)) +
                            px * g.Cout * ((addend ? 4.0 : 0.0) + (mask_y ? (m16 ? 2.0 : 4.0) : 0.0)) + 2.0 * 9 * (double)g.Cout * g.Cin;
    // (label = kernel symbol / role: which operand storage the launch read and wrote)
    ProfScope ps(g.store ? (input_is_c16 ? "conv_fwd_f16_ws/img_shuffle" : "conv_fwd_f16_ws/f32_shuffle") :
                 input_is_c16 == 2 ? "conv_fwd_f16_ws/p16_f32" : out16_planar ? (input_is_c16 ? "conv_fwd_f16_ws/img_p16" : "conv_fwd_f16_ws/f32_p16") :
                 input_is_c16 ? (out16 ? (output ? "conv_fwd_f16_ws/img_both" : "conv_fwd_f16_ws/img_img") : "conv_fwd_f16_ws/img_f32")
                              : (out16 ? "conv_fwd_f16_ws/f32_img" : "conv_fwd_f16_ws/f32_f32"), st, flops, io_bytes);
    const ScaleSlot isl{static_cast<float *>(in_slot)};
    const float *x = static_cast<const float *>(input), *bs = static_cast<const float *>(bias);
    const _Float16 *wp = static_cast<const _Float16 *>(packed16);
    float *o = static_cast<float *>(output);
#define EBFI_LAUNCH_F16WS(EX_, IN_)                                                                                      \
    do {                                                                                                                 \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_f16_ws<EX_, IN_>), 160 * 1024)) return rc_; \
        hipLaunchKernelGGL((conv_fwd_f16_ws<EX_, IN_>), grid, dim3(NTF16), lds, st, x, wp, bs, o, g, K16, act, slope, epi,   \
                           (int)tiles, isl, static_cast<const float *>(w_slot), FacEpi{nullptr, 0});                    \
    } while (0)
    if (input_is_c16 == 2) {
        if (extra || g.store) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_packed_f16: planar fp16 input with epilogue extras / a shuffled output");
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_f16_ws<false, false, true>), 160 * 1024)) return rc_;
        hipLaunchKernelGGL((conv_fwd_f16_ws<false, false, true>), grid, dim3(NTF16), lds, st, x, wp, bs, o, g, K16, act, slope, epi,
                           (int)tiles, isl, static_cast<const float *>(w_slot), FacEpi{nullptr, 0});
    } else if (g.store != 0) {             // output through PixelShuffle(2) / its inverse: the EXTRA form built with the layouts
#define EBFI_LAUNCH_F16ST(IN_)                                                                                           \
    do {                                                                                                                 \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_fwd_f16_ws<true, IN_, false, false, true>), 160 * 1024)) return rc_; \
        hipLaunchKernelGGL((conv_fwd_f16_ws<true, IN_, false, false, true>), grid, dim3(NTF16), lds, st, x, wp, bs, o, g, K16, act, slope, epi, \
                           (int)tiles, isl, static_cast<const float *>(w_slot), FacEpi{nullptr, 0});                    \
    } while (0)
        if (input_is_c16) EBFI_LAUNCH_F16ST(true);
        else EBFI_LAUNCH_F16ST(false);
#undef EBFI_LAUNCH_F16ST
    } else if (extra && input_is_c16) EBFI_LAUNCH_F16WS(true, true);
    else if (extra) EBFI_LAUNCH_F16WS(true, false);
    else if (input_is_c16) EBFI_LAUNCH_F16WS(false, true);
    else EBFI_LAUNCH_F16WS(false, false);
#undef EBFI_LAUNCH_F16WS
    return check_launch("conv_fwd_f16_ws");
}
}  // namespace

// weight / bias gradient of a (grouped) 3x3 convolution with fp16 operands: grad_output optionally times act'(saved_output)
// (side output grad_preact_out as in ebfi_conv2d_backward_weight_ex); Cin_per_group a multiple of 64.
extern "C" int ebfi_conv2d_backward_weight_f16g_ex(const void *input, const void *grad_output, const void *saved_output,
                                                   void *grad_weight, void *grad_bias, void *grad_preact_out, int grad_preact_is_c16,
                                                   int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups,
                                                   int act, float slope, void *x_slot, void *g_slot, void *workspace,
                                                   size_t workspace_bytes, void *stream);
extern "C" int ebfi_conv2d_backward_weight_f16g(const void *input, const void *grad_output, const void *saved_output,
                                                void *grad_weight, void *grad_bias, void *grad_preact_out, int B,
                                                int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                                float slope, void *x_slot, void *g_slot, void *workspace,
                                                size_t workspace_bytes, void *stream) {
    return ebfi_conv2d_backward_weight_f16g_ex(input, grad_output, saved_output, grad_weight, grad_bias, grad_preact_out, 0, B,
                                               Cin_per_group, H, W, Cout, ksize, pad, groups, act, slope, x_slot, g_slot, workspace,
                                               workspace_bytes, stream);
}

// grad_preact_is_c16 != 0: grad_preact_out (grad_output * act'(saved_output), the data gradient's input) leaves as a c16 IMAGE
// scaled by g_slot's scale instead of an fp32 tensor (Cout % 16 == 0; only with the pixel-major kernel: 3x3, pad 1, W % 4 == 0,
// 16-byte aligned tensors -- otherwise EBFI_ERR_UNSUPPORTED); ebfi_conv2d_packed_f16_c16(input_is_c16 = 1, in_slot = g_slot)
// reads it.
extern "C" int ebfi_conv2d_backward_weight_f16g_ex(const void *input, const void *grad_output, const void *saved_output,
                                                   void *grad_weight, void *grad_bias, void *grad_preact_out, int grad_preact_is_c16,
                                                   int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups,
                                                   int act, float slope, void *x_slot, void *g_slot, void *workspace,
                                                   size_t workspace_bytes, void *stream) {
    if (!input || !grad_output || !grad_weight) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16g: null argument");
    const bool gp16 = grad_preact_is_c16 != 0 && grad_preact_out != nullptr;
    if (gp16 && (act == ACT_NONE || Cout % 16 != 0 || !g_slot))
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16g: an fp16 image of grad * act' needs an activation, Cout %% 16 == 0 and g_slot");
    if (ksize != 3) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16g: k=%d", ksize);
    // input channels: multiples of 64 for the pair-word kernel; the pixel-major kernel zero-fills a partial 64-channel block
    // (loads past the sample's last channel read 0, columns past Cin are not stored), which serves the 32 / 48-channel layers
    // of the detail branch -- at some wasted matrix work, which is not what bounds these launches
    const bool ragged_ci = Cin_per_group % 64 != 0;
    if (ragged_ci && (groups != 1 || Cin_per_group < 16))
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16g: %d input channels per group", Cin_per_group);
    if (act < 0 || act > 2 || (act != ACT_NONE && !saved_output)) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16g: activation / saved_output");
    if (groups < 1 || Cout % groups != 0 || (groups > 1 && (Cout / groups) % 64 != 0))
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16g: %d output channels in %d groups", Cout, groups);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin_per_group, H, W, Cout, ksize, 1, pad)) return rc;
    g.groups = groups;
    const size_t need = ebfi_conv2d_backward_weight_workspace(B, Cin_per_group, H, W, Cout, ksize, 1, pad, EBFI_F32);
    if (!workspace || workspace_bytes < need)
        return fail(EBFI_ERR_WORKSPACE, "conv2d_backward_weight_f16g: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_weight = (int64_t)Cout * Cin_per_group * 9, n_total = n_weight + Cout;
    if (B == 0) {
        (void)hipMemsetAsync(grad_weight, 0, (size_t)n_weight * sizeof(float), st);
        if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), st);
        return EBFI_OK;
    }
    float *slab = static_cast<float *>(workspace);
    if (!gp16) {       // thin layers (64 -> 3, 64 -> 1): split precision on the matrix cores / the direct fp32 kernel, no operand scaling involved
        if (const ShiftPlan sp = shift_wgrad_plan(g, 3, 1, input, grad_output, saved_output, grad_preact_out); sp.kind != 0)
            return launch_wgrad_shift(st, sp, static_cast<const float *>(input), static_cast<const float *>(grad_output),
                                      static_cast<const float *>(saved_output), static_cast<float *>(grad_preact_out), slab, g, act, slope,
                                      static_cast<float *>(grad_weight), static_cast<float *>(grad_bias));
        if (const ThinPlan tp = thin_wgrad_plan(g, 3, 1, input, grad_output, saved_output, grad_preact_out); tp.kind != 0)
            return launch_wgrad_thin(st, tp, static_cast<const float *>(input), static_cast<const float *>(grad_output),
                                     static_cast<const float *>(saved_output), static_cast<float *>(grad_preact_out), slab, g, 3, act, slope,
                                     static_cast<float *>(grad_weight), static_cast<float *>(grad_bias));
    }
    int nsplit = wgrad_x3_splits(g, 3);
    // pre-activation gradients on quad-aligned rows: the pixel-major kernel with transposing LDS reads (conv_wgrad_f16_tr)
    // (EBFI_WGRAD_TR=0 keeps the pair-word kernel for A/B runs.  A first in-step measurement had this kernel at 102 us against
    // 72 us on the 64 / 128-channel layers although it was faster in isolation; after the later changes of the round -- operand
    // scales and their running maxima on separate cache lines, channel blocks placed per XCD -- it runs 54-63 us inside the
    // step as well: 18.95 -> 18.45 ms per step with it on every eligible layer.)
    const char *tr_env = dev_getenv("EBFI_WGRAD_TR");
    const int64_t tr_tiles = (int64_t)g.B * ceil_div(g.Ho, TRH) * ceil_div(g.Wo, TRW);
    const bool tr_pays = !(tr_env && tr_env[0] == '0');
    const bool tr_ok = pad == 1 && W % 4 == 0 && aligned16(input) && aligned16(grad_output) && tr_pays &&
                       (act == ACT_NONE || aligned16(saved_output)) && (!grad_preact_out || aligned16(grad_preact_out));
    if (tr_ok) {
        const int64_t tiles = tr_tiles;
        if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16g: too many tiles");
        if (nsplit > tiles) nsplit = (int)tiles;
        dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, 64));
        const ScaleSlot xs{static_cast<float *>(x_slot)}, gs{static_cast<float *>(g_slot)};
        ProfScope ps("conv_wgrad_f16_tr/f32", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * 9,
                     conv_bytes_wgrad(g, 9, act != ACT_NONE, grad_preact_out != nullptr) -
                         (gp16 ? 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout : 0.0));
#define EBFI_LAUNCH_WTR(DA_)                                                                                               \
    do {                                                                                                                   \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_f16_tr<DA_>), TR_LDS + KB_LDS_BYTES)) return rc_;       \
        hipLaunchKernelGGL((conv_wgrad_f16_tr<DA_>), grid, dim3(512), TR_LDS + KB_LDS_BYTES, st, static_cast<const float *>(input),           \
                           static_cast<const float *>(grad_output), static_cast<const float *>(saved_output),              \
                           static_cast<float *>(grad_preact_out), slab, g, slope, (int)tiles, grad_bias != nullptr ? 1 : 0,  \
                           xs, gs, gp16 ? 1 : 0);                                                                          \
    } while (0)
        if (act == ACT_LEAKY) EBFI_LAUNCH_WTR(ACT_LEAKY);
        else if (act == ACT_SIGMOID) EBFI_LAUNCH_WTR(ACT_SIGMOID);
        else EBFI_LAUNCH_WTR(ACT_NONE);
#undef EBFI_LAUNCH_WTR
        if (int rc = check_launch("conv_wgrad_f16_tr")) return rc;
    } else {
        if (gp16) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16g: the fp16 image of grad * act' needs the pixel-major kernel "
                                                    "(3x3, pad 1, W %% 4 == 0, 16-byte aligned tensors)");
        if (ragged_ci)
            return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16g: %d input channels need the pixel-major kernel (3x3, pad 1, W %% 4 == 0, "
                                              "16-byte aligned tensors)", Cin_per_group);
        using C = WCfg<3, 1, 32>;
        const size_t lds = (size_t)2 * (32 * GS + 33 * C::PS) * sizeof(unsigned);
        const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, C::WTX);
        dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, 64));
        const float *x = static_cast<const float *>(input), *go = static_cast<const float *>(grad_output);
        const float *yo = static_cast<const float *>(saved_output);
        float *gpre = static_cast<float *>(grad_preact_out);
        const ScaleSlot xs{static_cast<float *>(x_slot)}, gs{static_cast<float *>(g_slot)};
        const int need_bias = grad_bias != nullptr;
        ProfScope ps("conv_wgrad_f16_ws", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * 9,
                     conv_bytes_wgrad(g, 9, act != ACT_NONE, gpre != nullptr));
#define EBFI_LAUNCH_WF16(DA_)                                                                                              \
    do {                                                                                                                   \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_f16_ws<DA_>), (int)lds)) return rc_;     \
        hipLaunchKernelGGL((conv_wgrad_f16_ws<DA_>), grid, dim3(512), lds, st, x, go, yo, slab, gpre, g, slope, (int)tiles,    \
                           need_bias, xs, gs);                                                                             \
    } while (0)
        if (act == ACT_LEAKY) EBFI_LAUNCH_WF16(ACT_LEAKY);
        else if (act == ACT_SIGMOID) EBFI_LAUNCH_WF16(ACT_SIGMOID);
        else EBFI_LAUNCH_WF16(ACT_NONE);
#undef EBFI_LAUNCH_WF16
        if (int rc = check_launch("conv_wgrad_f16_ws")) return rc;
    }
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st, slab, nsplit, n_weight,
                           n_total, static_cast<float *>(grad_weight), static_cast<float *>(grad_bias), tr_ok ? Cin_per_group : 0, 9);
    }
    return check_launch("conv_wgrad_reduce_f32");
}

// Weight / bias gradient of a (grouped) 3x3 same-padded convolution from the fp16 c16 images (c16.hpp) of its input and of
// its PRE-activation gradient: input16 [B][groups*Cin/16][H][W][16] scaled by x_slot[0], grad16 [B][Cout/16][H][W][16] scaled
// by g_slot[0] (both written, and their |max| recorded, by the kernels that produced the tensors).  Same slabs, same
// deterministic reduction and the same workspace size as ebfi_conv2d_backward_weight_f16g.
extern "C" int ebfi_conv2d_backward_weight_f16c(const void *input16, const void *grad16, int grad_is_planar, void *grad_weight,
                                                void *grad_bias, int B, int Cin_per_group, int H, int W, int Cout, int groups,
                                                const void *x_slot, const void *g_slot, void *workspace, size_t workspace_bytes,
                                                void *stream) {
    if (!input16 || !grad16 || !grad_weight || !x_slot || !g_slot) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c: null argument");
    // grad_is_planar != 0: grad16 is a planar fp16 tensor [B, Cout, H, W] (scaled by g_slot) instead of a c16 image
    if (Cin_per_group % 16 != 0 || (!grad_is_planar && Cout % 16 != 0) || Cin_per_group < 16)
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16c: channel counts must be multiples of 16 (%d -> %d)", Cin_per_group, Cout);
    if (grad_is_planar && W % 4 != 0) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16c: planar gradient rows need W %% 4 == 0");
    if (groups < 1 || Cout % groups != 0 || (groups > 1 && ((Cout / groups) % 64 != 0 || Cin_per_group % 64 != 0)))
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c: %d output channels in %d groups", Cout, groups);
    if (!aligned16(input16) || !aligned16(grad16)) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c: images must be 16-byte aligned");
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin_per_group, H, W, Cout, 3, 1, 1)) return rc;
    g.groups = groups;
    if ((int64_t)(groups * Cin_per_group + 64) * H * W * 2 >= (1LL << 31) - (1LL << 26) || (int64_t)(Cout + 64) * H * W * 2 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    const size_t need = ebfi_conv2d_backward_weight_workspace(B, Cin_per_group, H, W, Cout, 3, 1, 1, EBFI_F32);
    if (!workspace || workspace_bytes < need)
        return fail(EBFI_ERR_WORKSPACE, "conv2d_backward_weight_f16c: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_weight = (int64_t)Cout * Cin_per_group * 9, n_total = n_weight + Cout;
    if (B == 0) {
        (void)hipMemsetAsync(grad_weight, 0, (size_t)n_weight * sizeof(float), st);
        if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), st);
        return EBFI_OK;
    }
    float *slab = static_cast<float *>(workspace);
    int nsplit = wgrad_x3_splits(g, 3);
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, TRH) * ceil_div(g.Wo, TRW);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c: too many tiles");
    if (nsplit > tiles) nsplit = (int)tiles;
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, 64));
    const ScaleSlot xs{const_cast<float *>(static_cast<const float *>(x_slot))}, gs{const_cast<float *>(static_cast<const float *>(g_slot))};
    {
        const double px = (double)g.B * g.Ho * g.Wo;
        ProfScope ps("conv_wgrad_f16_tr/img", st, 2.0 * px * (double)g.Cout * g.Cin * 9,
                     2.0 * px * (g.groups * g.Cin + g.Cout) + 4.0 * (double)n_total);
        if (grad_is_planar) {
            if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_f16_tr<ACT_NONE, true, true>), TR_LDS + KB_LDS_BYTES)) return rc_;
            hipLaunchKernelGGL((conv_wgrad_f16_tr<ACT_NONE, true, true>), grid, dim3(512), TR_LDS + KB_LDS_BYTES, st,
                               static_cast<const float *>(input16), static_cast<const float *>(grad16), static_cast<const float *>(nullptr),
                               static_cast<float *>(nullptr), slab, g, 0.f, (int)tiles, grad_bias != nullptr ? 1 : 0, xs, gs);
        } else {
            if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_f16_tr<ACT_NONE, true>), TR_LDS + KB_LDS_BYTES)) return rc_;
            hipLaunchKernelGGL((conv_wgrad_f16_tr<ACT_NONE, true>), grid, dim3(512), TR_LDS + KB_LDS_BYTES, st,
                               static_cast<const float *>(input16), static_cast<const float *>(grad16), static_cast<const float *>(nullptr),
                               static_cast<float *>(nullptr), slab, g, 0.f, (int)tiles, grad_bias != nullptr ? 1 : 0, xs, gs);
        }
        if (int rc = check_launch("conv_wgrad_f16_tr")) return rc;
    }
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st, slab, nsplit, n_weight,
                           n_total, static_cast<float *>(grad_weight), static_cast<float *>(grad_bias), Cin_per_group, 9);
    }
    return check_launch("conv_wgrad_reduce_f32");
}

// n (1..4) weight gradients over the SAME [B, H, W] pixels in one launch + one reduction launch (conv_wgrad_f16_tr_batch): the
// layers of one ResidualControl round.  Every layer must consist of exactly two 64 x 64 blocks (Cout x Cin_per_group = 128 x 64,
// 64 x 128, or 2 groups of 64 x 64); anything else: EBFI_ERR_UNSUPPORTED (callers then use ebfi_conv2d_backward_weight_f16c per
// layer).  Arguments are arrays of n entries; workspace: ebfi_conv2d_backward_weight_f16c_batch_workspace bytes.
static int wgrad_batch_per_xcd(int n) { return 32 / (2 * n); }
extern "C" size_t ebfi_conv2d_backward_weight_f16c_batch_workspace(int n, const int *Cin_per_group, const int *Cout) {
    if (n < 1 || n > WGRAD_BATCH_MAX || !Cin_per_group || !Cout) return 0;
    size_t total = 0;
    for (int k = 0; k < n; ++k) total += (size_t)8 * wgrad_batch_per_xcd(n) * ((size_t)Cout[k] * Cin_per_group[k] * 9 + Cout[k]) * sizeof(float);
    return total;
}
extern "C" int ebfi_conv2d_backward_weight_f16c_batch(int n, const void *const *input16, const void *const *grad16,
                                                      void *const *grad_weight, void *const *grad_bias, const int *Cin_per_group,
                                                      const int *Cout, const int *groups, void *const *x_slot, void *const *g_slot,
                                                      int B, int H, int W, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 1 || n > WGRAD_BATCH_MAX) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c_batch: %d layers (1..%d)", n, WGRAD_BATCH_MAX);
    if (!input16 || !grad16 || !grad_weight || !grad_bias || !Cin_per_group || !Cout || !groups || !x_slot || !g_slot)
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c_batch: null argument");
    const int per_xcd = wgrad_batch_per_xcd(n);
    ConvGeom g0;
    if (int rc = make_geom(g0, B, 64, H, W, 64, 3, 1, 1)) return rc;
    const int64_t tiles = (int64_t)B * ceil_div(g0.Ho, TRH) * ceil_div(g0.Wo, TRW);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c_batch: too many tiles");
    if (tiles < 8 * per_xcd) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16c_batch: %lld tiles do not fill %d splits", (long long)tiles, 8 * per_xcd);
    const size_t need = ebfi_conv2d_backward_weight_f16c_batch_workspace(n, Cin_per_group, Cout);
    if (!workspace || workspace_bytes < need)
        return fail(EBFI_ERR_WORKSPACE, "conv2d_backward_weight_f16c_batch: workspace %zu bytes < required %zu", workspace_bytes, need);
    WgradBatch bt{};
    ReduceBatch rb{};
    float *slab = static_cast<float *>(workspace);
    double flops = 0.0, bytes = 0.0;
    int64_t n_total_max = 0;
    const double px = (double)B * H * W;
    for (int k = 0; k < n; ++k) {
        const int ci = Cin_per_group[k], co = Cout[k], gr = groups[k];
        if (!input16[k] || !grad16[k] || !grad_weight[k] || !x_slot[k] || !g_slot[k])
            return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c_batch: null argument (layer %d)", k);
        if (gr < 1 || ci < 16 || ci % 16 != 0 || co % 16 != 0 || co % gr != 0 || (gr > 1 && ((co / gr) % 64 != 0 || ci % 64 != 0)) ||
            ceil_div(co, 64) * ceil_div(ci, 64) != 2)
            return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_f16c_batch: layer %d (%d -> %d, %d groups) is not two 64 x 64 blocks", k, ci, co, gr);
        if (!aligned16(input16[k]) || !aligned16(grad16[k])) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c_batch: images must be 16-byte aligned");
        if ((int64_t)(gr * ci + 64) * H * W * 2 >= (1LL << 31) - (1LL << 26) || (int64_t)(co + 64) * H * W * 2 >= (1LL << 31) - (1LL << 26))
            return fail(EBFI_ERR_ARG, "conv2d_backward_weight_f16c_batch: one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
        const int64_t n_weight = (int64_t)co * ci * 9, n_total = n_weight + co;
        bt.item[k] = WgradItem{input16[k], grad16[k], slab, static_cast<float *>(x_slot[k]), static_cast<float *>(g_slot[k]), ci, co, gr,
                               grad_bias[k] != nullptr ? 1 : 0};
        rb.item[k] = ReduceItem{slab, static_cast<float *>(grad_weight[k]), static_cast<float *>(grad_bias[k]), n_weight, n_total, ci};
        slab += (size_t)8 * per_xcd * n_total;
        n_total_max = n_total > n_total_max ? n_total : n_total_max;
        flops += 2.0 * px * (double)co * ci * 9;
        bytes += 2.0 * px * (gr * ci + co) + 4.0 * (double)n_total;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B == 0) {
        for (int k = 0; k < n; ++k) {
            (void)hipMemsetAsync(grad_weight[k], 0, (size_t)rb.item[k].n_weight * sizeof(float), st);
            if (grad_bias[k]) (void)hipMemsetAsync(grad_bias[k], 0, (size_t)Cout[k] * sizeof(float), st);
        }
        return EBFI_OK;
    }
    {
        ProfScope ps("conv_wgrad_f16_tr/img_batch", st, flops, bytes);
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wgrad_f16_tr_batch), TR_LDS + KB_LDS_BYTES)) return rc_;
        hipLaunchKernelGGL(conv_wgrad_f16_tr_batch, dim3((unsigned)(8 * 2 * n * per_xcd)), dim3(512), TR_LDS + KB_LDS_BYTES, st, bt, g0,
                           (int)tiles, per_xcd, n);
        if (int rc = check_launch("conv_wgrad_f16_tr_batch")) return rc;
    }
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_batch, dim3((unsigned)ceil_div(n_total_max, 64), (unsigned)n), dim3(256), 0, st, rb, 8 * per_xcd);
    }
    return check_launch("conv_wgrad_reduce_batch");
}

extern "C" int ebfi_f16_scales_finish(void *slots, int n, void *flag, void *stream) {
    if (!slots || !flag || n < 0) return fail(EBFI_ERR_ARG, "f16_scales_finish: bad argument");
    if (n == 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps("f16_scales_finish", st);
    hipLaunchKernelGGL(f16_scales_finish_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, static_cast<float *>(slots), n,
                       static_cast<int *>(flag));
    return check_launch("f16_scales_finish");
}

extern "C" int ebfi_pack_table_f16(const float *src, const int32_t *table, int64_t n, void *out, const int32_t *block_slot,
                                   void *slots, void *stream) {
    if (!src || !table || !out || !block_slot || !slots) return fail(EBFI_ERR_ARG, "pack_table_f16: null argument");
    if (n <= 0) return EBFI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps("pack_table_f16", st);
    hipLaunchKernelGGL(pack_table_f16_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, src, table, n,
                       static_cast<_Float16 *>(out), block_slot, static_cast<float *>(slots));
    return check_launch("pack_table_f16");
}

extern "C" int ebfi_conv2d_backward_weight_x3g(const void *input, const void *grad_output, void *grad_weight, void *grad_bias,
                                               int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups,
                                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!input || !grad_output || !grad_weight) return fail(EBFI_ERR_ARG, "conv2d_backward_weight_x3g: null argument");
    if (ksize != 1 && ksize != 3) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight_x3g: k=%d", ksize);
    if (groups < 1 || Cout % groups != 0 || (groups > 1 && (Cout / groups) % 64 != 0))
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight_x3g: %d output channels in %d groups", Cout, groups);
    ConvGeom g;
    if (int rc = make_geom(g, B, Cin_per_group, H, W, Cout, ksize, 1, pad)) return rc;
    g.groups = groups;
    const size_t need = ebfi_conv2d_backward_weight_workspace(B, Cin_per_group, H, W, Cout, ksize, 1, pad, EBFI_F32);
    if (!workspace || workspace_bytes < need)
        return fail(EBFI_ERR_WORKSPACE, "conv2d_backward_weight_x3g: workspace %zu bytes < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_weight = (int64_t)Cout * Cin_per_group * ksize * ksize, n_total = n_weight + Cout;
    if (B == 0) {
        (void)hipMemsetAsync(grad_weight, 0, (size_t)n_weight * sizeof(float), st);
        if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), st);
        return EBFI_OK;
    }
    float *slab = static_cast<float *>(workspace);
    const int nsplit = wgrad_x3_splits(g, ksize);
    if (int rc = launch_wgrad_x3(st, static_cast<const float *>(input), static_cast<const float *>(grad_output), nullptr, slab, nullptr,
                                 g, ksize, ACT_NONE, 0.f, nsplit, grad_bias != nullptr))
        return rc;
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st, slab, nsplit, n_weight,
                           n_total, static_cast<float *>(grad_weight), static_cast<float *>(grad_bias), 0, ksize * ksize);
    }
    return check_launch("conv_wgrad_reduce_f32");
}

// fp16 operand storage of the backward pass (round 4): helpers shared by conv2d.hip and fuse.hip.
//
// Layout "c16": a tensor [B, C, H, W] (C a multiple of 16) kept as fp16 [B][C/16][H][2][W][8] -- blocks of 16 channels; inside
// a block one image ROW holds first the channels 0..7 of its W pixels (16 bytes per pixel), then the channels 8..15 --
// multiplied by the power-of-two scale of its slot.  It is the STAGING layout of both backward kernels: the data gradient
// (conv_fwd_f16_ws) walks 16-channel chunks whose LDS image is [position][16 ch], the weight gradient (conv_wgrad_f16_tr)
// keeps [pixel][64 ch] images = four blocks; the unit both copy is the 16-byte piece (pixel, 8 channels), and a tile row is
// two contiguous runs of pieces, so the producers are plain 16-byte copies (no conversion, half the bytes of the fp32 NCHW
// planes, a quarter of the vector-memory instructions).  The half-row split is what makes the WRITERS coalesce: a 32x32 MFMA
// accumulator leaves channels 0..7 of 32 consecutive pixels in the lower lanes and channels 8..15 in the upper ones (after
// one v_permlane32_swap per register pair) -- two contiguous 512-byte runs per store instruction; the first form, with a
// pixel's 16 channels in 32 contiguous bytes, wrote half-lines and ran SLOWER than the fp32 stores it replaced.  The tensors
// are written in this form by whoever produces them: the epilogues of the forward / data-gradient kernels (store_out_tile),
// the fused ResidualControl stages (fuse.hip) or the standalone conversion (to_c16_kernel).  The WRITER applies the slot's
// scale and records |max|; readers only need the scale.
#pragma once
#include <hip/hip_runtime.h>

namespace ebfi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_f16(float a, float b) {   // v_cvt_pk_f16_f32: round to nearest even, a in the low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
}

// fp32 -> fp16 conversions of this wave saturate at +-65504 instead of producing inf (MODE.FP16_OVFL; true inf / NaN inputs
// stay what they are).  A scale that is too large for this step's data is caught by f16_scales_finish_kernel from the
// recorded maximum either way; saturating keeps the step's wrong gradients FINITE, so the maxima recorded further down the
// backward chain stay usable and every scale is repaired by that one finish launch instead of one layer per step.
//
// ROUND 6 -- the bit is NOT confined to conversions.  While MODE.FP16_OVFL is set the MATRIX CORES stop propagating non-finite
// operands: one NaN / +Inf / -Inf element in an operand of v_mfma_f32_32x32x16_bf16 or _f16 gives 32 non-finite results in the
// default mode and NONE with the bit set (tools/mfma_nan_probe.hip, profiles/r06/mfma_fp16_ovfl_nonfinite_probe.log).  That is
// the "NaN the forward kernel swallows" of round 5: the consumer waves of the image-writing forward ran their whole MFMA loop
// with the bit set for the sake of the epilogue's fp16 stores.  Rule: a wave that issues MFMAs sets the bit only AROUND its own
// fp16 conversions (saturate_fp16_conversions(true) ... (false) around an epilogue) and never across a matrix loop; waves that
// only convert (producers, elementwise kernels) may set it for good.
__device__ __forceinline__ void saturate_fp16_conversions(bool on = true) { __builtin_amdgcn_s_setreg(1 | (23 << 6), on ? 1u : 0u); }

// Scale slot: 64 floats (256 bytes); [0] = scale (power of two), [32] = running |max| of the fp32 values staged through it
// (float bits, ordered as unsigned for non-negative floats).  The two words sit in different 128-byte lines on purpose: the
// atomic that raises the maximum executes at the memory side and drops its line from L2 -- next to the scale, which every
// workgroup reads, that turned the 30 000-workgroup pack launch into a queue on one line (0.44 ms for 5.5 M elements).
// [1] = FLOOR of the running maximum (7/8 of the previous step's): a wave whose own maximum is below it sends no atomic.  Without
// it every launch started from 0 and nearly every wave of it (8192 in a fused stage) queued an atomic on the one address --
// 81 us instead of 33 us for a ResidualControl stage inside the step (the isolated benchmark, whose maximum is already there
// after the first iteration, never showed it).  A step whose values all stay below the floor keeps its scale (the true
// maximum is within 1/8 of the previous one) and lets the floor decay.
constexpr int SLOT_STRIDE = 64, SLOT_AMAX = 32, SLOT_FLOOR = 1;
constexpr int F16_TARGET_EXP = 2;          // next scale: |max| * scale in [2^(F16_TARGET_EXP-1), 2^F16_TARGET_EXP) (f16scale.TARGET_EXP)
// Running |max| of staged values, taken on the float BITS: non-negative floats order like unsigned integers and every NaN
// pattern orders above +inf, so ONE NaN element survives into the slot and raises the guard -- fmaxf returns its non-NaN
// operand and would drop it (round-4 advisory: with saturating conversions the recorded maximum is the only overflow signal).
__device__ __forceinline__ float amax_acc(float m, float v) {
    return __uint_as_float(max(__float_as_uint(m), __float_as_uint(v) & 0x7fffffffu));
}
__device__ __forceinline__ float amax_acc(float m, float a, float b) { return amax_acc(amax_acc(m, a), b); }

struct ScaleSlot {
    float *p;
    __device__ __forceinline__ float scale() const { return p ? p[0] : 1.f; }
    __device__ __forceinline__ void record(float wave_max_candidate) const {
        // one atomic per wave: butterfly over the 64 lanes on the float bits (amax_acc: NaN / Inf stay the largest), lane 0
        // publishes
        float m = wave_max_candidate;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = amax_acc(m, __shfl_xor(m, d, 64));
        // (many workgroups report into one word: only a value above the one already there needs the atomic -- NaN compares
        // false and goes through)
        if (p && (threadIdx.x & 63) == 0 && !(m <= fmaxf(__builtin_nontemporal_load(p + SLOT_AMAX), p[SLOT_FLOOR])))
            atomicMax(reinterpret_cast<unsigned *>(p + SLOT_AMAX), __float_as_uint(m));
    }
};


typedef unsigned u32x4_c16 __attribute__((ext_vector_type(4)));

// One pixel's 16 channels of a c16 block from the two lane halves of a 32x32 MFMA accumulator: lane (h = lane >> 5) holds
// channels 4h..4h+3 (a0, a1 = packed pairs) and 8+4h..8+4h+3 (b0, b1) of pixel (lane & 31).  v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, after which the lower lanes hold channels 0..7 and the upper
// lanes channels 8..15 of their pixel, each as 16 contiguous bytes: one 16-byte store per lane, 1 KB contiguous per wave.
__device__ __forceinline__ u32x4_c16 c16_gather_halves(unsigned a0, unsigned a1, unsigned b0, unsigned b1) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    return u32x4_c16{s0[0], s1[0], s0[1], s1[1]};
}

}  // namespace ebfi

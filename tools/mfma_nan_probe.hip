// Does MODE.FP16_OVFL change what the matrix cores do with a non-finite OPERAND?  (round 6: a NaN planted in the input of the
// image-writing forward kernel -- whose consumer waves run with FP16_OVFL set for their fp16 epilogue conversions -- never reached
// the output, while the same kernel without the epilogue extras passes it on.)  One wave, one v_mfma_f32_32x32x16_bf16 (and the
// f16 form), operand element [row 3][k 5] of A = NaN / +Inf, B = ones: row 3 of the result must be non-finite.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_nan_probe.hip -o tools/bin/mfma_nan_probe && tools/bin/mfma_nan_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool OVFL, bool HALF>
__global__ void probe(float special, float *out, float *cvt) {
    if (OVFL) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int lane = threadIdx.x;
    // A fragment of 32x32x16: lane holds row (lane & 31), k = 8 * (lane >> 5) .. + 7
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (HALF) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.5f; b[j] = (_Float16)1.f; }
        if ((lane & 31) == 3 && (lane >> 5) == 0) a[5] = (_Float16)special;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.5f; b[j] = (__bf16)1.f; }
        if ((lane & 31) == 3 && (lane >> 5) == 0) a[5] = (__bf16)special;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) out[lane * 16 + e] = acc[e];
    // and the conversions themselves under the same mode
    if (lane == 0) {
        cvt[0] = (float)(_Float16)special;
        cvt[1] = (float)(_Float16)1e6f;
        cvt[2] = (float)(__bf16)special;
        cvt[3] = special > 0.f ? special : special * 0.01f;
    }
}

template <bool OVFL, bool HALF>
static void run(const char *what, float special) {
    float *d, *c, h[64 * 16], hc[4];
    hipMalloc(&d, sizeof(h));
    hipMalloc(&c, sizeof(hc));
    hipLaunchKernelGGL((probe<OVFL, HALF>), dim3(1), dim3(64), 0, nullptr, special, d, c);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    int nonfinite = 0, nan = 0;
    for (int i = 0; i < 64 * 16; ++i) { nonfinite += !std::isfinite(h[i]); nan += std::isnan(h[i]); }
    printf("%-34s operand %5g: %3d non-finite results (%3d NaN) of 1024 [expected 32]; finite sample %g; cvt f16(special) %g  f16(1e6) %g  bf16(special) %g  leaky %g\n",
           what, special, nonfinite, nan, h[0], hc[0], hc[1], hc[2], hc[3]);
    hipFree(d);
    hipFree(c);
}

int main() {
    const float specials[3] = {NAN, INFINITY, -INFINITY};
    for (float s : specials) {
        run<false, false>("bf16 mfma, default mode", s);
        run<true, false>("bf16 mfma, MODE.FP16_OVFL = 1", s);
        run<false, true>("f16 mfma, default mode", s);
        run<true, true>("f16 mfma, MODE.FP16_OVFL = 1", s);
    }
    return 0;
}

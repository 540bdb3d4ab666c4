/*
 * ORACLE (test infrastructure only -- never linked into or called by the product path).
 *
 * CPU restatement of the reference's Filter-Adaptive Convolution (FAC / KernelConv2D).
 * The reference has NO CPU implementation (models/FAC/kernelconv2d/KernelConv2D.py:38-39,55-56
 * raise NotImplementedError) so this file restates the three CUDA kernels:
 *   forward      models/FAC/kernelconv2d/KernelConv2D_kernel.cu:25-53
 *   grad_input   models/FAC/kernelconv2d/KernelConv2D_kernel.cu:91-125
 *   grad_kernel  models/FAC/kernelconv2d/KernelConv2D_kernel.cu:128-150
 * Like the reference kernels, every tensor is addressed through explicit element strides
 * (the reference passes long4 shape/stride pairs), accumulation is a sequential ky-outer /
 * kx-inner loop in the storage precision.
 *
 * Pinning: the reference ships no golden vectors and no runnable test for this op (its
 * gradient_check, KernelConv2D.py:61-74, is written against a removed autograd API and is
 * CUDA only).  tests/test_oracle_fac.py re-runs that gradcheck protocol against this file and
 * cross-checks it with an independent unfold formulation.
 *
 * This header is included twice by fac_ref.c with REAL = float / double.
 */

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUFFIX)

static inline int64_t FN(at4)(const int64_t *st, int64_t a, int64_t b, int64_t c, int64_t d)
{
    return a * st[0] + b * st[1] + c * st[2] + d * st[3];
}

/* out[b,c,y,x] = sum_{ky,kx} in[b,c,y+ky,x+kx] * kern[b, c*K*K + ky*K + kx, y, x] */
void FN(fac_ref_forward)(const REAL *in, const int64_t *in_stride,
                         const REAL *kern, const int64_t *k_stride,
                         REAL *out, const int64_t *out_stride,
                         int64_t B, int64_t C, int64_t Ho, int64_t Wo, int K)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t y = 0; y < Ho; ++y)
                for (int64_t x = 0; x < Wo; ++x) {
                    REAL acc = 0;
                    for (int ky = 0; ky < K; ++ky)
                        for (int kx = 0; kx < K; ++kx) {
                            int64_t kc = (int64_t)K * K * c + (int64_t)K * ky + kx;
                            acc += in[FN(at4)(in_stride, b, c, y + ky, x + kx)] *
                                   kern[FN(at4)(k_stride, b, kc, y, x)];
                        }
                    out[FN(at4)(out_stride, b, c, y, x)] = acc;
                }
}

/*
 * grad_input[b,c,Y,X] = sum over taps (ky,kx) whose source pixel (Y-ky, X-kx) lies inside the
 * kernel plane of kern[b,kc,Y-ky,X-kx] * grad_out[b,c,Y-ky,X-kx]        (gather form, no atomics)
 * grad_kernel[b,c*K*K+ky*K+kx,y,x] = in[b,c,y+ky,x+kx] * grad_out[b,c,y,x]
 */
void FN(fac_ref_backward)(const REAL *in, const int64_t *in_stride,
                          const REAL *kern, const int64_t *k_stride,
                          const REAL *gout, const int64_t *go_stride,
                          REAL *gin, const int64_t *gi_stride,
                          REAL *gkern, const int64_t *gk_stride,
                          int64_t B, int64_t C, int64_t Ho, int64_t Wo, int K)
{
    const int64_t Hi = Ho + K - 1, Wi = Wo + K - 1;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t c = 0; c < C; ++c) {
            for (int64_t Y = 0; Y < Hi; ++Y)
                for (int64_t X = 0; X < Wi; ++X) {
                    REAL acc = 0;
                    for (int ky = 0; ky < K; ++ky)
                        for (int kx = 0; kx < K; ++kx) {
                            int64_t y = Y - ky, x = X - kx;
                            if (y < 0 || y > Ho - 1 || x < 0 || x > Wo - 1)
                                continue;
                            int64_t kc = (int64_t)K * K * c + (int64_t)K * ky + kx;
                            acc += kern[FN(at4)(k_stride, b, kc, y, x)] *
                                   gout[FN(at4)(go_stride, b, c, y, x)];
                        }
                    gin[FN(at4)(gi_stride, b, c, Y, X)] = acc;
                }
            for (int ky = 0; ky < K; ++ky)
                for (int kx = 0; kx < K; ++kx) {
                    int64_t kc = (int64_t)K * K * c + (int64_t)K * ky + kx;
                    for (int64_t y = 0; y < Ho; ++y)
                        for (int64_t x = 0; x < Wo; ++x)
                            gkern[FN(at4)(gk_stride, b, kc, y, x)] =
                                in[FN(at4)(in_stride, b, c, y + ky, x + kx)] *
                                gout[FN(at4)(go_stride, b, c, y, x)];
                }
        }
}

#undef FN
#undef CAT
#undef CAT2

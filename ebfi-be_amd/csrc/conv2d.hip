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
#include "common.hpp"

using namespace ebfi;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SIGMOID = 2 };

struct ConvGeom {
    int B, Cin, H, W, Cout, Ho, Wo, pad;
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

constexpr int TY = 4, TX = 64;

// ------------------------------------------------------------------------------------------------
// forward (and stride-1 data gradient when transposed != 0)
//   x    [B,Cin,H,W]   (for dgrad: grad_output [B,Cout_fwd,..])      dact_y: optional, same shape as x
//   w    forward: [Cout,Cin,KS,KS];  transposed: [Cin,Cout,KS,KS] of the FORWARD conv (its Cout = our Cin)
//   out  [B,Cout,Ho,Wo]
template <int KS, int S, int MT, int CK>
__global__ __launch_bounds__(256) void conv_fwd_f32(const float *__restrict__ x, const float *__restrict__ dact_y,
                                                    const float *__restrict__ w, const float *__restrict__ bias,
                                                    float *__restrict__ out, ConvGeom g, int transposed, int act,
                                                    float slope, int dact, float dslope) {
    constexpr int KK = KS * KS;
    constexpr int IH = S * (TY - 1) + KS, IW = S * (TX - 1) + KS;
    constexpr int PS = IH * IW;            // plane stride of the staged input tile
    constexpr int COS = 32 * MT;           // weight-slice row length (output channels of this block)
    __shared__ float sIn[CK * PS];
    __shared__ float sW[KK * CK * COS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + TX - 1) / TX, tiles_y = (g.Ho + TY - 1) / TY;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TY, x0 = tx * TX;
    const int co_base = blockIdx.y * COS;
    const int iy0 = S * y0 - g.pad, ix0 = S * x0 - g.pad;

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const float *xb = x + (int64_t)b * g.Cin * g.H * g.W;
    const float *yb = dact ? dact_y + (int64_t)b * g.Cin * g.H * g.W : nullptr;

    for (int c0 = 0; c0 < g.Cin; c0 += CK) {
        __syncthreads();
        // ---- input halo tile [CK][IH][IW], zero outside the image / beyond Cin
        for (int i = tid; i < CK * PS; i += 256) {
            const int ci = i / PS, rem = i - ci * PS;
            const int r = rem / IW, c = rem - r * IW;
            const int yy = iy0 + r, xx = ix0 + c;
            float v = 0.f;
            if (c0 + ci < g.Cin && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) {
                const int64_t o = ((int64_t)(c0 + ci) * g.H + yy) * g.W + xx;
                v = xb[o];
                if (dact) v *= act_grad(yb[o], dact, dslope);
            }
            sIn[i] = v;
        }
        // ---- weight slice sW[tap][ci][co]
        if (!transposed) {
            // W[co][ci][tap]: for fixed co the (ci,tap) run is contiguous
            for (int i = tid; i < COS * CK * KK; i += 256) {
                const int co = i / (CK * KK), rem = i - co * (CK * KK);
                const int ci = rem / KK, tap = rem - ci * KK;
                float v = 0.f;
                if (co_base + co < g.Cout && c0 + ci < g.Cin)
                    v = w[((int64_t)(co_base + co) * g.Cin + c0 + ci) * KK + tap];
                sW[(tap * CK + ci) * COS + co] = v;
            }
        } else {
            // data gradient: k = forward out-channel (our ci), m = forward in-channel (our co), flipped taps
            for (int i = tid; i < CK * COS * KK; i += 256) {
                const int ci = i / (COS * KK), rem = i - ci * (COS * KK);
                const int co = rem / KK, tap = rem - co * KK;
                float v = 0.f;
                if (co_base + co < g.Cout && c0 + ci < g.Cin)
                    v = w[((int64_t)(c0 + ci) * g.Cout + co_base + co) * KK + tap];
                sW[((KK - 1 - tap) * CK + ci) * COS + co] = v;
            }
        }
        __syncthreads();
        // ---- MFMA: wave owns output row `wave` of the tile (2 x-halves) x MT co tiles
        const float *bp = sIn + (lane >> 5) * PS + (S * wave) * IW + S * (lane & 31);
        const float *ap = sW + (lane >> 5) * COS + (lane & 31);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int cp = 0; cp < CK; cp += 2) {
                float a[MT], bv[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = ap[(tap * CK + cp) * COS + m * 32];
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
    const int yo = y0 + wave;
    if (yo < g.Ho) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int xo = x0 + n * 32 + (lane & 31);
                if (xo >= g.Wo) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (co < g.Cout) {
                        float v = acc[m][n][r];
                        if (bias) v += bias[co];
                        out[(((int64_t)b * g.Cout + co) * g.Ho + yo) * g.Wo + xo] = act_apply(v, act, slope);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: slab[split][co][ci*KK + tap] partial sums; slab[split][Cout*Cin*KK + co] bias partials
constexpr int WTY = 2, WTX = 32, WNP = WTY * WTX;   // pixel tile of the contraction
constexpr int GS = WNP + 1;                          // odd row stride of the grad_out image

template <int KS, int S, int CIB>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const float *__restrict__ x, const float *__restrict__ gout,
                                                      const float *__restrict__ yact, float *__restrict__ slab,
                                                      ConvGeom g, int dact, float dslope, int total_tiles,
                                                      int need_bias) {
    constexpr int KK = KS * KS;
    constexpr int IH = S * (WTY - 1) + KS, IW = S * (WTX - 1) + KS;
    constexpr int PS = (IH * IW) | 1;      // odd plane stride -> conflict-free across channels
    constexpr int NTW = (CIB * KK + 63) / 64;  // n-tiles (of 32) per wave: the CIB-channel block has <= 2*NTW tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sG = smem;                      // [64 co][GS]
    float *sIn = smem + 64 * GS;           // [CIB ci][PS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64, ci_base = blockIdx.z * CIB;
    const int ci_cnt = min(CIB, g.Cin - ci_base);
    const int ncols = ci_cnt * KK;         // valid (ci,tap) columns of this block
    const int mt = wave & 1, nh = wave >> 1;   // wave: co tile mt, n-tiles nh, nh+2, nh+4, ...
    const int tiles_x = (g.Wo + WTX - 1) / WTX, tiles_y = (g.Ho + WTY - 1) / WTY;

    f32x16 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // per-lane LDS offset of column n = (nh + 2*q)*32 + (lane&31): ci*PS + ky*IW + kx
    int boff[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = (nh + 2 * q) * 32 + (lane & 31);
        const int ci = n / KK, tap = n - ci * KK;
        const int ky = tap / KS, kx = tap - ky * KS;
        boff[q] = (n < ncols) ? ci * PS + ky * IW + kx : -1;
    }
    float bsum = 0.f;

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int y0 = ty * WTY, x0 = tx * WTX;
        const int iy0 = S * y0 - g.pad, ix0 = S * x0 - g.pad;
        __syncthreads();
        for (int i = tid; i < 64 * WNP; i += 256) {
            const int co = i / WNP, p = i - co * WNP;
            const int yy = y0 + p / WTX, xx = x0 + p % WTX;
            float v = 0.f;
            if (co_base + co < g.Cout && yy < g.Ho && xx < g.Wo) {
                const int64_t o = (((int64_t)b * g.Cout + co_base + co) * g.Ho + yy) * g.Wo + xx;
                v = gout[o];
                if (dact) v *= act_grad(yact[o], dact, dslope);
            }
            sG[co * GS + p] = v;
        }
        for (int i = tid; i < CIB * IH * IW; i += 256) {
            const int ci = i / (IH * IW), rem = i - ci * (IH * IW);
            const int r = rem / IW, c = rem - r * IW;
            const int yy = iy0 + r, xx = ix0 + c;
            float v = 0.f;
            if (ci < ci_cnt && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W)
                v = x[(((int64_t)b * g.Cin + ci_base + ci) * g.H + yy) * g.W + xx];
            sIn[ci * PS + rem] = v;
        }
        __syncthreads();
        if (need_bias && blockIdx.z == 0 && tid < 64) {
            float s = 0.f;
            for (int p = 0; p < WNP; ++p) s += sG[tid * GS + p];
            bsum += s;
        }
        const float *ap = sG + (mt * 32 + (lane & 31)) * GS + (lane >> 5);
#pragma unroll 4
        for (int p = 0; p < WNP; p += 2) {
            const int pp = p + (lane >> 5);
            const int poff = (S * (pp / WTX)) * IW + S * (pp % WTX);
            const float a = ap[p];
#pragma unroll
            for (int q = 0; q < NTW; ++q) {
                const float bv = boff[q] >= 0 ? sIn[boff[q] + poff] : 0.f;
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[q], 0, 0, 0);
            }
        }
    }
    // ---- write this workgroup's partial slab
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    float *my = slab + (int64_t)blockIdx.x * (wsz + g.Cout);
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = (nh + 2 * q) * 32 + (lane & 31);
        if (n >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < g.Cout) my[((int64_t)co * g.Cin + ci_base) * KK + n] = acc[q][r];
        }
    }
    if (need_bias && blockIdx.z == 0 && tid < 64 && co_base + tid < g.Cout) my[wsz + co_base + tid] = bsum;
}

// 64 consecutive elements per workgroup, 4 thread rows each summing every 4th slab (4 loads in
// flight), then a fixed-order combine through LDS: deterministic, and short dependent chains.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_f32(const float *__restrict__ slab, int nslabs,
                                                             int64_t n_weight, int64_t n_total,
                                                             float *__restrict__ gw, float *__restrict__ gb) {
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
        else if (gb) gb[j - n_weight] = s;
    }
}

int make_geom(ConvGeom &g, int B, int Cin, int H, int W, int Cout, int ks, int stride, int pad) {
    if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return fail(EBFI_ERR_ARG, "conv2d: non-positive dimension");
    if (ks != 1 && ks != 3 && ks != 7) return fail(EBFI_ERR_UNSUPPORTED, "conv2d: kernel size %d (1, 3 and 7 implemented)", ks);
    if (stride != 1 && stride != 2) return fail(EBFI_ERR_UNSUPPORTED, "conv2d: stride %d (1 and 2 implemented)", stride);
    if (pad < 0 || pad > ks) return fail(EBFI_ERR_ARG, "conv2d: padding %d out of range", pad);
    g = ConvGeom{B, Cin, H, W, Cout, (H + 2 * pad - ks) / stride + 1, (W + 2 * pad - ks) / stride + 1, pad};
    if (g.Ho <= 0 || g.Wo <= 0) return fail(EBFI_ERR_ARG, "conv2d: empty output");
    if ((int64_t)B * Cin * H * W > (1LL << 40) || (int64_t)B * Cout * g.Ho * g.Wo > (1LL << 40))
        return fail(EBFI_ERR_ARG, "conv2d: tensor too large");
    return EBFI_OK;
}

template <int KS, int S>
int launch_fwd(hipStream_t st, const float *x, const float *dact_y, const float *w, const float *bias, float *out,
               const ConvGeom &g, int transposed, int act, float slope, int dact, float dslope) {
    constexpr int CK = KS == 7 ? 2 : 8;      // input channels staged per chunk (LDS budget of the weight slice)
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, TY) * ceil_div(g.Wo, TX);
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d: too many tiles");
    const char *name = transposed ? "conv_dgrad_f32" : "conv_fwd_f32";
    if (g.Cout <= 32) {
        dim3 grid((unsigned)tiles, (unsigned)ceil_div(g.Cout, 32));
        ProfScope ps(name, st);
        hipLaunchKernelGGL((conv_fwd_f32<KS, S, 1, CK>), grid, dim3(256), 0, st, x, dact_y, w, bias, out, g, transposed,
                           act, slope, dact, dslope);
    } else {
        dim3 grid((unsigned)tiles, (unsigned)ceil_div(g.Cout, 64));
        ProfScope ps(name, st);
        hipLaunchKernelGGL((conv_fwd_f32<KS, S, 2, CK>), grid, dim3(256), 0, st, x, dact_y, w, bias, out, g, transposed,
                           act, slope, dact, dslope);
    }
    return check_launch(name);
}

constexpr int wgrad_cib(int ks) { return ks == 7 ? 8 : 64; }

int wgrad_splits(const ConvGeom &g, int ks) {
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, WTX);
    const int64_t blocks = ceil_div(g.Cout, 64) * ceil_div(g.Cin, wgrad_cib(ks));
    int64_t s = ceil_div(1024, blocks);          // aim at >= 1024 workgroups (4 per CU)
    if (s > tiles) s = tiles;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}

template <int KS, int S>
int launch_wgrad(hipStream_t st, const float *x, const float *gout, const float *yact, float *slab, const ConvGeom &g,
                 int dact, float dslope, int nsplit, int need_bias) {
    constexpr int CIB = wgrad_cib(KS);
    constexpr int IH = S * (WTY - 1) + KS, IW = S * (WTX - 1) + KS;
    constexpr int PS = (IH * IW) | 1;
    const size_t lds = (size_t)(64 * GS + CIB * PS) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_f32<KS, S, CIB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int64_t tiles = (int64_t)g.B * ceil_div(g.Ho, WTY) * ceil_div(g.Wo, WTX);
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(g.Cout, 64), (unsigned)ceil_div(g.Cin, CIB));
    ProfScope ps("conv_wgrad_f32", st);
    hipLaunchKernelGGL((conv_wgrad_f32<KS, S, CIB>), grid, dim3(256), lds, st, x, gout, yact, slab, g, dact, dslope,
                       (int)tiles, need_bias);
    return check_launch("conv_wgrad_f32");
}

}  // namespace

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
    if (ksize == 3 && stride == 1) return launch_fwd<3, 1>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f);
    if (ksize == 3 && stride == 2) return launch_fwd<3, 2>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f);
    if (ksize == 1 && stride == 1) return launch_fwd<1, 1>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f);
    if (ksize == 7 && stride == 1) return launch_fwd<7, 1>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f);
    if (ksize == 7 && stride == 2) return launch_fwd<7, 2>(st, x, nullptr, w, bs, o, g, 0, act, slope, 0, 0.f);
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
    if (ksize == 3) return launch_fwd<3, 1>(st, go, yo, w, nullptr, gi, g, 1, ACT_NONE, 0.f, act, slope);
    if (ksize == 7) return launch_fwd<7, 1>(st, go, yo, w, nullptr, gi, g, 1, ACT_NONE, 0.f, act, slope);
    return launch_fwd<1, 1>(st, go, yo, w, nullptr, gi, g, 1, ACT_NONE, 0.f, act, slope);
}

extern "C" size_t ebfi_conv2d_backward_weight_workspace(int B, int Cin, int H, int W, int Cout, int ksize, int stride,
                                                        int pad, int dtype) {
    (void)dtype;
    ConvGeom g;
    if (make_geom(g, B, Cin, H, W, Cout, ksize, stride, pad) != EBFI_OK) return 0;
    return (size_t)wgrad_splits(g, ksize) * ((size_t)Cout * Cin * ksize * ksize + Cout) * sizeof(float);
}

// grad_weight[Cout,Cin,k,k] (and grad_bias[Cout] when non-NULL), both fully overwritten, deterministic.
extern "C" int ebfi_conv2d_backward_weight(const void *input, const void *grad_output, const void *saved_output,
                                           void *grad_weight, void *grad_bias, int B, int Cin, int H, int W, int Cout,
                                           int ksize, int stride, int pad, int act, float slope, void *workspace,
                                           size_t workspace_bytes, int dtype, void *stream) {
    if (!input || !grad_output || !grad_weight) return fail(EBFI_ERR_ARG, "conv2d_backward_weight: null argument");
    if (dtype != EBFI_F32) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight: dtype %d not implemented", dtype);
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
    const int nsplit = wgrad_splits(g, ksize);
    const int need_bias = grad_bias != nullptr;
    int rc;
    if (ksize == 3 && stride == 1) rc = launch_wgrad<3, 1>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else if (ksize == 3 && stride == 2) rc = launch_wgrad<3, 2>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else if (ksize == 1 && stride == 1) rc = launch_wgrad<1, 1>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else if (ksize == 7 && stride == 1) rc = launch_wgrad<7, 1>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else if (ksize == 7 && stride == 2) rc = launch_wgrad<7, 2>(st, x, go, yo, slab, g, act, slope, nsplit, need_bias);
    else return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight: k=%d stride=%d not implemented", ksize, stride);
    if (rc) return rc;
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st, slab, nsplit,
                           n_weight, n_total, static_cast<float *>(grad_weight), static_cast<float *>(grad_bias));
    }
    return check_launch("conv_wgrad_reduce_f32");
}

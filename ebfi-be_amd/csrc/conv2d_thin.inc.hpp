// THIN convolution layers (included by conv2d.hip inside its anonymous namespace, round 5).
//
// Seven layers of the model have <= 6 channels on one side at full resolution (B = 8, 256 x 256): the model's first and last
// convolutions (3 -> 64 stride 2, 64 -> 3), ExposureDecision's (4 -> 64, 64 -> 1) and the detail branch's 7x7 stem / output
// convolution (6 -> 32 stride 2, 16 -> 3 on the reflection-padded 262 x 262 map).  On the matrix-core kernels they pad the
// thin side to a 32- or 64-row tile: the weight gradients alone took 70-160 us each for 6-130 MB of operands -- 1/6 of what
// HBM delivers.  Work per pixel is ~1700 multiply-adds, which the VECTOR pipe does in the time the thick tensor streams in
// once, so these kernels are direct convolutions in exact fp32 (no operand rounding at all: closer to the reference than the
// split-precision or fp16 forms they replace):
//
//   weight gradient  gw[co][ci][ky][kx] = sum_{b,y,x} gp[b][co][y][x] * X[b][ci][y S + ky - P][x S + kx - P],  gp = grad * act'(out)
//
// A workgroup owns 1-4 channels of the THICK tensor, one sample and a band of output rows; a thread owns ONE COLUMN of quads
// (4 consecutive output pixels) and walks THIN_RUN output rows down it with the KS input rows it needs in a sliding register
// window (one new row per output row), keeping the NT x KS x KS partial sums of its channels against all thin channels in
// registers.  The thick tensor is read once (halo rows of a run twice), the thin one -- a few MB -- from L2.  At the end: wave
// reduction by lane exchange, the four waves through LDS in a fixed order, one slab per (sample, band) summed by
// conv_wgrad_reduce_f32 in a fixed order: bit-reproducible like every other weight gradient here.
//   THIN_OUT: thick = input (Cin), thin = grad_out (Cout <= 4)      THIN_IN: thick = grad_out (Cout), thin = input (Cin <= 6)
//   KYSPLIT (7x7): ky is a grid dimension -- NT x 7 partial sums per thread instead of NT x 49.
// Loads go through buffer descriptors (out-of-range rows read 0 = zero padding); rows whose width is a multiple of 4 floats
// with a 16-byte aligned base are fetched as 16-byte quads aligned on the row, other widths (the 262-wide padded map) dword
// by dword.

constexpr int THIN_RUN = 8;            // output rows a thread walks (one column of quads, input rows kept in a sliding register window)

typedef float f32x4_thin __attribute__((ext_vector_type(4)));

struct ThinGeom {
    int B, Cin, H, W, Cout, Ho, Wo;
    int act;
    float slope;
    int bands, band_rows;               // band_rows = THIN_RUN * (256 / (Wo / 4)) output rows per workgroup
};

__device__ __forceinline__ float thin_dact(float g, float y, int act, float slope) {
    if (act == ACT_LEAKY) return y > 0.f ? g : g * slope;
    if (act == ACT_SIGMOID) return g * y * (1.f - y);
    return g;
}

// NG4 quads of one input row starting at column c0a (a multiple of 4, possibly negative): v[4 * q + e] = X[row][c0a + 4 q + e],
// 0 outside the image.  ALIGNED: every quad lies wholly inside or outside [0, W) (W % 4 == 0) and is one 16-byte load.
template <int NG4, bool ALIGNED>
__device__ __forceinline__ void thin_load_row(const __amdgpu_buffer_rsrc_t r, unsigned chan_off, int row, int c0a, int H, int W,
                                              float (&v)[NG4 * 4]) {
    const bool row_ok = row >= 0 && row < H;
    const unsigned rbase = chan_off + (unsigned)(row_ok ? row : 0) * (unsigned)W * 4u;
    if constexpr (ALIGNED) {
#pragma unroll
        for (int q = 0; q < NG4; ++q) {
            const int c = c0a + 4 * q;
            const unsigned off = sel_off(row_ok && c >= 0 && c < W, rbase + (unsigned)c * 4u);
            const f32x4_thin t = __builtin_bit_cast(f32x4_thin, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < NG4 * 4; ++e) {
            const int c = c0a + e;
            v[e] = buf_ld(r, sel_off(row_ok && c >= 0 && c < W, rbase + (unsigned)c * 4u));
        }
    }
}

// sum over the 64 lanes (every lane gets it), fixed exchange order
__device__ __forceinline__ float thin_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// KS x KS taps, stride S, padding P; NT thin channels (template: register arrays); NTH thick channels per workgroup.
template <int KS, int S, int P, int NT, int NTH, bool THIN_OUT, bool KYSPLIT, bool ALIGNED>
__global__ __launch_bounds__(256) void conv_wgrad_thin(const float *__restrict__ x, const float *__restrict__ gout,
                                                       const float *__restrict__ yact, float *__restrict__ gpre_out,
                                                       float *__restrict__ slab, ThinGeom g) {
    constexpr int KYN = KYSPLIT ? 1 : KS;                       // ky handled by one workgroup
    constexpr int NKEEP = KYN > S ? KYN - S : 0;                // window rows that survive a step to the next output row
    constexpr int PA = (P + 3) / 4 * 4;                         // the row window starts PA columns left of the quad's first input column
    constexpr int OFF = PA - P;                                 // window index of tap kx = 0 of output pixel j = 0
    constexpr int NG4 = (OFF + 3 * S + KS + 3) / 4;             // quads per row window
    constexpr int NXC = THIN_OUT ? NTH : NT;                    // input channels this workgroup reads
    constexpr int NGP = THIN_OUT ? NT : NTH;                    // grad_out channels this workgroup reads
    constexpr int NACC = NTH * NT * KYN * KS;
    __shared__ float red[4][NACC + NGP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int thick = THIN_OUT ? g.Cin : g.Cout;
    const int nblk = (thick + NTH - 1) / NTH;
    int t = blockIdx.x;
    int ky0 = 0;
    if constexpr (KYSPLIT) { ky0 = t % KS; t /= KS; }
    const int cblk = t % nblk; t /= nblk;
    const int band = t % g.bands;
    const int b = t / g.bands;
    const int c0 = cblk * NTH;                                  // first thick channel of this workgroup
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)b * g.Cin * HW, (unsigned)g.Cin * (unsigned)HW * 4u);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gout + (int64_t)b * g.Cout * HWo, (unsigned)g.Cout * (unsigned)HWo * 4u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(yact ? yact + (int64_t)b * g.Cout * HWo : gout, yact ? (unsigned)g.Cout * (unsigned)HWo * 4u : 0u);
    float *gp_out = gpre_out ? gpre_out + (int64_t)b * g.Cout * HWo : nullptr;
    // THIN_OUT: every workgroup needs gp of the NT output channels; the one with thick channel 0 (and ky 0) writes it out and
    // owns the bias sums.  THIN_IN: a workgroup's gp rows are its own thick channels: it writes them (ky 0) and sums its bias.
    const bool owner = (KYSPLIT ? ky0 == 0 : true) && (THIN_OUT ? c0 == 0 : true);

    float acc[NTH][NT][KYN][KS];
    float bsum[NGP];
#pragma unroll
    for (int a = 0; a < NTH; ++a)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int k = 0; k < KYN; ++k)
#pragma unroll
                for (int q = 0; q < KS; ++q) acc[a][n][k][q] = 0.f;
#pragma unroll
    for (int a = 0; a < NGP; ++a) bsum[a] = 0.f;

    // a thread = one column of quads (4 output pixels wide), THIN_RUN consecutive output rows
    const int tpr = g.Wo >> 2;                                  // quads per output row (a divisor of 256: launcher)
    const int run = tid / tpr, tx = tid - run * tpr;
    const int xo = tx * 4;
    const int y_begin = band * g.band_rows + run * THIN_RUN;
    const int y_end = min(y_begin + THIN_RUN, g.Ho);
    const int c0a = xo * S - PA;
    unsigned xch[NXC];
#pragma unroll
    for (int ci = 0; ci < NXC; ++ci) {
        const int cin = THIN_OUT ? c0 + ci : ci;
        xch[ci] = cin < g.Cin ? (unsigned)cin * (unsigned)HW * 4u : SENT;
    }
    float xw[NXC][KYN][NG4 * 4];
    // the window as it would be one output row earlier: its rows S .. KYN-1 are rows 0 .. KYN-S-1 of the first output row
    if constexpr (NKEEP > 0) {
        if (y_begin < y_end) {
#pragma unroll
            for (int ci = 0; ci < NXC; ++ci)
#pragma unroll
                for (int k = S; k < KYN; ++k) thin_load_row<NG4, ALIGNED>(rx, xch[ci], (y_begin - 1) * S + k - P, c0a, g.H, g.W, xw[ci][k]);
        }
    }
    for (int yo = y_begin; yo < y_end; ++yo) {
        // ---- input rows: shift the window by S rows, fetch the new ones
#pragma unroll
        for (int ci = 0; ci < NXC; ++ci) {
#pragma unroll
            for (int k = 0; k < NKEEP; ++k)
#pragma unroll
                for (int e = 0; e < NG4 * 4; ++e) xw[ci][k][e] = xw[ci][k + S][e];
#pragma unroll
            for (int k = NKEEP; k < KYN; ++k)
                thin_load_row<NG4, ALIGNED>(rx, xch[ci], yo * S + (KYSPLIT ? ky0 : k) - P, c0a, g.H, g.W, xw[ci][k]);
        }
        // ---- gp of this quad: NT channels (THIN_OUT) or the NTH thick channels (THIN_IN)
        float gp[NGP][4];
#pragma unroll
        for (int n = 0; n < NGP; ++n) {
            const int co = THIN_OUT ? n : c0 + n;
            const unsigned off = sel_off(co < g.Cout, ((unsigned)co * (unsigned)HWo + (unsigned)(yo * g.Wo + xo)) * 4u);
            const f32x4_thin gv = __builtin_bit_cast(f32x4_thin, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
            f32x4_thin yv = {0.f, 0.f, 0.f, 0.f};
            if (g.act != ACT_NONE) yv = __builtin_bit_cast(f32x4_thin, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
            gp[n][0] = thin_dact(gv.x, yv.x, g.act, g.slope);
            gp[n][1] = thin_dact(gv.y, yv.y, g.act, g.slope);
            gp[n][2] = thin_dact(gv.z, yv.z, g.act, g.slope);
            gp[n][3] = thin_dact(gv.w, yv.w, g.act, g.slope);
            if (owner) {
                bsum[n] += (gp[n][0] + gp[n][1]) + (gp[n][2] + gp[n][3]);
                if (gp_out != nullptr && co < g.Cout) {
                    const f32x4_thin o = {gp[n][0], gp[n][1], gp[n][2], gp[n][3]};
                    *reinterpret_cast<f32x4_thin *>(gp_out + (int64_t)co * HWo + (int64_t)yo * g.Wo + xo) = o;
                }
            }
        }
        // ---- the products
#pragma unroll
        for (int k = 0; k < KYN; ++k)
#pragma unroll
            for (int ci = 0; ci < NXC; ++ci)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    if constexpr (THIN_OUT) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            float s = acc[ci][n][k][kx];
#pragma unroll
                            for (int j = 0; j < 4; ++j) s = fmaf(gp[n][j], xw[ci][k][OFF + kx + j * S], s);
                            acc[ci][n][k][kx] = s;
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < NTH; ++a) {
                            float s = acc[a][ci][k][kx];
#pragma unroll
                            for (int j = 0; j < 4; ++j) s = fmaf(gp[a][j], xw[ci][k][OFF + kx + j * S], s);
                            acc[a][ci][k][kx] = s;
                        }
                    }
                }
    }
    // ---- reduction: lanes, then the four waves in a fixed order, then this workgroup's part of slab (b, band)
#pragma unroll
    for (int a = 0; a < NTH; ++a)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int k = 0; k < KYN; ++k)
#pragma unroll
                for (int q = 0; q < KS; ++q) {
                    const float s = thin_wave_sum(acc[a][n][k][q]);
                    if (lane == 0) red[wave][((a * NT + n) * KYN + k) * KS + q] = s;
                }
#pragma unroll
    for (int a = 0; a < NGP; ++a) {
        const float s = thin_wave_sum(bsum[a]);
        if (lane == 0) red[wave][NACC + a] = s;
    }
    __syncthreads();
    const int64_t n_weight = (int64_t)g.Cout * g.Cin * KS * KS, n_total = n_weight + g.Cout;
    float *my = slab + (int64_t)(b * g.bands + band) * n_total;
    for (int i = tid; i < NACC; i += 256) {
        const float s = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        int r = i;
        const int kx = r % KS; r /= KS;
        const int k = r % KYN; r /= KYN;
        const int n = r % NT; r /= NT;
        const int a = r;
        const int ky = KYSPLIT ? ky0 : k;
        const int co = THIN_OUT ? n : c0 + a, ci = THIN_OUT ? c0 + a : n;
        if (co < g.Cout && ci < g.Cin) my[(((int64_t)co * g.Cin + ci) * KS + ky) * KS + kx] = s;
    }
    if (owner) {
        for (int i = tid; i < NGP; i += 256) {
            const int co = THIN_OUT ? i : c0 + i;
            if (co < g.Cout) my[n_weight + co] = (red[0][NACC + i] + red[1][NACC + i]) + (red[2][NACC + i] + red[3][NACC + i]);
        }
    }
}

// Which shapes the thin weight gradient serves (everything else keeps the matrix-core kernels)
struct ThinPlan {
    int kind;      // 0 = none; 1 = thin out 3x3 s1 p1; 2 = thin in 3x3 s1 p1; 3 = thin out 7x7 s1 p0; 4 = thin in 7x7 s2 p3; 5 = thin in 3x3 s2 p1
    int nt;        // thin channels (register array size of the instance)
    bool aligned;
};

inline ThinPlan thin_wgrad_plan(const ConvGeom &g, int ks, int stride, const void *x, const void *go, const void *y, const void *gp) {
    ThinPlan p{0, 0, false};
    if (g.groups != 1 || g.Wo % 4 != 0 || g.B < 1) return p;
    const int tpr = g.Wo / 4;
    if (tpr > 256 || 256 % tpr != 0) return p;          // a thread = one column of quads: 256 threads are whole rows of them
    if (!aligned16(go) || (y && !aligned16(y)) || (gp && !aligned16(gp))) return p;
    if (dev_getenv("EBFI_NO_THIN") != nullptr) return p;
    p.aligned = g.W % 4 == 0 && aligned16(x);
    const int64_t px = (int64_t)g.B * g.Ho * g.Wo;
    if (px < 64 * 1024) return p;                       // small maps: the matrix-core kernels' fixed costs are not the problem
    if (ks == 3 && stride == 1 && g.pad == 1) {
        if (g.Cout <= 4 && g.Cin >= 16) { p.kind = 1; p.nt = g.Cout <= 1 ? 1 : (g.Cout <= 3 ? 3 : 4); }
        else if (g.Cin <= 4 && g.Cout >= 16) { p.kind = 2; p.nt = g.Cin <= 1 ? 1 : (g.Cin <= 3 ? 3 : 4); }
    } else if (ks == 7 && stride == 1 && g.pad == 0 && g.Cout <= 3 && g.Cin >= 8) {
        p.kind = 3; p.nt = 3;
    } else if (ks == 7 && stride == 2 && g.pad == 3 && g.Cin <= 6 && g.Cout >= 16) {
        p.kind = 4; p.nt = 6;
    } else if (ks == 3 && stride == 2 && g.pad == 1 && g.Cin <= 4 && g.Cout >= 16) {
        p.kind = 5; p.nt = g.Cin <= 3 ? 3 : 4;
    }
    if (p.kind == 0) return p;
    if (!p.aligned && p.kind != 3) p.kind = 0;          // (only the 7x7 output convolution's padded map has ragged rows)
    return p;
}

inline int thin_band_rows(const ConvGeom &g) { return THIN_RUN * (256 / (g.Wo / 4)); }
inline int thin_wgrad_slabs(const ConvGeom &g) { return g.B * (int)ceil_div(g.Ho, thin_band_rows(g)); }

template <int KS, int S, int P, int NT, int NTH, bool THIN_OUT, bool KYSPLIT, bool ALIGNED>
int launch_wgrad_thin_i(hipStream_t st, const float *x, const float *go, const float *y, float *gp, float *slab, const ThinGeom &tg,
                        const char *label, double flops, double bytes) {
    const int thick = THIN_OUT ? tg.Cin : tg.Cout;
    const int64_t wgs = (int64_t)tg.B * tg.bands * ceil_div(thick, NTH) * (KYSPLIT ? KS : 1);
    if (wgs > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_backward_weight (thin): too many workgroups");
    ProfScope ps(label, st, flops, bytes);
    hipLaunchKernelGGL((conv_wgrad_thin<KS, S, P, NT, NTH, THIN_OUT, KYSPLIT, ALIGNED>), dim3((unsigned)wgs), dim3(256), 0, st, x, go, y,
                       gp, slab, tg);
    return check_launch(label);
}

// grad_weight / grad_bias (+ grad_preact_out) of a thin layer: the kernel above, then the shared slab reduction
int launch_wgrad_thin(hipStream_t st, const ThinPlan &p, const float *x, const float *go, const float *y, float *gp, float *slab,
                      const ConvGeom &g, int ks, int act, float slope, float *gw, float *gb) {
    ThinGeom tg{g.B, g.Cin, g.H, g.W, g.Cout, g.Ho, g.Wo, act, slope, (int)ceil_div(g.Ho, thin_band_rows(g)), thin_band_rows(g)};
    const double flops = 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * ks * ks;
    const double bytes = conv_bytes_wgrad(g, ks * ks, act != ACT_NONE, gp != nullptr);
    int rc = EBFI_ERR_UNSUPPORTED;
    switch (p.kind) {
    case 1:
        if (p.nt == 1) rc = launch_wgrad_thin_i<3, 1, 1, 1, 4, true, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out", flops, bytes);
        else if (p.nt == 3) rc = launch_wgrad_thin_i<3, 1, 1, 3, 2, true, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out", flops, bytes);
        else rc = launch_wgrad_thin_i<3, 1, 1, 4, 2, true, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out", flops, bytes);
        break;
    case 2:
        if (p.nt == 1) rc = launch_wgrad_thin_i<3, 1, 1, 1, 4, false, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in", flops, bytes);
        else if (p.nt == 3) rc = launch_wgrad_thin_i<3, 1, 1, 3, 2, false, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in", flops, bytes);
        else rc = launch_wgrad_thin_i<3, 1, 1, 4, 2, false, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in", flops, bytes);
        break;
    case 3:
        // (Wo = W - 6 and Wo % 4 == 0 leave W % 4 == 2: the rows of this layer's input are never quad-aligned -- dword loads)
        rc = launch_wgrad_thin_i<7, 1, 0, 3, 1, true, true, false>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out7", flops, bytes);
        break;
    case 4:
        rc = launch_wgrad_thin_i<7, 2, 3, 6, 2, false, true, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in7s2", flops, bytes);
        break;
    case 5:
        if (p.nt == 3) rc = launch_wgrad_thin_i<3, 2, 1, 3, 2, false, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/ins2", flops, bytes);
        else rc = launch_wgrad_thin_i<3, 2, 1, 4, 2, false, false, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/ins2", flops, bytes);
        break;
    default:
        return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight (thin): no kernel for this shape");
    }
    if (rc) return rc;
    const int64_t n_weight = (int64_t)g.Cout * g.Cin * ks * ks, n_total = n_weight + g.Cout;
    {
        ProfScope ps("conv_wgrad_reduce_f32", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_f32, dim3((unsigned)ceil_div(n_total, 64)), dim3(256), 0, st, slab, thin_wgrad_slabs(g), n_weight,
                           n_total, gw, gb, 0, ks * ks);
    }
    return check_launch("conv_wgrad_reduce_f32");
}

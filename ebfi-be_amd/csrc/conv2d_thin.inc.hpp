// THIN 3x3 layers: weight gradients as direct fp32 kernels (included by conv2d.hip inside its anonymous namespace, round 5).
// (Round 6: in the split-precision mode the model's shapes take conv2d_shift.inc.hpp -- matrix cores, taps on the thin side's row
//  axis; these kernels remain the exact-fp32 form and serve the shapes that file does not build.)
//
// Four 3x3 layers of the model have <= 4 channels on one side at full resolution (B = 8, 256 x 256): the model's first and last
// convolutions (3 -> 64 stride 2, 64 -> 3) and ExposureDecision's (4 -> 64, 64 -> 1).  On the matrix-core kernels the thin side
// is padded to a 32- or 64-row tile: their weight gradients took 72-144 us each for 6-130 MB of operands.  Work per pixel is
// ~1700 multiply-adds, which the vector pipe does while the thick tensor streams in once, so these are direct convolutions in
// exact fp32 (no operand rounding: closer to the reference than the split-precision / fp16 forms they replace):
//
//   gw[co][ci][ky][kx] = sum_{b,y,x} gp[b][co][y][x] * X[b][ci][y S + ky - 1][x S + kx - 1],   gp = grad * act'(out)
//
// A workgroup owns 1-4 channels of the THICK tensor, one sample and a band of output rows; a thread owns ONE COLUMN of quads
// (4 consecutive output pixels) and walks THIN_RUN output rows down it with the 3 input rows it needs in a sliding register
// window (the next step's rows are requested before the current step's products), keeping the NT x 3 x 3 partial sums of its
// channels against all thin channels in registers.  The thick tensor is read once (halo rows of a run twice), the thin one
// -- a few MB -- from L2.  At the end: wave sums on the DPP path, the four waves through LDS in a fixed order, one slab per
// (sample, band) summed by conv_wgrad_reduce_f32 in a fixed order: bit-reproducible like every other weight gradient here.
//   THIN_OUT: thick = input (Cin), thin = grad_out (Cout <= 4)      THIN_IN: thick = grad_out (Cout), thin = input (Cin <= 4)
// Measured inside the step (B = 8, 256 x 256, entry point incl. the slab reduction): 64 -> 3  111 -> 79 us, 64 -> 1  103 -> 46,
// 4 -> 64  144 -> 122, 3 -> 64 stride 2  72 -> 52.  These kernels issue at ~1/6 of the vector pipe's peak (two waves per SIMD,
// chains of dependent packed FMAs), so there is room left; the first version -- a runtime `switch` on the activation inside
// the loop, wave sums through ds_bpermute -- was SLOWER than the matrix-core kernels (125 us for 64 -> 3).  The same scheme
// for the two 7x7 layers (16 -> 3 on the 262-wide padded map with ky as a grid dimension, 6 -> 32 stride 2) measured 192 and
// 435 us against 159 and 88: dropped, they stay on the matrix cores.
// Loads go through buffer descriptors (out-of-range rows read 0 = zero padding); rows are whole 16-byte quads (W % 4 == 0,
// aligned base: launcher), the quad-filling columns come as 16-byte loads and the two ragged end columns as dwords.

constexpr int THIN_RUN = 16;           // output rows a thread walks (one column of quads, input rows kept in a sliding register window)

typedef float f32x4_thin __attribute__((ext_vector_type(4)));

struct ThinGeom {
    int B, Cin, H, W, Cout, Ho, Wo;
    int act;
    float slope;
    int bands, band_rows;               // band_rows = THIN_RUN * (256 / (Wo / 4)) output rows per workgroup
};

// grad * act'(y) without a branch per element (a runtime `switch` on the activation compiled to a dozen scalar branches per quad,
// each with its own s_waitcnt): act' = (c0 + y (c1 + c2 y)) * (y > 0 ? 1 : sl) with
//   none: c = (1, 0, 0), sl = 1     LeakyReLU: c = (1, 0, 0), sl = slope     sigmoid: c = (0, 1, -1), sl = 1 (y in (0, 1))
struct ThinAct {
    float c0, c1, c2, sl;
};
__device__ __forceinline__ ThinAct thin_act(int act, float slope) {
    ThinAct a;
    const bool sg = act == ACT_SIGMOID;
    a.c0 = sg ? 0.f : 1.f;
    a.c1 = sg ? 1.f : 0.f;
    a.c2 = sg ? -1.f : 0.f;
    a.sl = act == ACT_LEAKY ? slope : 1.f;
    return a;
}
__device__ __forceinline__ float thin_dact(float g, float y, const ThinAct &a) {
    const float m = fmaf(y, fmaf(y, a.c2, a.c1), a.c0);
    return g * m * (y > 0.f ? 1.f : a.sl);
}

// The NV = 3 S + KS input columns a quad of output pixels needs from one input row: v[i] = X[row][c0a + OFF + i], 0 outside the
// image.  c0a is a multiple of 4 (possibly negative).  ALIGNED (W % 4 == 0, 16-byte aligned rows): the columns that fill whole
// quads of the row come as 16-byte loads (a quad lies wholly inside or outside [0, W)), the ragged ends dword by dword.
template <int I, int NV, int OFF, bool ALIGNED>
__device__ __forceinline__ void thin_load_cols(const __amdgpu_buffer_rsrc_t r, unsigned rbase, bool row_ok, int c0a, int W, float (&v)[NV]) {
    if constexpr (I < NV) {                     // (compile-time recursion: v[] must keep constant indices to stay in registers)
        const int c = c0a + OFF + I;
        const unsigned off = sel_off(row_ok && c >= 0 && c < W, rbase + (unsigned)c * 4u);
        if constexpr (ALIGNED && (OFF + I) % 4 == 0 && I + 4 <= NV) {
            const f32x4_thin t = __builtin_bit_cast(f32x4_thin, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
            v[I] = t.x; v[I + 1] = t.y; v[I + 2] = t.z; v[I + 3] = t.w;
            thin_load_cols<I + 4, NV, OFF, ALIGNED>(r, rbase, row_ok, c0a, W, v);
        } else {
            v[I] = buf_ld(r, off);
            thin_load_cols<I + 1, NV, OFF, ALIGNED>(r, rbase, row_ok, c0a, W, v);
        }
    }
}
template <int NV, int OFF, bool ALIGNED>
__device__ __forceinline__ void thin_load_row(const __amdgpu_buffer_rsrc_t r, unsigned chan_off, int row, int c0a, int H, int W,
                                              float (&v)[NV]) {
    const bool row_ok = row >= 0 && row < H;
    const unsigned rbase = chan_off + (unsigned)(row_ok ? row : 0) * (unsigned)W * 4u;
    thin_load_cols<0, NV, OFF, ALIGNED>(r, rbase, row_ok, c0a, W, v);
}

// Sum over the 64 lanes as a wave-uniform value, in a fixed order, on the VALU's data-parallel-primitive path: an inclusive scan
// along each row of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast15 / row_bcast31 carry the row totals up; lane 63 holds the sum.
// (The first version went through __shfl_xor = ds_bpermute: six LDS round trips with an s_waitcnt each per value, and with
// 50-80 partial sums per thread the reduction took as long as the eight rows of products before it.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float thin_dpp_add(float v) {
    const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(sh);
}
__device__ __forceinline__ float thin_wave_sum(float v) {
    v = thin_dpp_add<0x111, 0xf>(v);           // row_shr:1
    v = thin_dpp_add<0x112, 0xf>(v);           // row_shr:2
    v = thin_dpp_add<0x114, 0xf>(v);           // row_shr:4
    v = thin_dpp_add<0x118, 0xf>(v);           // row_shr:8  -> lane 15 of every row: the row's sum
    v = thin_dpp_add<0x142, 0xa>(v);           // row_bcast:15 into rows 1 and 3
    v = thin_dpp_add<0x143, 0xc>(v);           // row_bcast:31 into rows 2 and 3 -> lane 63: the wave's sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// KS x KS taps, stride S, padding P; NT thin channels (template: register arrays); NTH thick channels per workgroup.
template <int S, int NT, int NTH, bool THIN_OUT>
__global__ __launch_bounds__(256) void conv_wgrad_thin(const float *__restrict__ x, const float *__restrict__ gout,
                                                       const float *__restrict__ yact, float *__restrict__ gpre_out,
                                                       float *__restrict__ slab, ThinGeom g) {
    constexpr int KS = 3, P = 1;                                // 3x3 taps, padding 1 (the 7x7 layers stay on the matrix cores: see the header)
    constexpr int KYN = KS;
    constexpr bool ALIGNED = true;                              // rows are whole 16-byte quads (the launcher checks)
    constexpr int NKEEP = KYN > S ? KYN - S : 0;                // window rows that survive a step to the next output row
    constexpr int PA = (P + 3) / 4 * 4;                         // the row window starts PA columns left of the quad's first input column
    constexpr int OFF = PA - P;                                 // window index of tap kx = 0 of output pixel j = 0
    constexpr int NV = 3 * S + KS;                              // input columns a quad of output pixels touches in one row
    constexpr int NXC = THIN_OUT ? NTH : NT;                    // input channels this workgroup reads
    constexpr int NGP = THIN_OUT ? NT : NTH;                    // grad_out channels this workgroup reads
    constexpr int NACC = NTH * NT * KYN * KS;
    __shared__ float red[4][NACC + NGP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int thick = THIN_OUT ? g.Cin : g.Cout;
    const int nblk = (thick + NTH - 1) / NTH;
    int t = blockIdx.x;
    const int cblk = t % nblk; t /= nblk;
    const int band = t % g.bands;
    const int b = t / g.bands;
    const int c0 = cblk * NTH;                                  // first thick channel of this workgroup
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)b * g.Cin * HW, (unsigned)g.Cin * (unsigned)HW * 4u);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gout + (int64_t)b * g.Cout * HWo, (unsigned)g.Cout * (unsigned)HWo * 4u);
    const bool has_y = yact != nullptr && g.act != ACT_NONE;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(has_y ? yact + (int64_t)b * g.Cout * HWo : gout, has_y ? (unsigned)g.Cout * (unsigned)HWo * 4u : 0u);
    float *gp_out = gpre_out ? gpre_out + (int64_t)b * g.Cout * HWo : nullptr;
    // THIN_OUT: every workgroup needs gp of the NT output channels; the one with thick channel 0 (and ky 0) writes it out and
    // owns the bias sums.  THIN_IN: a workgroup's gp rows are its own thick channels: it writes them (ky 0) and sums its bias.
    const bool owner = (THIN_OUT ? c0 == 0 : true);

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
    float xw[NXC][KYN][NV];
    const ThinAct da = thin_act(g.act, g.slope);
    // first output row: its whole window, grad_out and (for act') the saved output
    auto load_g = [&](int yo, f32x4_thin (&gq)[NGP], f32x4_thin (&yq)[NGP]) {
#pragma unroll
        for (int n = 0; n < NGP; ++n) {
            const int co = THIN_OUT ? n : c0 + n;
            const unsigned off = sel_off(co < g.Cout && yo < g.Ho, ((unsigned)co * (unsigned)HWo + (unsigned)(yo * g.Wo + xo)) * 4u);
            gq[n] = __builtin_bit_cast(f32x4_thin, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
            yq[n] = __builtin_bit_cast(f32x4_thin, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));   // (no activation: empty descriptor, reads 0)
        }
    };
    f32x4_thin gv[NGP], yv[NGP];
#pragma unroll
    for (int ci = 0; ci < NXC; ++ci)
#pragma unroll
        for (int k = 0; k < KYN; ++k)
            thin_load_row<NV, OFF, ALIGNED>(rx, xch[ci], y_begin * S + k - P, c0a, g.H, g.W, xw[ci][k]);
    load_g(y_begin, gv, yv);
    constexpr int SN = KYN - NKEEP;                             // input rows that are new in every step
    for (int yo = y_begin; yo < y_end; ++yo) {
        // ---- the NEXT step's loads go out before this step's products: at two waves per SIMD nothing else hides the memory
        // latency (a step is ~200 instructions)
        float xn[NXC][SN][NV];
        f32x4_thin gn[NGP], yn[NGP];
#pragma unroll
        for (int ci = 0; ci < NXC; ++ci)
#pragma unroll
            for (int k = 0; k < SN; ++k)
                thin_load_row<NV, OFF, ALIGNED>(rx, xch[ci], (yo + 1) * S + (NKEEP + k) - P, c0a, g.H, g.W, xn[ci][k]);
        load_g(yo + 1, gn, yn);
        float gp[NGP][4];
#pragma unroll
        for (int n = 0; n < NGP; ++n) {
            gp[n][0] = thin_dact(gv[n].x, yv[n].x, da);
            gp[n][1] = thin_dact(gv[n].y, yv[n].y, da);
            gp[n][2] = thin_dact(gv[n].z, yv[n].z, da);
            gp[n][3] = thin_dact(gv[n].w, yv[n].w, da);
        }
        if (owner) {
#pragma unroll
            for (int n = 0; n < NGP; ++n) {
                const int co = THIN_OUT ? n : c0 + n;
                bsum[n] += (gp[n][0] + gp[n][1]) + (gp[n][2] + gp[n][3]);
                if (gp_out != nullptr && co < g.Cout) {
                    const f32x4_thin o = {gp[n][0], gp[n][1], gp[n][2], gp[n][3]};
                    *reinterpret_cast<f32x4_thin *>(gp_out + (int64_t)co * HWo + (int64_t)yo * g.Wo + xo) = o;
                }
            }
        }
        // ---- the products (pixel j outermost: consecutive multiply-adds go to different partial sums -- written sum by sum, the
        // compiler emitted chains of 3-4 dependent packed FMAs)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KYN; ++k)
#pragma unroll
                for (int ci = 0; ci < NXC; ++ci)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const float xv = xw[ci][k][kx + j * S];
                        if constexpr (THIN_OUT) {
#pragma unroll
                            for (int n = 0; n < NT; ++n) acc[ci][n][k][kx] = fmaf(gp[n][j], xv, acc[ci][n][k][kx]);
                        } else {
#pragma unroll
                            for (int a = 0; a < NTH; ++a) acc[a][ci][k][kx] = fmaf(gp[a][j], xv, acc[a][ci][k][kx]);
                        }
                    }
        // ---- the window moves down by S rows (register copies: ~1/8 of the step's instructions; a rotating window unrolled over
        // its period kept every slot live across the steps and cost more registers than it saved instructions)
#pragma unroll
        for (int ci = 0; ci < NXC; ++ci) {
#pragma unroll
            for (int k = 0; k < NKEEP; ++k)
#pragma unroll
                for (int e = 0; e < NV; ++e) xw[ci][k][e] = xw[ci][k + S][e];
#pragma unroll
            for (int k = 0; k < SN; ++k)
#pragma unroll
                for (int e = 0; e < NV; ++e) xw[ci][NKEEP + k][e] = xn[ci][k][e];
        }
#pragma unroll
        for (int n = 0; n < NGP; ++n) { gv[n] = gn[n]; yv[n] = yn[n]; }
    }
    // ---- reduction: lanes (wave-uniform sums, value i parked in lane i % 64), then the four waves in a fixed order, then this
    // workgroup's part of slab (b, band)
    constexpr int NRED = NACC + NGP;
    float park[(NRED + 63) / 64];
#pragma unroll
    for (int i = 0; i < (NRED + 63) / 64; ++i) park[i] = 0.f;
#pragma unroll
    for (int a = 0; a < NTH; ++a)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int k = 0; k < KYN; ++k)
#pragma unroll
                for (int q = 0; q < KS; ++q) {
                    const int i = ((a * NT + n) * KYN + k) * KS + q;
                    const float s = thin_wave_sum(acc[a][n][k][q]);
                    park[i / 64] = lane == (i & 63) ? s : park[i / 64];
                }
#pragma unroll
    for (int a = 0; a < NGP; ++a) {
        const int i = NACC + a;
        const float s = thin_wave_sum(bsum[a]);
        park[i / 64] = lane == (i & 63) ? s : park[i / 64];
    }
#pragma unroll
    for (int i = 0; i < (NRED + 63) / 64; ++i)
        if (i * 64 + lane < NRED) red[wave][i * 64 + lane] = park[i];
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
        const int ky = k;
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
    int kind;      // 0 = none; 1 = thin out 3x3 s1 p1; 2 = thin in 3x3 s1 p1; 5 = thin in 3x3 s2 p1
    int nt;        // thin channels (register array size of the instance)
    bool aligned;
};

// The geometry half of the plan -- channels, kernel, stride, padding, pixel count: everything known without the operand
// pointers.  Shared by thin_wgrad_plan and by ebfi_conv2d_backward_weight_workspace, which must size the slabs for exactly the
// layers the plan can select (a wide 512x512 layer at Wo = 32 used to get B * bands slabs it never touches: round-5 advisory).
inline ThinPlan thin_wgrad_geometry(const ConvGeom &g, int ks, int stride) {
    ThinPlan p{0, 0, false};
    if (g.groups != 1 || g.Wo % 4 != 0 || g.B < 1) return p;
    const int tpr = g.Wo / 4;
    if (tpr > 256 || 256 % tpr != 0) return p;          // a thread = one column of quads: 256 threads are whole rows of them
    const int64_t px = (int64_t)g.B * g.Ho * g.Wo;
    if (px < 64 * 1024) return p;                       // small maps: the matrix-core kernels' fixed costs are not the problem
    if (g.W % 4 != 0) return p;
    if (ks == 3 && stride == 1 && g.pad == 1) {
        if (g.Cout <= 4 && g.Cin >= 16) { p.kind = 1; p.nt = g.Cout <= 1 ? 1 : (g.Cout <= 3 ? 3 : 4); }
        else if (g.Cin <= 4 && g.Cout >= 16) { p.kind = 2; p.nt = g.Cin <= 1 ? 1 : (g.Cin <= 3 ? 3 : 4); }
    } else if (ks == 3 && stride == 2 && g.pad == 1 && g.Cin <= 4 && g.Cout >= 16) {
        p.kind = 5; p.nt = g.Cin <= 3 ? 3 : 4;
    }
    return p;
}

inline ThinPlan thin_wgrad_plan(const ConvGeom &g, int ks, int stride, const void *x, const void *go, const void *y, const void *gp) {
    ThinPlan p = thin_wgrad_geometry(g, ks, stride);
    if (p.kind == 0) return p;
    p.aligned = aligned16(x);
    if (!aligned16(go) || (y && !aligned16(y)) || (gp && !aligned16(gp)) || !p.aligned || dev_getenv("EBFI_NO_THIN") != nullptr)
        p.kind = 0;
    return p;
}

inline int thin_band_rows(const ConvGeom &g) { return THIN_RUN * (256 / (g.Wo / 4)); }
inline int thin_wgrad_slabs(const ConvGeom &g) { return g.B * (int)ceil_div(g.Ho, thin_band_rows(g)); }

template <int S, int NT, int NTH, bool THIN_OUT>
int launch_wgrad_thin_i(hipStream_t st, const float *x, const float *go, const float *y, float *gp, float *slab, const ThinGeom &tg,
                        const char *label, double flops, double bytes) {
    const int thick = THIN_OUT ? tg.Cin : tg.Cout;
    const int64_t wgs = (int64_t)tg.B * tg.bands * ceil_div(thick, NTH);
    if (wgs > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_backward_weight (thin): too many workgroups");
    ProfScope ps(label, st, flops, bytes);
    hipLaunchKernelGGL((conv_wgrad_thin<S, NT, NTH, THIN_OUT>), dim3((unsigned)wgs), dim3(256), 0, st, x, go, y,
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
        if (p.nt == 1) rc = launch_wgrad_thin_i<1, 1, 4, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out", flops, bytes);
        else if (p.nt == 3) rc = launch_wgrad_thin_i<1, 3, 2, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out", flops, bytes);
        else rc = launch_wgrad_thin_i<1, 4, 2, true>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/out", flops, bytes);
        break;
    case 2:
        if (p.nt == 1) rc = launch_wgrad_thin_i<1, 1, 4, false>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in", flops, bytes);
        else if (p.nt == 3) rc = launch_wgrad_thin_i<1, 3, 2, false>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in", flops, bytes);
        else rc = launch_wgrad_thin_i<1, 4, 1, false>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/in", flops, bytes);
        break;
    case 5:
        if (p.nt == 3) rc = launch_wgrad_thin_i<2, 3, 1, false>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/ins2", flops, bytes);
        else rc = launch_wgrad_thin_i<2, 4, 1, false>(st, x, go, y, gp, slab, tg, "conv_wgrad_thin/ins2", flops, bytes);
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

// ====================================================================================================================
// THIN-OUT 3x3 FORWARD on the matrix cores with the TAPS ON THE ROW AXIS.
//
// A 64 -> 3 convolution padded to a 32-row tile wastes 29 of 32 matrix rows, in each of 9 taps.  Written as
//   Q[(co, tap)][p'] = sum_ci w[co][ci][tap] * x[ci][p']          -- a 1x1 convolution with 9 Cout <= 27 output rows, K = Cin
//   out[co][p]       = act(bias[co] + sum_tap Q[(co, tap)][p + shift(tap)])
// the same tile carries 27 useful rows of 32 and the contraction is only Cin long: a ninth of the matrix work, and the
// B operand of v_mfma_f32_32x32x16_bf16 -- 8 consecutive k = 8 input CHANNELS of one pixel per lane -- comes straight from
// global memory (lanes = 32 consecutive pixels of a row: coalesced per channel plane), split into bf16 hi / lo in
// registers.  No LDS staging of the input at all; LDS only holds Q over the tile's window (8 x 64 pixels + halo = 10 x 66
// positions) for the shift-and-add.  Split precision like every forward convolution here (three products per k-step).
// Measured (B = 8, 256 x 256, 64 input channels; tools/thinfwd_time.py): 37 us (64 -> 3) and 33 us (64 -> 1) per launch against 72
// and 63 us on the padded 32-row tiles.  On the way: 58 us first; one / two tiles of lookahead and 16 instead of 8 waves per CU
// changed nothing; the pairwise hi / lo split (1700 -> 1250 vector instructions per wave) 50 us; then the two that mattered
// together, 50 -> 37 us: workgroup ids mapped so that an XCD owns a contiguous run of tiles (the halo rows / columns that
// neighbouring tiles share were fetched into two L2s: 176 MB of HBM reads for a 134 MB tensor) and the channel plane offset
// moved into the loads' scalar offset.  Ablation at the 50 us stage: 22 us without the loads (vector-pipe issue), 27 us more
// with them.  The same idea with the THIN tensor on the contraction axis (forward of 4 -> 64, data gradient of 64 -> 3) was
// built and measured slower than the kernels it would replace (139 vs 42 us, 56 vs 41 us: eight shifted, bounds-checked
// reads and a store address per output row cost more vector instructions than the padded tile wastes matrix work): removed.
constexpr int TF_TY = 8, TF_TX = 64, TF_WC = TF_TX + 2, TF_WPX = (TF_TY + 2) * TF_WC;        // window: 10 x 66 = 660 positions
constexpr int TF_NTILES = (TF_WPX + 31) / 32, TF_PITCH = TF_NTILES * 32, TF_ROWS = 27;       // 21 column tiles of 32
constexpr int TF_LDS = TF_ROWS * TF_PITCH * 4;                                               // 72.6 KB: two workgroups per CU
constexpr int TF_THREADS = 512, TF_WAVES = TF_THREADS / 64;                                  // 16 waves per CU keep enough loads in flight

// eight fp32 values -> their bf16 leading parts and bf16 remainders, PAIRWISE (v_cvt_pk_bf16_f32 converts two values per
// instruction and its result is already the packed pair the matrix fragment holds): ~3 vector instructions per value, where
// split_word + peel (one value at a time, then two byte permutes per pair) cost ~7 and made this kernel VALU-bound.
__device__ __forceinline__ void thin_split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    unsigned hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t p = {v[2 * i], v[2 * i + 1]};
        const bf16x2_t hp = __builtin_convertvector(p, bf16x2_t);
        const unsigned hb = __builtin_bit_cast(unsigned, hp);
        const f32x2_t hf = {__uint_as_float(hb << 16), __uint_as_float(hb & 0xffff0000u)};
        const bf16x2_t lp = __builtin_convertvector(p - hf, bf16x2_t);
        hw[i] = hb;
        lw[i] = __builtin_bit_cast(unsigned, lp);
    }
    const u32x4 hv = {hw[0], hw[1], hw[2], hw[3]}, lv = {lw[0], lw[1], lw[2], lw[3]};
    hi = __builtin_bit_cast(bf16x8, hv);
    lo = __builtin_bit_cast(bf16x8, lv);
}

__device__ __forceinline__ float thin_act_fwd(float v, int act, float slope) {
    const float lk = v > 0.f ? v : v * slope;
    const float sg = 1.f / (1.f + __expf(-v));
    return act == ACT_SIGMOID ? sg : (act == ACT_LEAKY ? lk : v);
}

template <int KSTEPS>                    // Cin / 16 (Cin % 16 == 0): the weight fragments of every k-step live in registers
__global__ __launch_bounds__(TF_THREADS, 4) void conv_thin_out_fwd(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ out, ThinGeom g,
                                                            int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float sQ[];                              // [27][TF_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    // (workgroup ids congruent mod 8 share an XCD and its L2: give each XCD a contiguous eighth of the tile sequence, so that the
    // halo rows / columns neighbouring tiles share are fetched into one L2 instead of two)
    int t = (gridDim.x & 7) == 0 ? xcd_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = tyi * TF_TY, x0 = txi * TF_TX;
    const int HW = g.H * g.W;
    const unsigned plane = (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)b * g.Cin * HW, (unsigned)g.Cin * plane);   // channels >= Cin: out of range, read 0
    const int rows_used = 9 * g.Cout;
    // ---- weight fragments: row i = lane & 31 = (co, tap), k = 16 ks + 8 h + e = input channel
    bf16x8 ah[KSTEPS], al[KSTEPS];
    {
        const int co = l31 / 9, tap = l31 - 9 * co;
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (unsigned)g.Cout * (unsigned)g.Cin * 36u);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float wd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ci = 16 * ks + 8 * h + e;
                wd[e] = buf_ld(rw, sel_off(l31 < rows_used && ci < g.Cin, (unsigned)((co * g.Cin + ci) * 9 + tap) * 4u));
            }
            thin_split8(wd, ah[ks], al[ks]);
        }
    }
    // ---- phase 1: Q over the window, a wave = every eighth column tile; the next tile's 32 loads go out before the current
    // tile is split and multiplied (two tiles ahead; without any prefetch the six tiles of a wave were six exposed memory round trips)
    auto issue = [&](int n, unsigned (&xw)[KSTEPS][8]) {
        const int p = 32 * n + l31;
        const int wr = p / TF_WC, wc = p - wr * TF_WC;
        const int yy = y0 - 1 + wr, xx = x0 - 1 + wc;
        const bool ok = n < TF_NTILES && p < TF_WPX && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const unsigned base = sel_off(ok, (unsigned)(yy * g.W + xx) * 4u + (unsigned)(8 * h) * plane);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            // the channel's plane offset is wave-uniform: it rides in the instruction's SCALAR offset (one vector add less per load).
            // The scalar offset is not part of the descriptor's range check: KSTEPS is exactly Cin / 16 (launcher), no channel
            // past the sample is ever addressed.
#pragma unroll
            for (int e = 0; e < 8; ++e) xw[ks][e] = __builtin_amdgcn_raw_buffer_load_b32(rx, base, (unsigned)(16 * ks + e) * plane, 0);
        }
    };
    constexpr int TPW = (TF_NTILES + TF_WAVES - 1) / TF_WAVES;  // column tiles per wave (3)
    unsigned xw[2][KSTEPS][8];                                  // one tile ahead: two register sets
    issue(wave, xw[0]);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int n = wave + TF_WAVES * i;
        if (n < TF_NTILES) {
            if (i + 1 < TPW) issue(n + TF_WAVES, xw[(i + 1) & 1]);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                float xv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = __uint_as_float(xw[i & 1][ks][e]);
                bf16x8 bh, bl;
                thin_split8(xv, bh, bl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc, 0, 0, 0);
            }
            const int p = 32 * n + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < TF_ROWS) sQ[row * TF_PITCH + p] = acc[r];
            }
        }
    }
    __syncthreads();
    // ---- phase 2: shift and add, bias, activation; a thread = one output pixel of the 8 x 64 tile
    const int HWo = g.Ho * g.Wo;
    for (int q = tid; q < TF_TY * TF_TX; q += TF_THREADS) {
        const int ty = q >> 6, tx = q & 63;
        const int yo = y0 + ty, xo = x0 + tx;
        if (yo >= g.Ho || xo >= g.Wo) continue;
        for (int co = 0; co < g.Cout; ++co) {
            float s = bias ? bias[co] : 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) s += sQ[(co * 9 + tap) * TF_PITCH + (ty + tap / 3) * TF_WC + tx + tap % 3];
            out[((int64_t)b * g.Cout + co) * HWo + (int64_t)yo * g.Wo + xo] = thin_act_fwd(s, g.act, g.slope);
        }
    }
}

// 3x3, stride 1, padding 1, Cout <= 3, Cin <= 64, at least 64 K output pixels: the tap-row forward
inline bool thin_out_fwd_ok(const ConvGeom &g, int ks, int stride) {
    return ks == 3 && stride == 1 && g.pad == 1 && g.groups == 1 && g.Cout <= 3 && g.Cin >= 16 && g.Cin <= 64 && g.Cin % 16 == 0 &&
           (int64_t)g.B * g.Ho * g.Wo >= 64 * 1024 && dev_getenv("EBFI_NO_THIN") == nullptr && dev_getenv("EBFI_NO_THIN_FWD") == nullptr;
}

int launch_thin_out_fwd(hipStream_t st, const float *x, const float *w, const float *bias, float *out, const ConvGeom &g, int act,
                        float slope) {
    ThinGeom tg{g.B, g.Cin, g.H, g.W, g.Cout, g.Ho, g.Wo, act, slope, 0, 0};
    const int tiles_x = ceil_div(g.Wo, TF_TX), tiles_y = ceil_div(g.Ho, TF_TY);
    const int64_t tiles = (int64_t)g.B * tiles_x * tiles_y;
    if (tiles > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d (thin forward): too many tiles");
    ProfScope ps("conv_thin_out_fwd", st, 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * 9, conv_bytes_fwd(g, 9, false));
#define EBFI_LAUNCH_TF(KS_)                                                                                                        \
    do {                                                                                                                           \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_thin_out_fwd<KS_>), TF_LDS)) return rc_;             \
        hipLaunchKernelGGL((conv_thin_out_fwd<KS_>), dim3((unsigned)tiles), dim3(TF_THREADS), TF_LDS, st, x, w, bias, out, tg, tiles_x, \
                           tiles_y);                                                                                               \
    } while (0)
    switch (g.Cin / 16) {
    case 1: EBFI_LAUNCH_TF(1); break;
    case 2: EBFI_LAUNCH_TF(2); break;
    case 3: EBFI_LAUNCH_TF(3); break;
    default: EBFI_LAUNCH_TF(4); break;
    }
#undef EBFI_LAUNCH_TF
    return check_launch("conv_thin_out_fwd");
}


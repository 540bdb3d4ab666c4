// THIN layers' weight gradients ON THE MATRIX CORES with the TAPS ON THE ROW AXIS OF THE THIN SIDE (round 6; included by conv2d.hip
// inside its anonymous namespace, after conv2d_thin.inc.hpp whose activation helpers it shares).
//
//   gw[co][ci][ky][kx] = sum_{b,y,x} gp[b][co][y][x] * X[b][ci][y + ky - P][x + kx - P],      gp = grad_out * act'(out)
//
// Four layers of the model have <= 4 channels on one side at full resolution and 16..64 on the other: the reconstruction's last
// convolution (64 -> 3 + sigmoid, model_singleframe.py:262), ExposureDecision's 4 -> 64 and 64 -> 1, and the detail branch's output
// convolution (16 -> 3, 7x7 on the reflection-padded 262 x 262 map, model_singleframe.py:207).  The generic matrix-core kernels pad
// the thin side to a 32- or 64-row tile in EVERY tap (157 us for the 7x7 layer: 3 live rows of 32, 49 taps); the direct fp32
// kernels of conv2d_thin.inc.hpp stream the thick tensor once but issue 1700 multiply-adds per pixel on the vector pipe (77-125 us
// per layer; they do not reach the 7x7 layer at all).  Here the product is written with the pixel as the CONTRACTION index and
// the shift on the thin operand:
//
//   THIN_OUT (Cout <= 4):  C[(co, ky, kx)][ci] = sum_{p' in X's domain} gp[co][p' - (ky, kx) + P] * X[ci][p']
//   THIN_IN  (Cin  <= 4):  C[(ci, ky, kx)][co] = sum_{p in gp's domain} X[ci][p + (ky, kx) - P]  * gp[co][p]
//
// A = the thin tensor, one matrix ROW per (thin channel, tap): NT * KS * KS rows (27 / 36 / 9 / 147 -- 2, 3, 1, 10 tiles of
// v_mfma_f32_16x16x32_bf16, nearly full), B = the thick tensor read ONCE, unshifted, 16 channels per tile.  Operands are bf16
// hi + lo pairs, three products per tile (hi.hi, hi.lo, lo.hi: the split-precision scheme of every other matrix-core kernel of this
// file, ~1e-5 of fp32); the exact-fp32 mode keeps the direct kernels.
//
// Built for the model's shapes: 3x3 same-padded layers with 1 or 3 output channels from 64, or 64 from 4, and the 7x7 unpadded
// 16 -> 3 layer (shift_wgrad_geometry); >= 64 K output pixels per launch -- below that the generic kernels' fixed costs are not the
// problem.
//
// Workgroup = (sample, band of 16 thick rows, segment of 64 thick columns), 256 threads:
//   * the thin tile the band can touch -- NT x (16 + KS - 1) x (64 + KS - 1) values, zero outside the thin tensor (= zero padding)
//     -- is staged ONCE, split, as four LDS arrays: hi / lo x two copies, the second shifted by one element, so that the 8
//     consecutive elements an A fragment needs (k = 8 consecutive pixels, from column `8 j + shift(kx)`) start on a dword in one of
//     the copies whatever the parity of the shift: four aligned ds_read_b32 per fragment, no byte permutes;
//   * per thick row: 16 / 64 channels x 64 pixels are loaded (THREE rows of loads in flight: see the row loop), multiplied
//     by act'(out) when the thick side is the gradient, split and written as [channel][72] bf16 rows (144-byte stride: the 16
//     lanes of a B fragment hit 16 different bank quads) into one of two buffers; two 32-pixel contraction steps per row;
//   * wave w owns the output tiles w, w + 4, ...; accumulators stay in registers for the whole band; the workgroup writes ONE slab
//     (conv_wgrad_reduce_f32 sums them in a fixed order: bit-reproducible like every other weight gradient here).
// The bias gradient: THIN_IN -- an extra A row of ones gives sum_p gp[co][p] from the same products; THIN_OUT -- summed from the
// thin values while they are staged (each thin pixel is "owned" by the workgroup whose band / segment contains its coordinates).
// grad_preact_out (the side tensor gp for the data gradient) is written by whoever loads the owned values.

constexpr int SH_R = 16;               // thick rows per workgroup
constexpr int SH_W = 64;               // thick columns per workgroup (two contraction steps of 32)
constexpr int SH_UP = 72;              // elements per thick channel row in LDS (64 + 8: 144-byte stride)
constexpr int SH_NUMAX = 64;           // thick channels (a multiple of 16)

struct ShiftGeom {
    int B, Cin, Cout;
    int Ht, Wt;                        // thin tensor's spatial size
    int Hu, Wu;                        // thick tensor's
    int NU;                            // thick channels
    int P;                             // padding
    int act;
    float slope;
    int bands, segs;
};

template <int KS, int NT>
struct ShiftDims {
    static constexpr int KK = KS * KS;
    static constexpr int TH = SH_R + KS - 1, TW = SH_W + KS - 1;
    static constexpr int TWP = (SH_W - 8 + KS - 1 + 8 + 1) / 2 * 2 + 2;           // >= 63 + KS, even (KS = 3: 68, KS = 7: 72)
    static constexpr int TSEL = NT * TH * TWP;                                     // elements of one of the four thin arrays
    static constexpr int THIN_BYTES = 4 * TSEL * 2;
    static constexpr int THICK_BYTES = 2 * 2 * SH_NUMAX * SH_UP * 2;
    static constexpr int LDS_BYTES = THIN_BYTES + THICK_BYTES + 64;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence over GLOBAL memory as well, which
// on gfx9 is `s_waitcnt vmcnt(0)`: it drained the row loads this kernel keeps in flight across its barriers (first version: one full
// memory round trip per thick row, 2.7 us of a 0.3 us step).  No thread of the workgroup reads global memory another one wrote.
__device__ __forceinline__ void sh_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned sh_pack_bf16(float a, float b) {
    typedef __bf16 bf16x2_sh __attribute__((ext_vector_type(2)));
    const bf16x2_sh v{(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// NCHR: 16-channel blocks of the thick tensor (1 or 4); ALIGNED: its rows are whole 16-byte quads on aligned bases (16-byte loads and
// stores; otherwise dwords); GPOUT (THIN_IN): the thick side tensor grad * act' is written out.  All three are compile-time so that
// the row loop is straight-line code: with runtime branches around every load the wait-count pass put `s_waitcnt vmcnt(0)` in
// front of each NEW load (a merge of control-flow paths is all-pending to it), i.e. one memory round trip per row again.
template <int KS, int NT, bool THIN_OUT, int NCHR, bool ALIGNED, bool GPOUT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_shift(const float *__restrict__ x, const float *__restrict__ gout,
                                                           const float *__restrict__ yact, float *__restrict__ gpre_out,
                                                           float *__restrict__ slab, ShiftGeom g) {
    using D = ShiftDims<KS, NT>;
    constexpr int KK = D::KK, TH = D::TH, TW = D::TW, TWP = D::TWP, TSEL = D::TSEL;
    constexpr int MW = NT * KK;                                  // weight rows of the product
    constexpr int MROWS = MW + (THIN_OUT ? 0 : 1);               // + the row of ones (bias of the thick = output channels)
    constexpr int MT = (MROWS + 15) / 16;                        // row tiles; also the most tiles a wave can own (<= 4 column tiles)
    extern __shared__ __attribute__((aligned(16))) char shm[];
    __bf16 *thin = reinterpret_cast<__bf16 *>(shm);              // [copy][hi | lo][NT][TH][TWP]
    char *thick = shm + D::THIN_BYTES;                           // [buffer][hi | lo][SH_NUMAX][SH_UP] bf16
    float *red = reinterpret_cast<float *>(shm + D::THIN_BYTES + D::THICK_BYTES);   // [4][NT] (THIN_OUT bias)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t_ = blockIdx.x;
    const int seg = t_ % g.segs; t_ /= g.segs;
    const int band = t_ % g.bands;
    const int b = t_ / g.bands;
    const int y0 = band * SH_R, x0 = seg * SH_W;
    const int HWt = g.Ht * g.Wt, HWu = g.Hu * g.Wu;
    // thin / thick tensors of this sample (THIN_OUT: thin = grad_out (+ saved output), thick = input; THIN_IN: the other way round)
    const float *thin_p = (THIN_OUT ? gout : x) + (int64_t)b * NT * HWt;
    const float *thick_p = (THIN_OUT ? x : gout) + (int64_t)b * g.NU * HWu;
    const bool has_y = yact != nullptr && g.act != ACT_NONE;
    const float *ythin_p = (THIN_OUT && has_y) ? yact + (int64_t)b * NT * HWt : nullptr;
    const float *ythick_p = (!THIN_OUT && has_y) ? yact + (int64_t)b * g.NU * HWu : nullptr;
    float *gp_thin = (THIN_OUT && gpre_out) ? gpre_out + (int64_t)b * NT * HWt : nullptr;
    float *gp_thick = (!THIN_OUT && gpre_out) ? gpre_out + (int64_t)b * g.NU * HWu : nullptr;
    const __amdgpu_buffer_rsrc_t rthin = make_rsrc(thin_p, (unsigned)NT * (unsigned)HWt * 4u);
    const __amdgpu_buffer_rsrc_t rythin = make_rsrc(ythin_p ? ythin_p : thin_p, ythin_p ? (unsigned)NT * (unsigned)HWt * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rthick = make_rsrc(thick_p, (unsigned)g.NU * (unsigned)HWu * 4u);
    const __amdgpu_buffer_rsrc_t rythick = make_rsrc(ythick_p ? ythick_p : thick_p, ythick_p ? (unsigned)g.NU * (unsigned)HWu * 4u : 0u);
    const ThinAct da = thin_act(g.act, g.slope);

    // ---- the thin tile, once: element (t, tr, tc) = T[t][ty0 + tr][tx0 + tc], 0 outside the tensor
    // A[(t, ky, kx)][thick (r, c)] = T[t][r + dy][c + dx]: THIN_OUT (dy, dx) = (P - ky, P - kx), THIN_IN (ky - P, kx - P)
    const int dmin = THIN_OUT ? g.P - (KS - 1) : -g.P;
    const int ty0 = y0 + dmin, tx0 = x0 + dmin;
    float bs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bs[t] = 0.f;
    {
        // every load of the tile goes out before the first value is used (a load per loop iteration was a chain of 18 dependent
        // round trips: ~27 us of the first version's 45 us per workgroup)
        constexpr int NEL = NT * TH * TW, NIT = (NEL + 255) / 256;
        float tvv[NIT], tyv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            const int t = idx / (TH * TW);
            const int rem = idx - t * (TH * TW);
            const int tr = rem / TW, tc = rem - tr * TW;
            const int ty = ty0 + tr, tx = tx0 + tc;
            const bool ok = idx < NEL && ty >= 0 && ty < g.Ht && tx >= 0 && tx < g.Wt;
            const unsigned off = sel_off(ok, ((unsigned)t * (unsigned)HWt + (unsigned)(ty * g.Wt + tx)) * 4u);
            tvv[it] = buf_ld(rthin, off);
            tyv[it] = THIN_OUT ? buf_ld(rythin, off) : 0.f;       // (no activation: empty descriptor, reads 0; thin_dact ignores it)
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            if (idx >= NEL) continue;
            const int t = idx / (TH * TW);
            const int rem = idx - t * (TH * TW);
            const int tr = rem / TW, tc = rem - tr * TW;
            float v = tvv[it];
            if constexpr (THIN_OUT) {
                const int ty = ty0 + tr, tx = tx0 + tc;
                v = thin_dact(v, tyv[it], da);                    // (0 outside the tensor: v is)
                // the workgroup whose band / segment holds (ty, tx) owns the pixel: bias sum and the side tensor
                const bool own = ty >= y0 && ty < min(y0 + SH_R, g.Ht) && tx >= x0 && tx < min(x0 + SH_W, g.Wt);
#pragma unroll
                for (int q = 0; q < NT; ++q) bs[q] += (own && t == q) ? v : 0.f;
                if (own && gp_thin) gp_thin[(int64_t)t * HWt + (int64_t)ty * g.Wt + tx] = v;
            }
            const __bf16 h = (__bf16)v;
            const __bf16 l = (__bf16)(v - (float)h);
            const int e = (t * TH + tr) * TWP + tc;
            thin[0 * TSEL + e] = h;                               // copy 0: hi, lo
            thin[1 * TSEL + e] = l;
            if (tc >= 1) {                                        // copy 1 = copy 0 shifted by one element
                thin[2 * TSEL + e - 1] = h;
                thin[3 * TSEL + e - 1] = l;
            }
        }
    }

    // ---- this lane's A fragments: per owned tile the LDS byte address of its row's first element for thick (row 0, column 8 kg)
    constexpr int ntl = NCHR;                                    // column tiles (16 thick channels each)
    constexpr int ntiles = MT * ntl;
    const int kg = lane >> 4;
    unsigned abase[MT];
    unsigned okmask = 0, onemask = 0;                            // bit i: tile i's row is a weight row / the row of ones
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int tile = wave + 4 * i;
        const int mt = tile / ntl;
        const int m = 16 * mt + (lane & 15);
        const bool wrow = tile < ntiles && m < MW;
        const int mm = wrow ? m : 0;
        const int t = mm / KK, tap = mm - t * KK;
        const int ky = tap / KS, kx = tap - ky * KS;
        const int roff = THIN_OUT ? KS - 1 - ky : ky, coff = THIN_OUT ? KS - 1 - kx : kx;
        const int col0 = 8 * kg + coff, par = col0 & 1;
        abase[i] = (unsigned)((2 * par) * TSEL + (t * TH + roff) * TWP + (col0 - par)) * 2u;
        okmask |= wrow ? (1u << i) : 0u;
        onemask |= (!THIN_OUT && tile < ntiles && m == MW) ? (1u << i) : 0u;
    }
    typedef float f32x4_sh __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4_sh __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_sh __attribute__((ext_vector_type(2)));
    f32x4_sh acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_sh{0.f, 0.f, 0.f, 0.f};

    // ---- thick rows: thread = (channel ub + 16 i, quad q of the 64 columns)
    const int q4 = (tid & 15) * 4, ub = tid >> 4;
    constexpr int NCH = NCHR;
    constexpr int NLD = NCHR * (ALIGNED ? 1 : 4) * (THIN_OUT ? 1 : 2);      // loads per thread and thick row
    // (THIN_IN: the saved output travels in registers beside the gradient and act' is applied when the row is written to LDS --
    //  applied at load time it made every row wait for its own loads)
    constexpr int NYC = THIN_OUT ? 1 : NCH;
    auto load_row = [&](int r, float (&tv)[NCH][4], float (&ty)[NYC][4]) {
        const int y = y0 + r;
        const bool row_ok = y < g.Hu;
        const int cx = x0 + q4;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int u = ub + 16 * i;
            const unsigned base = ((unsigned)u * (unsigned)HWu + (unsigned)(y * g.Wu + cx)) * 4u;
            float gq[4], yq[4] = {0.f, 0.f, 0.f, 0.f};
#if defined(EBFI_SHIFT_DIAG) && EBFI_SHIFT_DIAG == 2
            gq[0] = gq[1] = gq[2] = gq[3] = (float)(base & 7u);  // (diagnostic build: no global loads of the thick tensor)
            if (false)
#endif
            if constexpr (ALIGNED) {                             // whole quads inside or outside the row
                const unsigned off = sel_off(row_ok && cx < g.Wu, base);
                const f32x4_sh t4 = __builtin_bit_cast(f32x4_sh, __builtin_amdgcn_raw_buffer_load_b128(rthick, off, 0, 0));
                gq[0] = t4.x; gq[1] = t4.y; gq[2] = t4.z; gq[3] = t4.w;
                if constexpr (!THIN_OUT) {
                    const f32x4_sh y4 = __builtin_bit_cast(f32x4_sh, __builtin_amdgcn_raw_buffer_load_b128(rythick, off, 0, 0));
                    yq[0] = y4.x; yq[1] = y4.y; yq[2] = y4.z; yq[3] = y4.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = sel_off(row_ok && cx + j < g.Wu, base + 4u * (unsigned)j);
                    gq[j] = buf_ld(rthick, off);
                    if constexpr (!THIN_OUT) yq[j] = buf_ld(rythick, off);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tv[i][j] = gq[j];
                if constexpr (!THIN_OUT) ty[i][j] = yq[j];
            }
        }
    };
    auto store_row = [&](int r, int buf, float (&tv)[NCH][4], const float (&ty)[NYC][4]) {   // (act',) split, write [buf][hi | lo][u][q4 .. q4 + 3]; the side tensor
        const int y = y0 + r, cx = x0 + q4;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int u = ub + 16 * i;
            float hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!THIN_OUT) tv[i][j] = thin_dact(tv[i][j], ty[i][j], da);
                hi[j] = (float)(__bf16)tv[i][j];
                lo[j] = tv[i][j] - hi[j];
            }
            char *dst = thick + ((buf * 2) * SH_NUMAX * SH_UP + u * SH_UP + q4) * 2;
            *reinterpret_cast<u32x2_sh *>(dst) = u32x2_sh{sh_pack_bf16(hi[0], hi[1]), sh_pack_bf16(hi[2], hi[3])};
            *reinterpret_cast<u32x2_sh *>(dst + SH_NUMAX * SH_UP * 2) = u32x2_sh{sh_pack_bf16(lo[0], lo[1]), sh_pack_bf16(lo[2], lo[3])};
            if constexpr (!THIN_OUT && GPOUT) {
                if (y < g.Hu) {
                    float *o = gp_thick + (int64_t)u * HWu + (int64_t)y * g.Wu + cx;
                    if constexpr (ALIGNED) {
                        if (cx < g.Wu) *reinterpret_cast<f32x4_sh *>(o) = f32x4_sh{tv[i][0], tv[i][1], tv[i][2], tv[i][3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cx + j < g.Wu) o[j] = tv[i][j];
                    }
                }
            }
        }
    };
    auto products = [&](int r) {                                 // the two 32-pixel contraction steps of thick row r (buffer r & 1)
#if defined(EBFI_SHIFT_DIAG) && EBFI_SHIFT_DIAG == 1
        return;                                                  // (diagnostic build: no LDS reads, no matrix products)
#endif
        const char *tb = thick + ((r & 1) * 2) * SH_NUMAX * SH_UP * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int tile = wave + 4 * i;
                if (4 * i + 3 >= ntiles && tile >= ntiles) continue;   // (only the last round of tiles can be short: folds away elsewhere once unrolled)
                const int nt = tile - (tile / ntl) * ntl;
                // B fragment: 8 consecutive pixels of thick channel 16 nt + (lane & 15)
                const char *bp = tb + ((16 * nt + (lane & 15)) * SH_UP + 32 * s + 8 * kg) * 2;
                const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(bp);
                const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(bp + SH_NUMAX * SH_UP * 2);
                // A fragment: 8 consecutive thin elements, dword-aligned in the copy of the right parity
                const char *ap = reinterpret_cast<const char *>(thin) + abase[i] + (unsigned)(r * TWP + 32 * s) * 2u;
                u32x4_sh ah, al;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ah[j] = *reinterpret_cast<const unsigned *>(ap + 4 * j);
                    al[j] = *reinterpret_cast<const unsigned *>(ap + TSEL * 2 + 4 * j);
                }
                const bool wr = (okmask >> i) & 1u, one = (onemask >> i) & 1u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ah[j] = wr ? ah[j] : (one ? 0x3f803f80u : 0u);   // bf16 1.0 pairs in the row of ones, zeros in the padding rows
                    al[j] = wr ? al[j] : 0u;
                }
                const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah), a_l = __builtin_bit_cast(bf16x8, al);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, bl, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, bh, acc[i], 0, 0, 0);
            }
        }
    };
    // THREE rows of loads in flight (row j lives in register set j % 4): while row k multiplies, rows k + 1 .. k + 3 are in registers
    // or on their way -- with two workgroups of four waves per CU one row's products (~0.3 us) are much shorter than a round trip
    // to HBM under load (~2.5 us), and the first versions, one and then two rows ahead, moved 3.3 TB/s.
    // Steady state (rows up to k + 3 exist in all four phases): every load is issued UNCONDITIONALLY, so the body is one basic block
    // and the wait-count pass sees that row k + 1's NLD loads are complete once at most 2 NLD younger ones are outstanding (explicit
    // counts; with a branch around the younger loads it merged the two paths into "anything may be pending" and drained everything
    // before each LDS write).  With the side tensor's stores in flight loads and stores share the counter out of order: 0.
    const int nrows = min(SH_R, g.Hu - y0);
    float tv0[NCH][4], tv1[NCH][4], tv2[NCH][4], tv3[NCH][4];
    float ty0_[NYC][4], ty1_[NYC][4], ty2_[NYC][4], ty3_[NYC][4];
    constexpr int WOLD = (!THIN_OUT && GPOUT) ? 0 : (2 * NLD < 60 ? 2 * NLD : 60);   // (a smaller count only waits longer; the counter has 6 bits)
    load_row(0, tv0, ty0_);
    if (1 < nrows) load_row(1, tv1, ty1_);
    if (2 < nrows) load_row(2, tv2, ty2_);
    store_row(0, 0, tv0, ty0_);
    sh_lds_barrier();                                            // thin tile and thick row 0 are in LDS
#define SH_STEP(K_, SA_, YA_, SB_, YB_)                                                                                  \
    do {                                                                                                                 \
        load_row((K_) + 3, SA_, YA_);                                                                                    \
        products(K_);                                                                                                    \
        wait_vmcnt<WOLD>();                                                                                              \
        store_row((K_) + 1, ((K_) + 1) & 1, SB_, YB_);   /* the other buffer: its readers finished before the last barrier */ \
        sh_lds_barrier();                                                                                                \
    } while (0)
    int k = 0;
    for (; k + 6 < nrows; k += 4) {
        SH_STEP(k, tv3, ty3_, tv1, ty1_);
        SH_STEP(k + 1, tv0, ty0_, tv2, ty2_);
        SH_STEP(k + 2, tv1, ty1_, tv3, ty3_);
        SH_STEP(k + 3, tv2, ty2_, tv0, ty0_);
    }
#undef SH_STEP
    // tail (one to six rows; k is a multiple of 4: row k is in LDS, rows k + 1, k + 2 -- if any -- are in their sets)
#define SH_TAIL(P_, SA_, YA_, SB_, YB_)                                                                                  \
    if (k + (P_) < nrows) {                                                                                              \
        if (k + (P_) + 3 < nrows) load_row(k + (P_) + 3, SA_, YA_);                                                      \
        products(k + (P_));                                                                                              \
        if (k + (P_) + 1 < nrows) {                                                                                      \
            store_row(k + (P_) + 1, (k + (P_) + 1) & 1, SB_, YB_);                                                       \
            sh_lds_barrier();                                                                                            \
        }                                                                                                                \
    }
    SH_TAIL(0, tv3, ty3_, tv1, ty1_)
    SH_TAIL(1, tv0, ty0_, tv2, ty2_)
    SH_TAIL(2, tv1, ty1_, tv3, ty3_)
    SH_TAIL(3, tv2, ty2_, tv0, ty0_)
    SH_TAIL(4, tv3, ty3_, tv1, ty1_)
    SH_TAIL(5, tv0, ty0_, tv2, ty2_)
#undef SH_TAIL

    // ---- this workgroup's slab
    const int64_t n_weight = (int64_t)g.Cout * g.Cin * KK, n_total = n_weight + g.Cout;
    float *my = slab + (int64_t)blockIdx.x * n_total;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int tile = wave + 4 * i;
        if (tile >= ntiles) continue;
        const int mt = tile / ntl, nt = tile - mt * ntl;
        const int n = 16 * nt + (lane & 15);                      // thick channel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = 16 * mt + 4 * kg + j;
            if (m < MW) {
                const int t = m / KK, tap = m - t * KK;
                const int co = THIN_OUT ? t : n, ci = THIN_OUT ? n : t;
                my[((int64_t)co * g.Cin + ci) * KK + tap] = acc[i][j];
            } else if (!THIN_OUT && m == MW) {
                my[n_weight + n] = acc[i][j];                     // sum of gp over this workgroup's pixels
            }
        }
    }
    if constexpr (THIN_OUT) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float s = thin_wave_sum(bs[t]);
            if (lane == 0) red[wave * NT + t] = s;
        }
        __syncthreads();
        if (tid < NT) my[n_weight + tid] = (red[tid] + red[NT + tid]) + (red[2 * NT + tid] + red[3 * NT + tid]);
    }
}

struct ShiftPlan {
    int kind;      // 0 = none; 1 = thin out (Cout = nt); 2 = thin in (Cin = nt)
    int nt, ks;
};

// Geometry half of the plan (shared with the workspace size): which layers the kernel above serves
inline ShiftPlan shift_wgrad_geometry(const ConvGeom &g, int ks, int stride) {
    ShiftPlan p{0, 0, ks};
    if (g.groups != 1 || stride != 1 || g.B < 1) return p;
    if ((int64_t)g.B * g.Ho * g.Wo < 64 * 1024) return p;         // small maps: fixed costs, not the padded tiles, are the problem
    // (the instances that are built: the model's shapes.  64 thick channels for the 3x3 layers, 16 for the 7x7 one)
    if (ks == 3 && g.pad == 1) {
        if ((g.Cout == 1 || g.Cout == 3) && g.Cin == 64) { p.kind = 1; p.nt = g.Cout; }
        else if (g.Cin == 4 && g.Cout == 64) { p.kind = 2; p.nt = 4; }
    } else if (ks == 7 && g.pad == 0 && g.Cout == 3 && g.Cin == 16) {
        p.kind = 1; p.nt = 3;
    }
    return p;
}
inline int shift_wgrad_slabs(const ConvGeom &g, const ShiftPlan &p) {
    const int Hu = p.kind == 1 ? g.H : g.Ho, Wu = p.kind == 1 ? g.W : g.Wo;
    return g.B * (int)ceil_div(Hu, SH_R) * (int)ceil_div(Wu, SH_W);
}
inline ShiftPlan shift_wgrad_plan(const ConvGeom &g, int ks, int stride, const void *x, const void *go, const void *y, const void *gp) {
    ShiftPlan p = shift_wgrad_geometry(g, ks, stride);
    if (p.kind == 0) return p;
    // (dwords everywhere except the optional 16-byte thick accesses, which check their own alignment: only fp32 alignment is required)
    if (dev_getenv("EBFI_NO_SHIFT_WGRAD") != nullptr) p.kind = 0;
    (void)x; (void)go; (void)y; (void)gp;
    return p;
}

template <int KS, int NT, bool THIN_OUT, int NCHR, bool ALIGNED, bool GPOUT>
int launch_wgrad_shift_i(hipStream_t st, const float *x, const float *go, const float *y, float *gp, float *slab, const ShiftGeom &sg,
                         const char *label, double flops, double bytes) {
    using D = ShiftDims<KS, NT>;
    const int64_t wgs = (int64_t)sg.B * sg.bands * sg.segs;
    if (wgs > 2147483647LL) return fail(EBFI_ERR_ARG, "conv2d_backward_weight (shift): too many workgroups");
    if (sg.NU != 16 * NCHR) return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight (shift): %d thick channels", sg.NU);
    auto kern = &conv_wgrad_shift<KS, NT, THIN_OUT, NCHR, ALIGNED, GPOUT>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    ProfScope ps(label, st, flops, bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), D::LDS_BYTES, st, x, go, y, gp, slab, sg);
    return check_launch(label);
}

// grad_weight / grad_bias (+ grad_preact_out) of a thin layer on the matrix cores: the kernel above, then the shared slab reduction
int launch_wgrad_shift(hipStream_t st, const ShiftPlan &p, const float *x, const float *go, const float *y, float *gp, float *slab,
                       const ConvGeom &g, int act, float slope, float *gw, float *gb) {
    const bool out = p.kind == 1;
    ShiftGeom sg;
    sg.B = g.B; sg.Cin = g.Cin; sg.Cout = g.Cout;
    sg.Ht = out ? g.Ho : g.H; sg.Wt = out ? g.Wo : g.W;
    sg.Hu = out ? g.H : g.Ho; sg.Wu = out ? g.W : g.Wo;
    sg.NU = out ? g.Cin : g.Cout;
    sg.P = g.pad; sg.act = act; sg.slope = slope;
    sg.bands = (int)ceil_div(sg.Hu, SH_R); sg.segs = (int)ceil_div(sg.Wu, SH_W);
    if ((int64_t)sg.NU * sg.Hu * sg.Wu * 4 >= (1LL << 31) - (1LL << 26))
        return fail(EBFI_ERR_ARG, "conv2d_backward_weight (shift): one sample exceeds the 2 GiB reach of 32-bit buffer offsets");
    const float *thickp = out ? x : go;
    const bool al = sg.Wu % 4 == 0 && aligned16(thickp) && (out || act == ACT_NONE || !y || aligned16(y)) && (out || !gp || aligned16(gp));
    const int kk = p.ks * p.ks;
    const double flops = 2.0 * g.B * g.Ho * g.Wo * (double)g.Cout * g.Cin * kk;
    const double bytes = conv_bytes_wgrad(g, kk, act != ACT_NONE, gp != nullptr);
    int rc = EBFI_ERR_UNSUPPORTED;
#define EBFI_SHIFT(KS_, NT_, OUT_, NCHR_, GP_, LABEL_)                                                                              \
    (al ? launch_wgrad_shift_i<KS_, NT_, OUT_, NCHR_, true, GP_>(st, x, go, y, gp, slab, sg, LABEL_, flops, bytes)                  \
        : launch_wgrad_shift_i<KS_, NT_, OUT_, NCHR_, false, GP_>(st, x, go, y, gp, slab, sg, LABEL_, flops, bytes))
    if (p.ks == 7 && out && p.nt == 3)        // (dword loads serve quad-aligned rows too: one instance)
        rc = launch_wgrad_shift_i<7, 3, true, 1, false, false>(st, x, go, y, gp, slab, sg, "conv_wgrad_shift/out7", flops, bytes);
    else if (p.ks == 3 && out && p.nt == 3) rc = EBFI_SHIFT(3, 3, true, 4, false, "conv_wgrad_shift/out");
    else if (p.ks == 3 && out && p.nt == 1) rc = EBFI_SHIFT(3, 1, true, 4, false, "conv_wgrad_shift/out");
    else if (p.ks == 3 && !out && p.nt == 4 && gp) rc = EBFI_SHIFT(3, 4, false, 4, true, "conv_wgrad_shift/in");
    else if (p.ks == 3 && !out && p.nt == 4) rc = EBFI_SHIFT(3, 4, false, 4, false, "conv_wgrad_shift/in");
    else return fail(EBFI_ERR_UNSUPPORTED, "conv2d_backward_weight (shift): no kernel for this shape");
#undef EBFI_SHIFT
    if (rc) return rc;
    const int64_t n_weight = (int64_t)g.Cout * g.Cin * kk, n_total = n_weight + g.Cout;
    {
        ProfScope ps("conv_wgrad_reduce_tall", st);
        hipLaunchKernelGGL(conv_wgrad_reduce_tall, dim3((unsigned)ceil_div(n_total, 16)), dim3(256), 0, st, slab, shift_wgrad_slabs(g, p), n_weight,
                           n_total, gw, gb);
    }
    return check_launch("conv_wgrad_reduce_tall");
}

// ====================================================================================================================
// The DATA gradient of the 7x7 16 -> 3 layer (the detail branch's output convolution) with the same operand form:
//
//   gx[ci][y'][x'] = sum_{co, ky, kx} w[co][ci][ky][kx] * gp[co][y' + P - ky][x' + P - kx]
//
// conv7_x3 pads the three gradient channels to a 16-channel staging chunk and walks 49 taps of a 32-row tile: 108 us for 35 MB
// of output.  Here the contraction index is k = (co, ky, j), j = 0..7 one 8-element run ALONG x of the thin tile (j = 7 - kx; j = 0
// meets a zero weight): 21 runs = 168 -> 192 = six contraction steps of v_mfma_f32_16x16x32_bf16 per 16 output pixels, with
//   A[ci][(co, ky, j)] = w[co][ci][ky][7 - j]         the whole weight: 6 x (hi, lo) fragments a lane keeps in registers,
//   B[(co, ky, j)][x'] = gp[co][y' + P - ky][x' + P - 7 + j]   8 consecutive elements of the staged thin tile per lane -- from the
//                                                      copy whose parity makes them dword-aligned (see conv_wgrad_shift),
// and the 16 x 16 result tile is the 16 input channels of 16 pixels.  Workgroup = (sample, 16 rows, 64 columns) of grad_input;
// the only global traffic is the thin tile (once) and the result.
constexpr int D7_R = 16, D7_W = 64, D7_KS = 7, D7_NT = 3, D7_CI = 16;
constexpr int D7_TH = D7_R + D7_KS - 1, D7_TW = D7_W + D7_KS, D7_TWP = 72;
constexpr int D7_TSEL = D7_NT * D7_TH * D7_TWP;
constexpr int D7_LDS = 4 * D7_TSEL * 2;

struct D7Geom {
    int B, H, W, Ho, Wo, P, act;
    float slope;
    int bands, segs;
};

__global__ __launch_bounds__(256) void conv7_thin_dgrad(const float *__restrict__ gout, const float *__restrict__ yact,
                                                        const float *__restrict__ wgt, float *__restrict__ gx, D7Geom g) {
    constexpr int KS = D7_KS, NT = D7_NT, TH = D7_TH, TW = D7_TW, TWP = D7_TWP, TSEL = D7_TSEL, KK = KS * KS;
    __shared__ __attribute__((aligned(16))) __bf16 thin[4 * TSEL];       // [copy][hi | lo][NT][TH][TWP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t_ = blockIdx.x;
    const int seg = t_ % g.segs; t_ /= g.segs;
    const int band = t_ % g.bands;
    const int b = t_ / g.bands;
    const int y0 = band * D7_R, x0 = seg * D7_W;
    const int HWo = g.Ho * g.Wo, HW = g.H * g.W;
    const float *gp_p = gout + (int64_t)b * NT * HWo;
    const bool has_y = yact != nullptr && g.act != ACT_NONE;
    const __amdgpu_buffer_rsrc_t rgp = make_rsrc(gp_p, (unsigned)NT * (unsigned)HWo * 4u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(has_y ? yact + (int64_t)b * NT * HWo : gp_p, has_y ? (unsigned)NT * (unsigned)HWo * 4u : 0u);
    const ThinAct da = thin_act(g.act, g.slope);
    // thin tile: rows y0 + P - 6 .. y0 + 15 + P, columns x0 + P - 7 .. x0 + 63 + P; 0 outside the gradient
    const int ty0 = y0 + g.P - (KS - 1), tx0 = x0 + g.P - KS;
    {
        constexpr int NEL = NT * TH * TW, NIT = (NEL + 255) / 256;
        float tvv[NIT], tyv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            const int t = idx / (TH * TW);
            const int rem = idx - t * (TH * TW);
            const int tr = rem / TW, tc = rem - tr * TW;
            const int ty = ty0 + tr, tx = tx0 + tc;
            const bool ok = idx < NEL && ty >= 0 && ty < g.Ho && tx >= 0 && tx < g.Wo;
            const unsigned off = sel_off(ok, ((unsigned)t * (unsigned)HWo + (unsigned)(ty * g.Wo + tx)) * 4u);
            tvv[it] = buf_ld(rgp, off);
            tyv[it] = buf_ld(ry, off);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            if (idx >= NEL) continue;
            const int t = idx / (TH * TW);
            const int rem = idx - t * (TH * TW);
            const int tr = rem / TW, tc = rem - tr * TW;
            const float v = thin_dact(tvv[it], tyv[it], da);
            const __bf16 h = (__bf16)v;
            const __bf16 l = (__bf16)(v - (float)h);
            const int e = (t * TH + tr) * TWP + tc;
            thin[0 * TSEL + e] = h;
            thin[1 * TSEL + e] = l;
            if (tc >= 1) {
                thin[2 * TSEL + e - 1] = h;
                thin[3 * TSEL + e - 1] = l;
            }
        }
    }
    // the weight as A fragments: lane (ci = lane & 15, kg = lane >> 4), step s: run q = 4 s + kg = (co, ky); element j -> kx = 7 - j
    typedef float f32x4_d7 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4_d7 __attribute__((ext_vector_type(4)));
    const int ci = lane & 15, kg = lane >> 4;
    bf16x8 ah[6], al[6];
    unsigned boff[6];                                            // LDS element offset of run q's row: (co * TH + (6 - ky)) * TWP
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int q = 4 * s + kg;
        const bool live = q < NT * KS;
        const int qq = live ? q : 0;
        const int co = qq / KS, ky = qq - co * KS;
        float wv[8];
        wv[0] = 0.f;
#pragma unroll
        for (int j = 1; j < 8; ++j) wv[j] = live ? wgt[((int64_t)co * D7_CI + ci) * KK + ky * KS + (7 - j)] : 0.f;
        u32x4_d7 h4, l4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h0 = (float)(__bf16)wv[2 * j], h1 = (float)(__bf16)wv[2 * j + 1];
            h4[j] = sh_pack_bf16(h0, h1);
            l4[j] = sh_pack_bf16(wv[2 * j] - h0, wv[2 * j + 1] - h1);
        }
        ah[s] = __builtin_bit_cast(bf16x8, h4);
        al[s] = __builtin_bit_cast(bf16x8, l4);
        boff[s] = (unsigned)((co * TH + (KS - 1 - ky)) * TWP);     // (padding runs: any live row -- their weights are zero)
    }
    __syncthreads();
    // rows of this wave: r = wave, wave + 4, ...; per row four 16-pixel tiles
    const int par = lane & 1;
    const int nrows = min(D7_R, g.H - y0);
    float *gxb = gx + (int64_t)b * D7_CI * HW;
    for (int r = wave; r < nrows; r += 4) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4_d7 acc{0.f, 0.f, 0.f, 0.f};
            const int xl = 16 * nt + (lane & 15);                // pixel within the segment; tile column of j = 0 is xl
            const unsigned cbase = (unsigned)(2 * par) * TSEL + (unsigned)(r * TWP + xl - par);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const char *bp = reinterpret_cast<const char *>(thin) + (cbase + boff[s]) * 2u;
                u32x4_d7 bh, bl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[j] = *reinterpret_cast<const unsigned *>(bp + 4 * j);
                    bl[j] = *reinterpret_cast<const unsigned *>(bp + TSEL * 2 + 4 * j);
                }
                const bf16x8 b_h = __builtin_bit_cast(bf16x8, bh), b_l = __builtin_bit_cast(bf16x8, bl);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s], b_h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s], b_l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s], b_h, acc, 0, 0, 0);
            }
            // C[ci = 4 kg + j][pixel lane & 15]
            const int y = y0 + r, xg = x0 + xl;
            if (xg < g.W) {
#pragma unroll
                for (int j = 0; j < 4; ++j) gxb[(int64_t)(4 * kg + j) * HW + (int64_t)y * g.W + xg] = acc[j];
            }
        }
    }
}

// grad_input [B, 16, H, W] of a 7x7 stride-1 layer with 3 output channels from grad_output (times act'(saved_output) when given)
int launch_conv7_thin_dgrad(hipStream_t st, const float *go, const float *y, const float *w, float *gx, int B, int H, int W, int Ho, int Wo,
                            int pad, int act, float slope) {
    D7Geom g{B, H, W, Ho, Wo, pad, act, slope, (int)ceil_div(H, D7_R), (int)ceil_div(W, D7_W)};
    const int64_t wgs = (int64_t)B * g.bands * g.segs;
    if (wgs > 2147483647LL) return fail(EBFI_ERR_ARG, "conv7_thin_dgrad: too many workgroups");
    const double flops = 2.0 * B * Ho * Wo * 3.0 * 16.0 * 49.0;
    const double bytes = 4.0 * ((double)B * 3 * Ho * Wo * (y ? 2 : 1) + (double)B * 16 * H * W);
    ProfScope ps("conv7_thin_dgrad", st, flops, bytes);
    hipLaunchKernelGGL(conv7_thin_dgrad, dim3((unsigned)wgs), dim3(256), 0, st, go, y, w, gx, g);
    return check_launch("conv7_thin_dgrad");
}

// Single-product fp16 forms of the wave-specialised 3x3 kernels (included by conv2d.hip inside its anonymous namespace).
//
// Why: the split-precision kernels carry every operand as a bf16 pair and issue three MFMAs per product so that the
// FORWARD pass agrees with fp32 to ~1e-5 -- which the training step needs, because the L1 / census loss terms turn forward
// errors into sign flips of the gradient (measured with the oracle: fp16-rounded forward operands move the packed gradient
// by 9e-3 in norm, split-precision ones by 1e-3).  The BACKWARD products have no such amplifier: rounding grad_output,
// weights and saved inputs to fp16 (11-bit significand, 8x finer than bf16) moves the packed gradient from 0.95e-3 to
// 1.09e-3 ("precision of the backward products").  So the data gradient and the weight gradient run
// ONE v_mfma_f32_32x32x16_f16 per product on ONE 16-bit image per operand: a third of the matrix work, half the LDS
// bytes and a third of the conversion instructions of the split form.
//
// Range: fp16 spans 2^-24 .. 65504, and with the reference's x0.1 initialisation gradients fall to 1e-29 in the early
// layers.  Every operand is therefore multiplied by a POWER OF TWO on its way into LDS (exact in fp32) and the accumulator
// is multiplied by the inverse product on its way out.  The scale of an operand lives in a device slot
// {scale, running |max|}: the staging code records |max| of what it reads (it touches every element anyway), and one tiny
// kernel per step (ebfi_f16_scales_finish) turns the maxima into the next step's scales (max * scale in [2, 4): 14 binades
// of headroom above; below, fp16 is normal for 15 more and the MFMA takes subnormal operands -- measured on a 64 -> 64 layer,
// the weight gradient's error stays at 2.7e-4 until max * scale drops under 2^-10) and raises a flag if a value could have overflowed
// -- delayed scaling, as used for fp8 training; the first use of a slot is calibrated just in time by the host side
// (ebfi_amd/f16scale.py).

// (f16 vector types, pack_f16, saturate_fp16_conversions and ScaleSlot live in c16.hpp: the split-precision forward kernels
// and the fused stages write fp16 side images too)

// ------------------------------------------------------------------------------------------------
// conv_fwd_f16_ws: the wave-specialised 3x3 forward / data gradient (64 output channels per workgroup, 8 rows x 64 px
// tiles, 16-channel chunks, persistent tile walk: see conv_fwd_bf16x3_ws) with fp16 operand images.  With a third of the
// matrix work per chunk the kernel is paced by the arrival of its staging loads, so the roles are re-balanced: FOUR
// consumer waves (one per SIMD, two output rows each: an A fragment serves two rows) and four producer waves that, at 256
// registers per wave, hold TWO chunks of global loads in flight (the split-precision form had room for one).  Used as the
// DATA GRADIENT (transposed weight images) of the training step.
//   x       [B, groups*Cin, H, W] fp32, multiplied by in_slot.scale() while it is converted; |max| recorded into in_slot
//   wp      fp16 image [tap][Cout][K16], already scaled by w_slot[0] (the pack launch applied and recorded it)
//   out     act(acc / (in_scale * w_scale) + bias + addend) * act'(mask_y)
constexpr int NTF16 = 512;
//   IN16 (round 4): x is NOT fp32 NCHW but the scaled fp16 image of the tensor in the c16 layout (c16.hpp), written by its
//           producer with in_slot's scale: the staging becomes a plain 16-byte copy (6 loads + 6 LDS stores per thread and
//           chunk instead of 24 + 8 conversions + 12), half the bytes; in_slot is only read here (the writer recorded |max|)
//   epi.out16: the output additionally / only (out == NULL) as a c16 fp16 image for the next backward kernel
//   INP16:  x is a PLANAR fp16 tensor [B, groups*Cin, H, W] scaled by in_slot's scale (the grad_kernel of the FAC op, written by
//           fac_bwd_rows_f32<.., H16>): the quad staging of the fp32 form with 8-byte loads and byte permutes instead of
//           conversions, half the bytes; in_slot is only read
//   FAC (round 6; inference, with EXTRA = INP16 = false): the output rows are the per-pixel 5x5 filters of the filter-adaptive
//           convolution that follows (weight rows in the "facrows" layout: one FAC channel per 32-row tile); the epilogue applies
//           them to `fac.ev` and stores ONE value per pixel and channel (fac_epilogue_tile, conv2d.hip): the fused
//           KernelConv -> FAC kernel of conv_fwd_bf16x3_ws<0, true> at one matrix-core product per tap instead of three
//   ST (round 6, with EXTRA): the epilogue is built with the shuffled output layouts of ConvGeom::store (store_out_tile XM bit 3)
template <bool EXTRA, bool IN16 = false, bool INP16 = false, bool FAC = false, bool ST = false>
__global__ __launch_bounds__(NTF16) void conv_fwd_f16_ws(const float *__restrict__ x, const _Float16 *__restrict__ wp,
                                                         const float *__restrict__ bias, float *__restrict__ out, ConvGeom g, int K16,
                                                         int act, float slope, EpiExtra epi, int tiles_total, ScaleSlot in_slot,
                                                         const float *__restrict__ w_slot, FacEpi fac) {
    static_assert(!FAC || (!EXTRA && !INP16), "the FAC epilogue comes without epilogue extras (input: fp32 planes or a c16 image)");
    // (MODE.FP16_OVFL: the producers, which convert, set it for good below; the consumers only around their epilogue -- while it is
    //  set the matrix cores drop non-finite operands, c16.hpp)
    constexpr int KS = 3, KK = 9, MT = 2, RW = 2;              // RW: output rows per consumer wave
    constexpr int IH = TYB - 1 + KS, IW = TX - 1 + KS, PS = IH * IW, COS = 32 * MT;
    constexpr int NCW = TYB / RW, PT = 256;
    constexpr int WPIECES = KK * COS * 2, NWB = (WPIECES + PT - 1) / PT;
    constexpr int INB = PS * 32, WB = KK * COS * 32, BUFB = INB + WB;
    static_assert(NCW == 4 && NTF16 == 64 * NCW + PT, "wave roles");
    extern __shared__ __attribute__((aligned(16))) char smd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (g.Wo + TX - 1) / TX, tiles_y = (g.Ho + TYB - 1) / TYB;
    const int co_base = blockIdx.y * COS;
    const int grp = co_base / (g.Cout / g.groups);
    const int HW = g.H * g.W;
    constexpr unsigned ES = INP16 ? 2u : 4u;                        // bytes per input element of the quad-staging path
    const unsigned plane_bytes = (unsigned)HW * ES, x_bytes = (unsigned)g.Cin * plane_bytes;
    const int nchunks = K16 / CKB;
    const int G = gridDim.x;
    int ntiles_mine = 0;
    for (int t = blockIdx.x; t < tiles_total; t += G) ++ntiles_mine;
    const int nitems = ntiles_mine * nchunks;
    constexpr int NSTG = IN16 ? 3 : 2;                             // register stages of the producers = phases per round of their loop
    const int nitems_pad = (nitems + NSTG - 1) / NSTG * NSTG;
    const bool xcd_map = (gridDim.x & 7) == 0;
    auto tile_coords = [&](int tt, int &tb, int &ty0, int &tx0) {
        int u = xcd_map ? xcd_tile(tt, tiles_total) : tt;          // (neighbouring tiles on one XCD: shared halo lines hit its L2)
        const int txi = u % tiles_x; u /= tiles_x;
        const int tyi = u % tiles_y;
        tb = u / tiles_y; ty0 = tyi * TYB; tx0 = txi * TX;
    };
    const float sx = in_slot.scale();
    [[maybe_unused]] constexpr int KB_LDS_OFF = 2 * BUFB;
    KB_CLEAR_SELF();
    KB_STAMP(0);

    if (wave < NCW) {
        // static priority for the matrix-issuing waves: VALU issue is arbitrated by priority, then age (MI355X_MICROARCH.md, "two
        // waves per SIMD"); with the consumers above the converting producer wave of their SIMD the kernel measured 0.5-0.7 % faster,
        // with the producers above the consumers 2 % slower (same box, round 5)
        __builtin_amdgcn_s_setprio(1);
        // ------------------------------------------------------------------ consumers: output rows 2*wave, 2*wave + 1
        f32x16 acc[RW][MT][2];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][m][n][e] = 0.f;
        const float oscale = 1.f / (sx * (w_slot ? w_slot[0] : 1.f));
        [[maybe_unused]] float amax16 = 0.f;
        const int hsel = lane >> 5, l31 = lane & 31;
        const int a_lane = INB + l31 * 32 + ((hsel ^ ((l31 >> 3) & 1)) << 4);
        const int pbase = RW * wave * IW + l31;
        // bit (r*9 + tap): which 16-byte half of position (pbase + r*IW + ky*IW + kx) this lane's k-half lives in
        unsigned fbits = 0;
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int tap = 0; tap < KK; ++tap)
                fbits |= (unsigned)((((pbase + r * IW + (tap / KS) * IW + (tap % KS)) >> 3) & 1) ^ hsel) << (r * KK + tap);
        f16x8 af[2][MT], bf[2][RW][2];
        auto tap_read = [&](const char *base, int tap, int set) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const char *ap = base + a_lane + tap * COS * 32;
#pragma unroll
            for (int m = 0; m < MT; ++m) af[set][m] = *reinterpret_cast<const f16x8 *>(ap + m * 1024);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const char *bp = base + (pbase + r * IW) * 32 + (int)(((fbits >> (r * KK + tap)) & 1u) << 4) + (ky * IW + kx) * 32;
#pragma unroll
                for (int n = 0; n < 2; ++n) bf[set][r][n] = *reinterpret_cast<const f16x8 *>(bp + n * 1024);
            }
        };
        auto tap_mfma = [&](int set) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[r][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[set][m], bf[set][r][n], acc[r][m][n], 0, 0, 0);
        };
        __syncthreads();                   // (A) the first chunk is committed
        KB_STAMP(1);
        int item = 0, tcur = blockIdx.x;
        for (int ti = 0; ti < ntiles_mine; ++ti, tcur += G) {
            for (int chunk = 0; chunk < nchunks; ++chunk, ++item) {
                const char *base = smd + (item & 1) * BUFB;
                if (item < 12) KB_STAMP(2 + 2 * item);
                tap_read(base, 0, 0);
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) {
                    if (tap + 1 < KK) tap_read(base, tap + 1, (tap + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    tap_mfma(tap & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (item < 12) KB_STAMP(3 + 2 * item);
                __syncthreads();           // (B) this buffer has been read, the other one is complete
            }
            KB_STAMP(30);
            int cb_, cy0, cx0;
            tile_coords(tcur, cb_, cy0, cx0);
            // (written out per row: as a loop the EXTRA variant was not unrolled, `acc[r]` became a runtime index and the whole
            // accumulator array moved to scratch memory -- 410 instead of 45 us per launch)
            static_assert(RW == 2, "epilogue is written out for two rows");
            if constexpr (FAC) {
                fac_epilogue_tile<MT>(out, bias, acc[0], g, fac, cb_, co_base, cy0 + RW * wave, cx0, lane, slope, oscale);
                fac_epilogue_tile<MT>(out, bias, acc[1], g, fac, cb_, co_base, cy0 + RW * wave + 1, cx0, lane, slope, oscale);
            } else {
                if constexpr (EXTRA) saturate_fp16_conversions(true);      // (the epilogue's fp16 image stores; never across the MFMA loop)
                store_out_tile<MT, (EXTRA ? 3 : 0) | (ST ? 8 : 0)>(out, bias, acc[0], g, cb_, co_base, cy0 + RW * wave, cx0, lane, act, slope, epi, oscale, &amax16);
                store_out_tile<MT, (EXTRA ? 3 : 0) | (ST ? 8 : 0)>(out, bias, acc[1], g, cb_, co_base, cy0 + RW * wave + 1, cx0, lane, act, slope, epi, oscale, &amax16);
                if constexpr (EXTRA) saturate_fp16_conversions(false);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[r][m][n][e] = 0.f;
        }
        // The producers walk the items in groups of NSTG phases WITHOUT early exits (see there): the trailing dead phases' barriers
        // are matched here.
        for (int k = nitems; k < nitems_pad; ++k) __syncthreads();
        KB_STAMP(31);
        KB_FLUSH_SELF();
        if constexpr (EXTRA) {
            if (epi.out16 != nullptr) ScaleSlot{epi.slot16}.record(amax16);
        }
        return;
    }
    // ---------------------------------------------------------------------- producers
    saturate_fp16_conversions();           // (these waves convert and issue no MFMA)
    // 16-byte quads of 4 consecutive pixels x 8 channels (as conv_fwd_bf16x3_ws) and the chunk's weight pieces; two register
    // stages, so the loads of items i+2 and i+3 are in flight while item i+1 is converted and written.
    //
    // Round 5, read off the ISA: the rotating-stage loops below used to leave through `break` after any phase.  The structurizer
    // merges such exits into the loop latch, so in the control-flow graph every phase's end is a predecessor of the loop header;
    // the wait-count pass then has to assume that the stage committed FIRST in the loop body was issued LAST -- and emitted
    // s_waitcnt vmcnt(10) .. vmcnt(0) there: once per round of stages the producers waited for EVERY load in flight, the two
    // younger stages included.  The loops now run whole rounds (dead phases past the end stage empty descriptors and write a
    // buffer nobody reads; the consumers match their barriers), and every commit opens with an explicit s_waitcnt for exactly
    // its own stage, so that neither the guarded stores nor the re-use of a destination register as an address temporary makes
    // the compiler insert a wider wait.
    KB_STAMP(20);
    const int ptid = tid - 64 * NCW;
    const unsigned img_bytes = (unsigned)KK * (unsigned)g.Cout * (unsigned)K16 * 2u;
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(wp), 0, img_bytes, 0x00020000);
    if constexpr (IN16) {
        // ---- fp16 c16 input: 16-byte pieces (position, 8-channel half); a tile row is two contiguous runs of 66 pieces
        constexpr int NPIECE = PS * 2, NPK = (NPIECE + PT - 1) / PT;
        // piece id = (tile row, channel half, tile column): consecutive lanes walk the columns of one half-row run (c16.hpp)
        int p_row[NPK], p_col[NPK], p_half[NPK], p_dst[NPK];
#pragma unroll
        for (int k = 0; k < NPK; ++k) {
            const int id = ptid + k * PT;
            const int t = id / IW;
            p_col[k] = id - t * IW;
            p_half[k] = t & 1;
            p_row[k] = t >> 1;
            const int pos = p_row[k] * IW + p_col[k];
            p_dst[k] = pos * 32 + ((p_half[k] ^ ((pos >> 3) & 1)) << 4);
        }
        unsigned w_off[NWB];
        int w_dst[NWB];
#pragma unroll
        for (int it = 0; it < NWB; ++it) {
            const int j = ptid + it * PT;
            const int row = j >> 1, half = j & 1;
            const int tap = row / COS, co = row - tap * COS;
            w_off[it] = j < WPIECES ? (unsigned)(((tap * g.Cout + co_base + co) * K16 + half * 8) * 2) : SENT;
            w_dst[it] = INB + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
        }
        struct Stage16 {
            u32x4 rq[NPK];
            u32x4 rw[NWB];
        };
        Stage16 sa, sb, sc;                    // THREE chunks of loads in flight (10 registers of payload per chunk and thread)
        const unsigned blk_bytes = (unsigned)HW * 32u;                // one 16-channel block of one sample
        const _Float16 *x16 = reinterpret_cast<const _Float16 *>(x);
        const int cbg = g.Cin >> 4;                                   // 16-channel blocks per group (= chunks)
        int pf_tile = blockIdx.x, pf_chunk = 0;
        unsigned pf_off[NPK];
        const _Float16 *pf_src = x16;
        unsigned pf_bytes = 0u;
        auto pf_setup = [&]() {
            const bool live = pf_tile < tiles_total;
            int tb, ty0, tx0;
            tile_coords(live ? pf_tile : 0, tb, ty0, tx0);
#pragma unroll
            for (int k = 0; k < NPK; ++k) {
                const int yy = ty0 - g.pad + p_row[k], xx = tx0 - g.pad + p_col[k];
                const bool ok = live && ptid + k * PT < NPIECE && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                pf_off[k] = ok ? (unsigned)((yy * 2 + p_half[k]) * g.W + xx) * 16u : SENT;
            }
            pf_src = x16 + ((int64_t)tb * g.groups + grp) * cbg * HW * 16;
            pf_bytes = live ? (unsigned)cbg * blk_bytes : 0u;
        };
        auto prefetch = [&](Stage16 &s) {
            const uint64_t pa = reinterpret_cast<uint64_t>(pf_src);
            const uint64_t pu = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) |
                                (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<_Float16 *>(pu), 0, (unsigned)__builtin_amdgcn_readfirstlane(pf_bytes), 0x00020000);
            const unsigned cb = (unsigned)__builtin_amdgcn_readfirstlane(pf_chunk) * blk_bytes;
#pragma unroll
            for (int k = 0; k < NPK; ++k) s.rq[k] = __builtin_amdgcn_raw_buffer_load_b128(r, pf_off[k] + cb, 0, 0);
            const unsigned wb = (unsigned)__builtin_amdgcn_readfirstlane(pf_chunk) * (unsigned)(CKB * 2);
#pragma unroll
            for (int it = 0; it < NWB; ++it) s.rw[it] = __builtin_amdgcn_raw_buffer_load_b128(rwt, w_off[it] + wb, 0, 0);
            if (++pf_chunk == nchunks) {
                pf_chunk = 0;
                pf_tile += G;
                pf_setup();
            }
        };
        auto commit = [&](int buf, Stage16 &s) {
            char *base = smd + buf * BUFB;
            // this stage's NPK + NWB loads are the oldest in flight, two younger stages behind them
            wait_vmcnt<2 * (NPK + NWB)>();
#pragma unroll
            for (int k = 0; k < NPK; ++k)
                if (ptid + k * PT < NPIECE) *reinterpret_cast<u32x4 *>(base + p_dst[k]) = s.rq[k];
#pragma unroll
            for (int it = 0; it < NWB; ++it)
                if (ptid + it * PT < WPIECES) *reinterpret_cast<u32x4 *>(base + w_dst[it]) = s.rw[it];
        };
        pf_setup();
        prefetch(sa);                          // item 0
        prefetch(sb);                          // item 1
        prefetch(sc);                          // item 2
        commit(0, sa);
        __syncthreads();                       // (A)
        prefetch(sa);                          // item 3
        // steady state: before barrier (B) of item i the producers commit item i + 1 and request item i + 4; the stages rotate
        // sb -> sc -> sa (written out three times: a runtime stage index would move the registers to scratch)
        for (int item = 0; item < nitems_pad; item += 3) {
            commit((item + 1) & 1, sb);
            prefetch(sb);
            __syncthreads();                   // (B) item
            commit(item & 1, sc);
            prefetch(sc);
            __syncthreads();                   // (B) item + 1
            commit((item + 1) & 1, sa);
            prefetch(sa);
            __syncthreads();                   // (B) item + 2
        }
        return;
    }
    constexpr int SH = (4 - (KS / 2) % 4) % 4;
    constexpr int NQ = (IW + SH + 3) / 4;
    constexpr int NITEM = IH * NQ * 2;
    constexpr int NIT = (NITEM + PT - 1) / PT;
    int it_qh[NIT], it_qr[NIT], it_qq[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int id = ptid + k * PT;
        it_qh[k] = id & 1;
        it_qr[k] = (id >> 1) / NQ;
        it_qq[k] = (id >> 1) - it_qr[k] * NQ;
    }
    KB_STAMP(21);
    unsigned w_off[NWB];
    int w_dst[NWB];
#pragma unroll
    for (int it = 0; it < NWB; ++it) {
        const int j = ptid + it * PT;
        const int row = j >> 1, half = j & 1;
        const int tap = row / COS, co = row - tap * COS;
        w_off[it] = j < WPIECES ? (unsigned)(((tap * g.Cout + co_base + co) * K16 + half * 8) * 2) : SENT;
        w_dst[it] = INB + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    using QV = std::conditional_t<INP16, u32x2, u32x4>;            // 4 pixels of one channel: 8 or 16 bytes
    struct Stage {
        QV rq[NIT][8];
        u32x4 rw[NWB];
    };
    Stage sa, sb;
    int pf_tile = blockIdx.x, pf_chunk = 0;
    unsigned pf_off[NIT];
    const float *pf_src = x;
    unsigned pf_bytes = 0u;
    float amax = 0.f;
    auto pf_setup = [&]() {
        const bool live = pf_tile < tiles_total;
        int tb, ty0, tx0;
        tile_coords(live ? pf_tile : 0, tb, ty0, tx0);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int yy = ty0 - g.pad + it_qr[k], xq = tx0 - g.pad - SH + 4 * it_qq[k];
            const bool ok = live && ptid + k * PT < NITEM && yy >= 0 && yy < g.H && xq >= 0 && xq + 3 < g.W;
            pf_off[k] = ok ? (unsigned)(yy * g.W + xq) * ES + (unsigned)(8 * it_qh[k]) * plane_bytes : SENT;
        }
        // (INP16: `x` points at halves: the sample offset in bytes is half the fp32 one)
        pf_src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + ((int64_t)tb * g.groups + grp) * g.Cin * HW * (int64_t)ES);
        pf_bytes = live ? x_bytes : 0u;
    };
    auto prefetch = [&](Stage &s) {
        // the descriptor words are wave-uniform, but with two call sites the compiler no longer proves it and wraps every load
        // in a waterfall loop (measured: 214 instead of ~60 us per launch): make the uniformity explicit
        const uint64_t pa = reinterpret_cast<uint64_t>(pf_src);
        // (the builtin returns a signed int: widen through unsigned, or a low half with bit 31 set smears into the high half)
        const uint64_t pu = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) |
                            (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa);
        const __amdgpu_buffer_rsrc_t r = make_rsrc(reinterpret_cast<const float *>(pu), (unsigned)__builtin_amdgcn_readfirstlane(pf_bytes));
        const unsigned cb = (unsigned)__builtin_amdgcn_readfirstlane(pf_chunk) * (unsigned)CKB * plane_bytes;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if constexpr (INP16) s.rq[k][c] = __builtin_amdgcn_raw_buffer_load_b64(r, pf_off[k] + cb + (unsigned)c * plane_bytes, 0, 0);
                else s.rq[k][c] = __builtin_amdgcn_raw_buffer_load_b128(r, pf_off[k] + cb + (unsigned)c * plane_bytes, 0, 0);
            }
        const unsigned wb = (unsigned)__builtin_amdgcn_readfirstlane(pf_chunk) * (unsigned)(CKB * 2);
#ifndef KB_NO_WLOAD      // (timing ablation of tools/kbench only: results are wrong without the weight pieces)
#pragma unroll
        for (int it = 0; it < NWB; ++it) s.rw[it] = __builtin_amdgcn_raw_buffer_load_b128(rwt, w_off[it] + wb, 0, 0);
#else
#pragma unroll
        for (int it = 0; it < NWB; ++it) s.rw[it] = u32x4{0u, 0u, 0u, 0u};
        (void)wb;
#endif
        if (++pf_chunk == nchunks) {
            pf_chunk = 0;
            pf_tile += G;
            pf_setup();
        }
    };
    auto commit = [&](int buf, Stage &s) {
        char *base = smd + buf * BUFB;
        wait_vmcnt<NIT * 8 + NWB>();           // this stage's loads are the oldest in flight, one younger stage behind them
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (ptid + k * PT < NITEM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * it_qq[k] + j - SH;
                    if (c < 0 || c >= IW) continue;
                    u32x4 hv;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        if constexpr (INP16) {
                            // half (j & 1) of dword (j >> 1) of channels e, e + 1 -> one word (e in the low half)
                            hv[e >> 1] = __builtin_amdgcn_perm(s.rq[k][e + 1][j >> 1], s.rq[k][e][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
                        } else {
                            const float v0 = __uint_as_float(s.rq[k][e][j]), v1 = __uint_as_float(s.rq[k][e + 1][j]);
                            amax = amax_acc(amax, v0, v1);
                            hv[e >> 1] = pack_f16(v0 * sx, v1 * sx);
                        }
                    }
                    const int pos = it_qr[k] * IW + c;
                    const int d = pos * 32 + ((it_qh[k] ^ ((pos >> 3) & 1)) << 4);
                    *reinterpret_cast<u32x4 *>(base + d) = hv;
                }
            }
#pragma unroll
        for (int it = 0; it < NWB; ++it)
            if (ptid + it * PT < WPIECES) *reinterpret_cast<u32x4 *>(base + w_dst[it]) = s.rw[it];
    };
    // Order of the start-up (in-kernel stamps, tools/kbench f16): the loads are issued at the rate the memory system accepts them
    // (60 KB per workgroup and chunk: issuing two chunks took 5.7 us), so nothing but item 0's commit may stand between the
    // first loads and barrier (A) -- the consumers multiply item 0 while the later chunks are being requested.
    KB_STAMP(22);
    pf_setup();
    KB_STAMP(23);
    prefetch(sa);                          // item 0
    KB_STAMP(1);
    prefetch(sb);                          // item 1 (requested before item 0 is converted: its data arrives behind item 0's)
    commit(0, sa);
    KB_STAMP(2);
    __syncthreads();                       // (A)
    prefetch(sa);                          // item 2: two chunks of loads in flight from here on
    KB_STAMP(3);
    for (int item = 0; item < nitems_pad; item += 2) {
        if (item < 12) KB_STAMP(4 + 2 * item);
        commit((item + 1) & 1, sb);        // item + 1, while the consumers multiply item
        if (item < 12) KB_STAMP(5 + 2 * item);
        prefetch(sb);                      // item + 3 (past the end: empty descriptors, nothing is read)
        __syncthreads();                   // (B)
        if (item < 12) KB_STAMP(6 + 2 * item);
        commit(item & 1, sa);              // item + 2
        if (item < 12) KB_STAMP(7 + 2 * item);
        prefetch(sa);                      // item + 4
        __syncthreads();                   // (B)
    }
    KB_STAMP(31);
    KB_FLUSH_SELF();
    // (halo positions are read by several workgroups, padded / dead lanes contribute 0: the maximum is unaffected)
    if constexpr (!INP16) in_slot.record(amax);       // (an fp16 input was recorded by its writer)
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_f16_ws: conv_wgrad_x3_ws (3x3 weight gradient, 64 co x 64 ci per workgroup, 2 x 32-pixel contraction tiles,
// four producer + four consumer waves, split-K slabs) with fp16 operands.  The GEMM contracts over PIXELS, so a lane's 8
// k-slots are 8 consecutive pixels and a tap's kx shift moves them by ONE element: the images therefore stay one pixel per
// 32-bit LDS word (any shift is word-aligned), and the two halves of a word now carry the SAME pixel of channels c and
// c + 32 instead of the hi / lo halves of one value.  One 8-word operand read then yields the fragments of TWO output
// tiles (grad_out: both 32-row tiles; input: the n-tiles of channel blocks 0 and 1) through the same two `v_perm_b32` per
// word pair that used to peel hi from lo: half the LDS words written and read, a third of the MFMAs, and the staging
// code converts with one v_cvt_pk_f16_f32 per word instead of two roundings and a subtraction per value.
//   36 blocks of 32 x 32 per workgroup = 9 pair-column tiles x {c, c+32} x 2 row tiles; wave w owns pair tiles w and
//   w + 4 completely (8 blocks) and block (row tile w & 1, half w >> 1) of pair tile 8.
// Scales: grad_out * g_slot.scale(), input * x_slot.scale() on the way into LDS (|max| of both recorded), the accumulators
// leave multiplied by 1 / (g_scale * x_scale); bias sums and the grad * act' side output stay exact fp32.
template <int DACT>
__global__ __launch_bounds__(512) void conv_wgrad_f16_ws(const float *__restrict__ x, const float *__restrict__ gout,
                                                         const float *__restrict__ yact, float *__restrict__ slab,
                                                         float *__restrict__ gpre_out, ConvGeom g, float dslope, int total_tiles,
                                                         int need_bias, ScaleSlot x_slot, ScaleSlot g_slot) {
    using C = WCfg<3, 1, 32>;
    constexpr int KS = 3, KK = 9, WTX = C::WTX, IH = C::IH, IW = C::IW;
    constexpr int IWP = 32, EXC = IW - IWP, CIB = 64, CP = CIB / 2, PS = C::PS, IWS = C::IWS;
    constexpr int PT = 256, NW64 = PT / 64;
    constexpr int TROWS = PT / IWP, CPR = CIB / TROWS, NI = IH * CPR, NG = 64 / NW64;
    constexpr int NEX = (EXC * IH * CIB + PT - 1) / PT;
    constexpr int NQ = 4;                                   // consumer waves
    constexpr int NPT = (CP * KK + 31) / 32;                // pair-column tiles of 32 (9)
    constexpr int BUF = 32 * GS + (CP + 1) * PS;            // words per buffer: 32 grad_out pair rows, 32 pair planes + a zero plane
    static_assert(WTX == 32 && EXC == 2 && CIB % TROWS == 0 && NPT == 9 && CPR == 8 && NG == 16, "tile configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned smw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = blockIdx.y * 64, ci_base = blockIdx.z * CIB;
    const int grp = co_base / (g.Cout / g.groups);
    const int ci_cnt = min(CIB, g.Cin - ci_base);
    const int tiles_x = (g.Wo + WTX - 1) / WTX, tiles_y = (g.Ho + WTY - 1) / WTY;
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    const int G = gridDim.x;
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    float *my = slab + (int64_t)blockIdx.x * (wsz + g.Cout);
    const float sx = x_slot.scale(), sg = g_slot.scale();

    for (int i = tid; i < 2 * BUF; i += 512) smw[i] = 0u;   // pad slots, pad columns and the zero plane stay zero
    __syncthreads();

    if (wave < NQ) {
        // ------------------------------------------------------------------ consumers
        const int nq = wave;
        const int xm = nq & 1, xh = nq >> 1;
        f32x16 acc[2][2][2], accx;                          // [pair tile nq + 4 j][half: channel c / c + 32][row tile m]
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][hf][m][r] = 0.f;
        auto col_off = [&](int n) {                          // pair column n = cp * 9 + tap
            const int cp = n / KK, tap = n - cp * KK;
            const int ky = tap / KS, kx = tap - ky * KS;
            return 32 * GS + ((cp < CP) ? cp * PS + ky * IWS + kx : CP * PS) + 8 * (lane >> 5);
        };
        int boff[3];
        boff[0] = col_off(nq * 32 + (lane & 31));
        boff[1] = col_off((nq + NQ) * 32 + (lane & 31));
        boff[2] = col_off(8 * 32 + (lane & 31));
        const int aoff = (lane & 31) * GS + 8 * (lane >> 5);
        // low halves = channel c (row tile 0 / channel block 0), high halves = channel c + 32
        auto unzip = [&](const unsigned (&w)[8], f16x8 &c_lo, f16x8 &c_hi) {
            u32x4 hq, lq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hq[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);
                lq[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);
            }
            c_hi = __builtin_bit_cast(f16x8, hq);
            c_lo = __builtin_bit_cast(f16x8, lq);
        };
        __syncthreads();                   // (A) the first tile is committed
        int cur = 0;
        for (int tile = blockIdx.x; tile < total_tiles; tile += G) {
            const unsigned *sA = smw + cur * BUF + aoff;
            const unsigned *sB = smw + cur * BUF;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int row = kk >> 1, px0 = (kk & 1) * 16;
                unsigned aw[8], bw[3][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) aw[j] = sA[row * 32 + px0 + j];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bw[q][j] = sB[boff[q] + row * IWS + px0 + j];
                f16x8 a[2];
                unzip(aw, a[0], a[1]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f16x8 b[2];
                    unzip(bw[q], b[0], b[1]);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            acc[q][hf][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[hf], acc[q][hf][m], 0, 0, 0);
                }
                {
                    f16x8 b[2];
                    unzip(bw[2], b[0], b[1]);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm ? a[1] : a[0], xh ? b[1] : b[0], accx, 0, 0, 0);
                }
            }
            __syncthreads();               // (B) this image has been read, the other one is complete
            cur ^= 1;
        }
        const float oscale = 1.f / (sx * sg);
        const __amdgpu_buffer_rsrc_t rsl = make_rsrc(my, (unsigned)wsz * 4u);   // rows co >= Cout fall past the slab: dropped
        const unsigned co_row = (unsigned)(g.Cin * KK) * 4u;
        auto store_block = [&](const f32x16 &a, int m, int ptile, int hf) {
            const int n = ptile * 32 + (lane & 31);            // pair column: cp * 9 + tap
            const bool ok = n < CP * KK && n / KK + CP * hf < ci_cnt;
            const unsigned o0 = ok ? (unsigned)(((co_base + m * 32 + 4 * (lane >> 5)) * g.Cin + ci_base + CP * hf) * KK + n) * 4u : SENT;
#pragma unroll
            for (int r = 0; r < 16; ++r) buf_st(rsl, o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * co_row, a[r] * oscale);
        };
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int m = 0; m < 2; ++m) store_block(acc[j][hf][m], m, nq + NQ * j, hf);
        store_block(accx, xm, 8, xh);
        return;
    }
    // ---------------------------------------------------------------------- producers
    saturate_fp16_conversions();           // (MODE.FP16_OVFL in the converting waves only: under it the matrix cores drop non-finite operands, c16.hpp)
    const int ptid = tid - 64 * NQ;
    const int gslot = ptid & 63, gpy = gslot >> 5, gpx = gslot & 31, gco = ptid >> 6;
    const int icol = ptid & (IWP - 1), irow = ptid / IWP;
    const unsigned go_bytes = (unsigned)g.Cout * (unsigned)HWo * 4u, x_bytes = (unsigned)g.Cin * (unsigned)HW * 4u;
    struct Stage {
        float rg[NG], ry[DACT != 0 ? NG : 1], ri[NI], rex[NEX];
    };
    Stage sa, sb;
    float bacc[NG];
    float amax_g = 0.f, amax_x = 0.f;
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
        unsigned *sG = smw + buf * BUF, *sIn = sG + 32 * GS;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            if constexpr (DACT != 0) s.rg[it] *= act_grad_c<DACT>(s.ry[it], dslope);
            bacc[it] += s.rg[it];
            amax_g = amax_acc(amax_g, s.rg[it]);
        }
        // thread row gco holds channels gco + 4 it: `it` and `it + 8` are channels c and c + 32 of pair row gco + 4 it
#pragma unroll
        for (int it = 0; it < NG / 2; ++it)
            sG[(gco + NW64 * it) * GS + gslot] = pack_f16(s.rg[it] * sg, s.rg[it + NG / 2] * sg);
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
#pragma unroll
        for (int i = 0; i < NI; ++i) amax_x = amax_acc(amax_x, s.ri[i]);
        if (icol < IW) {
            // thread row irow holds channels irow + 8 k: k and k + 4 are channels c and c + 32 of pair plane irow + 8 k
#pragma unroll
            for (int r = 0; r < IH; ++r)
#pragma unroll
                for (int k = 0; k < CPR / 2; ++k)
                    sIn[(irow + k * TROWS) * PS + r * IWS + icol] = pack_f16(s.ri[r * CPR + k] * sx, s.ri[r * CPR + k + CPR / 2] * sx);
        }
#pragma unroll
        for (int i = 0; i < NEX; ++i) {
            // element e: channel e & 63; its pair partner (channel ^ 32) sits 32 lanes away in the same wave
            const int e = ptid + i * PT, rc = e / CIB, ch = e & (CIB - 1);
            const float v = s.rex[i];
            amax_x = amax_acc(amax_x, v);
            const float o = __shfl_xor(v, 32, 64);
            if (e < EXC * IH * CIB && ch < CP) sIn[ch * PS + (rc / EXC) * IWS + IWP + rc % EXC] = pack_f16(v * sx, o * sx);
        }
    };
    prefetch(blockIdx.x, sa);
    commit(blockIdx.x, 0, sa);
    prefetch(blockIdx.x + G, sa);          // two tiles of loads in flight from here on
    prefetch(blockIdx.x + 2 * G, sb);
    __syncthreads();                       // (A)
    int cur = 0;
    for (int tile = blockIdx.x; tile < total_tiles; tile += 2 * G) {
        commit(tile + G, cur ^ 1, sa);     // while the consumers multiply tile `tile` from image `cur`
        prefetch(tile + 3 * G, sa);        // (past the end: zero-record descriptors, nothing is read)
        __syncthreads();                   // (B)
        cur ^= 1;
        if (tile + G >= total_tiles) break;
        commit(tile + 2 * G, cur ^ 1, sb);
        prefetch(tile + 4 * G, sb);
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
    x_slot.record(amax_x);
    g_slot.record(amax_g);
}

// ------------------------------------------------------------------------------------------------
// Delayed scaling bookkeeping over slots 0 .. n-1 (SLOT_STRIDE floats each: scale at [0], |max| seen since the last call at [32]).  For every slot that saw
// data: flag[0] |= 1 when the data was not finite or max * scale could have left the fp16 range (the step's gradients are
// then suspect: the optimiser launch skips the update), the next scale puts max into [2, 4) (F16_TARGET_EXP), the maximum is cleared.
__global__ __launch_bounds__(256) void f16_scales_finish_kernel(float *__restrict__ slots, int n, int *__restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float *slot = slots + (int64_t)SLOT_STRIDE * i;
    const float a = slot[SLOT_AMAX], s = slot[0];
    if (!(a > 0.f)) {                      // nothing staged through this slot, or nothing above the floor: keep the scale
        if (a != a) atomicOr(flag, 1);
        slot[SLOT_AMAX] = 0.f;
        slot[SLOT_FLOOR] *= 0.5f;          // (decays fast: a tensor that shrank is measured again within a few steps)
        return;
    }
    if (!(a <= 3.0e38f) || a * s > 60000.f) atomicOr(flag, 1);
    if (a <= 3.0e38f) {
        int e;
        (void)frexpf(a, &e);               // a = m * 2^e, m in [0.5, 1)
        // (clamped like the host-side calibration, f16scale.calibrate: a recorded |max| below ~2^-125 must not turn the
        // next scale into +inf -- the x0.1 initialisation does produce 1e-29 gradients)
        slot[0] = ldexpf(1.f, min(F16_TARGET_EXP - e, 120));
        slot[SLOT_FLOOR] = 0.875f * a;     // next step: only waves above 7/8 of this maximum report (c16.hpp)
    } else {
        slot[SLOT_FLOOR] = 0.f;
    }
    slot[SLOT_AMAX] = 0.f;
}

// fp16 images from an index table (the bf16 pack launch's table format, hi entries only): packed[e] = fp16(src[table[e]] *
// scale(slot of e)), 0 for table[e] < 0.  Every site's image starts on a 256-element boundary, so a workgroup lies inside one
// image: block_slot[blockIdx.x] names its scale slot; |max| of the source values is recorded there (one atomic per wave).
__global__ __launch_bounds__(256) void pack_table_f16_kernel(const float *__restrict__ src, const int32_t *__restrict__ table,
                                                             int64_t n, _Float16 *__restrict__ out,
                                                             const int32_t *__restrict__ block_slot, float *__restrict__ slots) {
    saturate_fp16_conversions();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float *slot = slots + (int64_t)SLOT_STRIDE * block_slot[blockIdx.x];
    const int32_t t = e < n ? table[e] : -1;
    const float v = t >= 0 ? src[t & 0x3fffffff] : 0.f;
    if (e < n) out[e] = (_Float16)(v * slot[0]);
    // |max| over the wave on the float BITS (non-negative floats order like unsigned integers, and a NaN weight -- which
    // fmaxf would drop -- stays the largest value and reaches the slot: the finish launch then raises the guard)
    unsigned m = __float_as_uint(fabsf(v));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0 && !(__uint_as_float(m) <= fmaxf(__builtin_nontemporal_load(slot + SLOT_AMAX), slot[SLOT_FLOOR])))
        atomicMax(reinterpret_cast<unsigned *>(slot + SLOT_AMAX), m);
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_f16_tr: the 3x3 weight gradient on PIXEL-MAJOR fp16 images read with the transposing LDS load.
//
// The weight gradient contracts over pixels.  The kernels above therefore keep pixel-contiguous images ([channel][pixel],
// one pixel per 32-bit word so that a tap's one-pixel shift stays aligned), which forces dword-per-lane staging loads and
// a ds_write_b32 per word: conv_wgrad_f16_ws runs exactly as fast as the split-precision kernel although it issues a third of
// the MFMAs -- both are paced by their producers (~2.6 TB/s of 128-byte row segments through 50 load + 25-50 store
// instructions per thread and tile).  gfx950's ds_read_b64_tr_b16 removes the constraint: it delivers, to each lane, FOUR
// consecutive rows of ONE 16-bit column of a row-major LDS block -- a [pixel][channel] image is read as the
// [channel][8 pixels] fragment the MFMA wants.  With pixel-major images
//   * the staging is the forward kernel's: 16-byte global loads (4 pixels of one channel), eight of them packed into one
//     ds_write_b128 per pixel (8 channels): 24 loads + 12 stores per thread and 128-pixel tile;
//   * a tap shift is a ROW offset (128 bytes): no alignment problem, no shifted copies, no per-word permutes;
//   * both operands come from the same kind of image: A = grad_out^T and B = the shifted input, two transposing reads per
//     fragment, 14 reads for 9 MFMAs per consumer wave and k-step.
// Tile = 4 rows x 32 pixels (input halo 6 x 36 positions), 64 co x 64 ci per workgroup, 44 KB per LDS buffer, two register
// stages of loads in flight per producer thread.  Rows are 128 bytes (64 channels); the 16-byte chunk c of position p is
// stored at chunk c ^ 4*((p >> 1) & 1): the four rows of a transposing read then fall into four different bank quarters.
// Activation-free form only (pre-activation gradients: ResidualControl's fused node, KernelConv): the layers that fold
// act'(y) keep conv_wgrad_f16_ws.
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __fp16 hv8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
__device__ __forceinline__ f16x8 tr_frag(const char *lds_lo, const char *lds_hi) {
    typedef hv4 __attribute__((address_space(3))) *lp;
    const hv4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_lo));
    const hv4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_hi));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr int TRH = 4, TRW = 32, TRXR = TRH + 2, TRXQ = 10, TRXW = 36;      // tile rows / px, halo rows, quads and stored positions per row
constexpr int TR_XB = TRXR * TRXW * 128, TR_GB = TRH * TRW * 128, TR_BUFB = TR_XB + TR_GB;
constexpr int TR_LDS = 2 * TR_BUFB + 4 * 64 * 4;                             // two buffers + the bias scratch

// DACT != 0: grad_out is multiplied by act'(saved output) on the way in (the layer's own activation folded, as in the other
// weight-gradient kernels) and, with gpre_out, the product is written out for the data gradient; the saved-output quads take
// the registers of the second input stage, so this form keeps ONE tile of input loads in flight.
// IN16 (round 4, DACT == 0 only): x and gout are the scaled fp16 images of the two tensors in the c16 layout (c16.hpp): a
// [pixel][64 ch] LDS image is four 16-channel blocks whose tile rows are contiguous runs, so the producers copy 16-byte
// pieces (7 + 4 loads per thread and tile instead of 16 + 8 plus the conversions) and move half the bytes.
// GP16 (with IN16): grad_out is a PLANAR fp16 tensor [B, Cout, H, W] scaled by g_slot's scale instead of a c16 image (the
// grad_kernel the FAC backward writes plane by plane): staged like the fp32 form -- eight 8-byte loads of 4 pixels per thread,
// byte permutes into four (pixel, 8 channels) pieces.
// The body is shared by the one-layer launch (conv_wgrad_f16_tr) and the batched one (conv_wgrad_f16_tr_batch): the caller
// passes the workgroup's block (co_blk, ci_blk), its tile walk (first tile `split`, stride G, end total_tiles; xcd_map: the
// tile ids go through xcd_tile) and the slab `my` that receives its partial sums.
template <int DACT, bool IN16, bool GP16>
__device__ __forceinline__ void wgrad_tr_body(const float *__restrict__ x, const float *__restrict__ gout,
                                              const float *__restrict__ yact, float *__restrict__ gpre_out, float *__restrict__ my,
                                              const ConvGeom &g, float dslope, const int split, const int G, const int total_tiles,
                                              const bool xcd_map, int need_bias, ScaleSlot x_slot, ScaleSlot g_slot, const int co_blk,
                                              const int ci_blk, const int gpre_c16 = 0) {
    constexpr int KK = 9, PT = 256, NQ = 4;
    constexpr int TR_NSTG = IN16 ? 3 : 1;          // the image-reading producers run whole rounds of 3 phases (padding barriers below)
    constexpr bool WG_EXPLICIT_WAITS = false;
    constexpr int NXI = TRXR * TRXQ * 8, NXK = (NXI + PT - 1) / PT;          // input items (row, quad, 8-channel group): 480, 2 per thread
    static_assert(NXK == 2 && TRH * (TRW / 4) * 8 == PT, "one grad_out item and two input items per producer thread");
    extern __shared__ __attribute__((aligned(16))) char smt[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_base = co_blk * 64, ci_base = ci_blk * 64;
    const int grp = co_base / (g.Cout / g.groups);
    const int tiles_x = (g.Wo + TRW - 1) / TRW, tiles_y = (g.Ho + TRH - 1) / TRH;
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    const int64_t wsz = (int64_t)g.Cout * g.Cin * KK;
    const float sx = x_slot.scale(), sg = g_slot.scale();
    [[maybe_unused]] char *smd = smt;
    [[maybe_unused]] constexpr int KB_LDS_OFF = TR_LDS;
    KB_CLEAR_SELF();
    KB_STAMP(0);

    if (wave < NQ) {
        // ------------------------------------------------------------------ consumers
        // 36 blocks = 2 row tiles (co) x 18 column tiles (channel block cb, tap): wave w owns column tiles w, w+4, w+8, w+12
        // with both row tiles and row tile (w & 1) of column tile 16 + (w >> 1)
        const int nq = wave;
        const int xm = nq & 1, xt = 16 + (nq >> 1);
        f32x16 acc[4][2], accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][m][r] = 0.f;
        // lane geometry of the transposing read: 16-lane group G4 = k half (G4 >> 1) and 16-channel half (G4 & 1); inside a
        // group lane 4q + p addresses row q, columns 4p .. 4p + 3
        const int G4 = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int kh = G4 >> 1, cg = G4 & 1;
        const int rowoff = (8 * kh + q) * 128 + (p & 1) * 8;
        const int chunk0 = 2 * cg + (p >> 1);                    // 16-byte chunk inside a 32-channel half
        int aoff[2];                                             // grad_out image: pixel base is a multiple of 4 -> swizzle bit = q >> 1
#pragma unroll
        for (int m = 0; m < 2; ++m) aoff[m] = TR_XB + rowoff + (((4 * m + chunk0) ^ ((q >> 1) << 2)) << 4);
        int boff[5];                                             // input image: per column tile (cb, ky, kx)
        auto col_off = [&](int nt) {
            const int cb = nt / KK, tap = nt - cb * KK;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int swz = (((kx + 1 + q) >> 1) & 1) << 2;      // (position >> 1) & 1 of row q of the read (row base = 4 t + kx + 1)
            return (ky * TRXW + kx + 1) * 128 + rowoff + (((4 * cb + chunk0) ^ swz) << 4);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) boff[j] = col_off(nq + NQ * j);
        boff[4] = col_off(xt);
        __syncthreads();                   // (A) the first tile is committed
        KB_STAMP(1);
        int cur = 0;
        [[maybe_unused]] int kb_i = 0;
        int ntiles_done = 0;
        for (int tile = split; tile < total_tiles; tile += G, ++ntiles_done) {
            const char *base = smt + cur * TR_BUFB;
            if (kb_i < 12) KB_STAMP(2 + 2 * kb_i);
            // Software pipeline over the 8 k-steps of a tile: the 14 transposing reads of step ks + 1 are issued BEFORE the 9 MFMAs
            // of step ks and land in a second set of fragment registers.  Left to itself the compiler reuses one B fragment
            // for consecutive column tiles and waits for each read right before the MFMA pair that needs it: the in-kernel stamps
            // showed 490 cycles per k-step for 288 cycles of MFMA issue (LDS latency exposed six times per step).
            struct Frags {
                f16x8 a0, a1, b0, b1, b2, b3, b4;
            };
            auto load_frags = [&](int ks) {
                const int y = ks >> 1, px0 = (ks & 1) * 16;
                const char *ga = base + (y * TRW + px0) * 128;              // + 4 * 128 for the second half of a fragment
                const char *xa = base + (y * TRXW + px0) * 128;
                Frags f;
                f.a0 = tr_frag(ga + aoff[0], ga + aoff[0] + 512);
                f.a1 = tr_frag(ga + aoff[1], ga + aoff[1] + 512);
                f.b0 = tr_frag(xa + boff[0], xa + boff[0] + 512);
                f.b1 = tr_frag(xa + boff[1], xa + boff[1] + 512);
                f.b2 = tr_frag(xa + boff[2], xa + boff[2] + 512);
                f.b3 = tr_frag(xa + boff[3], xa + boff[3] + 512);
                f.b4 = tr_frag(xa + boff[4], xa + boff[4] + 512);
                return f;
            };
            auto multiply = [&](const Frags &f) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0, f.b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1, f.b0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0, f.b1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1, f.b1, acc[1][1], 0, 0, 0);
                acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0, f.b2, acc[2][0], 0, 0, 0);
                acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1, f.b2, acc[2][1], 0, 0, 0);
                acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0, f.b3, acc[3][0], 0, 0, 0);
                acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1, f.b3, acc[3][1], 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm ? f.a1 : f.a0, f.b4, accx, 0, 0, 0);
            };
            Frags f0 = load_frags(0), f1;
#pragma unroll
            for (int k2 = 0; k2 < TRH; ++k2) {
                f1 = load_frags(2 * k2 + 1);
                __builtin_amdgcn_sched_barrier(0);
                multiply(f0);
                __builtin_amdgcn_sched_barrier(0);
                if (k2 + 1 < TRH) f0 = load_frags(2 * k2 + 2);
                __builtin_amdgcn_sched_barrier(0);
                multiply(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kb_i < 12) KB_STAMP(3 + 2 * kb_i);
            ++kb_i;
            __syncthreads();               // (B) this image has been read, the other one is complete
            cur ^= 1;
        }
        KB_STAMP(29);
        const float oscale = 1.f / (sx * sg);
        const __amdgpu_buffer_rsrc_t rsl = make_rsrc(my, (unsigned)wsz * 4u);   // rows co >= Cout fall past the slab: dropped
        // slab layout of THIS kernel: [co][tap][ci] -- a block's lanes are consecutive input channels of one tap, so the channel
        // has to be the fastest index for the stores to be 128-byte segments (with the [co][ci][tap] layout of the other
        // kernels every store instruction scattered 64 dwords over 2.3 KB); conv_wgrad_reduce_f32 un-permutes (perm_cin)
        const unsigned co_row = (unsigned)(g.Cin * KK) * 4u;
        auto store_block = [&](const f32x16 &a, int m, int nt) {
            const int cb = nt / KK, tap = nt - cb * KK;
            const int ci = ci_base + cb * 32 + (lane & 31);
            const unsigned o0 = ci < g.Cin ? (unsigned)(((co_base + m * 32 + 4 * (lane >> 5)) * KK + tap) * g.Cin + ci) * 4u : SENT;
#pragma unroll
            for (int r = 0; r < 16; ++r) buf_st(rsl, o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * co_row, a[r] * oscale);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) store_block(acc[j][m], m, nq + NQ * j);
        store_block(accx, xm, xt);
        // (the producers' stage loops run whole rounds of TR_NSTG phases without early exits -- conv_fwd_f16_ws explains why --
        // and the barriers of their trailing dead phases are matched here)
        for (int k = ntiles_done; k < (ntiles_done + TR_NSTG - 1) / TR_NSTG * TR_NSTG; ++k) __syncthreads();
        KB_STAMP(30);
        __syncthreads();                   // (C) the producers' bias partials are in LDS
        KB_STAMP(31);
        KB_FLUSH_SELF();
        if (need_bias && ci_blk == 0 && wave == 0) {
            const float *scr = reinterpret_cast<const float *>(smt + 2 * TR_BUFB);
            const float v = ((scr[lane] + scr[64 + lane]) + scr[128 + lane]) + scr[192 + lane];
            if (co_base + lane < g.Cout) my[wsz + co_base + lane] = v;
        }
        return;
    }
    // ---------------------------------------------------------------------- producers
    saturate_fp16_conversions();           // (MODE.FP16_OVFL in the converting waves only, c16.hpp)
    const int ptid = tid - 64 * NQ;
    if constexpr (IN16) {
        static_assert(DACT == 0, "fp16 c16 operands: pre-activation gradients only");
        // input image: rows 0..5, stored columns 1..34 of the 36 (the two outer ones are never read), 4 blocks, 2 halves
        constexpr int XCOLS = TRW + 2, NXP = 4 * TRXR * 2 * XCOLS, NXK16 = (NXP + PT - 1) / PT;      // 1632 pieces, 7 per thread
        constexpr int NGK16 = 4;                                                               // 4 x 128 px x 2 = 1024 grad_out pieces
        const _Float16 *x16 = reinterpret_cast<const _Float16 *>(x), *g16 = reinterpret_cast<const _Float16 *>(gout);
        // input pieces: id = (block, row, half, column); grad_out pieces: id = ptid + 256 k -> pixel ptid & 31, half (ptid >> 5) & 1,
        // row ptid >> 6, block k: consecutive lanes walk the columns of one half-row run of the image (c16.hpp)
        int xb[NXK16], xr[NXK16], xc[NXK16], xh[NXK16], xd[NXK16];
#pragma unroll
        for (int k = 0; k < NXK16; ++k) {
            int id = ptid + k * PT;
            xc[k] = id % XCOLS; id /= XCOLS;
            xh[k] = id & 1; id >>= 1;
            xr[k] = id % TRXR;
            xb[k] = id / TRXR;                                       // 0..3 (>= 4: past the end)
            const int pos = xr[k] * TRXW + xc[k] + 1;
            xd[k] = pos * 128 + (((2 * xb[k] + xh[k]) ^ (((pos >> 1) & 1) << 2)) << 4);
        }
        const int p_half = (ptid >> 5) & 1;
        const int gp_x = ptid & 31, gp_y = ptid >> 6;
        const int gpix = gp_y * TRW + gp_x;
        int gd[NGK16];
#pragma unroll
        for (int k = 0; k < NGK16; ++k) gd[k] = TR_XB + gpix * 128 + (((2 * k + p_half) ^ (((gpix >> 1) & 1) << 2)) << 4);
        const int cbx = g.Cin >> 4, cbo = g.Cout >> 4;               // 16-channel blocks per group (input) / of grad_out
        const int cbx_rem = (g.Cin + 15) / 16 - (ci_base >> 4);      // blocks of this workgroup's 64-channel block that exist
        const unsigned xblk = (unsigned)HW * 32u, gblk = (unsigned)HWo * 32u;
        const unsigned xs_bytes = (unsigned)((g.Cin + 15) / 16) * xblk, gs_bytes = (unsigned)cbo * gblk;
        (void)cbx;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        struct S16 {
            u32x4 rx[NXK16];
            u32x4 rg[NGK16];                       // c16: four pieces; planar: eight 8-byte quads (4 pixels of channels 8 chg + e) as 4 x 16 B
        };
        S16 sa, sb, sc;
        float bacc[NGK16][8];
#pragma unroll
        for (int k = 0; k < NGK16; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) bacc[k][e] = 0.f;
        // planar grad_out item of this thread: 8-channel group chg, tile row pg_y, quad pg_q (as the fp32 form)
        const int chg = ptid & 7, pg_y = (ptid >> 3) >> 3, pg_q = (ptid >> 3) & 7;
        const unsigned gplane16 = (unsigned)HWo * 2u;
        auto tile_coords = [&](int tile, int &b, int &y0, int &x0) {
            int t = (xcd_map && tile < total_tiles) ? xcd_tile(tile, total_tiles) : tile;
            const int tx = t % tiles_x; t /= tiles_x;
            const int ty = t % tiles_y;
            b = t / tiles_y; y0 = ty * TRH; x0 = tx * TRW;
        };
        auto prefetch = [&](int tile, S16 &s) {
            int b, y0, x0;
            tile_coords(tile, b, y0, x0);
            const bool live = tile < total_tiles;
            const int bb = live ? b : 0;
            const __amdgpu_buffer_rsrc_t rxi = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<_Float16 *>(x16) + ((int64_t)bb * g.groups + grp) * ((g.Cin + 15) / 16) * HW * 16, 0, live ? xs_bytes : 0u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rgo = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<_Float16 *>(g16) + (int64_t)bb * cbo * HWo * 16, 0, live ? gs_bytes : 0u, 0x00020000);
#pragma unroll
            for (int k = 0; k < NXK16; ++k) {
                const int yy = y0 - 1 + xr[k], xx = x0 - 1 + xc[k];
                const bool ok = ptid + k * PT < NXP && xb[k] < cbx_rem && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const unsigned o = sel_off(ok, (unsigned)((ci_base >> 4) + xb[k]) * xblk + (unsigned)((yy * 2 + xh[k]) * g.W + xx) * 16u);
                s.rx[k] = __builtin_amdgcn_raw_buffer_load_b128(rxi, o, 0, 0);
            }
            if constexpr (GP16) {
                const __amdgpu_buffer_rsrc_t rgp = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<_Float16 *>(g16) + (int64_t)bb * g.Cout * HWo, 0, live ? (unsigned)g.Cout * gplane16 : 0u, 0x00020000);
                const int gy = y0 + pg_y, gx = x0 + 4 * pg_q;
                const unsigned o = sel_off(gy < g.Ho && gx + 3 < g.Wo, (unsigned)(co_base + 8 * chg) * gplane16 + (unsigned)(gy * g.Wo + gx) * 2u);
#pragma unroll
                for (int k = 0; k < NGK16; ++k) {          // channels 2k, 2k + 1 of the group in one register quad
                    const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rgp, o + (unsigned)(2 * k) * gplane16, 0, 0);
                    const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rgp, o + (unsigned)(2 * k + 1) * gplane16, 0, 0);
                    s.rg[k] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            } else {
                const int gy = y0 + gp_y, gx = x0 + gp_x;
                const bool gok = gy < g.Ho && gx < g.Wo;
#pragma unroll
                for (int k = 0; k < NGK16; ++k) {
                    const int cb = (co_base >> 4) + k;
                    const unsigned o = sel_off(gok && cb < cbo, (unsigned)cb * gblk + (unsigned)((gy * 2 + p_half) * g.Wo + gx) * 16u);
                    s.rg[k] = __builtin_amdgcn_raw_buffer_load_b128(rgo, o, 0, 0);
                }
            }
        };
        const float inv_sg = 1.f / sg;
        auto commit = [&](int buf, S16 &s) {
            char *img = smt + buf * TR_BUFB;
            wait_vmcnt<2 * (NXK16 + (GP16 ? 2 * NGK16 : NGK16))>();      // this stage is the oldest of three in flight
#pragma unroll
            for (int k = 0; k < NXK16; ++k)
                if (ptid + k * PT < NXP) *reinterpret_cast<u32x4 *>(img + xd[k]) = s.rx[k];
            if constexpr (GP16) {
                // rg[k] = {ch 2k: px 0-1, px 2-3, ch 2k+1: px 0-1, px 2-3}; piece of pixel j = channels 8 chg .. 8 chg + 7
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 hv;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        hv[k] = __builtin_amdgcn_perm(s.rg[k][2 + (j >> 1)], s.rg[k][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
                    const int pix = pg_y * TRW + 4 * pg_q + j;
                    *reinterpret_cast<u32x4 *>(img + TR_XB + pix * 128 + ((chg ^ (((pix >> 1) & 1) << 2)) << 4)) = hv;
                }
                if (need_bias && ci_blk == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f16x8 hv = __builtin_bit_cast(f16x8, s.rg[k]);     // 4 pixels of channel 2k, then 4 of channel 2k + 1
                        bacc[k][0] += ((float)hv[0] + (float)hv[1]) + ((float)hv[2] + (float)hv[3]);
                        bacc[k][1] += ((float)hv[4] + (float)hv[5]) + ((float)hv[6] + (float)hv[7]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < NGK16; ++k) {
                    *reinterpret_cast<u32x4 *>(img + gd[k]) = s.rg[k];
                    if (need_bias && ci_blk == 0) {                      // bias gradient: plain sums of the (scaled) grad_out values
                        const f16x8 hv = __builtin_bit_cast(f16x8, s.rg[k]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) bacc[k][e] += (float)hv[e];
                    }
                }
            }
        };
        // the two outer stored columns (0 and 35) of the input image are never read; zero-filled positions come from the loads
        prefetch(split, sa);
        prefetch(split + G, sb);
        prefetch(split + 2 * G, sc);
        commit(0, sa);
        __syncthreads();                   // (A)
        prefetch(split + 3 * G, sa);
        int cur = 0;
        for (int tile = split; tile < total_tiles; tile += 3 * G) {       // whole rounds: no early exit (see conv_fwd_f16_ws)
            commit(cur ^ 1, sb);           // tile + G
            prefetch(tile + 4 * G, sb);
            __syncthreads();               // (B)
            cur ^= 1;
            commit(cur ^ 1, sc);           // tile + 2 G
            prefetch(tile + 5 * G, sc);
            __syncthreads();               // (B)
            cur ^= 1;
            commit(cur ^ 1, sa);           // tile + 3 G
            prefetch(tile + 6 * G, sa);
            __syncthreads();               // (B)
            cur ^= 1;
        }
        if constexpr (GP16) {
            if (need_bias && ci_blk == 0) {
                // channel 8 chg + 2 k + i: summed over the 8 lanes-groups of a wave that share chg (lane bits 3..5), as the fp32 form
                float *scr = reinterpret_cast<float *>(smt + 2 * TR_BUFB);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v = bacc[k][i];
                        v += __shfl_xor(v, 8, 64);
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        if ((lane >> 3) == 0) scr[(wave - NQ) * 64 + 8 * chg + 2 * k + i] = v * inv_sg;
                    }
            }
            __syncthreads();               // (C)
            return;
        }
        if (need_bias && ci_blk == 0) {
            // channel 16 k + 8 half + e: summed over the lanes that share `half` (lane bits 0..4 = pixel), then the four waves
            // (= tile rows) go through LDS to the consumers, which add them in a fixed order after barrier (C)
            float *scr = reinterpret_cast<float *>(smt + 2 * TR_BUFB);
#pragma unroll
            for (int k = 0; k < NGK16; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = bacc[k][e];
                    v += __shfl_xor(v, 1, 64);
                    v += __shfl_xor(v, 2, 64);
                    v += __shfl_xor(v, 4, 64);
                    v += __shfl_xor(v, 8, 64);
                    v += __shfl_xor(v, 16, 64);
                    if ((lane & 31) == 0) scr[(wave - NQ) * 64 + 16 * k + 8 * p_half + e] = v * inv_sg;
                }
        }
        __syncthreads();                   // (C)
        return;
    }
    const unsigned go_bytes = (unsigned)g.Cout * (unsigned)HWo * 4u, x_bytes = (unsigned)g.Cin * (unsigned)HW * 4u;
    const unsigned xplane = (unsigned)HW * 4u, gplane = (unsigned)HWo * 4u;
    // item geometry (fixed per thread): 8-channel group `chg`; grad_out item (row gy, quad gq); input items (row, quad)
    const int chg = ptid & 7;
    const int g_y = (ptid >> 3) >> 3, g_q = (ptid >> 3) & 7;
    int x_r[NXK], x_q[NXK];
#pragma unroll
    for (int k = 0; k < NXK; ++k) {
        const int rq = (ptid + k * PT) >> 3;
        x_r[k] = rq / TRXQ;
        x_q[k] = rq - x_r[k] * TRXQ;
    }
    // register stages: the input quads (two thirds of a tile's bytes) of TWO tiles and the grad_out quads of one are in flight
    // (three full stages spill: 2 x 96 + addressing > 256 registers; measured 49 spilled registers inside the loop)
    struct XStage {
        u32x4 rx[NXK][8];
    };
    XStage sa, sb;
    u32x4 rg[8], ry[DACT != 0 ? 8 : 1];
    float bacc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bacc[e] = 0.f;
    float amax_g = 0.f, amax_x = 0.f;
    // (splits congruent mod 8 run on one XCD when gridDim.x is a multiple of 8 -- see the placement above -- so their tiles
    // split, split + G, .. are mapped onto one contiguous eighth of the tile sequence: shared halo lines hit that XCD's L2)
    auto tile_coords = [&](int tile, int &b, int &y0, int &x0) {
        int t = (xcd_map && tile < total_tiles) ? xcd_tile(tile, total_tiles) : tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        b = t / tiles_y; y0 = ty * TRH; x0 = tx * TRW;
    };
    auto prefetch_g = [&](int tile) {
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        const bool live = tile < total_tiles;
        const __amdgpu_buffer_rsrc_t rgo = make_rsrc(gout + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
        const int gy = y0 + g_y, gx = x0 + 4 * g_q;
        const unsigned o = (gy < g.Ho && gx + 3 < g.Wo) ? (unsigned)((co_base + 8 * chg) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
        for (int e = 0; e < 8; ++e) rg[e] = __builtin_amdgcn_raw_buffer_load_b128(rgo, o + (unsigned)e * gplane, 0, 0);
        if constexpr (DACT != 0) {
            const __amdgpu_buffer_rsrc_t rya = make_rsrc(yact + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e) ry[e] = __builtin_amdgcn_raw_buffer_load_b128(rya, o + (unsigned)e * gplane, 0, 0);
        }
    };
    auto prefetch_x = [&](int tile, XStage &s) {
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        const bool live = tile < total_tiles;
        const __amdgpu_buffer_rsrc_t rxi = make_rsrc(x + ((int64_t)(live ? b : 0) * g.groups + grp) * g.Cin * HW, live ? x_bytes : 0u);
#pragma unroll
        for (int k = 0; k < NXK; ++k) {
            const int yy = y0 - 1 + x_r[k], xq = x0 - 4 + 4 * x_q[k];
            const bool ok = ptid + k * PT < NXI && yy >= 0 && yy < g.H && xq >= 0 && xq + 3 < g.W;
            const unsigned o = ok ? (unsigned)((ci_base + 8 * chg) * HW + yy * g.W + xq) * 4u : SENT;
#pragma unroll
            for (int e = 0; e < 8; ++e) s.rx[k][e] = __builtin_amdgcn_raw_buffer_load_b128(rxi, o + (unsigned)e * xplane, 0, 0);
        }
    };
    auto commit_g = [&](int buf, int tile) {
        char *gi = smt + buf * TR_BUFB + TR_XB;
        // grad_out (and the saved output) of this tile: the input quads of the NEXT tile were requested after them
        if constexpr (WG_EXPLICIT_WAITS) wait_vmcnt<NXK * 8>();
        if constexpr (DACT != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rg[e][j] = __float_as_uint(__uint_as_float(rg[e][j]) * act_grad_c<DACT>(__uint_as_float(ry[e][j]), dslope));
            if (gpre_out != nullptr && ci_blk == 0 && !gpre_c16) {   // grad * act' for the data gradient: one writer per output-channel block
                int b, y0, x0;
                tile_coords(tile, b, y0, x0);
                const bool live = tile < total_tiles;
                const __amdgpu_buffer_rsrc_t rgp = make_rsrc(gpre_out + (int64_t)(live ? b : 0) * g.Cout * HWo, live ? go_bytes : 0u);
                const int gy = y0 + g_y, gx = x0 + 4 * g_q;
                const unsigned o = (gy < g.Ho && gx + 3 < g.Wo) ? (unsigned)((co_base + 8 * chg) * HWo + gy * g.Wo + gx) * 4u : SENT;
#pragma unroll
                for (int e = 0; e < 8; ++e) __builtin_amdgcn_raw_buffer_store_b128(rg[e], rgp, o + (unsigned)e * gplane, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 hv;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float v0 = __uint_as_float(rg[e][j]), v1 = __uint_as_float(rg[e + 1][j]);
                bacc[e] += v0;
                bacc[e + 1] += v1;
                amax_g = amax_acc(amax_g, v0, v1);
                hv[e >> 1] = pack_f16(v0 * sg, v1 * sg);
            }
            const int pix = g_y * TRW + 4 * g_q + j;
            *reinterpret_cast<u32x4 *>(gi + pix * 128 + ((chg ^ (((pix >> 1) & 1) << 2)) << 4)) = hv;
            if constexpr (DACT != 0) {
                // gpre_c16: grad * act' leaves as the c16 IMAGE the data gradient stages (c16.hpp) -- `hv` IS its piece (this
                // pixel, channels 8 chg .. 8 chg + 7, times the slot's scale): one 16-byte store per pixel instead of eight
                // fp32 quads per 4 pixels, and the data gradient copies pieces instead of converting planes
                if (gpre_c16 && gpre_out != nullptr && ci_blk == 0) {
                    int b, y0, x0;
                    tile_coords(tile, b, y0, x0);
                    const bool live = tile < total_tiles;
                    const int cb16 = g.Cout >> 4, cblk = (co_base >> 4) + (chg >> 1);
                    const __amdgpu_buffer_rsrc_t rgi = __builtin_amdgcn_make_buffer_rsrc(
                        reinterpret_cast<_Float16 *>(gpre_out) + (int64_t)(live ? b : 0) * cb16 * HWo * 16, 0,
                        live ? (unsigned)cb16 * (unsigned)HWo * 32u : 0u, 0x00020000);
                    const int gy = y0 + g_y, gx = x0 + 4 * g_q + j;
                    const unsigned o = (gy < g.Ho && gx < g.Wo && cblk < cb16)
                                           ? ((unsigned)cblk * (unsigned)HWo * 2u + (unsigned)((gy * 2 + (chg & 1)) * g.Wo + gx)) * 16u : SENT;
                    __builtin_amdgcn_raw_buffer_store_b128(hv, rgi, o, 0, 0);
                }
            }
        }
    };
    auto commit_x = [&](int buf, XStage &s) {
        char *xi = smt + buf * TR_BUFB;
        // this stage's input quads are the oldest loads in flight; behind them: one tile of grad_out quads (+ the saved output
        // with a folded derivative) and, with two input stages, the other stage
        if constexpr (WG_EXPLICIT_WAITS) wait_vmcnt<(DACT != 0 ? 16 : NXK * 8 + 8)>();
#pragma unroll
        for (int k = 0; k < NXK; ++k)
            if (ptid + k * PT < NXI) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * x_q[k] + j - 2;                   // stored positions: tile columns -2 .. 33 of the quad grid
                    u32x4 hv;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float v0 = __uint_as_float(s.rx[k][e][j]), v1 = __uint_as_float(s.rx[k][e + 1][j]);
                        amax_x = amax_acc(amax_x, v0, v1);
                        hv[e >> 1] = pack_f16(v0 * sx, v1 * sx);
                    }
                    if (c < 0 || c >= TRXW) continue;
                    const int pos = x_r[k] * TRXW + c;
                    *reinterpret_cast<u32x4 *>(xi + pos * 128 + ((chg ^ (((pos >> 1) & 1) << 2)) << 4)) = hv;
                }
            }
    };
    // Start-up order as in conv_fwd_f16_ws: only the first tile's commit between the first loads and barrier (A) (the three
    // prefetches that used to stand there took 9 us to issue: the consumers started 18.7 us into a 44 us kernel).
    prefetch_x(split, sa);
    prefetch_g(split);
    KB_STAMP(1);
    if constexpr (DACT == 0) {
        prefetch_x(split + G, sb);         // (requested before tile 0 is converted: arrives behind it)
        commit_x(0, sa);
        commit_g(0, split);
        KB_STAMP(2);
        __syncthreads();                   // (A)
        prefetch_g(split + G);
        prefetch_x(split + 2 * G, sa);     // from here on: input quads of two tiles, grad_out quads of one in flight
        KB_STAMP(3);
        int cur = 0;
        [[maybe_unused]] int kb_i = 0;
        // (This loop keeps its early exit between the phases, although the structurizer routes it through the loop latch and the
        // wait-count pass then drains the other input stage once per round -- s_waitcnt vmcnt(17) .. (8) where vmcnt(33) would
        // do; see conv_fwd_f16_ws.  Both cures that work for the image-reading producers -- dead padding phases, or whole rounds
        // plus a tail phase after the loop -- push this 240-register variant into scratch: 166 / 223 spill instructions.)
        for (int tile = split; tile < total_tiles; tile += 2 * G) {
            if (kb_i < 6) KB_STAMP(4 + 4 * kb_i);
            commit_x(cur ^ 1, sb);         // tile + G, while the consumers multiply tile `tile` from image `cur`
            prefetch_x(tile + 3 * G, sb);  // (past the end: zero-record descriptors, nothing is read)
            commit_g(cur ^ 1, tile + G);
            prefetch_g(tile + 2 * G);
            if (kb_i < 6) KB_STAMP(5 + 4 * kb_i);
            __syncthreads();               // (B)
            cur ^= 1;
            if (tile + G >= total_tiles) break;
            if (kb_i < 6) KB_STAMP(6 + 4 * kb_i);
            commit_x(cur ^ 1, sa);         // tile + 2 G
            prefetch_x(tile + 4 * G, sa);
            commit_g(cur ^ 1, tile + 2 * G);
            prefetch_g(tile + 3 * G);
            if (kb_i < 6) KB_STAMP(7 + 4 * kb_i);
            ++kb_i;
            __syncthreads();               // (B)
            cur ^= 1;
        }
    } else {
        commit_x(0, sa);
        commit_g(0, split);
        KB_STAMP(2);
        __syncthreads();                   // (A)
        prefetch_x(split + G, sa);
        prefetch_g(split + G);
        int cur = 0;
        for (int tile = split; tile < total_tiles; tile += G) {
            commit_x(cur ^ 1, sa);         // tile + G
            prefetch_x(tile + 2 * G, sa);
            commit_g(cur ^ 1, tile + G);
            prefetch_g(tile + 2 * G);
            __syncthreads();               // (B)
            cur ^= 1;
        }
    }
    x_slot.record(amax_x);
    g_slot.record(amax_g);
    // bias partial of this workgroup: channel 8 chg + e summed over the 32 threads of a wave that share chg (lane bits 3..5);
    // the four waves' sums go through LDS to the consumers, which add them in a fixed order after barrier (C)
    if (need_bias && ci_blk == 0) {
        float *scr = reinterpret_cast<float *>(smt + 2 * TR_BUFB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bacc[e];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if ((lane >> 3) == 0) scr[(wave - NQ) * 64 + 8 * chg + e] = v;
        }
    }
    KB_STAMP(30);
    __syncthreads();                       // (C)
    KB_STAMP(31);
    KB_FLUSH_SELF();
}

template <int DACT, bool IN16 = false, bool GP16 = false>
__global__ __launch_bounds__(512) void conv_wgrad_f16_tr(const float *__restrict__ x, const float *__restrict__ gout,
                                                         const float *__restrict__ yact, float *__restrict__ gpre_out,
                                                         float *__restrict__ slab, ConvGeom g, float dslope, int total_tiles,
                                                         int need_bias, ScaleSlot x_slot, ScaleSlot g_slot, int gpre_c16 = 0) {
    // Workgroup -> (split, co block, ci block).  The ci blocks of one (split, co block) read the SAME grad_out tiles (839 MB at
    // 128 -> 1600): they are placed 8 workgroup ids apart, i.e. on the same XCD under the round-robin dispatch (speed only:
    // MI355X_MICROARCH.md, workgroup dispatch), so that the second reader finds the tile in that XCD's L2.  PMC before:
    // 2.2 GB fetched per launch for 0.91 GB of operands.
    int split, co_blk, ci_blk;
    {
        const int nz = gridDim.z, ny = gridDim.y, nx = gridDim.x;
        const int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), total = nx * ny * nz;
        const int T8 = total / (8 * nz) * (8 * nz);          // ids below T8: groups of 8 * nz; the remainder pairs up in id order
        int rest;                                            // 0 .. nx * ny - 1: (split, co block)
        if (L < T8) {
            const int xcd = L & 7, i = L >> 3;
            ci_blk = i % nz;
            rest = (i / nz) * 8 + xcd;
        } else {
            ci_blk = (L - T8) % nz;
            rest = T8 / nz + (L - T8) / nz;
        }
        split = rest % nx;
        co_blk = rest / nx;
    }
    const int64_t wsz = (int64_t)g.Cout * g.Cin * 9;
    // (splits congruent mod 8 run on one XCD when gridDim.x is a multiple of 8, so their tiles split, split + G, .. are mapped
    // onto one contiguous eighth of the tile sequence -- xcd_tile: shared halo lines hit that XCD's L2)
    wgrad_tr_body<DACT, IN16, GP16>(x, gout, yact, gpre_out, slab + (int64_t)split * (wsz + g.Cout), g, dslope, split, (int)gridDim.x,
                                    total_tiles, (gridDim.x & 7) == 0, need_bias, x_slot, g_slot, co_blk, ci_blk, gpre_c16);
}

// Several weight gradients over the SAME pixels in one launch (round 4): the three layers of a ResidualControl round, each
// exactly two 64 x 64 blocks.  One launch of one layer splits the pixels 128 ways to fill the chip and every workgroup leaves a
// 147 KB slab of partial sums: 37.7 MB written and read again by the reduction per layer -- as much HBM traffic as the operands,
// and a quarter of the kernel's time (in-kernel stamps: 4.5 us until the first tile is staged, 5.5 us in the slab stores of a
// 36 us launch).  Batched, the 256 CUs are shared by the 6 blocks: 40 splits per layer, a third of the slab traffic and of the
// start-up per layer, 12 launches per step instead of 36.
// Workgroup L: XCD L & 7 owns one contiguous eighth of the tiles; its 2 * nunits * per_xcd workgroups are (split within the XCD,
// layer, block) -- the two blocks of a layer read a common operand (the input image for two co blocks, the gradient image for
// two ci blocks) and find it in that XCD's L2, neighbouring tiles share their halo lines there.
struct WgradItem {
    const void *x16, *g16;
    float *slab, *x_slot, *g_slot;
    int Cin, Cout, groups, need_bias;
};
constexpr int WGRAD_BATCH_MAX = 4;
struct WgradBatch {
    WgradItem item[WGRAD_BATCH_MAX];
};
__global__ __launch_bounds__(512) void conv_wgrad_f16_tr_batch(WgradBatch bt, ConvGeom g0, int total_tiles, int per_xcd, int nunits) {
    const int L = blockIdx.x, xcd = L & 7, i = L >> 3;
    if (i >= 2 * nunits * per_xcd) return;                 // (whole workgroup, before any barrier)
    const int which = i & 1, pu = i >> 1;
    const int unit = pu % nunits, s_local = pu / nunits;
    WgradItem it = bt.item[0];
#pragma unroll
    for (int k = 1; k < WGRAD_BATCH_MAX; ++k)
        if (unit == k) it = bt.item[k];
    ConvGeom g = g0;
    g.Cin = it.Cin; g.Cout = it.Cout; g.groups = it.groups;
    const int t0 = (int)((int64_t)total_tiles * xcd / 8), t1 = (int)((int64_t)total_tiles * (xcd + 1) / 8);
    const bool two_co = g.Cout > 64;                       // (the launcher admits layers of exactly two blocks)
    const int64_t wsz = (int64_t)g.Cout * g.Cin * 9;
    wgrad_tr_body<ACT_NONE, true, false>(static_cast<const float *>(it.x16), static_cast<const float *>(it.g16), nullptr, nullptr,
                                         it.slab + (int64_t)(xcd * per_xcd + s_local) * (wsz + g.Cout), g, 0.f, t0 + s_local, per_xcd, t1,
                                         false, it.need_bias, ScaleSlot{it.x_slot}, ScaleSlot{it.g_slot}, two_co ? which : 0,
                                         two_co ? 0 : which);
}

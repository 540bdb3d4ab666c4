/*
 * ORACLE (test infrastructure only -- never linked into or called by the product path).
 *
 * CPU restatement of the reference's modulated deformable convolution v2 (DCNv2):
 *   bilinear sample            models/DCNv2/src/cpu/dcn_v2_im2col_cpu.cpp:27-57
 *   d(sample)/d(pixel) weight  models/DCNv2/src/cpu/dcn_v2_im2col_cpu.cpp:59-84
 *   d(sample)/d(coord) weight  models/DCNv2/src/cpu/dcn_v2_im2col_cpu.cpp:86-127
 *   deformable im2col          models/DCNv2/src/cpu/dcn_v2_im2col_cpu.cpp:129-198
 *   col2im (grad_input)        models/DCNv2/src/cpu/dcn_v2_im2col_cpu.cpp:200-257
 *   col2im_coord (grad_offset, grad_mask)  dcn_v2_im2col_cpu.cpp:259-330
 *   forward / backward composition follows the CUDA wrapper
 *       models/DCNv2/src/cuda/dcn_v2_cuda.cu:64-94 (fwd), :138-211 (bwd)
 *   because the CPU wrapper accumulates into an uninitialised at::empty output
 *   (models/DCNv2/src/cpu/dcn_v2_cpu.cpp:65,110,127) and is therefore not a usable statement
 *   of the op.
 *
 * Reference quirks kept on purpose:
 *   - a tap contributes only when  -1 < h_im < H  and  -1 < w_im < W  (strict), corners outside
 *     the image read as zero;
 *   - col2im is launched with pad_h in place of pad_w (dcn_v2_im2col_cpu.cpp:364, same in
 *     dcn_v2_im2col_cuda.cu:368): grad_input uses pad_h for BOTH axes;
 *   - col2im scans a 5x5 window around the C-truncated sample position and keeps grid points
 *     closer than 1 in both axes; that is the set {floor, floor+1}^2 clipped to the image, which
 *     is what this file iterates directly (the extra window cells always carry weight 0);
 *   - grad_mask does not include the mask factor, grad_offset does.
 *
 * The reference's C++ cannot be built in this image (it includes TH/TH.h and THC headers that
 * torch 2.10 no longer ships), so this restatement is pinned by the reference's own
 * known-answer tests (models/DCNv2/testcpu.py:32-67 zero-offset identity, :69-97 gradcheck),
 * restated in tests/test_oracle_dcn.py, plus an independent grid_sample formulation.
 *
 * Layouts (all contiguous): im [B,C,H,W]; offset [B, dg*2*kh*kw, Ho, Wo] with channel
 * 2*(i*kw+j) = dy and +1 = dx inside each group's block; mask [B, dg*kh*kw, Ho, Wo];
 * col [B, C*kh*kw, Ho*Wo]; weight [Co, C, kh, kw].
 *
 * Included twice by dcn_ref.c with REAL = float / double.
 */

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef struct {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo;
} FN(dcn_geom);

static REAL FN(bilinear)(const REAL *plane, int H, int W, REAL h, REAL w)
{
    int h0 = (int)floor((double)h), w0 = (int)floor((double)w);
    int h1 = h0 + 1, w1 = w0 + 1;
    REAL lh = h - h0, lw = w - w0;
    REAL hh = 1 - lh, hw = 1 - lw;
    REAL v1 = (h0 >= 0 && w0 >= 0) ? plane[h0 * W + w0] : 0;
    REAL v2 = (h0 >= 0 && w1 <= W - 1) ? plane[h0 * W + w1] : 0;
    REAL v3 = (h1 <= H - 1 && w0 >= 0) ? plane[h1 * W + w0] : 0;
    REAL v4 = (h1 <= H - 1 && w1 <= W - 1) ? plane[h1 * W + w1] : 0;
    REAL w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;
    return (w1_ * v1 + w2_ * v2 + w3_ * v3 + w4_ * v4);
}

static REAL FN(pixel_weight)(REAL ah, REAL aw, int h, int w, int H, int W)
{
    if (ah <= -1 || ah >= H || aw <= -1 || aw >= W)
        return 0;
    int h0 = (int)floor((double)ah), w0 = (int)floor((double)aw);
    int h1 = h0 + 1, w1 = w0 + 1;
    REAL wt = 0;
    if (h == h0 && w == w0) wt = (h + 1 - ah) * (w + 1 - aw);
    if (h == h0 && w == w1) wt = (h + 1 - ah) * (aw + 1 - w);
    if (h == h1 && w == w0) wt = (ah + 1 - h) * (w + 1 - aw);
    if (h == h1 && w == w1) wt = (ah + 1 - h) * (aw + 1 - w);
    return wt;
}

static REAL FN(coord_weight)(REAL ah, REAL aw, int H, int W, const REAL *plane, int dir)
{
    if (ah <= -1 || ah >= H || aw <= -1 || aw >= W)
        return 0;
    int h0 = (int)floor((double)ah), w0 = (int)floor((double)aw);
    int h1 = h0 + 1, w1 = w0 + 1;
    REAL wt = 0;
    if (dir == 0) { /* d/dh */
        if (h0 >= 0 && w0 >= 0) wt += -1 * (w0 + 1 - aw) * plane[h0 * W + w0];
        if (h0 >= 0 && w1 <= W - 1) wt += -1 * (aw - w0) * plane[h0 * W + w1];
        if (h1 <= H - 1 && w0 >= 0) wt += (w0 + 1 - aw) * plane[h1 * W + w0];
        if (h1 <= H - 1 && w1 <= W - 1) wt += (aw - w0) * plane[h1 * W + w1];
    } else { /* d/dw */
        if (h0 >= 0 && w0 >= 0) wt += -1 * (h0 + 1 - ah) * plane[h0 * W + w0];
        if (h0 >= 0 && w1 <= W - 1) wt += (h0 + 1 - ah) * plane[h0 * W + w1];
        if (h1 <= H - 1 && w0 >= 0) wt += -1 * (ah - h0) * plane[h1 * W + w0];
        if (h1 <= H - 1 && w1 <= W - 1) wt += (ah - h0) * plane[h1 * W + w1];
    }
    return wt;
}

/* one sample: im [C,H,W], offset [dg*2*kk,Ho,Wo], mask [dg*kk,Ho,Wo] -> col [C*kk, Ho*Wo] */
static void FN(im2col_1)(const FN(dcn_geom) *g, const REAL *im, const REAL *off, const REAL *msk,
                         REAL *col)
{
    const int kk = g->kh * g->kw, cpg = g->C / g->dg, HWo = g->Ho * g->Wo;
    for (int c = 0; c < g->C; ++c) {
        const int grp = c / cpg;
        const REAL *plane = im + (size_t)c * g->H * g->W;
        const REAL *goff = off + (size_t)grp * 2 * kk * HWo;
        const REAL *gmsk = msk + (size_t)grp * kk * HWo;
        for (int ho = 0; ho < g->Ho; ++ho)
            for (int wo = 0; wo < g->Wo; ++wo) {
                const int p = ho * g->Wo + wo;
                const int h_in = ho * g->sh - g->ph, w_in = wo * g->sw - g->pw;
                for (int i = 0; i < g->kh; ++i)
                    for (int j = 0; j < g->kw; ++j) {
                        const int t = i * g->kw + j;
                        const REAL dy = goff[(size_t)(2 * t) * HWo + p];
                        const REAL dx = goff[(size_t)(2 * t + 1) * HWo + p];
                        const REAL m = gmsk[(size_t)t * HWo + p];
                        const REAL h_im = h_in + i * g->dh + dy;
                        const REAL w_im = w_in + j * g->dw + dx;
                        REAL v = 0;
                        if (h_im > -1 && w_im > -1 && h_im < g->H && w_im < g->W)
                            v = FN(bilinear)(plane, g->H, g->W, h_im, w_im);
                        col[(size_t)(c * kk + t) * HWo + p] = v * m;
                    }
            }
    }
}

/* batched im2col, exported for tests: col [B, C*kk, Ho*Wo] */
void FN(dcn_ref_im2col)(const REAL *im, const REAL *off, const REAL *msk, REAL *col,
                        int B, int C, int H, int W, int kh, int kw, int sh, int sw,
                        int ph, int pw, int dh, int dw, int dg)
{
    FN(dcn_geom) g = {B, C, H, W, 0, kh, kw, sh, sw, ph, pw, dh, dw, dg, 0, 0};
    g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    const int kk = kh * kw, HWo = g.Ho * g.Wo;
    for (int b = 0; b < B; ++b)
        FN(im2col_1)(&g, im + (size_t)b * C * H * W, off + (size_t)b * dg * 2 * kk * HWo,
                     msk + (size_t)b * dg * kk * HWo, col + (size_t)b * C * kk * HWo);
}

/* out[b] = bias (+) weight_flat[Co, C*kk] @ col[b]      (dcn_v2_cuda.cu:64-94) */
void FN(dcn_ref_forward)(const REAL *im, const REAL *weight, const REAL *bias, const REAL *off,
                         const REAL *msk, REAL *out,
                         int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                         int ph, int pw, int dh, int dw, int dg)
{
    FN(dcn_geom) g = {B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, 0, 0};
    g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    const int kk = kh * kw, HWo = g.Ho * g.Wo, Kd = C * kk;
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)Kd * HWo);
    for (int b = 0; b < B; ++b) {
        FN(im2col_1)(&g, im + (size_t)b * C * H * W, off + (size_t)b * dg * 2 * kk * HWo,
                     msk + (size_t)b * dg * kk * HWo, col);
        REAL *o = out + (size_t)b * Co * HWo;
        for (int co = 0; co < Co; ++co) {
            REAL *orow = o + (size_t)co * HWo;
            for (int p = 0; p < HWo; ++p) orow[p] = 0;
            const REAL *wrow = weight + (size_t)co * Kd;
            for (int k = 0; k < Kd; ++k) {
                const REAL wv = wrow[k];
                const REAL *crow = col + (size_t)k * HWo;
                for (int p = 0; p < HWo; ++p) orow[p] += wv * crow[p];
            }
            for (int p = 0; p < HWo; ++p) orow[p] += bias[co];
        }
    }
    free(col);
}

/* grad_input scatter for one sample; colg [C*kk, Ho*Wo]; NOTE pad_h used for both axes. */
static void FN(col2im_1)(const FN(dcn_geom) *g, const REAL *colg, const REAL *off,
                         const REAL *msk, REAL *gim)
{
    const int kk = g->kh * g->kw, cpg = g->C / g->dg, HWo = g->Ho * g->Wo;
    for (int c = 0; c < g->C; ++c) {
        const int grp = c / cpg;
        REAL *gplane = gim + (size_t)c * g->H * g->W;
        const REAL *goff = off + (size_t)grp * 2 * kk * HWo;
        const REAL *gmsk = msk + (size_t)grp * kk * HWo;
        for (int i = 0; i < g->kh; ++i)
            for (int j = 0; j < g->kw; ++j) {
                const int t = i * g->kw + j;
                for (int ho = 0; ho < g->Ho; ++ho)
                    for (int wo = 0; wo < g->Wo; ++wo) {
                        const int p = ho * g->Wo + wo;
                        const int w_in = wo * g->sw - g->ph; /* reference quirk: pad_h */
                        const int h_in = ho * g->sh - g->ph;
                        const REAL dy = goff[(size_t)(2 * t) * HWo + p];
                        const REAL dx = goff[(size_t)(2 * t + 1) * HWo + p];
                        const REAL m = gmsk[(size_t)t * HWo + p];
                        const REAL fh = h_in + i * g->dh + dy;
                        const REAL fw = w_in + j * g->dw + dx;
                        const REAL top = colg[(size_t)(c * kk + t) * HWo + p] * m;
                        const int ch = (int)fh, cw = (int)fw; /* C truncation, as the reference */
                        for (int ddy = -2; ddy <= 2; ++ddy)
                            for (int ddx = -2; ddx <= 2; ++ddx) {
                                const int yy = ch + ddy, xx = cw + ddx;
                                if (yy >= 0 && yy < g->H && xx >= 0 && xx < g->W &&
                                    fabs((double)(fh - yy)) < 1 && fabs((double)(fw - xx)) < 1) {
                                    REAL wt = FN(pixel_weight)(fh, fw, yy, xx, g->H, g->W);
                                    gplane[yy * g->W + xx] += wt * top;
                                }
                            }
                    }
            }
    }
}

/* grad_offset / grad_mask for one sample */
static void FN(col2im_coord_1)(const FN(dcn_geom) *g, const REAL *colg, const REAL *im,
                               const REAL *off, const REAL *msk, REAL *goffo, REAL *gmsko)
{
    const int kk = g->kh * g->kw, cpg = g->C / g->dg, HWo = g->Ho * g->Wo;
    for (int grp = 0; grp < g->dg; ++grp) {
        const REAL *goff = off + (size_t)grp * 2 * kk * HWo;
        const REAL *gmsk = msk + (size_t)grp * kk * HWo;
        for (int oc = 0; oc < 2 * kk; ++oc) {
            const int t = oc / 2, dir = oc % 2;
            const int i = t / g->kw, j = t % g->kw;
            for (int ho = 0; ho < g->Ho; ++ho)
                for (int wo = 0; wo < g->Wo; ++wo) {
                    const int p = ho * g->Wo + wo;
                    const int h_in = ho * g->sh - g->ph, w_in = wo * g->sw - g->pw;
                    const REAL dy = goff[(size_t)(2 * t) * HWo + p];
                    const REAL dx = goff[(size_t)(2 * t + 1) * HWo + p];
                    const REAL m = gmsk[(size_t)t * HWo + p];
                    REAL val = 0, mval = 0;
                    for (int cl = 0; cl < cpg; ++cl) {
                        const int c = grp * cpg + cl;
                        const REAL *plane = im + (size_t)c * g->H * g->W;
                        const REAL cg = colg[(size_t)(c * kk + t) * HWo + p];
                        REAL fh = h_in + i * g->dh + dy;
                        REAL fw = w_in + j * g->dw + dx;
                        if (fh <= -1 || fw <= -1 || fh >= g->H || fw >= g->W) {
                            fh = fw = -2;
                        } else {
                            mval += cg * FN(bilinear)(plane, g->H, g->W, fh, fw);
                        }
                        const REAL wt = FN(coord_weight)(fh, fw, g->H, g->W, plane, dir);
                        val += wt * cg * m;
                    }
                    goffo[(size_t)(grp * 2 * kk + oc) * HWo + p] = val;
                    if (dir == 0)
                        gmsko[(size_t)(grp * kk + t) * HWo + p] = mval;
                }
        }
    }
}

/* per-sample loop of dcn_v2_cuda.cu:150-203; all grads are zero-initialised here. */
void FN(dcn_ref_backward)(const REAL *im, const REAL *weight, const REAL *bias, const REAL *off,
                          const REAL *msk, const REAL *gout, REAL *gim, REAL *goff, REAL *gmsk,
                          REAL *gweight, REAL *gbias,
                          int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                          int ph, int pw, int dh, int dw, int dg)
{
    (void)bias;
    FN(dcn_geom) g = {B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, 0, 0};
    g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    const int kk = kh * kw, HWo = g.Ho * g.Wo, Kd = C * kk;
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)Kd * HWo);
    memset(gim, 0, sizeof(REAL) * (size_t)B * C * H * W);
    memset(gweight, 0, sizeof(REAL) * (size_t)Co * Kd);
    memset(gbias, 0, sizeof(REAL) * (size_t)Co);
    memset(goff, 0, sizeof(REAL) * (size_t)B * dg * 2 * kk * HWo);
    memset(gmsk, 0, sizeof(REAL) * (size_t)B * dg * kk * HWo);
    for (int b = 0; b < B; ++b) {
        const REAL *im_n = im + (size_t)b * C * H * W;
        const REAL *off_n = off + (size_t)b * dg * 2 * kk * HWo;
        const REAL *msk_n = msk + (size_t)b * dg * kk * HWo;
        const REAL *go_n = gout + (size_t)b * Co * HWo;
        /* columns = weight_flat^T @ grad_output_n */
        for (int k = 0; k < Kd; ++k) {
            REAL *crow = col + (size_t)k * HWo;
            for (int p = 0; p < HWo; ++p) crow[p] = 0;
            for (int co = 0; co < Co; ++co) {
                const REAL wv = weight[(size_t)co * Kd + k];
                const REAL *grow = go_n + (size_t)co * HWo;
                for (int p = 0; p < HWo; ++p) crow[p] += wv * grow[p];
            }
        }
        FN(col2im_coord_1)(&g, col, im_n, off_n, msk_n, goff + (size_t)b * dg * 2 * kk * HWo,
                           gmsk + (size_t)b * dg * kk * HWo);
        FN(col2im_1)(&g, col, off_n, msk_n, gim + (size_t)b * C * H * W);
        /* grad_weight += grad_output_n @ im2col(input_n)^T ; grad_bias += rowsum(grad_output_n) */
        FN(im2col_1)(&g, im_n, off_n, msk_n, col);
        for (int co = 0; co < Co; ++co) {
            const REAL *grow = go_n + (size_t)co * HWo;
            for (int k = 0; k < Kd; ++k) {
                const REAL *crow = col + (size_t)k * HWo;
                REAL s = 0;
                for (int p = 0; p < HWo; ++p) s += grow[p] * crow[p];
                gweight[(size_t)co * Kd + k] += s;
            }
            REAL s = 0;
            for (int p = 0; p < HWo; ++p) s += grow[p];
            gbias[co] += s;
        }
    }
    free(col);
}

#undef FN
#undef CAT
#undef CAT2

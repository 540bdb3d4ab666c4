/*
 * ebfi_hip.h -- C ABI of libebfi_hip.so: the MI355X (gfx950) kernels behind the EBFI-BE
 * frame-synthesis hot path.  Plain pointers and sizes only; every pointer marked "device" is a
 * HIP device address, `stream` is a hipStream_t (NULL = the legacy default stream).
 *
 * Conventions
 *   - every entry point returns EBFI_OK (0) or a negative ebfi_status; it never throws and never
 *     allocates device memory.  ebfi_last_error() returns a thread-local description of the last
 *     failure on the calling thread.
 *   - launches are asynchronous on `stream`; the library keeps no global mutable state besides the
 *     optional profiler (ebfi_prof_*), so distinct streams/threads may call concurrently.
 *   - dtype: every floating tensor of a call is fp32 in memory (EBFI_F32), like the reference's ops; the
 *     other ebfi_dtype values select the matrix-core OPERAND precision of the convolution kernels
 *     (fp32 accumulation always).  Shapes/strides are in ELEMENTS, NCHW order of the LOGICAL dims
 *     (a channels-last tensor is passed with its real strides).
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   ebfi_fac_forward / ebfi_fac_backward
 *       pybind module `kernelconv2d_cuda`: forward / backward
 *       models/FAC/kernelconv2d/KernelConv2D_cuda.cpp:10-61, kernels KernelConv2D_kernel.cu:25-204
 *   ebfi_dcn_forward / ebfi_dcn_backward
 *       pybind module `_ext`: dcn_v2_forward / dcn_v2_backward
 *       models/DCNv2/src/dcn_v2.h:9-92, src/vision.cpp:4-9, src/cuda/dcn_v2_cuda.cu:20-216,
 *       src/cuda/dcn_v2_im2col_cuda.cu:125-402
 *   ebfi_conv2d_*               nn.Conv2d + activation inside ConvLayer (models/model_misc/submodules.py:159-200)
 *   ebfi_scale_residual_cat_*   exposure/time-scaled residual + concat of ResidualControl (model_singleframe.py:124-134)
 *   ebfi_prodmean_*             AdaptiveAvgPool2d(1) of a product of two maps (ExposureDecision, model_singleframe.py:66-68)
 *   ebfi_se_gate_*              SEGating (+ residual + ReLU / LeakyReLU) of the detail branch (models/model_misc/resnet_3D.py:89-141)
 *   ebfi_ed_head_*              GroupNorm x2 -> pooled product -> sigmoid -> scaled concat of ExposureDecision (model_singleframe.py:66-72)
 *   ebfi_groupnorm_*            nn.GroupNorm of ExposureDecision (models/Ours/model_singleframe.py:36,66-67)
 *   ebfi_census_*               Ternary census loss (loss/restore.py:108-145)
 *   ebfi_gauss5_*               GaussianConv of the Laplacian-pyramid loss (loss/restore.py:149-163)
 *   ebfi_laploss_*              whole Laplacian-pyramid L1 term of a step as one difference pyramid (loss/restore.py:166-213)
 *   ebfi_adam_step              optimizer.step() of train_ours.py:276-277 over the flat parameter buffer
 *   ebfi_grad_gather            gradient bucket packing of DistributedDataParallel (train_ours.py:754) as one launch
 *   ebfi_gather_sum             weight re-layouts of the depth-2 Conv3d / ConvTranspose3d (models/model_misc/resnet_3D.py)
 *   ebfi_events_to_stack        dataloader/encodings.py:307-350 (events_to_stack)
 *   ebfi_frame2lap / _frame2dcp myutils/utils.py:34-49 / :15-31
 */
#ifndef EBFI_HIP_H
#define EBFI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (bumped whenever an entry point is added or changed; the Python binding refuses any other value):
 *   3  round 2/3 additions (ebfi_ed_head_*, ebfi_laploss_*, ebfi_se_gate_*, ebfi_adam_step, ebfi_pack_table_bf16,
 *      ebfi_conv2d_packed_x3, the fused KernelConv -> FAC forward, ...)
 *   4  ebfi_se_gate_forward takes a workspace (ebfi_se_gate_workspace)
 *   5  (round 4) the never-implemented EBFI_BF16 storage value left ebfi_dtype; fp16 filter storage and the fused-gradient
 *      entry points of the KernelConv -> FAC training path (ebfi_fac_forward_p16 / _backward_p16, ebfi_conv2d_packed_*_c16,
 *      ebfi_conv2d_backward_weight_f16c*, ebfi_to_c16, ebfi_scale_residual_cat_*_c16)
 *   6  fp16-operand FORWARD: ebfi_pack_table_f16 writes forward weight images too; ebfi_conv2d_packed_f16_c16 takes the
 *      planar-fp16 output form (the 128 -> 1600 KernelConv of the training step)
 *   7  EpiExtra masks from fp16 images: ebfi_conv2d_packed_f16_c16 takes mask_is_c16 (the LeakyReLU derivative read from the
 *      sign of the c16 image instead of an fp32 tensor)
 *   8  ebfi_conv2d_backward_weight_f16g_ex: the weight gradient writes grad * act'(y) as a c16 image for the data gradient
 *   9  (round 5) ebfi_grad_gather (gradient packing + overflow flag in the wire buffer); ebfi_adam_step_guarded takes the
 *      all-reduced flag; ebfi_fac_*_p16 take the unpadded input (replicate padding inside); ebfi_pad2d_backward;
 *      ebfi_conv2d_packed_x3_rc / ebfi_scale_residual_cat_backward_c16a (ResidualControl's tail in the convolution's epilogue)
 *  10  ebfi_conv2d_thin_forward (3x3 layers with <= 3 output channels: taps on the matrix row axis)
 *  11  ebfi_scalar_conv_forward / _backward (ResidualControl's scalar-conditioned channel scales: a bank of 1x1 convolutions on
 *      [B,K,1,1] inputs in one launch each way); ebfi_kernelconv_fac_fused_f16 (the fused KernelConv -> FAC kernel of inference
 *      on fp16 operands)
 *  12  ebfi_se_gate_forward_ps / _backward_ps (the squeeze-excite gate of an up-convolution stage reading the transposed
 *      convolution's output through the pixel shuffle)
 *  13  ebfi_conv2d_packed_x3_shuffled / ebfi_conv2d_packed_f16_shuffled (convolutions storing through PixelShuffle(2) / its inverse)
 *  14  ebfi_to_c16_cat2 (the fp16 image of a two-part channel concatenation without the concatenated tensor) */
#define EBFI_ABI_VERSION 14

typedef enum {
    EBFI_OK = 0,
    EBFI_ERR_ARG = -1,         /* shape / stride / pointer precondition violated */
    EBFI_ERR_LAUNCH = -2,      /* HIP reported a launch error */
    EBFI_ERR_UNSUPPORTED = -3, /* dtype or configuration not implemented */
    EBFI_ERR_WORKSPACE = -4    /* workspace missing or too small */
} ebfi_status;

/* EBFI_F32_BF16MMA: fp32 tensors, operands rounded to bf16 for the matrix cores, fp32 accumulation
 * (accepted by ebfi_conv2d_backward_weight; the forward / data-gradient have *_bf16mma entry points) */
/* (value 1 was a bf16-storage mode that no entry point ever implemented: removed in ABI 5, the number stays unused) */
typedef enum { EBFI_F32 = 0, EBFI_F32_BF16MMA = 2, EBFI_F32_BF16X3MMA = 3 } ebfi_dtype;

int ebfi_abi_version(void);
const char *ebfi_last_error(void);

/* ------------------------------------------------------------------ FAC (filter-adaptive conv)
 * out[b,c,y,x] = sum_{ky,kx} input[b,c,y+ky,x+kx] * kernel[b, c*K*K + ky*K + kx, y, x]
 * input  [B, C, Ho+K-1, Wo+K-1]  (already padded by the caller, KernelConv2D.py:85-87)
 * kernel [B, C*K*K, Ho, Wo],  output [B, C, Ho, Wo].  Any strides; K >= 1.
 * Unlike the reference the outputs need NOT be pre-zeroed: every element is written. */
int ebfi_fac_forward(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                     const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                     int kernel_size,
                     void *output, const int64_t output_shape[4], const int64_t output_stride[4],
                     int dtype, void *stream);

/* grad_input [B,C,Ho+K-1,Wo+K-1], grad_kernel [B,C*K*K,Ho,Wo]; both fully overwritten.
 * Either grad pointer may be NULL to skip it. */
int ebfi_fac_backward(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                      const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                      int kernel_size,
                      const void *grad_output, const int64_t grad_output_stride[4],
                      void *grad_input, const int64_t grad_input_stride[4],
                      void *grad_kernel, const int64_t grad_kernel_stride[4],
                      int dtype, void *stream);

/* Same, with grad_kernel leaving as the gradient of the PRE-activation of the LeakyReLU(kernel_leaky_slope) layer that produced
 * the filters (KernelConv + FAC in Modification, reference models/Ours/model_singleframe.py:161-162): grad_kernel is
 * multiplied by 1 where kernel > 0 and by the slope elsewhere.  kernel_leaky_slope = 1 is ebfi_fac_backward exactly. */
int ebfi_fac_backward_ex(const void *input, const int64_t input_shape[4], const int64_t input_stride[4],
                         const void *kernel, const int64_t kernel_shape[4], const int64_t kernel_stride[4],
                         int kernel_size,
                         const void *grad_output, const int64_t grad_output_stride[4],
                         void *grad_input, const int64_t grad_input_stride[4],
                         void *grad_kernel, const int64_t grad_kernel_stride[4],
                         float kernel_leaky_slope, int dtype, void *stream);

/* ------------------------------------------------------------------ DCNv2 (modulated deformable conv)
 * All tensors contiguous NCHW:
 *   input [B,C,H,W]  weight [Co,C,kh,kw]  bias [Co]
 *   offset [B, dg*2*kh*kw, Ho, Wo]  (channel 2*(i*kw+j) = dy, +1 = dx inside each group block)
 *   mask   [B, dg*kh*kw, Ho, Wo]    output [B,Co,Ho,Wo]
 *   Ho = (H + 2*ph - (dh*(kh-1)+1))/sh + 1, likewise Wo.
 * The column tensor of the reference never exists in memory.
 * ebfi_dcn_forward accepts dtype EBFI_F32 (exact fp32 matrix cores: the kernel the reference's known-answer tests pin)
 * or EBFI_F32_BF16X3MMA (same fp32 tensors; the 64 x C*kh*kw x 64 product runs on the bf16 matrix cores in split
 * precision, ~1e-5 of the exact result).  The backward is fp32 only. */
int ebfi_dcn_forward(const void *input, const void *weight, const void *bias, const void *offset,
                     const void *mask, void *output,
                     int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                     int ph, int pw, int dh, int dw, int deformable_group,
                     int dtype, void *stream);

/* Bytes of scratch ebfi_dcn_backward needs (partial weight-gradient slabs). */
size_t ebfi_dcn_backward_workspace(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                                   int ph, int pw, int dh, int dw, int deformable_group, int dtype);

/* All five gradients are fully (over)written; grad_input is accumulated with float atomics after
 * being zeroed by the library (run-to-run differences in the last bits, like the reference's
 * atomicAdd col2im, dcn_v2_im2col_cuda.cu:249).  Reference quirk kept: grad_input uses pad_h for
 * both axes (dcn_v2_im2col_cuda.cu:368). */
int ebfi_dcn_backward(const void *input, const void *weight, const void *bias, const void *offset,
                      const void *mask, const void *grad_output,
                      void *grad_input, void *grad_offset, void *grad_mask, void *grad_weight,
                      void *grad_bias,
                      int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                      int ph, int pw, int dh, int dw, int deformable_group,
                      void *workspace, size_t workspace_bytes, int dtype, void *stream);

/* ------------------------------------------------------------------ conv + bias + activation (ConvLayer)
 * What the reference gets from nn.Conv2d + nn.LeakyReLU / nn.Sigmoid inside ConvLayer
 * (models/model_misc/submodules.py:159-200).  Contiguous NCHW fp32.
 *   input [B,Cin,H,W]  weight [Cout,Cin,k,k]  bias [Cout] or NULL  output [B,Cout,Ho,Wo]
 *   k in {1,3}, stride in {1,2}, Ho = (H + 2*pad - k)/stride + 1.
 *   act: 0 none, 1 LeakyReLU(slope), 2 Sigmoid -- fused into the epilogue. */
int ebfi_conv2d_forward(const void *input, const void *weight, const void *bias, void *output,
                        int B, int Cin, int H, int W, int Cout, int ksize, int stride, int pad,
                        int act, float slope, int dtype, void *stream);

/* grad_input = conv^T(grad_output * act'(saved_output)); stride 1, pad = k/2 only.
 * saved_output is the forward OUTPUT (post-activation); may be NULL when act == 0. */
int ebfi_conv2d_backward_data(const void *grad_output, const void *saved_output, const void *weight,
                              void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize,
                              int stride, int pad, int act, float slope, int dtype, void *stream);

size_t ebfi_conv2d_backward_weight_workspace(int B, int Cin, int H, int W, int Cout, int ksize,
                                             int stride, int pad, int dtype);

/* grad_weight [Cout,Cin,k,k] and (if non-NULL) grad_bias [Cout]; fully overwritten; deterministic
 * (per-workgroup partial slabs in `workspace`, fixed-order reduction). */
int ebfi_conv2d_backward_weight(const void *input, const void *grad_output, const void *saved_output,
                                void *grad_weight, void *grad_bias,
                                int B, int Cin, int H, int W, int Cout, int ksize, int stride, int pad,
                                int act, float slope, void *workspace, size_t workspace_bytes,
                                int dtype, void *stream);

/* Same as ebfi_conv2d_backward_weight plus an optional side output
 * grad_preact_out [B,Cout,Ho,Wo] = grad_output * act'(saved_output) (fp32 kernels only; NULL to skip):
 * the caller can then run ebfi_conv2d_backward_data on it with act = 0 and no saved activation. */
int ebfi_conv2d_backward_weight_ex(const void *input, const void *grad_output, const void *saved_output,
                                   void *grad_weight, void *grad_bias, void *grad_preact_out,
                                   int B, int Cin, int H, int W, int Cout, int ksize, int stride, int pad,
                                   int act, float slope, void *workspace, size_t workspace_bytes,
                                   int dtype, void *stream);

/* bf16 matrix-core variants: fp32 tensors in memory, operands rounded to bf16 for v_mfma_f32_32x32x16_bf16,
 * fp32 accumulation (16x the fp32 MFMA rate).  k in {1,3}, stride 1.  `workspace` receives the weight
 * re-packed to bf16 [tap][co][ci16] (ebfi_conv2d_bf16_workspace bytes: room for the hi and lo images of the
 * split-precision variants below). */
size_t ebfi_conv2d_bf16_workspace(int Cin, int Cout, int ksize);
int ebfi_conv2d_forward_bf16mma(const void *input, const void *weight, const void *bias, void *output,
                                int B, int Cin, int H, int W, int Cout, int ksize, int stride, int pad,
                                int act, float slope, void *workspace, size_t workspace_bytes, void *stream);
int ebfi_conv2d_backward_data_bf16mma(const void *grad_output, const void *saved_output, const void *weight,
                                      void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize,
                                      int stride, int pad, int act, float slope,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* Split-precision ("bf16x3") variants: every fp32 operand is carried as hi = bf16(v), lo = bf16(v - hi) and each
 * product is accumulated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (fp32 accumulation).  Results agree with
 * the exact fp32 kernels to ~1e-5 relative (tests/test_gpu_conv.py), at 3/16 of their matrix-core time.  Same
 * arguments, workspace (ebfi_conv2d_bf16_workspace) and limits as the *_bf16mma functions. */
int ebfi_conv2d_forward_bf16x3(const void *input, const void *weight, const void *bias, void *output,
                               int B, int Cin, int H, int W, int Cout, int ksize, int stride, int pad,
                               int act, float slope, void *workspace, size_t workspace_bytes, void *stream);
int ebfi_conv2d_backward_data_bf16x3(const void *grad_output, const void *saved_output, const void *weight,
                                     void *grad_input, int B, int Cin, int H, int W, int Cout, int ksize,
                                     int stride, int pad, int act, float slope,
                                     void *workspace, size_t workspace_bytes, void *stream);

/* Data gradient of a 7x7 STRIDE-2 convolution with at most 16 input channels (the detail branch's stem,
 * models/model_misc/resnet_3D.py BasicStem folded to 2-D: 6 <- 64 channels) from grad_preact = grad_output * act'(output),
 * split precision, evaluated per output parity class (no zero-inserted gradient tensor).  pad = 3. */
int ebfi_conv2d_backward_data_s2_bf16x3(const void *grad_preact, const void *weight, void *grad_input, int B, int Cin,
                                        int H, int W, int Cout, int ksize, int pad, void *stream);

/* Weights packed ahead of the call.  ebfi_conv2d_forward_bf16x3 / ebfi_conv2d_backward_data_bf16x3 re-pack `weight` into
 * `workspace` on every call; when `weight` is NULL they take `workspace` as ALREADY holding the packed images
 * ([hi | lo], bf16 [tap][M][K16]: forward M = Cout, K = Cin; data gradient `transposed`: M = Cin, K = Cout, taps flipped;
 * ebfi_conv2d_packed_bytes bytes) -- weights change once per optimiser step, not once per launch.
 * ebfi_conv2d_pack_bf16x3 packs one weight; ebfi_pack_table_bf16 packs ANY number of weights in one launch from a host-built
 * table: packed element e = bf16 of src[table[e] & 0x3fffffff] (its rounding remainder when bit 30 is set, 0 when
 * table[e] < 0), which also carries folded / concatenated weight layouts (ebfi_amd/weightbank.py). */
size_t ebfi_conv2d_packed_bytes(int Cin, int Cout, int ksize, int transposed);
int ebfi_conv2d_pack_bf16x3(const void *weight, int Cin, int Cout, int ksize, int transposed, void *packed,
                            size_t packed_bytes, void *stream);
int ebfi_pack_table_bf16(const float *src, const int32_t *table, int64_t n, void *out, void *stream);

/* 3x3, stride 1, padding 1, Cout <= 3, 16 <= Cin <= 64, >= 64 K output pixels (the model's last convolution 64 -> 3 + sigmoid,
 * ExposureDecision's 64 -> 1; reference model_misc/submodules.py:159-200 ConvLayer): Q[(co, tap)][p] = sum_ci w[co][ci][tap] x[ci][p]
 * on the matrix cores (27 of 32 rows used, K = Cin; the input's channel fragments straight from global memory, split into
 * bf16 hi / lo in registers: three products per k-step like every forward convolution here), then out = act(bias + sum_tap
 * Q shifted by the tap) from LDS.  `weight` is the fp32 [Cout, Cin, 3, 3] tensor.  Any other shape: EBFI_ERR_UNSUPPORTED without an
 * error record -- the caller takes ebfi_conv2d_forward_bf16x3 instead. */
int ebfi_conv2d_thin_forward(const void *input, const void *weight, const void *bias, void *output, int B, int Cin, int H, int W,
                             int Cout, int ksize, int stride, int pad, int act, float slope, void *stream);

/* Split-precision convolution on packed weight images with the options hand-scheduled layer chains need
 * (ebfi_amd/rc_fused.py; ResidualControl, reference models/Ours/model_singleframe.py:115-136):
 *   groups      grouped convolution: input [B, groups*Cin_per_group, H, W], output channel co reads the input channels
 *               of group co / (Cout/groups); Cout/groups a multiple of 64 when groups > 1
 *   addend      [B,Cout,Ho,Wo] added before the activation (NULL: none)
 *   mask_y      [B,Cout,Ho,Wo]: the result is multiplied by act'(mask_y) for activation mask_act / mask_slope (NULL: none)
 * out = act(conv(input) + bias + addend) * act'(mask_y); stride 1.  With transposed images it is the data gradient of a
 * layer; addend + mask then turn its output into the PRE-activation gradient of the layer below.
 * Restriction (since ABI 5): with addend or mask_y, `act` and `mask_act` must be none or LeakyReLU -- a sigmoid there returns
 * EBFI_ERR_UNSUPPORTED (the epilogue with extras compiles ONE activation form, LeakyReLU with slope 1 = none; the model never
 * combines a sigmoid layer with them).  Without addend / mask_y every activation is accepted. */
int ebfi_conv2d_packed_x3(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                          int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                          float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope, void *stream);
/* KernelConv -> FAC as ONE kernel (inference).  Replaces, for the pair
 *   filters = LeakyReLU(KernelConv(cat))                 reference models/Ours/model_singleframe.py:161 (ConvLayer 3x3, 2C -> C*K*K)
 *   out     = KernelConv2D(K)(feat, filters)             :162, models/FAC/kernelconv2d/KernelConv2D.py:77-87 (ReplicationPad2d(K//2)
 *                                                        + kernelconv2d_cuda.forward, KernelConv2D_kernel.cu:25-53)
 * the conv launch, the [B, C*K*K, H, W] filter tensor and the FAC launch: the filters exist only in the accumulators of
 * the conv's workgroups, whose epilogue applies them to the replicate-clamped `feat`.
 *   input   [B, Cin, H, W] fp32 (cat([feat, frame_feat]) in the reference model)
 *   packed  split-precision forward images ([hi | lo], bf16 [tap][C*32][Cin16]) of the conv weight re-tiled to 32 rows per
 *           FAC channel: row c*32 + t = weight row c*K*K + t for t < K*K, zero otherwise (ebfi_amd/weightbank.py "facrows")
 *   bias32  [C*32] fp32, same row layout (zeros in the pad rows)
 *   feat    [B, C, H, W] fp32, output [B, C, H, W] fp32; K = 5; W % 4 == 0. */
int ebfi_kernelconv_fac_fused_x3(const void *input, const void *packed, size_t packed_bytes, const void *bias32,
                                 const void *feat, void *output, int B, int Cin, int H, int W, int C, int fac_ksize,
                                 float slope, void *stream);

/* The same fused pair on fp16 operands (round 6; ABI 11): one matrix-core product per tap.  `input` fp32 NCHW, multiplied by
 * in_slot[0] (a power of two the caller sets from the tensor right before the launch) while it is staged, |max| recorded in
 * in_slot -- or, input_is_c16 = 1, the c16 image of that tensor written by ebfi_to_c16 with in_slot's scale (Cin % 16 == 0: half
 * the bytes per staged chunk; every one of the C / 2 output-channel blocks stages the whole input); `packed16`: the "facrows" fp16 image [tap][C*32][K16] scaled by w_slot[0] (ebfi_pack_table_f16).  Replaces the same
 * reference lines as ebfi_kernelconv_fac_fused_x3 (model_singleframe.py:161-162, KernelConv2D.py:82-87,
 * KernelConv2D_kernel.cu:25-53) for inference. */
int ebfi_kernelconv_fac_fused_f16(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                                  const void *bias32, const void *feat, void *output, int B, int Cin, int H, int W, int C,
                                  int fac_ksize, float slope, void *in_slot, const void *w_slot, void *stream);
/* ------------------------------------------------------------------ fp16 single-product backward (training step)
 * The data gradient and the weight gradient of the 3x3 layers with ONE fp16 MFMA per product (the forward keeps the
 * split-precision kernels: DESIGN.md section 4).  Every operand is scaled by a power of two kept in a device SLOT of
 * 64 floats (256 bytes): slot[0] = scale, slot[32] = running |max| of the values staged through it (separate cache lines);
 * kernels apply slot[0] and atomically raise slot[32]; slot i of a book starts at slots + 64 i;
 * ebfi_f16_scales_finish (once per step, after the backward pass) turns the maxima of all `n` slots into the next step's
 * scales (|max| * scale in [2, 4)), clears them, and sets flag[0] when a value was not finite or |max| * scale could
 * have left the fp16 range.  A new slot is initialised by the caller (ebfi_amd/f16scale.py calibrates it just in time).
 *   ebfi_pack_table_f16      fp16 weight images from the index table of ebfi_pack_table_bf16 (hi entries); images start on
 *                            256-element boundaries, elements [256 k, 256 k + 256) are scaled by slot block_slot[k]
 *   ebfi_conv2d_packed_f16   ebfi_conv2d_packed_x3 on an fp16 image (3x3, pad 1, W % 4 == 0): out = act(conv / (in_scale *
 *                            w_scale) + bias + addend) * act'(mask_y); with transposed images: the data gradient
 *   ebfi_conv2d_backward_weight_f16g   ebfi_conv2d_backward_weight_x3g / _ex (act'(saved_output) folded, optional
 *                            grad_preact_out) with fp16 operands; Cin_per_group a multiple of 64 */
int ebfi_f16_scales_finish(void *slots, int n, void *flag, void *stream);
int ebfi_pack_table_f16(const float *src, const int32_t *table, int64_t n, void *out, const int32_t *block_slot,
                        void *slots, void *stream);
int ebfi_conv2d_packed_f16(const void *input, const void *packed16, size_t packed_bytes, const void *bias, void *output,
                           int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                           float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                           void *in_slot, const void *w_slot, void *stream);
int ebfi_conv2d_backward_weight_f16g(const void *input, const void *grad_output, const void *saved_output,
                                     void *grad_weight, void *grad_bias, void *grad_preact_out, int B,
                                     int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                     float slope, void *x_slot, void *g_slot, void *workspace,
                                     size_t workspace_bytes, void *stream);
/* _ex: grad_preact_is_c16 != 0 writes grad_preact_out (grad_output * act'(saved_output)) as a c16 image scaled by g_slot's scale
 * (see "fp16 operand STORAGE" below) for ebfi_conv2d_packed_f16_c16 to stage; needs an activation, Cout % 16 == 0, 3x3 / pad 1,
 * W % 4 == 0 and 16-byte aligned tensors (EBFI_ERR_UNSUPPORTED otherwise) */
int ebfi_conv2d_backward_weight_f16g_ex(const void *input, const void *grad_output, const void *saved_output,
                                        void *grad_weight, void *grad_bias, void *grad_preact_out, int grad_preact_is_c16, int B,
                                        int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                                        float slope, void *x_slot, void *g_slot, void *workspace,
                                        size_t workspace_bytes, void *stream);
/* ------------------------------------------------------------------ fp16 operand STORAGE of the backward pass (round 4)
 * Layout "c16": a tensor [B, C, H, W] (C % 16 == 0, W % 4 == 0) as fp16 [B][C/16][H][2][W][8] -- 16-channel blocks; an image
 * row holds the channels 0..7 of its W pixels, then the channels 8..15 -- multiplied by the power-of-two scale in slot[0] of
 * its scale slot (ebfi_f16_scales_finish).  It is the staging
 * layout of both fp16 backward kernels: their producers copy 16-byte pieces instead of converting fp32 NCHW planes, half
 * the bytes.  The WRITER of an image applies the scale and records |max| into the slot; readers only read the scale.
 * Replaces, inside ResidualControl's backward chain (models/Ours/model_singleframe.py:115-136), the fp32 tensors autograd
 * would hand from layer to layer.
 *   ebfi_to_c16                        fp32 [B,C,HW] -> image (optionally times LeakyReLU'(mask_y): a pre-activation gradient)
 *   ebfi_conv2d_packed_x3_c16          ebfi_conv2d_packed_x3 writing its output also as an image (out16, slot16)
 *   ebfi_conv2d_packed_f16_c16         ebfi_conv2d_packed_f16 reading an image (input_is_c16 = 1; 2 = planar fp16; scale in in_slot) and / or
 *                                      writing its output as one (out16 / slot16; `output` may then be NULL); mask_is_c16: mask_y is
 *                                      the c16 image of the mask tensor (its signs are read) instead of the fp32 tensor
 *   ebfi_conv2d_backward_weight_f16c   weight / bias gradient from the images of the input and of the pre-activation gradient
 *   ebfi_scale_residual_cat_forward_c16 / _backward_c16   the fused ResidualControl stages writing images: forward out + out16;
 *                                      backward [grad_a0 | grad_a1] * LeakyReLU'(a) as ONE image of 2C channels, grad_x fp32,
 *                                      scale gradients as per-slice partial sums [slices][B][C] (summed in order by the caller) */
int ebfi_to_c16(const float *src, const float *mask_y, float mask_slope, void *dst16, void *slot, int B, int C, int H, int W,
                void *stream);
/* Round 6 (ABI 14): the image of cat([src0, src1], 1) -- [B, C0 + C1, H, W] -- written from its two parts: the concatenated fp32 tensor
 * of Modification (`torch.cat([ev, FrameTensor], 1)`, model_singleframe.py:159-160) is never materialised when its only consumer, the
 * 128 -> 1600 KernelConv of the training step, reads the image.  C0, C1 multiples of 8, C0 + C1 a multiple of 16; same bits as
 * ebfi_to_c16 of the concatenation. */
int ebfi_to_c16_cat2(const float *src0, int C0, const float *src1, int C1, void *dst16, void *slot, int B, int H, int W, void *stream);
int ebfi_conv2d_packed_x3_c16(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                              int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups, int act,
                              float slope, const void *addend, const void *mask_y, int mask_act, float mask_slope,
                              void *out16, void *slot16, int out16_planar, void *stream);
int ebfi_conv2d_packed_f16_c16(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                               const void *bias, void *output, int B, int Cin_per_group, int H, int W, int Cout, int ksize,
                               int pad, int groups, int act, float slope, const void *addend, const void *mask_y,
                               int mask_act, float mask_slope, void *in_slot, const void *w_slot, void *out16,
                               void *slot16, int out16_planar, int mask_is_c16, void *stream);
/* Round 6 (ABI 13): the two packed convolutions with the fp32 output stored THROUGH PixelShuffle(2) (out_layout 1: `output` is
 * [B, Cout/4, 2H, 2W], channel 4c + 2py + px of pixel (y, x) at (c, 2y + py, 2x + px); Cout % 4 == 0) or through its inverse
 * (out_layout 2: [B, 4*Cout, H/2, W/2]; H, W even); 0 = the plain layout.  3x3 same-padded layers on the wave-specialised kernels
 * (W % 4 == 0, Cout > 32, 16-byte aligned input), no groups, no addend, no fp16 side output.  Replaces the PixelShuffle copy of the
 * reconstruction head (models/Ours/model_singleframe.py:257-260: conv 64 -> 256, nn.PixelShuffle(2), LeakyReLU) and its backward:
 *   forward   ebfi_conv2d_packed_x3_shuffled(.., act = LeakyReLU, out_layout 1) writes what the next layer reads;
 *   backward  ebfi_conv2d_packed_f16_shuffled on that next layer's TRANSPOSED fp16 image with mask_y = that layer's own input (the
 *             shuffled activation, natural layout of this launch), out_layout 2: the pre-activation gradient of the 64 -> 256
 *             convolution in ITS layout.  mask_y / mask_act / mask_slope as in ebfi_conv2d_packed_f16_c16 (fp32 mask tensor). */
int ebfi_conv2d_packed_x3_shuffled(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output,
                                   int B, int Cin, int H, int W, int Cout, int act, float slope, int out_layout, void *stream);
int ebfi_conv2d_packed_f16_shuffled(const void *input, int input_is_c16, const void *packed16, size_t packed_bytes,
                                    const void *bias, void *output, int B, int Cin, int H, int W, int Cout, int act, float slope,
                                    const void *mask_y, int mask_act, float mask_slope, void *in_slot, const void *w_slot,
                                    int out_layout, void *stream);
int ebfi_conv2d_backward_weight_f16c(const void *input16, const void *grad16, int grad_is_planar, void *grad_weight,
                                     void *grad_bias, int B, int Cin_per_group, int H, int W, int Cout, int groups,
                                     const void *x_slot, const void *g_slot, void *workspace, size_t workspace_bytes,
                                     void *stream);
/* n (1..4) weight gradients over the SAME [B, H, W] pixels as ONE launch + one reduction (the three layers of a ResidualControl
 * round, models/Ours/model_singleframe.py:127-133): arguments are arrays of n entries; every layer must be exactly two 64 x 64
 * blocks (128 x 64, 64 x 128, or 2 groups of 64 x 64) -- otherwise EBFI_ERR_UNSUPPORTED and the caller uses the per-layer entry.
 * A third of the partial-sum slab traffic and of the start-up per layer. */
size_t ebfi_conv2d_backward_weight_f16c_batch_workspace(int n, const int *Cin_per_group, const int *Cout);
int ebfi_conv2d_backward_weight_f16c_batch(int n, const void *const *input16, const void *const *grad16, void *const *grad_weight,
                                           void *const *grad_bias, const int *Cin_per_group, const int *Cout, const int *groups,
                                           void *const *x_slot, void *const *g_slot, int B, int H, int W, void *workspace,
                                           size_t workspace_bytes, void *stream);
/* The 1600-channel tensors of the KernelConv -> FAC pair in the TRAINING step (reference
 * models/Ours/model_singleframe.py:161-162, KernelConv2D_kernel.cu:25-150): the filters and grad_kernel as PLANAR fp16
 * [B, C*K*K, Ho, Wo] scaled by a slot's power of two -- the layout the FAC kernels stream plane by plane.  The 128 -> 1600
 * convolution writes the filters in that form (ebfi_conv2d_packed_x3_c16 with out16_planar, output NULL), the FAC forward /
 * backward read them and the backward writes grad_kernel (times the LeakyReLU derivative) likewise; the weight / data
 * gradient of the convolution stage the planes (ebfi_conv2d_backward_weight_f16c grad_is_planar, ebfi_conv2d_packed_f16_c16
 * input_is_c16 = 2).  No fp32 [B, 1600, h, w] tensor is written or read. */
/* input_is_unpadded (round 5): `input` (and grad_input) are the UNPADDED [B, C, Ho, Wo] tensors and the replicate padding of
 * KernelConv2D.py:82-86 (nn.ReplicationPad2d(K / 2)) is applied inside the kernels -- clamped reads in the forward; in the
 * backward clamped reads plus the padding's ADJOINT folded into grad_input with a fixed summation order (no atomics: torch's
 * replication_pad2d_backward adds atomically).  0: `input` is the padded [B, C, Ho+K-1, Wo+K-1] tensor as before. */
int ebfi_fac_forward_p16(const float *input, int input_is_unpadded, const void *filters16, const void *f_slot, float *output, int B,
                         int C, int Ho, int Wo, int K, void *stream);
int ebfi_fac_backward_p16(const float *input, int input_is_unpadded, const void *filters16, const void *f_slot,
                          const float *grad_output, float *grad_input, void *grad_kernel16, void *g_slot, float kernel_leaky_slope,
                          int B, int C, int Ho, int Wo, int K, void *stream);
int ebfi_scale_residual_cat_forward_c16(const float *a0, const float *s0, const float *a1, const float *s1, const float *x,
                                        float *out, void *out16, void *slot, int B, int C, int H, int W,
                                        int64_t a_batch_stride, void *stream);
int ebfi_scale_residual_cat_backward_slices(void);
int ebfi_scale_residual_cat_backward_c16(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                         const float *s1, void *grad_a16, void *slot, float *grad_x, float *grad_s0_part,
                                         float *grad_s1_part, int B, int C, int H, int W, int64_t a_batch_stride,
                                         float mask_slope, void *stream);
/* ... with `a` as the c16 image of [a0 | a1] (2C channels, scale in a_slot) written by ebfi_conv2d_packed_x3_rc (round 5). */
int ebfi_scale_residual_cat_backward_c16a(const float *grad_out, const void *a16, const void *a_slot, const float *s0, const float *s1,
                                          void *grad_a16, void *slot, float *grad_x, float *grad_s0_part, float *grad_s1_part, int B,
                                          int C, int H, int W, float mask_slope, void *stream);
/* The grouped second-layer convolution of a ResidualControl round with the round's tail in its epilogue
 * (models/Ours/model_singleframe.py:127-133): a = LeakyReLU(conv3x3(input) + bias); pre16 = c16 image of a (scale / |max| in
 * pre_slot); output[b, co] = a * post_scale[b, co] + post_res[b, co % res_channels] -- the exposure- / time-scaled residual and
 * concatenation `cat(s_ex * a0 + x, s_t * a1 + x)` -- and out16 = c16 image of output.  Replaces the separate
 * ebfi_scale_residual_cat_forward_c16 launch; no fp32 `a` is written.  3x3, same padding, 64-channel blocks, W % 4 == 0.
 * Inference form: pre16, pre_slot, out16 and slot16 all NULL -- only `output` is written (replaces the launch of
 * ebfi_scale_residual_cat_forward_ex after the convolution, and the fp32 `a` between them). */
int ebfi_conv2d_packed_x3_rc(const void *input, const void *packed, size_t packed_bytes, const void *bias, void *output, int B,
                             int Cin_per_group, int H, int W, int Cout, int groups, float slope, const void *post_scale,
                             const void *post_res, int res_channels, void *pre16, void *pre_slot, void *out16, void *slot16,
                             void *stream);
/* weight / bias gradient of such a (grouped) convolution from a pre-activation gradient: grad_weight
 * [Cout, Cin_per_group, k, k]; workspace as ebfi_conv2d_backward_weight_workspace(B, Cin_per_group, H, W, Cout, k, 1, pad) */
int ebfi_conv2d_backward_weight_x3g(const void *input, const void *grad_output, void *grad_weight, void *grad_bias,
                                    int B, int Cin_per_group, int H, int W, int Cout, int ksize, int pad, int groups,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ event voxel binning
 * xs, ys, ts: float64[n] device (ts sorted, normalised as h5dataset.py:334), ps: float32[n].
 * out: float32 [2, bins, H, W], fully overwritten (index 0 = positive, 1 = negative counts).
 * Bit-exact with events_to_stack incl. its shared-edge and out-of-range behaviour.
 * workspace: ebfi_events_workspace(bins) bytes. */
size_t ebfi_events_workspace(int bins);
int ebfi_events_to_stack(const double *xs, const double *ys, const double *ts, const float *ps,
                         int64_t n, int bins, int H, int W, float *out,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ blur-level maps
 * frame: float32 [B,3,H,W] contiguous in [0,1]; out: float32 [B,1,H,W]. */
int ebfi_frame2lap(const float *frame, float *out, int B, int H, int W, void *stream);
int ebfi_frame2dcp(const float *frame, float *out, float *scratch /* [B,H,W] */, int B, int H, int W,
                   int window, void *stream);

/* ------------------------------------------------------------------ scalar-conditioned channel scales
 * ResidualControl's Conv1[i](Ex) / Conv2[i](T) (models/Ours/model_singleframe.py:85-94, :127-129: ConvLayer(k=1) + LeakyReLU on a
 * [B,K,1,1] input, reference submodules.py:159-200) for a BANK of S such layers in one launch:
 *   out[s][b][c] = leaky_relu(bias_s[c] + sum_k v[b][k] * weight_s[c][k], slope)        v [B,K], weight_s [C,K], out [S,B,C]
 * weights / biases: host arrays of S device pointers (the layers' own parameter tensors; bias entries may be NULL); they are
 * copied into the kernel arguments, so a captured launch keeps them.  S <= 32, K <= 8, S*B*C <= 2^20.
 * backward: grad_weight [S,C,K], grad_bias [S,C], grad_v [B,K] (each may be NULL), all fully overwritten, fixed summation
 * order (bit-reproducible); `out` is the forward's output (its sign selects the LeakyReLU derivative). */
int ebfi_scalar_conv_forward(const float *v, const void *const *weights, const void *const *biases, float *out, int S, int B,
                             int K, int C, float slope, void *stream);
int ebfi_scalar_conv_backward(const float *v, const void *const *weights, const float *out, const float *grad_out,
                              float *grad_weight, float *grad_bias, float *grad_v, int S, int B, int K, int C, float slope,
                              void *stream);

/* ------------------------------------------------------------------ fused stages between the convolutions
 * One round of ResidualControl (models/Ours/model_singleframe.py:124-134):
 *   out[:, :C] = s0[b,c] * a0 + x,  out[:, C:] = s1[b,c] * a1 + x      (a0/a1/x [B,C,H,W], s0/s1 [B,C], out [B,2C,H,W])
 * backward is its adjoint (grad_s* reduced per plane in fixed order).  HW = H*W, a multiple of 4.
 * x == NULL drops the residual (then grad_x == NULL), s1 == NULL means s1 = 1 (then a1 / grad_s1 are not read / written
 * by the backward): cat([atten * ev, bl]) of ExposureDecision (model_singleframe.py:68-72) is the same stage. */
int ebfi_scale_residual_cat_forward(const float *a0, const float *s0, const float *a1, const float *s1, const float *x,
                                    float *out, int B, int C, int64_t HW, void *stream);
int ebfi_scale_residual_cat_backward(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                     const float *s1, float *grad_a0, float *grad_a1, float *grad_x,
                                     float *grad_s0, float *grad_s1, int B, int C, int64_t HW, void *stream);

/* The same stage with a0 / a1 (and grad_a0 / grad_a1) given as channel slices of wider [B, *, H, W] tensors (batch strides
 * in elements, multiples of 4).  mask_leaky != 0: a0 / a1 are LeakyReLU(mask_slope) outputs and grad_a0 / grad_a1 leave
 * multiplied by that activation's derivative, i.e. as gradients of the PRE-activations. */
int ebfi_scale_residual_cat_forward_ex(const float *a0, const float *s0, const float *a1, const float *s1, const float *x,
                                       float *out, int B, int C, int64_t HW, int64_t a_batch_stride, void *stream);
int ebfi_scale_residual_cat_backward_ex(const float *grad_out, const float *a0, const float *s0, const float *a1,
                                        const float *s1, float *grad_a0, float *grad_a1, float *grad_x, float *grad_s0,
                                        float *grad_s1, int B, int C, int64_t HW, int64_t a_batch_stride,
                                        int64_t grad_a_batch_stride, int mask_leaky, float mask_slope, void *stream);

/* out[plane] = mean over HW of a*b for contiguous [planes, HW] maps (AdaptiveAvgPool2d(1) of a product: the event / blur
 * correlation of ExposureDecision, model_singleframe.py:66-68) and its adjoint.  HW a multiple of 4. */
int ebfi_prodmean_forward(const float *a, const float *b, float *out, int64_t planes, int64_t HW, void *stream);
int ebfi_prodmean_backward(const float *a, const float *b, const float *grad_out, float *grad_a, float *grad_b,
                           int64_t planes, int64_t HW, void *stream);

/* Sparse 0/1 linear map: out[i] = sum_{r<R} src[idx[i*R+r]], negative indices skipped.  Carries the weight
 * re-layouts that turn the depth-2 Conv3d / ConvTranspose3d of the detail branch (models/model_misc/resnet_3D.py,
 * model_singleframe.py:170-223) into 2-D convolutions, and their adjoints. */
int ebfi_gather_sum(const float *src, const int32_t *idx, float *out, int64_t n_out, int R, void *stream);

/* ------------------------------------------------------------------ squeeze-excite gate of the detail branch
 * SEGating (models/model_misc/resnet_3D.py:89-105) fused with what follows it: out = act(x * sigmoid(W mean(x) + b) (+ res)).
 * x, res, out: [B*C planes][N] contiguous fp32 (a [B,C,D,H,W] tensor as it stands), N % 4 == 0, B*C <= 4096;
 * weight [C,C] (the 1x1x1 conv), bias [C] or NULL; act: 0 none, 1 LeakyReLU(slope) (slope 0 = ReLU).
 * forward writes mean [B*C] and gate [B*C] for the backward; both directions take a workspace of
 * ebfi_se_gate_workspace(B, C, N) floats (slice sums of the plane reductions); grad_res / grad_bias may be NULL; `out` may be
 * NULL when act == 0 (backward).  Two launches each way, deterministic (fixed-order reductions). */
size_t ebfi_se_gate_workspace(int B, int C, int64_t N);
int ebfi_se_gate_forward(const float *x, const float *weight, const float *bias, const float *res, float *out, float *mean,
                         float *gate, float *workspace, int B, int C, int64_t N, int act, float slope, void *stream);
int ebfi_se_gate_backward(const float *grad_out, const float *out, const float *x, const float *weight, const float *gate,
                          const float *mean, float *grad_x, float *grad_res, float *grad_weight, float *grad_bias,
                          float *workspace, int B, int C, int64_t N, int act, float slope, void *stream);

/* Round 6 (ABI 12): the same gate behind a transposed convolution whose output has NOT been pixel-shuffled yet -- upConv3D of the
 * detail branch (model_singleframe.py:200-223, resnet_3D / model_3DUnet upConv3D: ConvTranspose3d (3,4,4)/(1,2,2) -> SEGating ->
 * LeakyReLU), run here as a 3x3 convolution to 8*C channels.  x: that convolution's output [B, 8*C, h, w] (channel = (c, d, py, px));
 * out / grad_out: [B, C, 2, 2h, 2w]; grad_x in x's layout.  The gate kernels address x through the shuffle: the PixelShuffle copy
 * of the stage's largest tensor (forward and backward) never runs.  w even; no residual input. */
int ebfi_se_gate_forward_ps(const float *x, const float *weight, const float *bias, float *out, float *mean, float *gate,
                            float *workspace, int B, int C, int h, int w, int act, float slope, void *stream);
int ebfi_se_gate_backward_ps(const float *grad_out, const float *out, const float *x, const float *weight, const float *gate,
                             const float *mean, float *grad_x, float *grad_weight, float *grad_bias, float *workspace, int B, int C,
                             int h, int w, int act, float slope, void *stream);

/* ------------------------------------------------------------------ GroupNorm (exposure-decision head)
 * nn.GroupNorm(groups, C) on contiguous NCHW fp32 (reference models/Ours/model_singleframe.py:36,66-67).
 * HW = H*W must be a multiple of 4.  mean / rstd [B*groups] are written by forward and read by backward.
 * workspace: ebfi_groupnorm_workspace(B, C) bytes for forward, that + 2*B*groups*4 for backward.
 * gamma / beta / grad_gamma / grad_beta may be NULL.  Deterministic (fixed-order double-precision sums). */
size_t ebfi_groupnorm_workspace(int B, int C);
int ebfi_groupnorm_forward(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                           int B, int C, int64_t HW, int groups, float eps,
                           void *workspace, size_t workspace_bytes, void *stream);
int ebfi_groupnorm_backward(const float *grad_y, const float *x, const float *gamma, const float *mean,
                            const float *rstd, float *grad_x, float *grad_gamma, float *grad_beta,
                            int B, int C, int64_t HW, int groups,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ExposureDecision head (models/Ours/model_singleframe.py:66-72) for contiguous [B,C,H,W] maps ev, bl and ONE GroupNorm
 * (gamma, beta [C], `groups`, eps) applied to both:
 *   atten[B*C] = sigmoid(mean_HW(GN(ev) * GN(bl)));   out[B,2C,H,W] = cat([ev * atten, bl], 1).
 * GN is affine per plane, so the pooled product follows from five plane moments: one pass over the maps, no normalised
 * map is written.  forward also writes `stats` (B*C*5 + B*groups*4 doubles: plane means of ev, ev^2, bl, bl^2, ev*bl,
 * then (mean, rstd) of both maps per group) for backward, which returns the gradients of the whole head
 *   grad_ev, grad_bl [B,C,H,W], grad_gamma, grad_beta [C] (may be NULL)
 * from grad_out [B,2C,H,W] in one reduction + one elementwise pass (closed form in the moments).
 * HW % 4 == 0, C % groups == 0, C <= 1024; workspace: ebfi_ed_head_workspace(B, C, HW) bytes.  Deterministic. */
size_t ebfi_ed_head_workspace(int B, int C, int64_t HW);
int ebfi_ed_head_forward(const float *ev, const float *bl, const float *gamma, const float *beta, float *out, float *atten,
                         double *stats, int B, int C, int64_t HW, int groups, float eps,
                         void *workspace, size_t workspace_bytes, void *stream);
int ebfi_ed_head_backward(const float *grad_out, const float *ev, const float *bl, const float *gamma, const float *beta,
                          const float *atten, const double *stats, float *grad_ev, float *grad_bl, float *grad_gamma,
                          float *grad_beta, int B, int C, int64_t HW, int groups,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ loss image operators
 * 5x5 binomial blur ([1,4,6,4,1]/16 twice, times `factor`) with reflect padding over `planes` contiguous
 * H x W planes: the GaussianConv of the reference's Laplacian-pyramid loss (loss/restore.py:149-163).
 * backward is its exact adjoint.  H, W >= 3. */
int ebfi_gauss5_forward(const float *input, float *output, int64_t planes, int H, int W, float factor, void *stream);
int ebfi_gauss5_backward(const float *grad_output, float *grad_input, int64_t planes, int H, int W, float factor,
                         void *stream);

/* Census (Ternary, 7x7) loss of loss/restore.py:108-145 on [B,C,H,W] fp32 images, gray = channel mean.
 * forward writes ebfi_census_partials(B,H,W) per-tile sums; loss = sum(partial) / (B*H*W) (summed by the caller
 * in fixed order -> deterministic).  backward: grad_x = grad_loss[0] * d loss / d x; y is a constant (the
 * reference detaches the target transform, restore.py:138). */
int64_t ebfi_census_partials(int B, int H, int W);
int ebfi_census_forward(const float *x, const float *y, float *partial, int B, int C, int H, int W, void *stream);
int ebfi_census_backward(const float *x, const float *y, const float *grad_loss, float *grad_x,
                         int B, int C, int H, int W, void *stream);

/* Laplacian-pyramid L1 term of the training loss for up to two predictions against one target
 * (loss/restore.py:166-213 LaplacianPyramid + LaplacianLoss: sum_i 2^i * L1sum(lap_i(pred), lap_i(target)), `levels` = 5;
 * combined for (Sharp, SharpPre) with the coefficients of train_ours.py:258-268).  The pyramid operators are linear,
 * so the library builds ONE pyramid of [pred_a - target ; pred_b - target] (planes_per_term = B*C planes of H x W each;
 * pred_b may be NULL).  forward fills `workspace` (ebfi_laploss_workspace_floats(planes, ...) floats, planes = 1 or 2
 * x planes_per_term) and writes ebfi_laploss_partials(planes, ...) block sums:
 *   loss = coef_a * Lap(pred_a, target) + coef_b * Lap(pred_b, target) = sum(partial)   (summed by the caller, fixed order).
 * backward turns the workspace left by forward (consumed in place: once per forward) into
 *   grad_pred[planes, H, W] = grad_loss[0] * d loss / d [pred_a ; pred_b].
 * H, W multiples of 2^(levels-1), >= 3 pixels on the coarsest blurred level, 2 <= levels <= 8. */
int64_t ebfi_laploss_workspace_floats(int64_t planes, int H, int W, int levels);
int64_t ebfi_laploss_partials(int64_t planes, int H, int W, int levels);
int ebfi_laploss_forward(const float *pred_a, const float *pred_b, const float *target, float coef_a, float coef_b,
                         float *workspace, float *partial, int64_t planes_per_term, int H, int W, int levels, void *stream);
int ebfi_laploss_backward(const float *grad_loss, float *workspace, float *grad_pred, int64_t planes, int H, int W,
                          int levels, void *stream);

/* Adjoints of nn.ReflectionPad2d(pad) (the detail branch's output conv: ReflectionPad2d(3) + 7x7 convolution,
 * models/Ours/model_singleframe.py:207; replicate = 0) and nn.ReplicationPad2d(pad) (the FAC module,
 * models/FAC/kernelconv2d/KernelConv2D.py:82-86; replicate = 1): grad_padded [planes, H+2*pad, W+2*pad] -> grad_input
 * [planes, H, W] as a gather in a fixed order (bit-reproducible; torch's backward of either pad accumulates with atomics).
 * Reflection needs pad < H, W. */
int ebfi_pad2d_backward(const float *grad_padded, float *grad_input, int64_t planes, int H, int W, int pad, int replicate, void *stream);

/* Adam update (torch.optim.Adam of train_ours.py:276-277, amsgrad / weight decay off as in config/train_ours.yml:59-65)
 * over one flat fp32 buffer of n elements, in place: exp_avg <- exp_avg + (1-beta1)(grad - exp_avg);
 * exp_avg_sq <- beta2 exp_avg_sq + (1-beta2) grad^2; param <- param - lr/(1-beta1^t) * exp_avg / (sqrt(exp_avg_sq)/sqrt(1-beta2^t) + eps)
 * with t = step[0] (a device scalar holding the ALREADY incremented step count: no host synchronisation).
 * All buffers 16-byte aligned. */
int ebfi_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *step, int64_t n,
                   double lr, double beta1, double beta2, double eps, void *stream);
/* Same update, guarded: when guard[0] != 0 (set by ebfi_f16_scales_finish: an fp16 operand of this step's backward pass
 * left its range) nothing is updated, step[0] is decremented again and guard[1] counts the skipped step.  guard may be NULL.
 * flag (optional, needs guard): one device float, the ranks' guard flags summed by the gradient all-reduce (ebfi_grad_gather
 * wrote this rank's into the wire buffer): anything but an exact 0 skips the step too and sets guard[0] = 1, so every rank
 * of a data-parallel job takes the same branch without a collective of its own for the flag. */
int ebfi_adam_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, int64_t n,
                           double lr, double beta1, double beta2, double eps, int *guard, const float *flag, void *stream);

/* Gradient packing of a training step (replaces the bucket copy of DistributedDataParallel, train_ours.py:754, and the
 * 255-piece concatenation of earlier rounds): `grads` / `numels` are HOST arrays of `count` device pointers (NULL: that
 * parameter has no gradient, its range is zero-filled) and element counts (sum = total); tensor k is copied to
 * flat[sum of numels[0..k-1] ...].  Then flat[total] = (guard && guard[0] != 0) ? 1 : 0 and flat[total + 1 .. total + pad - 1]
 * = 0 (1 <= pad <= 256: the wire buffer's trailer).  The pointers travel by value in the kernel arguments, 128 per launch:
 * a captured launch replays with the addresses the capture saw (graph-pool allocations keep them). */
int ebfi_grad_gather(const void *const *grads, const int64_t *numels, int count, float *flat, int64_t total, int pad,
                     const int *guard, void *stream);


/* ------------------------------------------------------------------ per-kernel device timing
 * When enabled, every launch made by this library is bracketed by a hipEvent pair recorded on the
 * launch stream.  ebfi_prof_collect() must be called after the stream(s) are synchronised; it
 * folds the pending pairs into per-kernel totals and frees their slots (totals keep accumulating until
 * ebfi_prof_reset), so a long run collects once per step.  Pending slots between two collects are bounded
 * (default EBFI_PROF_MAX_PENDING, ebfi_prof_set_capacity changes it); launches beyond the bound are not
 * timed and are counted in *dropped -- callers must treat dropped != 0 as a failed measurement.
 * Labels are "<kernel symbol>" or "<kernel symbol>/<role>" when one kernel serves several ops
 * (e.g. "conv_fwd_bf16x3_db/fwd" and "conv_fwd_bf16x3_db/dgrad"). */
#define EBFI_PROF_MAX_PENDING 65536
void ebfi_prof_enable(int on);
int ebfi_prof_set_capacity(int max_pending);
void ebfi_prof_reset(void);
int ebfi_prof_collect(int *dropped);
int ebfi_prof_num_kernels(void);
int ebfi_prof_get(int index, const char **name, int64_t *launches, double *total_ms);
/* algorithmic work (SURVEY.md 8(d) formulas) summed over the timed launches of kernel `index` */
int ebfi_prof_get_work(int index, double *flops, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* EBFI_HIP_H */

/*
 * cfun_hip.h -- C ABI of libcfun_hip.so: MI355X (gfx950) kernels for CFUN's volumetric hot path.
 *
 * The reference (Wuziyi616/CFUN) has no native code and no FFI: every entry point below replaces a
 * stock torch / numpy call made by the reference's Python on the hot path (file:line cited per
 * function, paths relative to the reference root).  INTEGRATION.md shows the ctypes stub a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs of int32/int64/float; no torch / HIP types
 *     (`cfun_stream_t` is a hipStream_t passed as void*; NULL = the null stream);
 *   - every pointer is DEVICE memory owned by the caller (incl. workspaces); the library never
 *     allocates, frees or keeps a pointer after return;
 *   - activations are fp32, NDHWC ("channels last 3d"), dense; weights are the packed layouts
 *     described at cfun_conv3d_fwd;
 *   - every call only enqueues on `stream` and returns; no hidden synchronisation;
 *   - return 0 on success, a hipError_t (>0) or CFUN_E* (<0) otherwise; never throws;
 *   - semantic "errors" follow the reference: a degenerate RoI gives zeros (model.py:281-287),
 *     empty inputs give empty outputs.
 */
#ifndef CFUN_HIP_H
#define CFUN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cfun_stream_t;

#define CFUN_VERSION 100          /* 0.1.0 */
#define CFUN_OK 0
#define CFUN_EINVAL (-1)          /* bad argument / unsupported shape */
#define CFUN_EWORKSPACE (-2)      /* workspace too small */
#define CFUN_EALIGN (-3)          /* pointer not 16-byte aligned where required */

#define CFUN_ACT_NONE 0
#define CFUN_ACT_RELU 1
#define CFUN_ACT_LRELU 2

#define CFUN_ALGO_AUTO 0
#define CFUN_ALGO_DIRECT 1        /* generic VALU direct convolution (any shape) */
#define CFUN_ALGO_MFMA 2          /* LDS-tiled implicit GEMM on v_mfma_f32_16x16x4_f32 (Ci%4==0, Co%4==0) */
#define CFUN_ALGO_WINO 4          /* as AUTO, but every supported 3x3x3 stride-1 conv runs the F(2,3)-along-x MFMA kernel */
#define CFUN_ALGO_WINO2 5         /* as WINO with y in the Winograd domain as well (F(2x2,3x3) per z tap) in forward / dgrad */

int cfun_version(void);
const char* cfun_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Direct 3-D convolution, NDHWC fp32, im2col-free, fused epilogue.
 * Replaces nn.Conv3d (+ BatchNorm3d(eval) + ReLU / residual add) of backbone.py:14-23,38-54,123-128,
 * model.py:131-134,713-717 and every bias-free Conv3d of mask_branch.py:23-88,91-122, plus the
 * nn.Upsample(scale_factor=2) that feeds a conv (mask_branch.py:112,120) and F.upsample at
 * model.py:144 (as the x2-nearest residual).
 *
 *   y[n,zo,yo,xo,co] = act( s * sum_{tap,ci} X[n, zo*stride+dz-pd, ..., ci] * W[tap][ci][co] + t + r )
 *     X      = x, or nearest-x2 upsample of x when up2 (x is stored at Di,Hi,Wi; X has 2*Di,...)
 *     s      = scale[co] (scale_mode 1) | scale[n*Co+co] (scale_mode 2: Dropout3d channel mask) | 1
 *     t      = shift[co] (bias and/or folded BatchNorm) | 0
 *     r      = res[...] (res_mode 1) read at (zo>>1,yo>>1,xo>>1) when res_up2 | 0
 *   bwd_data / bwd_weight take g = dL/d(conv sum) in the layout of y: [N,Do,Ho,Wo,Co], or for d2s convs the
 *   high-resolution [N,2Do,2Ho,2Wo,Cq] tensor (the kernels gather the parities themselves).
 *
 * Packed weights: wp[tap][ci][CoP] fp32, tap = (dz*kh+dy)*kw+dx, CoP = Co rounded up to 16, pad = 0.
 * bwd_data takes the transposed pack wpT[tap][co][CiP] (same tap order, CiP = Ci rounded up to 16).
 * bwd_weight writes dwp in the wp layout (pad columns are written as zeros).
 * ---------------------------------------------------------------------------------------------- */
typedef struct CfunConv3dParams {
  int32_t N;
  int32_t Di, Hi, Wi, Ci;       /* stored input */
  int32_t Do, Ho, Wo, Co;       /* output */
  int32_t CoP, CiP;             /* padded row lengths of wp / wpT */
  int32_t kd, kh, kw;
  int32_t stride;               /* same on the three axes (1 or 2) */
  int32_t pd, ph, pw;
  int32_t up2;
  int32_t act;
  float slope;                  /* LeakyReLU negative slope */
  int32_t scale_mode;
  int32_t has_shift;
  int32_t res_mode;             /* 0 none, 1 add before the activation */
  int32_t res_up2;
  int32_t d2s;                  /* 1: depth-to-space x2 epilogue: Co = 8*Cq, channel (pz,py,px,o) of low-res voxel
                                   (z,y,x) is written to y[n, 2z+pz, 2y+py, 2x+px, o]; y is [N,2Do,2Ho,2Wo,Cq];
                                   the residual (if any) is [N,Do,Ho,Wo,Cq] = nearest-x2 up-sampled into y.
                                   Used to run "nearest x2 upsample -> 5x5x5 conv" (mask_branch.py:118-122) as a
                                   3x3x3 conv with parity-folded weights: 27/125 of the FLOPs, same result. */
  int32_t d2s_cq;               /* d2s: valid channels per parity (channels of y); 0 = Co/8.  Co/8 is the padded
                                   per-parity column stride of the packed weights. */
  int32_t tap_skip;             /* d2s + 3x3x3: the weights are a parity-folded "nearest x2 -> 3x3x3" kernel
                                   (ops.fold_up2_weight): only the 2x2x2 taps {p,p+1}^3 of each output parity are
                                   non-zero and the kernels skip the rest (8/27 of the FLOPs). */
  int32_t algo;                 /* CFUN_ALGO_* */
  int32_t w_prepared;           /* bit 0: the `wp` given to cfun_conv3d_fwd* is the forward operand of cfun_weight_prepare
                                   (for Winograd shapes the transformed U, not the plain pack); bit 1: the `wpT` given to
                                   cfun_conv3d_bwd_data is its data-gradient operand.  0: plain packs (the kernels transform
                                   them per call). */
} CfunConv3dParams;

/* ws: cfun_conv3d_fwd_workspace_bytes(p) bytes (split-K partials for volumes too small to fill the chip);
 * ws may be NULL -- the kernel then runs unsplit. */
size_t cfun_conv3d_fwd_workspace_bytes(const CfunConv3dParams* p);
/* Which kernel family cfun_conv3d_fwd runs for p when given that workspace (labels for profiles / bench.py). */
#define CFUN_KERNEL_DIRECT 0      /* generic VALU direct conv (conv3d_direct.hip) */
#define CFUN_KERNEL_MFMA 1        /* k_conv_mfma: implicit GEMM, every tap on the fp32 matrix cores */
#define CFUN_KERNEL_WINO 2        /* k_conv_wino: 3x3x3 stride 1, x axis in the Winograd F(2,3) domain (2/3 of the MFMAs) */
#define CFUN_KERNEL_STEM 3        /* k_conv_stem*: C_in = 1, HBM-bound */
#define CFUN_KERNEL_POINTWISE 4   /* k_conv_pointwise: 1x1x1 -> 8 channels, streaming */
int cfun_conv3d_fwd_kernel(const CfunConv3dParams* p);
/* For CFUN_KERNEL_WINO launches: out = {2-D (y in the Winograd domain too: 4/9 of the MFMAs) ? 1 : 0, 16-channel subtiles
 * per output-channel tile, two-waves-per-SIMD loop (2-D tiles of 16 / 32 channels) ? 1 : 0, output-channel columns computed
 * (a multiple of 16 >= C_out)}.  Returns 0, or CFUN_EINVAL if p does not run on the Winograd kernels.  (Labels and the
 * executed-MFMA count of bench.py; not needed to call the conv.) */
int cfun_conv3d_wino_plan(const CfunConv3dParams* p, int32_t out[4]);
int cfun_conv3d_fwd(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                    float* y, const CfunConv3dParams* p, void* ws, size_t ws_bytes, cfun_stream_t stream);
/* InstanceNorm3d(affine=False) + LeakyReLU folded into the convolutions around it (mask_branch.py:23-25,91-116: every
 * "conv -> [Dropout3d] -> InstanceNorm3d -> LeakyReLU -> conv" chain of the U-Net).  All hooks are optional:
 *   out_stats  [N][Cy][2] {mean, rstd} of the conv's OUTPUT y per (sample, channel), biased variance + out_eps (what
 *              cfun_instnorm_stats computes in a pass of its own), taken from the epilogue that writes y: per-tile fp32
 *              sums combined in fp64 in a fixed order (deterministic).  Cy = Co, or d2s_cq for depth-to-space outputs.
 *   in_stats   [N][Ci][2] {mean, rstd}: the conv reads in_act((x - mean) * rstd) in place of x, applied while the input
 *              tile is staged (zero padding stays zero) -- the normalised tensor is never written to memory.
 *   in_act     CFUN_ACT_* applied to the (normalised) input; in_slope its LeakyReLU slope.  in_act without in_stats
 *              folds a plain LeakyReLU (mask_branch.py:18) into the consumer.
 * cfun_conv3d_fused_support(p): which hooks the kernel that runs p implements (CFUN_FUSE_* bits; 0 = none, the caller
 * then keeps the separate cfun_instnorm_* passes).  Workspace: cfun_conv3d_fwd_fused_workspace_bytes. */
typedef struct CfunConvFusion {
  const float* in_stats;
  int32_t in_act;
  float in_slope;
  float* out_stats;
  float out_eps;
} CfunConvFusion;
#define CFUN_FUSE_OUT_STATS 1
#define CFUN_FUSE_IN_NORM 2       /* in_stats / in_act in cfun_conv3d_fwd_fused */
#define CFUN_FUSE_IN_NORM_WGRAD 4 /* ... and in cfun_conv3d_bwd_weight_fused (the weight gradient needs the same input) */
int cfun_conv3d_fused_support(const CfunConv3dParams* p);
size_t cfun_conv3d_fwd_fused_workspace_bytes(const CfunConv3dParams* p, const CfunConvFusion* f);
int cfun_conv3d_fwd_fused(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                          float* y, const CfunConv3dParams* p, const CfunConvFusion* f, void* ws, size_t ws_bytes,
                          cfun_stream_t stream);
/* The weight gradient of a conv whose forward ran with in_stats / in_act: x is the RAW tensor the forward read, the
 * kernel applies the same prologue while staging it (only f->in_* are used).  oidhw: 0 = dw in the packed layout of
 * cfun_conv3d_bwd_weight, 1 = torch's OIDHW (cfun_conv3d_bwd_weight_oidhw).  Same workspace as the plain calls. */
int cfun_conv3d_bwd_weight_fused(const float* x, const float* g, float* dw, int32_t oidhw, const CfunConv3dParams* p,
                                 const CfunConvFusion* f, void* ws, size_t ws_bytes, cfun_stream_t stream);
/* dx[n,zi,yi,xi,ci] (stored-input resolution) = sum over outputs/taps that read it of g * W. g = dL/d(conv sum). */
size_t cfun_conv3d_bwd_data_workspace_bytes(const CfunConv3dParams* p);
int cfun_conv3d_bwd_data(const float* g, const float* wpT, float* dx, const CfunConv3dParams* p, void* ws,
                         size_t ws_bytes, cfun_stream_t stream);
size_t cfun_conv3d_bwd_weight_workspace_bytes(const CfunConv3dParams* p);
int cfun_conv3d_bwd_weight(const float* x, const float* g, float* dwp, const CfunConv3dParams* p, void* ws,
                           size_t ws_bytes, cfun_stream_t stream);
/* Same gradient (same kernels, same workspace, bit-identical sums) delivered in torch's OIDHW layout
 * dw [Co, Ci, kd, kh, kw] -- what `nn.Conv3d.weight.grad` holds (backbone.py:14-23, mask_branch.py:23-89): the
 * reduction of the per-chunk partial sums and the packed -> OIDHW transposition are one pass. */
int cfun_conv3d_bwd_weight_oidhw(const float* x, const float* g, float* dw, const CfunConv3dParams* p, void* ws,
                                 size_t ws_bytes, cfun_stream_t stream);

/* g = dy * act'(y) * s  -- the epilogue's derivative (y is the saved conv output); also the gradient of `res`.
 * `vox_per_n` = Do*Ho*Wo (only used by scale_mode 2). */
int cfun_act_bwd(const float* y, const float* dy, const float* scale, float* g, int64_t nvox, int32_t C,
                 int64_t vox_per_n, int32_t act, float slope, int32_t scale_mode, cfun_stream_t stream);
/* out[c] = sum over voxels of g[v,c]  (bias gradients).  ws: cfun_channel_sum_workspace_bytes. */
size_t cfun_channel_sum_workspace_bytes(int64_t nvox, int32_t C);
int cfun_channel_sum(const float* g, float* out, int64_t nvox, int32_t C, void* ws, size_t ws_bytes,
                     cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Classifier head GEMMs (model.py:750-784): Conv3d(C -> fc, kernel = pool size) on a pool-sized input, the 1x1x1 conv
 * and the two nn.Linear layers are all  y[R][O] = act(scale[o] * (x[R][K] . w[O][K]^T) + shift[o])  with few rows
 * (R <= 64 RoIs) and, for conv1, a weight that dominates the model (K = C*pd*ph*pw = 221 184, 113 MB): it is streamed
 * once, in the checkpoint's OIDHW layout (w = conv1.weight viewed [O][K], x = the RoI-aligned features [R][C][pd][ph][pw]
 * viewed [R][K]).  K % 4 == 0; x, w, dw, dx 16-byte aligned; scale / shift [O] or NULL; act NONE or RELU.
 * Replaces F.conv3d / F.linear + batch_norm + relu of model.py:767-779.  Deterministic (fixed summation order).
 * ---------------------------------------------------------------------------------------------- */
size_t cfun_fc_workspace_bytes(int32_t R, int32_t K, int32_t O);
int cfun_fc_fwd(const float* x, const float* w, const float* scale, const float* shift, float* y, int32_t R, int32_t K,
                int32_t O, int32_t act, void* ws, size_t ws_bytes, cfun_stream_t stream);
/* dw[O][K] = g[R][O]^T . x[R][K]   (g = dL/d(x.w^T), i.e. already multiplied by act' and scale: cfun_act_bwd).
 * The kernel keeps g and its x slice in LDS: R <= cfun_fc_bwd_weight_max_rows(O) (12+ RoIs at every O of the path; 64
 * rows x O = 1024, the reference's default FC width at inference batch sizes, do not fit 160 KB) -- CFUN_EINVAL beyond
 * that; callers split the rows and add the partial gradients (cfun_amd.ops._FC does). */
int32_t cfun_fc_bwd_weight_max_rows(int32_t O);
int cfun_fc_bwd_weight(const float* x, const float* g, float* dw, int32_t R, int32_t K, int32_t O, cfun_stream_t stream);
/* dx[R][K] = g[R][O] . w[O][K] */
int cfun_fc_bwd_data(const float* g, const float* w, float* dx, int32_t R, int32_t K, int32_t O, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / normalisation (mask_branch.py:18,28,34,40,46,57,94,99,110,115; backbone.py:56,78,89).
 * ---------------------------------------------------------------------------------------------- */
int cfun_lrelu_fwd(const float* x, float* y, int64_t n, float slope, cfun_stream_t stream);
int cfun_lrelu_bwd(const float* x, const float* dy, float* dx, int64_t n, float slope, cfun_stream_t stream);
/* Channel-strided forms (C % 4 == 0, strides in floats, % 4 == 0): the rows of C channels live inside a wider NDHWC
 * buffer -- the producers of a concatenation write straight into its halves and the consumers of its gradient read
 * straight out of them (zero-copy torch.cat, mask_branch.py:184-206).  x / dx dense unless a stride is given. */
int cfun_lrelu_fwd_strided(const float* x, float* y, int64_t nvox, int32_t C, int64_t x_stride, int64_t y_stride,
                           float slope, cfun_stream_t stream);
int cfun_lrelu_bwd_strided(const float* x, const float* dy, float* dx, int64_t nvox, int32_t C, int64_t dy_stride,
                           float slope, cfun_stream_t stream);
/* dx = lrelu'(x) * dy + add: x also feeds a second branch (a residual add, mask_branch.py:131-176; the level-1 concat,
 * mask_branch.py:203) whose gradient `add` (dense, or NULL) is summed in the same pass instead of by autograd's own.
 * dy may be channel-strided (dy_stride floats per voxel row; = C when dense).  C % 4 == 0. */
int cfun_lrelu_bwd_add(const float* x, const float* dy, const float* add, float* dx, int64_t nvox, int32_t C,
                       int64_t dy_stride, float slope, cfun_stream_t stream);
int cfun_add(const float* a, const float* b, float* out, int64_t n, cfun_stream_t stream);
/* lo[n,z,y,x,c] = sum of the 8 children of hi (backward of nearest x2 upsampling). lo dims D,H,W. */
int cfun_upsample2_bwd(const float* hi, float* lo, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C,
                       cfun_stream_t stream);

/* InstanceNorm3d(affine=False, eps, biased variance) fused with LeakyReLU.
 * stats[n,c] = {mean, rstd}.  x is [N, V voxels, C].  Two-stage deterministic fp64 reduction. */
size_t cfun_instnorm_workspace_bytes(int32_t N, int64_t V, int32_t C);
int cfun_instnorm_stats(const float* x, float* stats, int32_t N, int64_t V, int32_t C, float eps, void* ws,
                        size_t ws_bytes, cfun_stream_t stream);
int cfun_instnorm_lrelu_fwd(const float* x, const float* stats, float* y, int32_t N, int64_t V, int32_t C,
                            float slope, cfun_stream_t stream);
int cfun_instnorm_lrelu_bwd(const float* x, const float* stats, const float* dy, float* dx, int32_t N, int64_t V,
                            int32_t C, float slope, void* ws, size_t ws_bytes, cfun_stream_t stream);
int cfun_instnorm_lrelu_fwd_strided(const float* x, const float* stats, float* y, int32_t N, int64_t V, int32_t C,
                                    int64_t y_stride, float slope, cfun_stream_t stream);
int cfun_instnorm_lrelu_bwd_strided(const float* x, const float* stats, const float* dy, float* dx, int32_t N, int64_t V,
                                    int32_t C, int64_t dy_stride, float slope, void* ws, size_t ws_bytes,
                                    cfun_stream_t stream);
/* ... + add (dense [N,V,C] or NULL): the second gradient of x when x also feeds a residual (see cfun_lrelu_bwd_add). */
int cfun_instnorm_lrelu_bwd_add(const float* x, const float* stats, const float* dy, const float* add, float* dx, int32_t N,
                                int64_t V, int32_t C, int64_t dy_stride, float slope, void* ws, size_t ws_bytes,
                                cfun_stream_t stream);
/* cfun_instnorm_lrelu_bwd in two halves, for a volume depth-sharded over several GPUs: the
 * means of gn = dy * lrelu'(xhat) and gn * xhat over the LOCAL voxels -> means [N,C,2]; the caller combines them across
 * the ranks (local voxel count weights, one all-reduce of 2*C floats per sample) and applies
 * dx = rstd * (gn - means[0] - xhat * means[1]).  `stats` must already be the GLOBAL (mean, rstd). */
int cfun_instnorm_bwd_means(const float* x, const float* stats, const float* dy, float* means, int32_t N, int64_t V,
                            int32_t C, int64_t dy_stride, float slope, void* ws, size_t ws_bytes, cfun_stream_t stream);
int cfun_instnorm_lrelu_bwd_apply(const float* x, const float* stats, const float* means, const float* dy, float* dx,
                                  int32_t N, int64_t V, int32_t C, int64_t dy_stride, float slope, cfun_stream_t stream);

/* MaxPool3d(kernel 2, stride 2) (backbone.py:127).  idx[v,c] = argmax child 0..7 (uint8). */
int cfun_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int32_t N, int32_t Do, int32_t Ho, int32_t Wo,
                      int32_t C, cfun_stream_t stream);
int cfun_maxpool2_bwd(const float* dy, const uint8_t* idx, float* dx, int32_t N, int32_t Do, int32_t Ho, int32_t Wo,
                      int32_t C, cfun_stream_t stream);

/* Inference tail: utils.unmold_mask (utils.py:443-460: F.interpolate(mode='trilinear', align_corners=False) of the
 * detection's class probabilities to its box, pasted into a zero volume) fused with the class arg-max of
 * unmold_detections (model.py:1853-1858).  probs [md,mh,mw,C] fp32 (device), box = HOST int32[6] (z1,y1,x1,z2,y2,x2)
 * inside the volume, out [D,H,W] uint8 class ids (0 outside the box). */
int cfun_unmold_argmax(const float* probs, uint8_t* out, int32_t D, int32_t H, int32_t W, int32_t md, int32_t mh,
                       int32_t mw, int32_t C, const int32_t* box, cfun_stream_t stream);

/* LiTS fork inference tail: the overlap-tile utils.unmold_mask (LiTS_2017/utils.py:383-408) -- every detection's
 * class probabilities resized (trilinear, align_corners=False) to its own box, summed into the volume in detection
 * order, divided by (hit count + 1e-6), clipped to [0,1] -- fused with the class arg-max of unmold_detections
 * (LiTS_2017/model.py:1828-1829).  probs [n,md,mh,mw,C] fp32 (device; C in {2,3,8}), boxes = HOST int32[n*6]
 * (z1,y1,x1,z2,y2,x2) inside the volume, n <= 64.  labels [D,H,W] uint8 and/or full [D,H,W,C] fp32 (either may be
 * NULL).  n == 0 yields zeros. */
int cfun_unmold_overlap(const float* probs, const int32_t* boxes, int32_t n, uint8_t* labels, float* full, int32_t D,
                        int32_t H, int32_t W, int32_t md, int32_t mh, int32_t mw, int32_t C, cfun_stream_t stream);

/* LiTS fork mask losses (LiTS_2017/model.py:907-979).
 * Weighted cross entropy: nn.CrossEntropyLoss(weight = w) (model.py:926, w = [1, 1, 100]) on logits [nvox, C] and uint8
 * labels: loss = sum w[y] * (-log softmax[y]) / sum w[y]; wsum receives sum w[y] (device float, needed by the backward).
 * weights: device float[C].  ws: cfun_ce_weighted_workspace_bytes(). */
size_t cfun_ce_weighted_workspace_bytes(void);
int cfun_softmax_ce_weighted_fwd(const float* logits, const uint8_t* labels, const float* weights, float* loss,
                                 float* wsum, int64_t nvox, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream);
int cfun_softmax_ce_weighted_bwd(const float* logits, const uint8_t* labels, const float* weights, const float* gscale,
                                 const float* wsum, float* dlogits, int64_t nvox, int32_t C, cfun_stream_t stream);
/* Edge loss of the fork (model.py:936-979): MSE between the RAW three Sobel responses of predicted probabilities
 * [n,D,H,W,C] and target labels, classes 1..C-1, summed over RoIs and classes, / n.  dc (cfun_edge_raw_dc_bytes, may be
 * NULL when no backward follows) keeps the response differences; cfun_edge_raw_bwd turns them into dL/dprobs. */
size_t cfun_edge_raw_dc_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C);
int cfun_edge_raw_fwd(const float* probs, const uint8_t* labels, float* loss, float* dc, int32_t n, int32_t D, int32_t H,
                      int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream);
int cfun_edge_raw_bwd(const float* dc, const float* gscale, float* dprobs, int32_t n, int32_t D, int32_t H, int32_t W,
                      int32_t C, cfun_stream_t stream);

/* GT mask targets of detection_target_layer (model.py:481-493, utils.py:318-339) as uint8 class labels:
 * labels [D,H,W] (class id per voxel = argmax of the one-hot GT), bounds [R,6] int32 voxel crop
 * (z1,y1,x1,z2,y2,x2) = int(shape * normalised coordinate), out [R,md,mh,mw] = nearest-resized crop. */
int cfun_mask_target_labels(const uint8_t* labels, const int32_t* bounds, uint8_t* out, int32_t R, int32_t D,
                            int32_t H, int32_t W, int32_t md, int32_t mh, int32_t mw, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 3-D RoIAlign = crop + trilinear(align_corners=True) resize (model.py:265-289, utils.py:160-174).
 * fm [D,H,W,C]; boxes [R,6] normalised (z1,y1,x1,z2,y2,x2); out [R,pd,ph,pw,C];
 * bounds [R,6] int32 receives the integer crop (floor lo / ceil hi, python-slice clamped) -- the
 * bit-exact part of the contract.  Empty crop => zeros.
 * ---------------------------------------------------------------------------------------------- */
int cfun_roi_align3d_fwd(const float* fm, const float* boxes, float* out, int32_t* bounds, int32_t R, int32_t D,
                         int32_t H, int32_t W, int32_t C, int32_t pd, int32_t ph, int32_t pw, cfun_stream_t stream);
/* Every element of dfm is written (a gather over the RoIs in index order: no atomics, run-to-run reproducible). */
int cfun_roi_align3d_bwd(const float* dout, const int32_t* bounds, float* dfm, int32_t R, int32_t D, int32_t H,
                         int32_t W, int32_t C, int32_t pd, int32_t ph, int32_t pw, cfun_stream_t stream);
/* The same on ONE depth slab of a depth-sharded map: fm / dfm hold planes [z0, z0 + dl) of the
 * [D,H,W,C] map; the crop bounds come from the whole map's size, planes outside the slab count as zeros.  RoIAlign is
 * linear in the map, so the ranks' partial [R,pd,ph,pw,C] results ADD UP to the RoIAlign of the whole map: one
 * all-reduce of the pooled crops (10.6 MB for 12 RoIs) replaces the all-gather of p2 / p3 (66 MB at 512x512x256). */
int cfun_roi_align3d_slab_fwd(const float* fm, const float* boxes, float* out, int32_t* bounds, int32_t R, int32_t D,
                              int32_t H, int32_t W, int32_t C, int32_t z0, int32_t dl, int32_t pd, int32_t ph, int32_t pw,
                              cfun_stream_t stream);
int cfun_roi_align3d_slab_bwd(const float* dout, const int32_t* bounds, float* dfm, int32_t R, int32_t D, int32_t H,
                              int32_t W, int32_t C, int32_t z0, int32_t dl, int32_t pd, int32_t ph, int32_t pw,
                              cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy 3-D NMS (utils.py:122-157 + compute_iou utils.py:50-70): fp32 IoU in numpy's operation
 * order (no FMA contraction), iou > threshold suppresses, stops after max_num picks.
 * boxes [n,6], scores [n], n <= 4096.  keep [max(n,1)] int32 (pick order), count [1] int32.
 * Ties in score are ordered higher-original-index first.
 * ---------------------------------------------------------------------------------------------- */
size_t cfun_nms3d_workspace_bytes(int32_t n);
int cfun_nms3d(const float* boxes, const float* scores, int32_t n, float threshold, int32_t max_num, int32_t* keep,
               int32_t* count, void* ws, size_t ws_bytes, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Mask-head losses on NDHWC logits [n,V,C] with uint8 class labels [n,V].
 *   softmax            model.py:794,799
 *   cross-entropy      model.py:909-935 (target = argmax of the one-hot GT, mean over n*V)
 *   Sobel edge loss    model.py:938-981 (valid 3x3x3, channels 0,1,0 magnitude, MSE, sum over classes 1..C-1, / n)
 * ---------------------------------------------------------------------------------------------- */
int cfun_softmax_fwd(const float* logits, float* probs, int64_t nvox, int32_t C, cfun_stream_t stream);
int cfun_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, int64_t nvox, int32_t C,
                     cfun_stream_t stream);
size_t cfun_loss_workspace_bytes(int64_t nvox);
/* loss[0] = mean over voxels of -log softmax(logits)[label] */
int cfun_softmax_ce_fwd(const float* logits, const uint8_t* labels, float* loss, int64_t nvox, int32_t C, void* ws,
                        size_t ws_bytes, cfun_stream_t stream);
/* dlogits = gscale[0] * (softmax(logits) - onehot(label)) / nvox */
int cfun_softmax_ce_bwd(const float* logits, const uint8_t* labels, const float* gscale, float* dlogits,
                        int64_t nvox, int32_t C, cfun_stream_t stream);
/* probs [n,D,H,W,C], labels [n,D,H,W]; loss[0] = (1/n) sum_{i,j>=1} mean_{valid voxels} (|grad p| - |grad t|)^2 */
int cfun_edge_loss_fwd(const float* probs, const uint8_t* labels, float* loss, int32_t n, int32_t D, int32_t H,
                       int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream);
size_t cfun_edge_loss_bwd_workspace_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C);
int cfun_edge_loss_bwd(const float* probs, const uint8_t* labels, const float* gscale, float* dprobs, int32_t n,
                       int32_t D, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream);

/* Backward of BOTH mask losses in one pass over the hi-res tensors (the 'finetune' stage, model.py:909-981):
 *   dlogits = g_ce[0]/nvox * (probs - onehot(label)) + softmax_bwd(probs, d edge_loss / d probs * g_edge[0])
 * probs = softmax(logits) as produced by cfun_softmax_fwd; ws as for cfun_edge_loss_bwd. */
int cfun_mask_losses_bwd(const float* probs, const uint8_t* labels, const float* g_ce, const float* g_edge,
                         float* dlogits, int32_t n, int32_t D, int32_t H, int32_t W, int32_t C, void* ws,
                         size_t ws_bytes, cfun_stream_t stream);

/* Training variant: the forward also stores the unit-gradient coefficient field dc
 * (cfun_edge_loss_bwd_workspace_bytes(n,D,H,W,C) bytes, 2*(C-1) floats per valid voxel), and the backward of both
 * mask losses is then ONE pass (no second Sobel march over the probabilities). */
int cfun_edge_loss_fwd_save(const float* probs, const uint8_t* labels, float* loss, float* dc, int32_t n, int32_t D,
                            int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream);
int cfun_mask_losses_bwd_saved(const float* probs, const uint8_t* labels, const float* g_ce, const float* g_edge,
                               const float* dc, float* dlogits, int32_t n, int32_t D, int32_t H, int32_t W, int32_t C,
                               cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight layout conversion between the reference's state-dict layout and the kernels' packs (one launch
 * each; replaces the permute / pad / transpose glue around nn.Conv3d.weight):
 *   pack:       w  OIDHW [Co][Ci][T]  ->  wp  [T][Ci][CoP]   (pad columns = 0)
 *   transpose:  wp [T][Ci][CoP]       ->  wpT [T][Co][CiP]   (pad columns = 0; the pack bwd_data takes)
 *   unpack:     dwp [T][Ci][CoP]      ->  dw  OIDHW [Co][Ci][T]   (the weight gradient back in state-dict layout)
 * T = kd*kh*kw taps in (dz,dy,dx) order; CoP / CiP = Co / Ci rounded up to 16.
 * ---------------------------------------------------------------------------------------------- */
int cfun_weight_pack(const float* w, float* wp, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream);
int cfun_weight_pack_transpose(const float* wp, float* wpT, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream);

/* Round 6: the 'finetune' mask losses as ONE forward and ONE backward pass (loss_fused.hip).  Replaces, for the reference's
 * Mask.forward softmax (model.py:799) + compute_mrcnn_mask_loss (model.py:909-935) + compute_mrcnn_mask_edge_loss
 * (model.py:938-981), the three forward passes cfun_softmax_fwd / cfun_softmax_ce_fwd / cfun_edge_loss_fwd_save and the
 * 27-neighbour gather cfun_mask_losses_bwd_saved:
 *   fwd: logits [n,D,H,W,C], labels uint8 [n,D,H,W] -> probs [n,D,H,W,C] = softmax(logits), losses[0] = cross entropy,
 *        losses[1] = Sobel edge loss.  u: NULL (forward only), or a cfun_mask_fused_u_bytes buffer that receives the
 *        backward's operand: per output column (y,x) in [0,H-2) x [0,W-2) and input plane z in [0,D), 2(C-1) floats =
 *        the z part of the transposed Sobel stencil applied to d(edge loss)/d(Sobel responses) for an upstream gradient of 1.
 *        ws: cfun_mask_fused_workspace_bytes().
 *   bwd: dlogits = g2[0] * dCE/dlogits + g2[1] * dEdge/dlogits: a 2-D stencil over u per plane + the softmax backward.
 * C in {8, 3}, D, H, W >= 3 (cfun_mask_fused_supported); C % 4 == 0 needs 16-byte aligned tensors. */
size_t cfun_mask_fused_workspace_bytes(void);
int cfun_mask_fused_supported(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C);
size_t cfun_mask_fused_u_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C);
int cfun_mask_fused_fwd(const float* logits, const uint8_t* labels, float* probs, float* losses, float* u, int32_t n,
                        int32_t D, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream);
int cfun_mask_fused_bwd(const float* u, const float* probs, const uint8_t* labels, const float* g2, float* dlogits, int32_t n,
                        int32_t D, int32_t H, int32_t W, int32_t C, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight operands of MANY convolutions in ONE launch, straight from their OIDHW tensors.
 * cfun_weight_prepare_kinds(p): which operand the forward (kinds[0]) and the data gradient (kinds[1]) of conv p read
 * when p->w_prepared says so, and its size in bytes -- the plain packs, or what the kernels would otherwise derive from
 * them per call: the Winograd-transformed weights of k_conv_wino (1-D / 2-D; the data gradient's with mirrored taps), the
 * parity-folded 2x2x2 weights of a stride-2 conv's data gradient.  CFUN_WOP_NONE: not preparable (more than 27 taps) --
 * use cfun_weight_pack*.  The host fills a table of jobs, cfun_weight_prepare_plan assigns the workgroups
 * (block_begin), the caller copies the table to the device and cfun_weight_prepare runs all jobs as one grid.  A job may
 * gather output channels (co_idx) / input channels (ci_idx) of its source: the per-RoI Dropout3d weight slices of
 * mask_branch.py:130-175 without index_select launches.  Values are bit-identical to pack + per-call transform.
 * ---------------------------------------------------------------------------------------------- */
#define CFUN_WOP_NONE 0
#define CFUN_WOP_PACK 1           /* [tap][Ci][CoP] */
#define CFUN_WOP_PACKT 2          /* [tap][Co][CiP] */
#define CFUN_WOP_WINO1 3          /* [9 (dz,dy)][Ci][CoP] x 4 x-points */
#define CFUN_WOP_WINO2 4          /* [12 (dz,py)][Ci][CoP] x 4 x-points */
#define CFUN_WOP_WINO1_T 5        /* the data gradient's: [9][Co][CiP] x 4, taps mirrored */
#define CFUN_WOP_WINO2_T 6        /* [12][Co][CiP] x 4, taps mirrored */
#define CFUN_WOP_S2FOLD 7         /* [8][Co][round16(8*Ci)]: stride-2 3x3x3 data gradient as a 2x2x2 conv + depth-to-space */
typedef struct CfunWeightJob {
  const float* w;                 /* source [*, src_ci, T] (OIDHW viewed 3-D), device */
  float* fwd;                     /* forward operand (fwd_kind), 16-byte aligned, or NULL */
  float* dgrad;                   /* data-gradient operand (dgrad_kind) or NULL */
  const int64_t* co_idx;          /* NULL, or Co source rows (gathered output channels), device */
  const int64_t* ci_idx;          /* NULL, or Ci source columns (gathered input channels), device */
  int32_t Co, Ci, T;              /* the conv's channels (after gathering) and taps (<= 27) */
  int32_t src_ci;                 /* input channels of the source tensor (its row length is src_ci * T) */
  int32_t fwd_kind, dgrad_kind;   /* CFUN_WOP_* */
  uint32_t block_begin;           /* filled by cfun_weight_prepare_plan */
  int32_t reserved;
} CfunWeightJob;
int cfun_weight_prepare_kinds(const CfunConv3dParams* p, int32_t kinds[2], size_t bytes[2]);
int cfun_weight_prepare_plan(CfunWeightJob* jobs_host, int32_t njobs, int64_t* nblocks);
int cfun_weight_prepare(const CfunWeightJob* jobs_dev, int32_t njobs, int64_t nblocks, cfun_stream_t stream);

/* The weight fold of "nn.Upsample(scale_factor=2) -> Conv3d(k, padding=k/2)" (mask_branch.py:108-116: k = 3; 216-218: k = 5) into
 * a 3x3x3 conv on the low-resolution input that produces the 8 output parities as channels: hi-res tap t of parity p reads the
 * low-res offset floor((p + t - k/2) / 2) per axis, taps sharing an offset are summed.  w [Co][Ci][k][k][k] -> wf [8 * cqp][Ci][27]
 * (channel = parity * cqp + co; rows co >= Co are zero: cqp pads a parity group to the kernels' tile width); _bwd is the transpose,
 * g [8 * cqp][Ci][27] -> dw [Co][Ci][k^3].  k in {3, 5}. */
int cfun_fold_up2_fwd(const float* w, float* wf, int32_t Co, int32_t Ci, int32_t k, int32_t cqp, cfun_stream_t stream);
int cfun_fold_up2_bwd(const float* g, float* dw, int32_t Co, int32_t Ci, int32_t k, int32_t cqp, cfun_stream_t stream);

/* cfun_weight_pack and cfun_weight_pack_transpose of the same OIDHW weight in ONE launch (a training step needs both
 * layouts of every conv weight: wp for the forward / weight-gradient kernels, wpT for the data gradient). */
int cfun_weight_pack_both(const float* w, float* wp, float* wpT, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream);
int cfun_weight_unpack(const float* dwp, float* dw, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer tail of train_epoch (model.py:1538-1545, 1641-1645) over FLAT fp32 buffers (cfun_amd/optim.py keeps
 * parameters, gradients and momentum in a few large arenas, so a step is 2 + #arenas launches):
 *   sumsq_partials: partials[0 .. cfun_sumsq_partials_count()) = per-workgroup fp64 sums of g^2 (fixed order)
 *   norm_finalize:  norm[0] = sqrt(sum of `count` partials)         -- torch.nn.utils.clip_grad_norm_'s total norm
 *   sgd_momentum_step (torch.optim.SGD, dampening 0, no nesterov), in place on p and m:
 *       c = max_norm > 0 ? min(1, max_norm / (norm[0] + 1e-6)) : 1 ;  d = c*g + weight_decay*p ;
 *       m = first_step ? d : momentum*m + d ;  p -= lr*m
 * ---------------------------------------------------------------------------------------------- */
int32_t cfun_sumsq_partials_count(void);
int cfun_sumsq_partials(const float* g, int64_t n, double* partials, cfun_stream_t stream);
int cfun_norm_finalize(const double* partials, int32_t count, float* norm, cfun_stream_t stream);
int cfun_sgd_momentum_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float weight_decay,
                           float max_norm, const float* norm, int32_t first_step, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Depth-sharding helpers: copy `planes` depth planes at z0 of a [N,D,H,W,C]
 * tensor into a dense send buffer / write a received buffer into a padded tensor.
 * ---------------------------------------------------------------------------------------------- */
int cfun_halo_pack(const float* x, float* buf, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, int32_t z0,
                   int32_t planes, cfun_stream_t stream);
int cfun_halo_unpack(const float* buf, float* x, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, int32_t z0,
                     int32_t planes, cfun_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline: the volume resize in front of the network (utils.resize_image mode 'self', utils.py:389-393 ->
 * skimage.transform.resize order 1, 'constant'; LiTS_2017/model.py:1741-1761: zero-pad to PAD_IMAGE_SHAPE, then order 0).
 * in: the source volume addressed by element `strides[3]` over `dims[3]` (both in OUTPUT axis order, so a [H,W,D] array
 * is resized into [D,H,W] without a transpose); `frame[3]` / `offset[3]` (NULL: none) place it inside a virtual zero
 * frame that is resized instead; out: dense [out_dims[0], out_dims[1], out_dims[2]].  order 0: nearest
 * (floor((o + 0.5) * n / m)); order 1: linear, samples outside are 0 ('grid-constant'), c = (o + 0.5) * n / m - 0.5;
 * clip_minmax: device pointer to {min, max} of the source for skimage's clip = True, or NULL.  All array arguments
 * except in / out / clip_minmax are HOST pointers, read before the call returns.
 * ---------------------------------------------------------------------------------------------- */
int cfun_resize3d(const float* in, const int64_t* strides, const int32_t* dims, const int32_t* frame,
                  const int32_t* offset, float* out, const int32_t* out_dims, int32_t order, const float* clip_minmax,
                  cfun_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CFUN_HIP_H */

/*
 * seam_hip.h -- C ABI of libseam_hip.so: the MI355X (gfx950) kernels behind the forward
 * hot path of SEAM Match-RCNN (BASELINE.json north_star).
 *
 * Conventions (every entry point):
 *   - returns int = hipError_t of the launch (0 = success); never throws, never
 *     allocates, never synchronises; asynchronous on `stream` (a hipStream_t, e.g.
 *     torch.cuda.current_stream().cuda_stream passed as void*).
 *   - all pointers are DEVICE pointers into caller-owned, contiguous fp32 storage
 *     (int32/int64 where stated); shapes are int32.
 *   - activations are NHWC ("channels last") fp32 inside the library; the reference's
 *     NCHW tensors cross the boundary through seam_nchw_to_nhwc_f32 / seam_nhwc_to_nchw_f32.
 *   - thread-safe for distinct streams.
 *
 * Each entry cites the reference interface (file:line under the reference tree) whose
 * arithmetic it replaces.  "[TV]" = the arithmetic lives in torchvision, which the
 * reference only configures (models/video_matchrcnn.py:6-9,337-338).
 */
#ifndef SEAM_HIP_H
#define SEAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* seam_stream_t; /* hipStream_t */

/* ABI version (major*1000 + minor). */
int seam_version(void);
/* hipGetErrorString for a code returned by any entry point. */
const char* seam_error_string(int code);

/* Variant selectors of the launchers (kernel-form switches the parity tests and the A/B tools flip: which Winograd form,
 * persistent or one tile per block, a forced implicit-GEMM tile ...).  A launcher never reads the environment; these are
 * process-wide ints, named like the environment variables the Python host applies once at load time ("SEAM_W24_PC", ...;
 * the table is csrc/seam_opts.h).  Every variant of a kernel computes the same results (bit-identical where the tests say so).
 * seam_set_option / seam_get_option return 0, or hipErrorInvalidValue for an unknown name. */
int seam_option_count(void);
const char* seam_option_name(int index);           /* NULL outside [0, count) */
int seam_set_option(const char* name, int value);
int seam_get_option(const char* name, int* value);

/* ---------------------------------------------------------------------------------
 * Implicit-GEMM convolution, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces every conv / linear on the path:
 *   ResNet-50 body + FrozenBN [TV]            (ctor models/video_matchrcnn.py:337)
 *   FPN lateral/output convs, RPNHead [TV]    (models/video_matchrcnn.py:337-338)
 *   TwoMLPHead / FastRCNNPredictor [TV]       (call models/video_matchrcnn.py:226-227)
 *   MaskRCNNHeads / MaskRCNNPredictor [TV]    (call models/video_matchrcnn.py:278-279)
 *   MatchPredictor.conv_seq / .linear         (models/match_head.py:50-62,67-69,93-95)
 *
 *   y[n,ho,wo,k] = act( scale[k] * sum_{r,s,c} x[n,ho*stride+r-pad,wo*stride+s-pad,c]
 *                                   * w[k][(r*S+s)*C+c]  + shift[k] + residual[n,ho,wo,k] )
 *
 * x        NHWC [N,H,W,C], C multiple of 4 and, when C >= 32, of 32 (pad channels zero-weighted)
 * w_packed rows_padded*kred floats from seam_pack_conv_weight_f32 (tile-contiguous slabs
 *          [n_tile][chunk][BN][32]; opaque to the caller)
 * scale    [K] or NULL (=1);  shift [K] or NULL (=0)   (bias / folded BatchNorm)
 * residual NHWC [N,Ho,Wo,K] or NULL, added before the activation
 * relu     0 none | 1 ReLU | 2 (backward passes) `residual` is NOT added: the result is zeroed where
 *          residual <= 0 -- the ReLU mask of the forward activation fused into the input-gradient conv
 * kred     padded reduction length = seam_conv_kred(C,R,S)
 */
int seam_conv_kred(int C, int R, int S);           /* host helper: ceil(R*S*C / 32) * 32 */
int seam_conv_rows_padded(int K);                  /* host helper: ceil(K / 64) * 64      */
int seam_conv_tile(int M, int K);                  /* host helper: BM*1000+BN of the block tile the fp32 launcher
                                                      picks for an [M x K] output (128 or 64 each) */
int seam_conv_tile_prec(int prec, int M, int K);   /* same for prec 0 = fp32, 1 = fp16, 2 = split-bf16 (these two also
                                                      have a 256x128 tile run by 8 waves) */
int seam_conv_tile_taps(int prec, int M, int K, int taps);   /* ... for a layer with R*S = taps (the fp16 choice depends on it) */

/* Pack an OIHW weight [K,Cin,R,S] (PyTorch layout) into the kernel's slab layout
 * (rows_padded*kred floats; reduction chunks ordered (r, c-chunk, s); channels >= Cin zero-filled).
 * mode 0: Conv2d / Linear (Linear = R=S=1; fc6 = a 7x7 "valid" conv over the 7x7 ROI tile)
 * mode 1: ConvTranspose2d(k=2,s=2) weight [Cin,Cout,2,2] -> rows ((a*2+b)*Cout + co),
 *         i.e. a 1x1 conv producing the 4 sub-pixels as channel groups; K = 4*Cout.
 * mode 2: input-gradient weights of a Conv2d whose OIHW weight is w [Cin, K, R, S] (note the roles:
 *         K = the forward conv's INPUT channels = rows produced, Cin = its output channels = channels
 *         reduced): taps rotated by 180 degrees, channels swapped, so that
 *         dX = seam_conv2d_f32(dY, packed, pad = R-1-pad_fwd) for a stride-1 conv (Linear: W^T).   */
int seam_pack_conv_weight_f32(const float* w, float* w_packed, int K, int Cin, int R, int S,
                              int Cstore, int mode, seam_stream_t stream);

int seam_conv2d_f32(const float* x, const float* w_packed, const float* scale,
                    const float* shift, const float* residual, float* y,
                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                    int relu, seam_stream_t stream);

/* The same convolution with the FPN top-down merge [TV FeaturePyramidNetwork.forward: inner_lateral +
 * F.interpolate(last_inner, size=feat_shape, mode="nearest")] fused into its epilogue (SURVEY 8b
 * `seam_fpn_topdown_add_f32`): y = act(conv(x)*scale + shift + top[n, floor(ho*Ht/Ho), floor(wo*Wt/Wo), :]),
 * top NHWC [N,Ht,Wt,K], K % 4 == 0.  Same rounding order as seam_conv2d_f32 followed by seam_upsample_add_f32
 * (bit-identical), without writing and re-reading the lateral map. */
int seam_conv2d_upres_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                          const float* top, float* y, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int Ht, int Wt, int relu, seam_stream_t stream);

/* seam_conv2d_f32 with the output grid cropped to its top-left [Ho, Wo] (Ho, Wo <= the full output size): `pad` rows /
 * columns of zeros before the map and fewer after it.  Call site: the ResNet stem [TV ResNet.conv1, 7x7 / stride 2 / pad 3]
 * on the space-to-depth input written by seam_preprocess_s2d_batch_f32 -- a 4x4 / stride-1 convolution over 12 channels
 * with 2 rows of padding before and 1 after (weights re-indexed on the host: w'[k,(dy,dx,c),r',s'] = w[k,c,2r'+dy-1,2s'+dx-1]). */
int seam_conv2d_crop_f32(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                         int relu, seam_stream_t stream);
/* fp16 twin (operands fp16, fp32 accumulate): the stem of the fp16 path on seam_preprocess_s2d_batch_f16's 16-channel cells. */
int seam_conv2d_crop_f16(const void* x, const void* w_packed, const float* scale, const float* shift, void* y,
                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                         int relu, seam_stream_t stream);

/* ResNet downsample blocks [TV Bottleneck.forward: out = relu(bn3(conv3(h)) + bn_d(conv_d(x)))] as ONE GEMM over two 1x1
 * sources: y[n,ho,wo,:] = act(W[:, :C1] . x1[n,ho,wo,:] + W[:, C1:] . x2[n,ho*stride2,wo*stride2,:] (* scale) + shift).
 * x1 NHWC [N,Ho,Wo,C1], x2 NHWC [N,H2,W2,C2], C1 and C2 multiples of 32; w_packed = seam_pack_conv_weight_f32 of the
 * [K, C1+C2, 1, 1] weight (the host folds the two FrozenBN scales into it and adds the shifts).  The shortcut branch is
 * never written to memory and read back as a residual. */
int seam_conv2d_dual_f32(const float* x1, const float* x2, const float* w_packed, const float* scale,
                         const float* shift, float* y, int N, int Ho, int Wo, int C1, int H2, int W2, int C2,
                         int stride2, int K, int relu, seam_stream_t stream);

/* Pointwise (1x1, stride 1, pad 0) convolution with a SHORT reduction, weights stationary in LDS (csrc/seam_pw.hip) -- the same
 * contract as seam_conv2d_f32 / seam_conv2d_dual_f32 (stride2 = 1) / seam_conv2d_upres_f32 on those shapes [TV Bottleneck conv1 /
 * conv3, FeaturePyramidNetwork.inner_blocks, MaskRCNNPredictor.conv5_mask], exact fp32 on v_mfma_f32_32x32x2_f32:
 *   y[M,K] = act( [x | x2][M, C1 + C2] . w[K, C1 + C2]^T + shift [+ residual] )
 * x [M,C1], x2 [M,C2] or NULL (C2 = 0), w row-major [K, C1 + C2] with any per-channel scale already folded in (one fp32
 * rounding per weight), shift [K] (required).  res_mode 0: no residual; 1: residual [M,K]; 2: residual = coarse NHWC map
 * [N,rH,rW,K] added through a nearest-neighbour upsample to the [Ho,Wo] output grid (M = N*Ho*Wo, ATen's index rule).
 * Shapes served: C1, C2 multiples of 32, C1 + C2 <= 256, K a multiple of 64 (of 128 / 256 for the wider wave tiles), any M > 0;
 * seam_conv1x1_sw_config returns 0 for anything else (MT*100 + NT of the wave tile otherwise) and the launcher then returns
 * hipErrorInvalidValue.  An output pixel is one wave's fixed fma chain: results are deterministic and independent of M. */
int seam_conv1x1_sw_config(int M, int C1, int C2, int K);
int seam_conv1x1_sw_f32(const float* x, const float* x2, const float* w, const float* shift, const float* residual, float* y,
                        int M, int C1, int C2, int K, int relu, int res_mode, int Ho, int Wo, int rH, int rW,
                        seam_stream_t stream);

/* conv1x1_f16pc (csrc/seam_pwhpc.hip, round 6): the fp16 1x1 / stride-1 layers with a LONG reduction (C >= 512, a multiple of 256; K a
 * multiple of 128) as a producer / consumer block -- the fp16 twin of seam_conv1x1_pc_f32.  Replaces, for those shapes, the reference's
 * `torch.nn.Conv2d(C, K, 1)` + FrozenBatchNorm2d + ReLU of torchvision's Bottleneck.conv1 (reached from
 * /root/reference/models/video_matchrcnn.py:57-66 through `resnet_fpn_backbone`) under autocast-free fp16 operands.
 *   _supported: 1 when the shape is served;  _weight_halves: fp16 elements of the packed weights;
 *   seam_pack_conv1x1_weight_f16pc: w [K, C] fp32 row-major -> fragment order;
 *   seam_conv1x1_f16pc: y[M, K] = act(scale * (x[M, C] . w^T) + shift), fp16 in / out, fp32 accumulation; `residual` must be NULL. */
int seam_conv1x1_f16pc_supported(long long M, int C, int K);
long long seam_conv1x1_f16pc_weight_halves(int K, int C);
int seam_pack_conv1x1_weight_f16pc(const float* w, void* w_packed, int K, int C, seam_stream_t stream);
int seam_conv1x1_f16pc(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                       long long M, int C, int K, int relu, seam_stream_t stream);

/* fp16 twin of seam_conv1x1_sw_f32 (csrc/seam_pwh.hip, round 6): the config-5 path's 1x1 layers [TV Bottleneck conv1 / conv3 /
 * stride-1 projection shortcut, FeaturePyramidNetwork.inner_blocks; behind models/video_matchrcnn.py:337] as a STREAMING kernel --
 * weights stationary in LDS, independent waves, 16-byte NHWC pieces in and out -- with the contract of seam_conv2d_f16 /
 * seam_conv2d_dual_f16 (stride2 = 1) on those shapes:
 *   y[M,K] = fp16( act( [x | x2][M, C1 + C2] . w[K, C1 + C2]^T * scale + shift [+ residual] ) )
 * res_mode 0: no residual (residual NULL); 1: residual [M,K]; 2 (single source only): residual = coarse NHWC map [N,rH,rW,K] added
 * through a nearest-neighbour upsample to the [Ho,Wo] output grid (M = N*Ho*Wo, Ho*Wo >= 128; ATen's index rule) -- the FPN top-down
 * merge [TV FeaturePyramidNetwork.forward] in the conv epilogue, as seam_conv1x1_sw_f32 / seam_conv2d_upres_f32 do in fp32.
 * x, x2, residual, y fp16; w fp16 row-major [K, C1 + C2]; scale / shift fp32 [K] or NULL; fp32 accumulation, fp32 epilogue, one
 * rounding.  Shapes served (seam_conv1x1_swh_config != 0): C1, C2 multiples of 64, C1 + C2 <= 512 (<= 256 for the 256-channel slab
 * K % 256 == 0 takes, <= 128 when K is not a multiple of 128), K a multiple of 64 with K / slab dividing 32; any M > 0.  relu 0 | 1.  Same products as seam_conv2d_f16 in a
 * different accumulation order (agreement to fp32 rounding, not bit for bit); deterministic and independent of M. */
int seam_conv1x1_swh_config(long long M, int C1, int C2, int K);
int seam_conv1x1_swh_f16(const void* x, const void* x2, const void* w, const float* scale, const float* shift,
                         const void* residual, void* y, long long M, int C1, int C2, int K, int relu, int res_mode,
                         int Ho, int Wo, int rH, int rW, seam_stream_t stream);

/* The ResNet stem [TV ResNet.conv1 7x7 / stride 2 / pad 3 + bn1 + relu; behind models/video_matchrcnn.py:337] of the fp16 path on the
 * PADDED space-to-depth frame, as the streaming kernel above (csrc/seam_pwh.hip, round 6): xpad fp16 [N, Ho + 3, Wo + 3, 16] = the
 * frame of seam_preprocess_s2d_batch_f16 with 2 zero cells before and 1 after in both directions; w fp16 [64, 256] row-major with
 * k = (4 r + s) * 16 + channel (the re-indexed weights of seam_conv2d_crop_f16's call site); scale / shift fp32 [64] or NULL;
 * y fp16 [N, Ho, Wo, 64].  A tap is a constant shift of the flattened padded index, so every A fragment is one contiguous 1 KB run.
 * Same products as seam_conv2d_crop_f16, fp32 accumulation in tap-major order (agreement to fp32 rounding); deterministic. */
int seam_stem_s2d_swh_f16(const void* xpad, const void* w, const float* scale, const float* shift, void* y, int N, int Ho, int Wo,
                          int relu, seam_stream_t stream);

/* Pointwise (1x1, stride 1, pad 0) convolution with a LONG reduction as producer / consumer waves (csrc/seam_pwpc.hip, round 5):
 * the bottleneck reductions of ResNet layer2-4 and the layer4 expansions [TV Bottleneck conv1 / conv3 behind
 * models/video_matchrcnn.py:337], exact fp32 on v_mfma_f32_32x32x2_f32, the same contract as seam_conv2d_f32 on those shapes:
 *   y[M,K] = act( x[M,C] . W[K,C]^T * scale + shift [+ residual[M,K]] )
 * Shapes served (seam_conv1x1_pc_supported): C >= 256 and a multiple of 128, K a multiple of 128, any M > 0.  Weights:
 * seam_conv1x1_pc_weight_floats(K, C) floats from seam_pack_conv1x1_pc_f32 (row-major [K, C] in; MFMA fragment order).  scale /
 * shift [K] or NULL.  An output is one wave's fixed fma chain over k: results are deterministic and independent of M; they differ
 * from seam_conv2d_f32's by the order of the fp32 accumulation only. */
int seam_conv1x1_pc_supported(long long M, int C, int K);
long long seam_conv1x1_pc_weight_floats(int K, int C);
int seam_pack_conv1x1_pc_f32(const float* w, float* w_packed, int K, int C, seam_stream_t stream);
int seam_conv1x1_pc_f32(const float* x, const float* w_packed, const float* scale, const float* shift, const float* residual,
                        float* y, long long M, int C, int K, int relu, seam_stream_t stream);

/* fp16 twin of seam_conv2d_dual_f32 (x1, x2, w_packed, y fp16; C1 and C2 multiples of 64; fp32 accumulation and epilogue). */
int seam_conv2d_dual_f16(const void* x1, const void* x2, const void* w_packed, const float* scale,
                         const float* shift, void* y, int N, int Ho, int Wo, int C1, int H2, int W2, int C2,
                         int stride2, int K, int relu, seam_stream_t stream);

/* fp16 variant (BASELINE config 5: "fp16 MFMA path with fp32 ... accumulation"): x, w_packed,
 * residual are IEEE fp16 (NHWC, C multiple of 8 and, when C >= 64, of 64); v_mfma_f32_32x32x16_f16
 * with fp32 accumulators; scale/shift stay fp32; y is fp16, or fp32 when y_f32 != 0 (the last trunk
 * Linear hands fp32 descriptors to the fp32 heads).  Weights come from seam_pack_conv_weight_f16
 * (source still the fp32 OIHW parameter; rows_padded * seam_conv_kred_f16(C,R,S) halves). */
int seam_conv_kred_f16(int C, int R, int S);
int seam_pack_conv_weight_f16(const float* w, void* w_packed, int K, int Cin, int R, int S,
                              int Cstore, int mode, seam_stream_t stream);
int seam_conv2d_f16(const void* x, const void* w_packed, const float* scale, const float* shift,
                    const void* residual, void* y, int N, int H, int W, int C, int K, int R, int S,
                    int stride, int pad, int relu, int y_f32, seam_stream_t stream);

/* ---------------------------------------------------------------------------------
 * GeneralizedRCNNTransform [TV] (reached from GeneralizedRCNN.forward; callers
 * stuffs/engine.py:115, evaluate_movingfashion.py:31): per image
 *   (x-mean)/std -> bilinear resize (align_corners=False, scale=in/out) to [out_h,out_w]
 *   -> zero-pad to [Hp,Wp] -> NHWC with 4 stored channels (4th = 0).
 * img  CHW [3,in_h,in_w] in [0,1];  out NHWC4 slot [Hp,Wp,4] of the batch tensor. */
int seam_preprocess_f32(const float* img, float* out, int in_h, int in_w, int out_h, int out_w,
                        int Hp, int Wp, seam_stream_t stream);

int seam_preprocess_f16(const float* img, void* out /* fp16 NHWC8 */, int in_h, int in_w, int out_h,
                        int out_w, int Hp, int Wp, seam_stream_t stream);

/* The same transform for n images of ONE shape laid out at a constant stride (img_stride floats between
 * consecutive images: the frames of a clip tensor [T,3,H,W], which the reference hands over as a list of views):
 * one launch writes the whole NHWC batch [n,Hp,Wp,4] (fp16: [n,Hp,Wp,8]).  n <= 65535. */
int seam_preprocess_batch_f32(const float* imgs, size_t img_stride, float* out, int n, int in_h, int in_w,
                              int out_h, int out_w, int Hp, int Wp, seam_stream_t stream);
int seam_preprocess_batch_f16(const float* imgs, size_t img_stride, void* out, int n, int in_h, int in_w,
                              int out_h, int out_w, int Hp, int Wp, seam_stream_t stream);

/* The same transform written space-to-depth for the stem: out [n, Hp/2, Wp/2, 12], channel (dy*2+dx)*3 + c = pixel
 * (2Y+dy, 2X+dx), colour c (Hp, Wp even). */
int seam_preprocess_s2d_batch_f32(const float* imgs, size_t img_stride, float* out, int n, int in_h, int in_w,
                                  int out_h, int out_w, int Hp, int Wp, seam_stream_t stream);
/* fp16 output: [n, Hp/2, Wp/2, 16] -- the 12 channels above + 4 zeros (the fp16 GEMM reads 8-channel vectors). */
int seam_preprocess_s2d_batch_f16(const float* imgs, size_t img_stride, void* out, int n, int in_h, int in_w,
                                  int out_h, int out_w, int Hp, int Wp, seam_stream_t stream);
/* The same frame with pad_lo zero cells before and pad_hi after it in both directions: out [n, Hp/2 + pad_lo + pad_hi,
 * Wp/2 + pad_lo + pad_hi, 16] -- the input layout of seam_stem_s2d_swh_f16 (pad_lo 2, pad_hi 1: the 4x4 / pad-2 form of the stem). */
int seam_preprocess_s2d_pad_batch_f16(const float* imgs, size_t img_stride, void* out, int n, int in_h, int in_w, int out_h,
                                      int out_w, int Hp, int Wp, int pad_lo, int pad_hi, seam_stream_t stream);

/* Same transform fed by a uint8 HWC RGB frame [in_h,in_w,3]: fuses ToTensor (x/255, stuffs/transform.py:46-49)
 * so a clip crosses PCIe at 1 byte per sample (device side).  out: fp32 NHWC4 or, when
 * out_f16 != 0, fp16 NHWC8. */
int seam_preprocess_u8(const uint8_t* img, void* out, int in_h, int in_w, int out_h, int out_w,
                       int Hp, int Wp, int out_f16, seam_stream_t stream);

/* max_pool2d on NHWC [TV: ResNet stem 3x3/s2/p1; LastLevelMaxPool k=1,s=2]. C % 4 == 0. */
int seam_maxpool2d_f32(const float* x, float* y, int N, int H, int W, int C, int k, int stride,
                       int pad, seam_stream_t stream);

int seam_maxpool2d_f16(const void* x, void* y, int N, int H, int W, int C, int k, int stride,
                       int pad, seam_stream_t stream);   /* fp16, C % 8 == 0 */

/* FPN top-down [TV]: lat[n,h,w,:] += top[n, floor(h*Ht/H), floor(w*Wt/W), :] (nearest). */
int seam_upsample_add_f32(float* lat, const float* top, int N, int H, int W, int Ht, int Wt, int C,
                          seam_stream_t stream);

int seam_upsample_add_f16(void* lat, const void* top, int N, int H, int W, int Ht, int Wt, int C,
                          seam_stream_t stream);

/* ---------------------------------------------------------------------------------
 * MultiScaleRoIAlign(['0','1','2','3'], P, sampling_ratio) + roi_align(aligned=False) [TV]
 * call sites models/video_matchrcnn.py:225 (P=7), :277 (P=14); models/matchrcnn.py:463.
 * feat[l] NHWC [N,Hl,Wl,C] (C % 4 == 0, C <= 1024);  rois [K,5] = (batch_idx,x1,y1,x2,y2)
 * in resized-image pixels;  levels int32 [K] or NULL (NULL: LevelMapper on device:
 * clamp(floor(4+log2(sqrt(area)/224)+1e-6), k_min, k_min+3) - k_min);
 * scales[l] = 2^round(log2(Hl/Himg)) passed by value;  out NHWC [K,P,P,C]. */
int seam_roi_align_f32(const float* feat0, const float* feat1, const float* feat2, const float* feat3,
                       const int* hw /* host int[8]: H0,W0,...,H3,W3 */, int C,
                       float scale0, float scale1, float scale2, float scale3, int k_min,
                       const float* rois, const int* levels, float* out, int K, int P,
                       int sampling_ratio, seam_stream_t stream);

int seam_roi_align_f16(const void* feat0, const void* feat1, const void* feat2, const void* feat3,
                       const int* hw, int C, float scale0, float scale1, float scale2, float scale3,
                       int k_min, const float* rois, const int* levels, void* out, int K, int P,
                       int sampling_ratio, seam_stream_t stream);   /* fp16 maps in, fp16 out */

/* TEST / BENCHMARK HOOK ONLY -- not part of the per-stream thread-safety contract of this header: a process-wide selector of the
 * kernel the two entry points above launch (also env SEAM_ROIALIGN_LDS at first use): 2 (default) = LDS-staged ROI quadrant tiles,
 * 1 = row-staged tiles, 0 = one wave per bin gathering its 16 taps from L1/L2 (also what any shape outside sampling_ratio 2,
 * P <= 16, C % 64 == 0 takes).  All three give bit-identical results, so a concurrent change can never alter an output -- but
 * production callers should not call it (tools/roialign_ab.py and tests/test_gpu_ops.py do); measurements in
 * profiles/r03_roialign_ab.txt. */
void seam_roi_align_set_lds(int on);

/* Layout bridges at the module boundary: x [B,C,L] <-> y [B,L,C]. */
int seam_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int L, seam_stream_t stream);
int seam_nhwc_to_nchw_f32(const float* x, float* y, int B, int L, int C, seam_stream_t stream);

/* fp16 path: the reference-facing side stays fp32 NCHW, the library side is fp16 NHWC. */
int seam_nchw_f32_to_nhwc_f16(const float* x, void* y, int B, int C, int L, seam_stream_t stream);
int seam_nhwc_f16_to_nchw_f32(const void* x, float* y, int B, int L, int C, seam_stream_t stream);

/* AvgPool2d((6,6)) (+ no-op ReLU) of models/match_head.py:59-60: x [K,L,C] -> y [K,C]. */
int seam_avgpool_f32(const float* x, float* y, int K, int L, int C, seam_stream_t stream);

int seam_avgpool_f16(const void* x, void* y, int K, int L, int C, seam_stream_t stream);

/* ---------------------------------------------------------------------------------
 * NONLocalBlock1D(256, sub_sample=False, bn_layer=False) + attention pooling, batched
 * over sequences: models/nlb.py:66-101 via models/match_head.py:114-121 / :144-151.
 *   per sequence X[T,256]:  TH/PH/G = X W^T + b ; a = TH.wc[:128] ; b = PH.wc[128:]
 *   f = ReLU(a_i+b_j)/T ; Y = f G ; Z = Y Ww^T + bw + X      (T == 1: Z = X, ref :115-117)
 *   s = Z.wa + ba ; p = softmax_t(s) ; out = sum_t p_t Z_t   (T == 0: out = 0)
 * seq      element (t,s,c) at seq[t*t_stride + s*s_stride + c], t < len[s]
 *          (pass x3_1_seq + S*256, t_stride = S*256, s_stride = 256: row 0 is the dummy)
 * len      int32 [S] (device)
 * w_proj_t [256][384] = concat(theta,phi,g).weight transposed;  b_proj [384]
 * w_cat    [256] concat_project weight;  w_out_t [128][256] = W.weight^T;  b_out [256]
 * w_att [256], b_att [1]
 * out [S,256];  att [S,Tmax] or NULL (softmax weights; entries >= len[s] untouched)
 * z   [S,Tmax,256] or NULL: the non-local block's own output Z (= NONLocalBlock1D.forward)
 * ws   workspace, >= seam_nlb_workspace_floats(S,Tmax) floats
 * use_nlb  0: no block; 1: the module's `.nlb` flag (models/match_head.py:88), length-1
 *          sequences bypass the block (ref :115-117); 2: apply the block to every sequence */
int64_t seam_nlb_workspace_floats(int S, int Tmax);
int seam_nlb_attnpool_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len,
                          int S, int Tmax, const float* w_proj_t, const float* b_proj,
                          const float* w_cat, const float* w_out_t, const float* b_out,
                          const float* w_att, const float* b_att, float* out, float* att,
                          float* z, float* ws, int use_nlb, seam_stream_t stream);

/* The same block with its GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32): G = X Wg + bg, Y = f G and
 * Z = Y Ww^T + bw + X per 32-row tile, one workgroup per sequence, sequences of at most seam_nlb_mfma_max_len() (96) rows;
 * seq rows 16-byte aligned (t_stride, s_stride multiples of 4 floats).  theta / phi enter the block only through
 * a_i = theta_i . wc[:128] and b_j = phi_j . wc[128:] (models/nlb.py:80-90), so the caller folds them once:
 *   u[256] = W_theta^T wc[:128], v[256] = W_phi^T wc[128:], cd[2] = (b_theta . wc[:128], b_phi . wc[128:]).
 * wg_frag [4][32][64][4] / wo_frag [8][16][64][4]: the g and W projections in MFMA B-fragment order -- element e of lane
 * (h = lane >> 5, n = lane & 31) of fragment [n_tile][j] = W[k = 8 j + 4 h + e][32 n_tile + n] (W as [k][n]).
 * out / att / z / use_nlb as above. */
int seam_nlb_mfma_max_len(void);
int seam_nlb_attnpool_mfma_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                               const float* wg_frag, const float* b_g, const float* u, const float* v,
                               const float* cd, const float* wo_frag, const float* b_out, const float* w_att,
                               const float* b_att, float* out, float* att, float* z, int use_nlb,
                               seam_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Pairwise match classifier `last((a_i - b_j)^2)`: models/match_head.py:73-74,161-162;
 * NumPy twin evaluate_movingfashion.py:94-100,263-264.
 * a [Q,D], b [G,D], w [2,D], bias [2] -> out [Q,G,2].   D % 32 == 0, D <= 1024. */
int seam_pair_logits_f32(const float* a, const float* b, const float* w, const float* bias,
                         float* out, int Q, int G, int D, seam_stream_t stream);

/* Score + rank: evaluate_movingfashion.py:97-99,265-269.  Reads logits [Q,G,2]; ranks by
 * softmax(x)[...,1] (== monotone in x1-x0), descending, ties -> lower index first.
 * idx int64 [Q,k], score [Q,k] = softmax(x)[...,1] of the selected entries. k <= G. */
int seam_rank_topk_f32(const float* logits, int64_t* idx, float* score, int Q, int G, int k,
                       seam_stream_t stream);

/* Evaluator-side scoring.  score = softmax(logits)[...,1] for n_pairs rows of 2
 * (compute_distances / compute_selfdist, evaluate_movingfashion.py:102-121);  rank[q] = position of
 * product target[q] in the descending ranking of query q (what `(rankings == shop_prod_index)
 * .nonzero()` extracts at evaluate_movingfashion.py:228,268), same tie rule as seam_rank_topk_f32. */
int seam_match_scores_f32(const float* logits, float* score, int64_t n_pairs, seam_stream_t stream);
/* HOST helper (host pointers, no device work, no stream): the greedy tracklet linking of evaluate_movingfashion.py:166-202 for all
 * products of an evaluator pass.  blocks = the products' n_s x n_s self-similarity blocks concatenated (what
 * seam_pair_scores_blockdiag_f32 wrote, copied to the host); seg int64 [n_seg + 1] detection offsets; imgs int64 / scores double
 * [seg[n_seg]] frame index / confidence per detection.  Out: members_out [seg[n_seg]] -- per product, at seg[s], the LOCAL detection
 * indices tracklet after tracklet (creation order), members in link order; track_len_out [seg[n_seg]] -- at seg[s] the lengths of
 * that product's tracklets; n_tracks_out [n_seg].  Same decisions as the reference's loops (first maximum in row-major order). */
int seam_host_build_tracklets(const float* blocks, const int64_t* seg, const int64_t* imgs, const double* scores, int n_seg,
                              double threshold, int32_t* members_out, int32_t* track_len_out, int32_t* n_tracks_out);

/* Block-diagonal self-similarity for the evaluator's tracking step (`compute_selfdist` once per product,
 * evaluate_movingfashion.py:102-121,165-176) in ONE launch: x [rows, Dd] holds the detections' descriptors grouped by product,
 * seg int32 [n_seg + 1] the group boundaries (row offsets), out_off int64 [n_seg + 1] the running sum of n_s^2; for every group
 * out[out_off[s] + i * n_s + j] = softmax(W (x_i - x_j)^2 + b)[1].  Only the n_s x n_s diagonal blocks are computed (the all-pairs
 * matrix of a pass is rows^2).  max_rows >= every n_s.  Bit-identical to seam_pair_logits_f32 + seam_match_scores_f32 per group. */
int seam_pair_scores_blockdiag_f32(const float* x, const int* seg, const int64_t* out_off, const float* w, const float* bias,
                                   float* out, int n_seg, int max_rows, int Dd, seam_stream_t stream);
int seam_rank_of_f32(const float* logits, const int64_t* target, int64_t* rank, int Q, int G,
                     seam_stream_t stream);

/* Rankings over per-frame score rows (evaluate_movingfashion.py:293-315): out[g] = mean (mode 0) or max
 * (mode 1) over the n rows of score [n,G];  seam_rank_of_scores_f32: rank of target[q] in the descending
 * order of the plain score row q of [Q,G] (ties -> lower index first; what `np.argsort(x)[::-1] ==
 * shop_prod_index` extracts at :296-297,306-307). */
int seam_score_reduce_f32(const float* score, float* out, int n, int G, int mode, seam_stream_t stream);
/* ... over row segments: rows seg[p] .. seg[p+1]-1 (seg: P+1 int32 offsets on the device) -> out [P,G], one launch for all
 * products of an evaluator pass; bit-identical to P calls of seam_score_reduce_f32 (P <= 65535). */
int seam_score_reduce_seg_f32(const float* score, const int* seg, float* out, int P, int G, int mode, seam_stream_t stream);
int seam_rank_of_scores_f32(const float* score, const int64_t* target, int64_t* rank, int Q, int G,
                            seam_stream_t stream);

/* torchvision.ops.box_iou as the evaluator calls it to pick the tracklet that follows the ground truth
 * (evaluate_movingfashion.py:205-209): a [Na,4], b [Nb,4] xyxy -> out [Na,Nb]. */
int seam_box_iou_f32(const float* a, const float* b, float* out, int Na, int Nb, seam_stream_t stream);

/* Fused pairwise logits + top-k (a13 + a14 in one pass; no [Q,G,2] round trip through HBM):
 * same ranking rule and bit-identical x1-x0 as seam_pair_logits_f32 + seam_rank_topk_f32.
 * k <= 256, k <= G; ws: >= seam_pair_topk_workspace_floats(Q,G,k) floats of scratch. */
int64_t seam_pair_topk_workspace_floats(int Q, int G, int k);
int seam_pair_topk_f32(const float* a, const float* b, const float* w, const float* bias,
                       int64_t* idx, float* score, int Q, int G, int D, int k, float* ws,
                       seam_stream_t stream);

/* The same ranking for a LARGE bank (configs[2]/[3]: G = 20 000 / 50 000) on the fp32 matrix cores: candidates are found with the
 * expanded form of the logit difference (one [Q,256] x [256,G] v_mfma_f32_16x16x4_f32 GEMM + two rank-1 terms, per-query
 * threshold from a 4096-row sample, no [Q,G] tensor in HBM), the k + margin best are re-scored with the direct form in the
 * operation order of seam_pair_logits_f32, and a per-query rounding-error bound proves that no other product can enter the top k
 * (a query that fails the proof is redone with the direct form over the whole bank).  idx / score are bit-identical to
 * seam_pair_logits_f32 + seam_rank_topk_f32 (ref models/match_head.py:161-162; evaluate_movingfashion.py:94-100,263-269).
 * Requires D == 256, G >= seam_pair_topk_mfma_min_gallery(), k <= seam_pair_topk_mfma_max_k(), 16-byte aligned a / b / ws;
 * ws: >= seam_pair_topk_mfma_workspace_floats(Q,G,k) floats.  flags bit 0: direct-form path for every query (test hook).
 * stats: device int[4] or NULL -- [0] queries that took the direct-form path, [1] largest candidate count, [2] overflowed lists. */
int seam_pair_topk_mfma_min_gallery(void);
int seam_pair_topk_mfma_max_k(void);
int64_t seam_pair_topk_mfma_workspace_floats(int Q, int G, int k);
int seam_pair_topk_mfma_f32(const float* a, const float* b, const float* w, const float* bias,
                            int64_t* idx, float* score, int Q, int G, int D, int k, float* ws, int flags,
                            int* stats, seam_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Detection post-processing [TV] + models/video_matchrcnn.py:154-205.
 * BoxCoder.decode (weights wx,wy,ww,wh; dw,dh clamped to log(1000/16)) + clip to image.
 * deltas [N,ncls*4], anchors/proposals [N,4] -> boxes [N,ncls*4]. */
int seam_decode_boxes_f32(const float* deltas, const float* boxes_in, float* boxes_out, int N,
                          int ncls, float wx, float wy, float ww, float wh, float clip_h,
                          float clip_w, seam_stream_t stream);

/* Greedy NMS, batched over B images, over boxes [B,N,4] ALREADY SORTED by descending score inside
 * each image: keep[b,i]=1/0 (int32 [B,N]).  IoU > thr suppresses (strict), areas without +1.
 * 64x64 bitmask tiles + one-wave scan per image, N <= 16384.
 * mask_ws: uint64 workspace of B*N*ceil(N/64) words. */
int seam_nms_sorted_f32(const float* boxes, int* keep, int B, int N, float thr, uint64_t* mask_ws,
                        seam_stream_t stream);

/* Same, stopping after the first max_keep survivors of each image (0 = no limit): greedy NMS decides a box from
 * higher-scored boxes only, so the result equals the full scan truncated to max_keep survivors -- what the callers keep
 * anyway ([TV] filter_proposals post_nms_top_n; postprocess_detections detections_per_img,
 * models/video_matchrcnn.py:199-201). */
int seam_nms_sorted_topn_f32(const float* boxes, int* keep, int B, int N, float thr, int max_keep,
                             uint64_t* mask_ws, seam_stream_t stream);

/* RPN filter_proposals, per pyramid level [TV RegionProposalNetwork._get_top_n_idx + BoxCoder.decode + clip + sigmoid]
 * (SURVEY.md 8b `seam_rpn_decode_topk`): for each of n_img images the exact top-k of its n = H*W*A objectness logits
 * (logit descending, anchor index ascending = the first k of a stable descending sort), and for the winners in that order
 * the decoded (weights 1,1,1,1), clipped box, sigmoid(logit) and (optionally, index != NULL) the anchor index.
 * obj / deltas are read in place from the head output: logit of anchor j = (pixel j / A, a = j % A) at
 * obj[img*obj_img_stride + pixel*obj_pix_stride + a], its deltas at deltas[img*dlt_img_stride + pixel*dlt_pix_stride + 4a ..].
 * anchors [n,4]; clip_hw [n_img,2] = (height, width) of each resized image; outputs are written at row
 * img*out_img_stride + out_offset + rank (boxes [.,4], scores [.], index [.] int64).  k <= seam_rpn_topk_max() (1024), k <= n. */
int seam_rpn_topk_max(void);
int seam_rpn_topk_decode_f32(const float* obj, const float* deltas, const float* anchors,
                             const float* clip_hw, float* boxes, float* scores, int64_t* index,
                             int n_img, int n, int A, int k, int64_t obj_img_stride,
                             int obj_pix_stride, int64_t dlt_img_stride, int dlt_pix_stride,
                             int64_t out_img_stride, int out_offset, seam_stream_t stream);

/* paste_masks_in_image [TV] (transform.postprocess, reached from model(images)): masks [K,1,28,28]
 * probabilities, boxes [K,4] (original-image px) -> out [K,1,H,W]. */
int seam_paste_masks_f32(const float* masks, const float* boxes, float* out, int K, int H, int W,
                         seam_stream_t stream);

/* maskrcnn_inference [TV] (call models/video_matchrcnn.py:291): logits laid out
 * [K,14,14,(a,b),ncls] (sub-pixel groups from the transposed conv) -> prob [K,1,28,28] of the
 * channel labels[k] (int64), after sigmoid. */
int seam_mask_select_f32(const float* logits, const int64_t* labels, float* prob, int K, int ncls,
                         seam_stream_t stream);

int seam_mask_select_f16(const void* logits, const int64_t* labels, float* prob, int K, int ncls,
                         seam_stream_t stream);   /* fp16 logits in, fp32 probabilities out */

/* ---------------------------------------------------------------------------------------------------
 * Split-bf16 ("bx3") convolution: the same contraction as seam_conv2d_f32 on fp32 activations, computed as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with v_mfma_f32_32x32x16_bf16 and fp32 accumulation (relative error ~1e-5 per
 * product, fp32 exponent range; well inside the 1e-3 contract of BASELINE.json).  Same call sites as
 * seam_conv2d_f32 (opt-in: model.set_compute_dtype("bf16x3")).  Weights: seam_pack_conv_weight_bx3 (same size as
 * the fp32 pack; tmp = rows_padded*kred floats of scratch). */
int seam_pack_conv_weight_bx3(const float* w, void* w_packed, float* tmp, int K, int Cin, int R, int S,
                              int Cstore, int mode, seam_stream_t stream);
int seam_conv2d_bx3(const float* x, const void* w_packed, const float* scale, const float* shift,
                    const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S,
                    int stride, int pad, int relu, seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Winograd F(2x2,3x3) convolution, fp32 MFMA: the stride-1 3x3 layers of the same call sites as seam_conv2d_f32
 * (ResNet-50 bottleneck 3x3s, FPN output convs, RPNHead conv [TV]; MaskRCNNHeads [TV]; MatchPredictor.conv_seq,
 * models/match_head.py:50-60) with 2.25x fewer matrix-core issues.  Same contract as seam_conv2d_f32 with R = S = 3,
 * stride = 1 (x NHWC [N,H,W,C], y NHWC [N,H+2*pad-2,W+2*pad-2,K], scale / shift / residual / relu identical);
 * C must be a multiple of 8 and K of 32 (seam_wino_supported).  All arithmetic is fp32; results differ from
 * seam_conv2d_f32 only by the rounding of the +/- transforms (~1e-6 relative to the output scale).
 * u_packed: seam_wino_weight_floats(K, Cstore) floats from seam_pack_conv_weight_wino_f32 (U = G g Gt per (k, c),
 * computed in fp64 and rounded once, stored in MFMA fragment order; mode 0 = forward weights [K,Cin,3,3],
 * mode 2 = input-gradient weights as in seam_pack_conv_weight_f32). */
int seam_wino_supported(int C, int K, int R, int S, int stride);
long long seam_wino_weight_floats(int K, int Cstore);
int seam_wino_slot_fill_pct(int N, int H, int W, int C, int K, int pad);   /* host helper: % of the launch's 2x2-tile slots
                                                                             that hold real output (the caller's switch
                                                                             between this kernel and seam_conv2d_f32) */
int seam_wino_tile_variant(int N, int H, int W, int C, int K, int pad);   /* host helper: MT of the conv3x3_wino<MT>
                                                                            kernel the launcher picks (32*MT tiles per block) */
int seam_pack_conv_weight_wino_f32(const float* w, float* u_packed, int K, int Cin, int Cstore, int mode,
                                   seam_stream_t stream);
int seam_conv3x3_wino_f32(const float* x, const float* u_packed, const float* scale, const float* shift,
                          const float* residual, float* y, int N, int H, int W, int C, int K, int pad, int relu,
                          seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * 1x1 convolutions / Linear layers with at most 16 outputs (csrc/seam_narrow.hip): y[M,K] = act(x[M,C] . w[K,C]^T + bias),
 * C a multiple of 16, C <= 256, K <= 16 (seam_linear_narrow_supported).  Call sites: RPNHead.cls_logits + bbox_pred [TV]
 * (3 + 12 outputs per pixel, packed as one [15, 256] weight) and MaskRCNNPredictor.mask_fcn_logits [TV] (14 classes).
 * HBM-bound row stream on v_mfma_f32_16x16x4_f32 with the weights register-resident; w_packed = 64 * C floats from
 * seam_pack_linear_narrow_f32 (source: the fp32 [K, C] weight). */
int seam_linear_narrow_supported(int C, int K);
int seam_pack_linear_narrow_f32(const float* w, float* w_packed, int K, int C, seam_stream_t stream);
int seam_linear_narrow_f32(const float* x, const float* w_packed, const float* bias, float* y, long long M, int C, int K,
                           int relu, seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Winograd F(2x4,3x3) convolution, fp32 MFMA (csrc/seam_wino24.hip): the same layers and the same contract as
 * seam_conv3x3_wino_f32, with output tiles of 2 rows x 4 columns (F(2,3) down the rows, F(4,3) with the points
 * {0, +-1, +-2, inf} along the columns): 24 multiplies per 8 outputs -- 3x fewer matrix-core issues than
 * seam_conv2d_f32, 1.33x fewer than seam_conv3x3_wino_f32; rounding ~2x that of F(2x2,3x3) (~1e-6 of the output
 * scale).  u_packed: seam_wino24_weight_floats(K, Cstore) floats from seam_pack_conv_weight_wino24_f32 (U = G2 g G4t in
 * fp64, rounded once, MFMA fragment order).  seam_wino_issue_slots / seam_wino24_issue_slots: MFMA work of a launch
 * (block tile slots x positions) for the host's choice between the two forms (maps whose width is not a multiple of 4
 * can favour F(2x2)). */
long long seam_wino24_weight_floats(int K, int Cstore);
long long seam_wino24_issue_slots(int N, int H, int W, int C, int K, int pad);
int seam_wino24_variant(int N, int H, int W, int C, int K, int pad);   /* n-tiles per block (kernel variant conv3x3_wino24<NT>) picked for this shape; 0 = unsupported */
int seam_wino24_form(int N, int H, int W, int C, int K, int pad);      /* 1: the launch runs on the producer / consumer kernel conv3x3_wino24pc (round 5: the NT = 2 block as four MFMA-only waves + four transform / load waves, two per SIMD; bit-identical results), 0: conv3x3_wino24<NT>, -1: unsupported.  SEAM_W24_PC=0 in the environment keeps every launch on conv3x3_wino24<NT> */
long long seam_wino_issue_slots(int N, int H, int W, int C, int K, int pad);
int seam_pack_conv_weight_wino24_f32(const float* w, float* u_packed, int K, int Cin, int Cstore, int mode,
                                     seam_stream_t stream);
int seam_conv3x3_wino24_f32(const float* x, const float* u_packed, const float* scale, const float* shift,
                            const float* residual, float* y, int N, int H, int W, int C, int K, int pad, int relu,
                            seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * 3x3 / stride-1 convolution, fp16 operands / fp32 accumulation, producer / consumer form (csrc/seam_f16pc.hip, round 5): the
 * config-5 path's 3x3 layers (ResNet / FPN / RPN / mask head / match trunk; ref models/video_matchrcnn.py:337-338,
 * models/match_head.py:50-58) with the input patch staged in LDS once per 64-channel chunk instead of gathered nine times.  Same
 * contract as seam_conv2d_f16 on those shapes (NHWC fp16 in / out, per-channel fp32 scale / shift, ReLU) except that a residual
 * operand is NOT taken (`residual` must be NULL, else hipErrorInvalidValue: no 3x3 layer of the path has one, and the kernel's
 * epilogue rounds to fp16 in the accumulator registers); fp32 accumulation in a different order than seam_conv2d_f16 (chunk-major
 * instead of tap-major): results agree to fp32 rounding of the accumulation, not bit for bit.  Weights:
 * seam_f16pc_weight_halves(K, Cstore) fp16 values from seam_pack_conv_weight_f16pc (OIHW fp32 in; MFMA fragment order).  Shapes:
 * seam_conv3x3_f16pc_supported (C a multiple of 128, K a multiple of 128, pad 0 | 1; maps of >= 24 output columns or whole maps
 * of <= 16 x 16 outputs); seam_conv3x3_f16pc_pays = supported AND expected faster than seam_conv2d_f16 (tiles >= 3/4 full; a
 * function of the map geometry only, not of N) -- the rule the host side dispatches by. */
int seam_conv3x3_f16pc_supported(int N, int H, int W, int C, int K, int pad);
int seam_conv3x3_f16pc_pays(int N, int H, int W, int C, int K, int pad);
long long seam_f16pc_weight_halves(int K, int Cstore);
int seam_pack_conv_weight_f16pc(const float* w, void* w_packed, int K, int Cin, int Cstore, seam_stream_t stream);
int seam_conv3x3_f16pc(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                       int N, int H, int W, int C, int K, int pad, int relu, seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * The match trunk as one call (SURVEY.md 8b `seam_match_trunk_f32`): MatchPredictor / TemporalAggregationNLB's
 * conv_seq (4 valid 3x3 convs + ReLU, 14 -> 12 -> 10 -> 8 -> 6) -> AvgPool2d(6,6) + ReLU -> Linear(1024,256) + BatchNorm1d
 * (ref models/match_head.py:50-62,67-69,93-95).  roi NHWC [K,14,14,256] -> x3 [K,256].  Six launches on `stream` (why the
 * kernels are not fused further: csrc/seam_trunk.hip).  conv[4] / linear: the packed weights of the layers -- `w` from
 * seam_pack_conv_weight_f32 (required for `linear`; for a conv it may be NULL when a Winograd form is given), `u` / `u24` from
 * seam_pack_conv_weight_wino_f32 / _wino24_f32 or NULL, `scale` / `shift` the [K] epilogue vectors (bias; for `linear` the folded
 * BatchNorm1d) or NULL.  form: int[4] per conv -- 0 implicit GEMM, 1 F(2x2,3x3), 2 F(2x4,3x3) -- or NULL: chosen on the map
 * geometry as the Python layer does.  ws: >= seam_match_trunk_workspace_floats(K) floats. */
typedef struct {
    const float* w;
    const float* u;
    const float* u24;
    const float* scale;
    const float* shift;
} seam_trunk_layer_t;
int64_t seam_match_trunk_workspace_floats(int K);
int seam_match_trunk_f32(const float* roi, const seam_trunk_layer_t* conv, const seam_trunk_layer_t* linear, float* x3,
                         int K, float* ws, const int* form, seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Gradient kernels of the match heads: the grad-enabled pass of the training loop,
 * stuffs/engine.py:120-121,158-168,183-185 -> MatchPredictor / TemporalAggregationNLB in .train()
 * (models/match_head.py:66-76,90-169,339).  fp32, fixed-order reductions (bit-reproducible).
 *
 * seam_conv_wgrad_f32: dw [K,C,R,S] (OIHW) = sum_{n,ho,wo} dy[n,ho,wo,k] * x[n,ho*stride+r-pad,wo*stride+s-pad,c]
 *   for x NHWC [N,H,W,C], dy NHWC [N,Ho,Wo,K]; C,K multiples of 4.  fp32-MFMA GEMM split over the pixel
 *   axis; ws: >= seam_conv_wgrad_workspace_floats(M = N*Ho*Wo, C, K, R, S) floats.  (Linear: R=S=1, H=W=1.)
 * seam_colsum_f32: out[k] = sum_m x[m,k]  (bias gradients); ws: >= seam_colsum_workspace_floats(M,K) floats. */
int64_t seam_conv_wgrad_workspace_floats(int M, int C, int K, int R, int S);
int seam_conv_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K,
                        int R, int S, int stride, int pad, float* ws, seam_stream_t stream);
int64_t seam_colsum_workspace_floats(int M, int K);
int seam_colsum_f32(const float* x, float* out, int M, int K, float* ws, seam_stream_t stream);

/* Backward of conv_seq's last ReLU -> AvgPool2d(6,6) -> ReLU (models/match_head.py:56-60):
 * dy[n,hw,c] = y[n,hw,c] > 0 ? dpool[n,c] / HW : 0   (y = the ReLU'd conv output, NHWC [N,HW,C]). */
int seam_avgpool_relu_bwd_f32(const float* dpool, const float* y, float* dy, int N, int HW, int C,
                              seam_stream_t stream);

/* nn.BatchNorm1d(256) in training mode (models/match_head.py:62): batch statistics over the M rows (M >= 2),
 * running_mean/var updated in place with `momentum` (unbiased variance), or left alone when NULL;
 * save_mean / save_invstd [F] feed seam_bn1d_bwd_f32 (dx, dgamma, dbeta). */
int seam_bn1d_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                            float* save_mean, float* save_invstd, float* running_mean,
                            float* running_var, int M, int F, float momentum, float eps,
                            seam_stream_t stream);
int seam_bn1d_bwd_f32(const float* dy, const float* x, const float* save_mean, const float* save_invstd,
                      const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int F,
                      int frozen /* 1: eval-mode statistics, dx = gamma*invstd*dy */, seam_stream_t stream);

/* nn.CrossEntropyLoss(weight=[w0,w1]) over [n,2] logits with int64 targets, mean reduction -- the criterion of
 * every loss of models/match_head.py (:213,257,367,386): loss [1] and dloss/dlogits [n,2] in one launch. */
int seam_ce2_fwd_bwd_f32(const float* logits, const int64_t* target, const float* weight, float* loss,
                         float* dlogits, int64_t n, seam_stream_t stream);

/* Gradients of seam_pair_logits_f32: g [Q,G,2] -> da [Q,256], db [G,256], dw [2,256], dbias [2]. */
int seam_pair_logits_bwd_f32(const float* a, const float* b, const float* w, const float* g, float* da,
                             float* db, float* dw, float* dbias, int Q, int G, int D,
                             seam_stream_t stream);

/* Gradients of seam_nlb_attnpool_f32 (same operand layouts; Tmax <= 64): dout [S,256] -> dseq (same strides
 * as seq; only rows t < len[s] are written) and the parameter gradients, grads[11] = device pointers in the
 * reference's layouts: theta.weight [128,256], theta.bias [128], phi.weight, phi.bias, g.weight, g.bias,
 * concat_project.0.weight [256], W.weight [256,128], W.bias [256], attention_scorer.weight [256], .bias [1].
 * ws: >= seam_nlb_bwd_workspace_floats(S,Tmax) floats. */
int64_t seam_nlb_bwd_workspace_floats(int S, int Tmax);
int seam_nlb_attnpool_bwd_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S,
                              int Tmax, const float* w_proj_t, const float* b_proj, const float* w_cat,
                              const float* w_out_t, const float* b_out, const float* w_att,
                              const float* b_att, const float* dout, float* dseq, float* const* grads,
                              float* ws, int use_nlb, seam_stream_t stream);

/* Gradients of the non-local block ALONE -- the z output of seam_nlb_attnpool_f32, i.e. NONLocalBlock1D.forward called
 * directly and grad-enabled (ref models/nlb.py:66-101): dz rows at dz + s*dz_s_stride + t*dz_t_stride (256 floats each)
 * -> dseq (as above) and grads[9] = the first nine pointers of the list above (no attention scorer behind a direct call).
 * Same workspace as seam_nlb_attnpool_bwd_f32. */
int seam_nlb_block_bwd_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                           const float* w_proj_t, const float* b_proj, const float* w_cat,
                           const float* w_out_t, const float* b_out, const float* dz, int64_t dz_t_stride,
                           int64_t dz_s_stride, float* dseq, float* const* grads, float* ws, int use_nlb,
                           seam_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Input pipeline on the device: what MovingFashionDataset.__getitem__ does to a decoded
 * frame (datasets/MFDataset.py:79-93), on uint8 [H,W,3] device images.
 *
 * seam_frame_noise_u8: rgb = uint8(clip((bgr[...,::-1] / 255.0 + n * sigma) * 255.0, 0, 255)), float64 like NumPy;
 *   n = noise[H,W,3] float64 standard-normal draws, or (noise NULL) counter-based draws keyed by `seed`;
 *   sigma 0 with noise NULL = the noise=False branch (channel flip only).
 * seam_resize_bicubic_u8: PIL Image.resize((OW,OH)) -- BICUBIC, antialiased, 8-bit fixed point, horizontal then
 *   vertical pass -- bit-identical to Pillow (MFDataset.py:92: half resolution); ws: seam_resize_workspace_bytes. */
int seam_frame_noise_u8(const uint8_t* bgr, const double* noise, uint8_t* rgb, int H, int W, double sigma,
                        uint64_t seed, seam_stream_t stream);
int64_t seam_resize_workspace_bytes(int H, int W, int OH, int OW);
int seam_resize_bicubic_u8(const uint8_t* in, uint8_t* out, int H, int W, int OH, int OW, void* ws,
                           seam_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEAM_HIP_H */

/*
 * mrcnn_hip.h - C ABI of libmrcnn_hip.so: the MI355X (gfx950) kernels of the FPN Mask R-CNN
 * training path of katotetsuro/chainer-maskrcnn.
 *
 * The reference is pure Python on Chainer/CuPy and has no FFI of its own; every entry point
 * below names the reference interface (file:line under the reference tree) whose device work
 * it replaces.  The binding a reference maintainer would add is a ctypes stub - see
 * INTEGRATION.md.  The in-repo host layer (chainer-maskrcnn_amd/chainer_maskrcnn/_hip) is
 * exactly such a stub over torch tensors.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless marked "host".
 *   - the caller owns every buffer, including workspaces (query *_workspace_bytes first);
 *     the library never allocates or frees device memory and keeps no pointer past return.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream); no call synchronises the device - with ONE exception, the explicit status query
 *     mrcnn_roi_align_bwd_plan_status (ABI v9).  The composite calls (mrcnn_bottleneck_*) also enqueue on the
 *     caller's side stream, behind events of a small library-owned pool (no device memory).
 *   - return value: 0 = success; negative = MRCNN_E_* argument error; positive = hipError_t.
 *     mrcnn_last_error() returns a thread-local message for the last failure.
 *   - no C++ exceptions cross this boundary.
 *   - layouts: MRCNN_LAYOUT_NCHW is the reference's (Chainer) layout; MRCNN_LAYOUT_NHWC
 *     (channel innermost) is the layout the fast kernels are written for.  A torch tensor
 *     of logical shape (N,C,H,W) in torch.channels_last memory format IS MRCNN_LAYOUT_NHWC.
 */
#ifndef MRCNN_HIP_H
#define MRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRCNN_ABI_VERSION 10

enum {
    MRCNN_OK = 0,
    MRCNN_E_INVALID = -1,      /* bad argument (null pointer, non-positive size, ...) */
    MRCNN_E_UNSUPPORTED = -2,  /* valid but not implemented for this shape/layout */
    MRCNN_E_WORKSPACE = -3     /* workspace too small */
};

enum { MRCNN_LAYOUT_NCHW = 0, MRCNN_LAYOUT_NHWC = 1 };

#define MRCNN_MAX_LEVELS 8

int mrcnn_abi_version(void);
const char *mrcnn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * ROIAlign.  Replaces chainer_maskrcnn.functions.roi_align.roi_align_2d.roi_align_2d(x, rois,
 * outh, outw, spatial_scale) -- the un-vendored submodule called through
 * chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7 -- and its autograd backward.
 *
 *   x     (N,C,H,W) f32 in `layout`
 *   rois  (R,5) f32 rows (batch_idx, x1, y1, x2, y2) in image pixels (already xy order,
 *         i.e. AFTER the [0,2,1,4,3] permutation of roi_align_2d_yx.py:5)
 *   y     (R,C,PH,PW) f32 in `layout` (NHWC => (R,PH,PW,C))
 *   sampling_ratio  >0: fixed grid per bin; 0: adaptive ceil(roi_size/pooled_size)
 * Algorithm: Caffe2/Detectron RoIAlign (legacy, non-aligned), float32, no FMA contraction in
 * the coordinate arithmetic - identical, operation for operation, to oracle/roi_align.py.
 * ---------------------------------------------------------------------------------------- */
int mrcnn_roi_align_fwd_f32(const float *x, int layout, int N, int C, int H, int W,
                            const float *rois, int R, int PH, int PW, float spatial_scale,
                            int sampling_ratio, float *y, void *stream);
/* The same call with the caller's scratch of mrcnn_roi_align_fwd_workspace_bytes(R) bytes (ABI v6).  With it the NHWC fast path
 * walks the RoIs in MAP ORDER: a first small kernel ranks them by (level, image, 64-row band of the box centre, centre column)
 * and the forward kernel processes RoI perm[i] at position i, writing ITS output rows - the reference's per-RoI loop
 * (fpn_roi_mask_head.py:59-61) has no order dependence, the results are the same bits in the same rows, and co-resident waves
 * read neighbouring map lines instead of random ones.  Without scratch (or R < 128, or R > 8192): caller order. */
int mrcnn_roi_align_fwd_ws_f32(const float *x, int layout, int N, int C, int H, int W,
                               const float *rois, int R, int PH, int PW, float spatial_scale,
                               int sampling_ratio, float *y, void *ws, size_t ws_bytes, void *stream);
size_t mrcnn_roi_align_fwd_workspace_bytes(int R);
/* A/B switch of the map-order walk (process-wide; default 1 = on when scratch is given). */
int mrcnn_roi_align_set_fwd_map_order(int on);

/* Backward (adjoint scatter).  gx is fully overwritten (callee zero-fills cells no RoI touches).
 * Fast path (NHWC, C%4==0, PH,PW<=16, sampling_ratio>0): owner-computes tiles, no atomics,
 * bit-reproducible.  Other shapes: memset + atomic scatter.
 * mrcnn_roi_align_bwd_ws_f32 takes the caller's scratch of mrcnn_roi_align_bwd_workspace_bytes() bytes (the RoI-split slabs of small
 * maps); mrcnn_roi_align_bwd_f32 = the same call without scratch.  Both give identical bits. */
int mrcnn_roi_align_bwd_f32(const float *gy, int layout, int N, int C, int H, int W,
                            const float *rois, int R, int PH, int PW, float spatial_scale,
                            int sampling_ratio, float *gx, void *stream);
int mrcnn_roi_align_bwd_ws_f32(const float *gy, int layout, int N, int C, int H, int W,
                               const float *rois, int R, int PH, int PW, float spatial_scale,
                               int sampling_ratio, float *gx, void *ws, size_t ws_bytes, void *stream);
size_t mrcnn_roi_align_bwd_workspace_bytes(int N, int C, int H, int W, int R, int PH, int PW, int sampling_ratio);

/* Multi-level (FPN) batched variants: ONE launch for all RoIs over all pyramid levels.
 * Replace the per-RoI Python loops of chainer_maskrcnn/model/head/fpn_roi_mask_head.py:59-61,
 * 75-77,93-94 (and fpn_roi_keypoint_head.py:63-68,85-86,102-103).
 *   xs / gxs        host arrays of L device base pointers (level l: (N,C,Hs[l],Ws[l]) NHWC)
 *   Hs, Ws, scales  host arrays of length L  (scales[l] = spatial_scales[l])
 *   levels          (R,) int32 device, level of each RoI (already clipped to [0,L))
 *   rois            (R,5) f32 device (batch_idx,x1,y1,x2,y2)
 *   y / gy          (R,PH,PW,C) f32
 *   accumulate      backward: 0 = every gxs[l] is overwritten (zero where no RoI lands),
 *                   1 = gradients are added to gxs[l] (second pooled size over the same pyramid)
 *   ws, ws_bytes    backward: optional device scratch of mrcnn_roi_align_fpn_bwd_workspace_bytes() bytes
 *                   ([slabs][per-RoI tables of variant 3]).  With it, levels with few 8x8 tiles (the coarse ones, where map_rois_to_fpn_levels puts most RoIs) are
 *                   computed by several workgroups per tile over disjoint RoI subsets and summed in fixed order;
 *                   without it (NULL) one workgroup per tile does all the RoIs.  Results are bit-reproducible
 *                   either way (the two modes differ from each other by summation order only).
 * Only MRCNN_LAYOUT_NHWC with C%4==0 is supported. */
int mrcnn_roi_align_fpn_fwd_f32(const float *const *xs, const int *Hs, const int *Ws,
                                const float *scales, int L, int N, int C, const float *rois,
                                const int32_t *levels, int R, int PH, int PW,
                                int sampling_ratio, float *y, void *stream);
/* forward with scratch of mrcnn_roi_align_fwd_workspace_bytes(R) bytes: map-order walk, see mrcnn_roi_align_fwd_ws_f32 */
int mrcnn_roi_align_fpn_fwd_ws_f32(const float *const *xs, const int *Hs, const int *Ws,
                                   const float *scales, int L, int N, int C, const float *rois,
                                   const int32_t *levels, int R, int PH, int PW,
                                   int sampling_ratio, float *y, void *ws, size_t ws_bytes, void *stream);
int mrcnn_roi_align_fpn_bwd_f32(const float *gy, float *const *gxs, const int *Hs, const int *Ws,
                                const float *scales, int L, int N, int C, const float *rois,
                                const int32_t *levels, int R, int PH, int PW,
                                int sampling_ratio, int accumulate, void *ws, size_t ws_bytes, void *stream);
size_t mrcnn_roi_align_fpn_bwd_workspace_bytes(const int *Hs, const int *Ws, int L, int N, int C, int R, int PH, int PW,
                                               int sampling_ratio);
/* (ABI v9) The backward in two launches.  Which (RoI, bin) pairs land on which 4 x 4 patch of the gradient map - two thirds of a wave's
 * life in the fused backward - depends on the RoIs only, and in a training step the RoIs are known a whole head before the backward
 * (the reference runs the forward with them: fpn_roi_mask_head.py:59-61,75-77).  mrcnn_roi_align_fpn_bwd_plan_f32 writes every patch's
 * entry list ((gy row, 4 row weights, 4 column weights), in (RoI, ph, pw) order) into the caller's `plan` buffer of
 * mrcnn_roi_align_fpn_bwd_plan_bytes() bytes - on any stream, any time after the RoIs exist; mrcnn_roi_align_fpn_bwd_planned_f32 is
 * mrcnn_roi_align_fpn_bwd_f32 that streams gy along those lists: same entry order, same weights, same FMAs = the same bits FOR FINITE gy
 * (the lean kernel multiplies every row of a patch by its weight, zero weights included, where the fused kernel skips a zero-weight row: an
 * Inf / NaN in gy reaches all 16 cells of the patches it touches instead of the rows it lands on - still non-finite output, other cells).
 * The plan is checked on the device (magic, level count and every level's H x W, N, RoI count, pooled size, sampling ratio, node capacity
 * against this call's plan_bytes; a tile whose entries did not fit the pool is flagged): what the plan does not hold is computed by the
 * fused kernel in a second launch.
 * The buffer must not be modified between the two calls, and both must see the same rois / levels / scales; split_levels != 0 iff the
 * backward call is given its mrcnn_roi_align_fpn_bwd_workspace_bytes() scratch (the coarse levels' RoI split).  L == 1 with
 * levels == NULL is the single-level form (configs[1]). */
size_t mrcnn_roi_align_fpn_bwd_plan_bytes(const int *Hs, const int *Ws, int L, int N, int R, int PH, int PW, int split_levels);
int mrcnn_roi_align_fpn_bwd_plan_f32(const int *Hs, const int *Ws, const float *scales, int L, int N, int C, const float *rois,
                                     const int32_t *levels, int R, int PH, int PW, int sampling_ratio, int split_levels,
                                     void *plan, size_t plan_bytes, void *stream);
/* measurement knob of the lean backward: gy rows in flight per wave / waves per SIMD / threads per workgroup: 0 = 10 / 8 / 512, 1 = 16 / 7 / 512,
 * 2 = 8 / 8 / 512 (a workgroup = one 8 x 8 tile), 8 = 8 / 8 / 256 (half a tile: the default), 9 = 8 / 8 / 128 (one patch); + 256 x bits
 * (1: no gy loads, 2: no gx stores - wrong results; 16: plain instead of non-temporal stores) */
int mrcnn_debug_roi_align_lean_variant(int v);
/* measurement: device buffer of 10 x u64 per wave (8 waves per 8 x 8 tile, tile-major) that the lean backward of the next planned calls fills -
 * s_memtime at entry / first node's loads back / end of the entry loop / stores acknowledged, entries, HW_ID | XCC_ID << 32, s_memrealtime
 * at entry / at the end, s_memtime when the kernel arguments / the node's scalar loads are back; null = off */
int mrcnn_debug_roi_align_lean_stamps(unsigned long long *stamps);
int mrcnn_roi_align_fpn_bwd_planned_f32(const float *gy, float *const *gxs, const int *Hs, const int *Ws, const float *scales, int L,
                                        int N, int C, const float *rois, const int32_t *levels, int R, int PH, int PW,
                                        int sampling_ratio, int accumulate, void *ws, size_t ws_bytes, void *plan,
                                        size_t plan_bytes, int plan_verified, void *stream);
/* plan_verified != 0: the caller has read mrcnn_roi_align_bwd_plan_status() for this plan after the builder finished and found
 * status3[0] == 1 (the builder's header) and status3[1] == 0 (no tile flagged): the call is then the lean kernel alone.  With 0 a second
 * launch follows it that computes the tiles the plan could not hold with the fused kernel - normally none, every workgroup leaves at
 * once, but their dispatch costs 3.5 - 5 us.  A training step passes 0 (it cannot afford the status query's synchronisation and does not
 * feel 5 us); the configs[1] microbenchmark verifies once, outside the timed region, and reports both.  status3 (host, 3 ints): header
 * valid, tiles flagged, pool nodes used.  mrcnn_roi_align_bwd_plan_status is the one entry point that WAITS for the device. */
int mrcnn_roi_align_bwd_plan_status(const void *plan, size_t plan_bytes, int *status3, void *stream);

/* Process-wide choice of the fast backward kernel: 2 (default) = one independent wave per 4x4 cell patch that derives the
 * geometry itself; 1 = the barrier-synchronised 8x8 tile kernel of round 1 (what tensors >= 4 GiB fall back to; selectable so
 * that tests reach it on small inputs).  Same results contract.  (ABI v10: the table-driven variant 3 and the forward-built
 * work plan of ABI v8 - mrcnn_roi_align_plan_workspace_bytes, mrcnn_roi_align_set_bwd_plan - are gone: both were measured
 * slower for the forward + backward pair and never shipped.) */
int mrcnn_roi_align_set_bwd_variant(int variant);

/* Diagnostic build of the backward kernel with s_memtime stamps at the phase boundaries of every wave (tools/roi_stamps.py;
 * measurement only).  stamps: 12 x u64 per wave, (8 * ceil(tiles / 8)) workgroups x 4 waves. */
int mrcnn_debug_roi_align_bwd_stamps(const float *gy, int N, int C, int H, int W, const float *rois, int R, int PH, int PW,
                                     float spatial_scale, int sampling_ratio, float *gx, unsigned long long *stamps,
                                     void *stream);

/* Measurement: block -> (XCD, CU) placement of a launch shaped like the ROIAlign backward (256 threads per block, all blocks co-resident
 * for spin_us): out[b] = HW_ID | XCC_ID << 32, out[nblocks + b] = s_memrealtime at the start of block b. */
int mrcnn_debug_dispatch_census(unsigned long long *out, int nblocks, int spin_us, void *stream);

/* Verification hook for the "ROIAlign indices bit-exact" contract: dumps, for every RoI and
 * both axes, the integer corner cells and float weights of every sample exactly as the
 * kernels compute them (same __device__ function).  Shapes as oracle.roi_align.
 * roi_align_sample_tables: cnt (R,2) i32, idx (R,2,smax,2) i32, wgt (R,2,smax,2) f32. */
int mrcnn_roi_align_sample_tables(const float *rois, int R, int H, int W, int PH, int PW,
                                  float spatial_scale, int sampling_ratio, int smax,
                                  int32_t *cnt, int32_t *idx, float *wgt, void *stream);

/* ------------------------------------------------------------------------------------------
 * Convolution (fp32 MFMA implicit GEMM).  Replaces the cuDNN/cuBLAS calls Chainer issues for
 * L.Convolution2D / L.Linear / L.Deconvolution2D in
 *   chainer_maskrcnn/model/extractor/feature_pyramid_network.py:22-40,48-68
 *   chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:80-86,131-141
 *   chainer_maskrcnn/model/head/fpn_roi_mask_head.py:24-49,65-83
 * and their autograd backward passes.
 *   x  (N,H,W,Cin) NHWC      w  (Cout,KH,KW,Cin)      y  (N,Ho,Wo,Cout) NHWC
 *   Ho = (H + 2*pad - KH)/stride + 1.   Cout must be a multiple of 32 and Cin a multiple of 32 or
 *   exactly 4 (the image layer: K axis = (tap, 4 channels)); the host layer zero-pads channel
 *   counts such as 3, 18, 80, 81.  bias (Cout) may be NULL; relu != 0
 *   fuses max(.,0) into the forward epilogue.  Linear layers are 1x1 convolutions with
 *   H = W = 1; the 2x2/2 deconvolution is a 1x1 convolution to 4*Cout channels + a host view.
 * fwd / bwd_data take an optional workspace (mrcnn_conv2d_workspace_bytes; may be NULL/0): deep layers with few
 * pixels and a long K axis are split over K into slabs summed in fixed order (deterministic), so all CUs work.
 * bwd_data supports stride 1 only (a strided 1x1 convolution is a stride-1 one on the subsampled
 * lattice followed by mrcnn_subsample_bwd_f32); accumulate != 0 adds into gx instead of overwriting.  bwd_filter accumulates over pixels with a deterministic
 * split-K (slabs in the caller's workspace, fixed summation order); gbias may be NULL; accumulate != 0
 * adds into gw / gbias (layers shared by several inputs, e.g. the RPN head over 5 pyramid levels).
 * ---------------------------------------------------------------------------------------- */
size_t mrcnn_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
/* Multiply-accumulates the MFMA pipes execute for ONE pass (forward, backward-data or backward-filter) of this layer:
 * N*Ho*Wo*KH*KW*Cin*Cout for the direct kernels, (m+2)^2 * tiles * Cin * Cout (2.25x / 4x fewer) where the 3x3 /
 * stride 1 / pad 1 layer takes the Winograd F(m x m,3x3) path (>= 256 channels, >= 2048 pixels) in that pass (pass: 0 forward,
 * 1 backward-data, 2 backward-filter).  For roofline accounting. */
long long mrcnn_conv2d_executed_macs(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int pass);
/* Process-wide selection of the Winograd path: thresholds (defaults 256 channels, 2048 pixels: measured break-even on
 * gfx950) and output tile (0 = per layer, whichever of F(2x2,3x3) / F(4x4,3x3) needs fewer multiplications; 2 or 4 =
 * forced).  Lowering the thresholds is how the tests run whole small networks through the Winograd kernels. */
int mrcnn_conv2d_set_winograd_thresholds(int min_channels, int min_pixels, int tile);
/* Per-pass override of the Winograd tile (forward, backward-data, backward-filter): 0 = follow the global tile above,
 * 2 / 4 = forced, -1 = that pass never takes the Winograd path.  Default (2, 0, 0): F(4x4,3x3) amplifies float32 rounding
 * far more than F(2x2,3x3).  In the forward pass of the ResNet conv2 layers that shows up as parameter-gradient errors of
 * up to 2e-2 (300x the float32 noise floor) in the layers fed by c4 / c5; in the backward passes, and in the forward pass of
 * the layers behind the backbone, it does not (profiles/r02_winograd_pass_probe.txt, r02_winograd_layer_probe.txt) - the
 * host layer therefore brackets the calls of its FPN / RPN / head convolutions with (0, ., .) (nn/core.py: Conv(fwd_tile=0)).
 * (0, 0, 0) = F(4x4) wherever it is cheaper in every pass of every layer: activations still within 2.2e-4 of their scale.
 * The setting is read on the host when a convolution entry point is called.  The getter writes the three current values. */
int mrcnn_conv2d_set_winograd_pass_tiles(int fwd, int bwd_data, int bwd_filter);
int mrcnn_conv2d_get_winograd_pass_tiles(int *tiles3);
/* EXPLORATORY, opt-in, never the default (and never bench.py's `value`): the forward-kind GEMM launches (forward convolutions and
 * both Winograd batched GEMMs) of a pass (forward, backward-data, backward-filter) with three-term split operands on the 16-bit
 * MFMA - every float32 operand staged as hi + lo planes, float32 accumulation of al*bh + ah*bl + ah*bh: 0 = float32 MFMA,
 * 1 = bf16 planes (16 significant bits, float32 range; ~4e-6 per product), 2 = IEEE-half planes (22 bits, ~5e-7 per product;
 * operands must stay below 65504 and lose relative precision below 6e-5), 3 = THREE bf16 planes hi + mid + lo (= the float32
 * operand exactly) and the six products of weight >= 2^-16 - a float32-ACCURATE emulation (the dropped terms are <= 3 x 2^-24 of
 * |ab|, the size of the float32 MFMA's own accumulation rounding) at 3/8 of the float32 MFMA cycles.  gfx950 has no xf32: this is
 * what challenging the 157.3 TF/s fp32-MFMA ceiling costs in accuracy and buys in time.  Values outside 0..3: MRCNN_E_ARG. */
int mrcnn_conv2d_set_split_operands(int fwd, int bwd_data, int bwd_filter);
/* The three modes in force (forward, backward-data, backward-filter), for callers that bracket a call with their own setting. */
int mrcnn_conv2d_get_split_operands(int *modes3);
/* Measurement knob (tools/gemm_only_profile.py): workgroups per CU the tile choice and the forward / backward-data split-K plan aim
 * for (default 2), the HALF rounds of workgroup slots the filter-gradient split-K fills (default 2 = one round), and a forced forward /
 * backward-data tile (0 = the planner's choice, 1 = 128x64, 2 = 64x64). */
int mrcnn_debug_conv_plan(int fill, int filter_rounds, int force_tile);
/* Measurement knob, tile order of the Winograd input transforms (speed only, same results): 0 = launch order, 1 = XCD-banded raster order,
 * n >= 2 = XCD-banded column panels n tiles wide (default 16). */
int mrcnn_debug_wino_banded(int on);
/* Measurement knob, split-operand GEMM kernels only: 1 = the MFMAs are skipped, 2 = the global loads inside the K loop are skipped,
 * 4 = the epilogue is skipped (results are garbage while one of these bits is set; where does such a kernel's time go?);
 * 8 = the three-plane (bf16x6) kernels run the plain K loop (split + LDS stores between the barriers) instead of the pipelined one
 * (split in registers in the MFMAs' shadow, loads two steps ahead) - same results, for A/B.  Bits 1 and 2 act on the plain loop. */
int mrcnn_debug_conv_parts(int mask);
/* Test / measurement entry points of the plane GEMMs (csrc/planes_gemm.h), the kernels behind split mode 3 on the Winograd path.
 * split_planes: float32 (R, C), C % 16 == 0 -> "P16" planes, unsigned short [R][C/16][3][16]: hi, mid, lo bf16 planes of 16
 * consecutive channels in 96 contiguous bytes (hi + mid + lo == x exactly).
 * planes_gemm, kind 0 (F): c (M, N) = sum_k a[m][k] b[batch(m)][n][k] - a = P16 (M, K), b = P16 (nbatch * N, K), batch(m) =
 * m / batch_rows, M = nbatch * batch_rows, batch_rows % 128 == 0; tile bm x bn = 128 x 128 | 64.
 * kind 1 (G): c (ksplit, M, nbatch, N) partial sums over the row ranges of a = P16 (nbatch * K, M), b = P16 (nbatch * K, N),
 * K = batch_rows, K % (16 * ksplit) == 0; tiles 128 | 64 each way.  Six v_mfma_f32_32x32x16_bf16 per product, float32 accumulate. */
int mrcnn_debug_split_planes_f32(const float *x, void *planes, int R, int C, int r4, void *stream);
int mrcnn_debug_planes_gemm_stamps(unsigned long long *stamps);
int mrcnn_debug_planes_gemm(int kind, const void *a, const void *b, float *c, int M, int N, int K, int batch_rows, int nbatch,
                            int ksplit, int bm, int bn, void *stream);

/* Measurement knob for bench.py's roofline split (never set on a product path): bit 0 skips the MFMA GEMM launches of
 * the convolution calls, bit 1 skips every other kernel they launch (Winograd transforms, slab / tail / column sums),
 * bit 2 makes the filter-gradient call return at once, bit 3 skips only that call's split-K slab sums (tools/ab_step.py:
 * what the weight-gradient stream costs the step).  Outputs are garbage while a bit is set; 0 restores normal operation. */
int mrcnn_conv2d_set_debug_skip(int mask);
/* The same kind of knob for the BatchNorm calls: bit 0 leaves out the backward's finalisation launch, bit 1 the forward's (the form fed by
 * the convolution epilogue's partial sums); the outputs of the skipped launch keep whatever their buffers held. */
int mrcnn_debug_bn_skip(int mask);
/* Measurement knobs of the channel-wise reductions (A/B in one process, tools/bn_plan_ab.py; the defaults are 1024, 4): most row
 * blocks = partial rows a reduction pass may use (also sizes mrcnn_bn_workspace_bytes: set it before the sizes are asked for), and
 * the channel quads one finalisation workgroup sums (4: 64 row slices x 16 channels; 1: 256 slices of one quad - slower). */
int mrcnn_debug_bn_plan(int red_cap, int fin_quads);
/* wino_v (nullable): caller-owned buffer of mrcnn_conv2d_winograd_v_bytes() bytes (0 = the layer does not take the
 * Winograd path).  The forward pass leaves its transformed input there and the filter-gradient pass of the same layer
 * reads it instead of transforming x again (same call geometry, same Winograd settings). */
size_t mrcnn_conv2d_winograd_v_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mrcnn_conv2d_fwd_f32(const float *x, const float *w, const float *bias, float *y, int N, int H,
                         int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                         float *wino_v, void *ws, size_t ws_bytes, void *stream);
/* Forward with a rectangular kernel and per-axis padding: the (15,1) / (1,15) separable pairs of the Light-Head R-CNN
 * head (chainer_maskrcnn/model/head/light_roi_mask_head.py:29-44).  Workspace as mrcnn_conv2d_workspace_bytes() of the
 * same geometry with pad = max(pad_h, pad_w). */
/* Convolution feeding a training-mode BatchNorm (no bias, no ReLU: extractor/feature_pyramid_network.py:48-66, Chainer's
 * ResNet50Layers): the GEMM epilogue also leaves per-channel sums / sums of squares of every block of output rows in
 * bn_part (rows, 2, Cout); mrcnn_bn_train_fwd_stats_f32 finishes BatchNorm from them without a statistics pass over the
 * activation.  Winograd layers produce the same partials in their output transform (wino_v / ws as for
 * mrcnn_conv2d_fwd_f32).  mrcnn_conv2d_bnstats_rows = rows for this geometry, or 0 when the call would take a path without
 * the fused statistics (split-K or tail-split launches): use the plain entry points then. */
size_t mrcnn_conv2d_bnstats_rows(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mrcnn_conv2d_fwd_bnstats_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                 int stride, int pad, float *bn_part, float *wino_v, void *ws, size_t ws_bytes, void *stream);
int mrcnn_conv2d_fwd_rect_f32(const float *x, const float *w, const float *bias, float *y, int N, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int relu, void *ws,
                              size_t ws_bytes, void *stream);
/* relu_x (nullable, same shape as gx): the layer's input when it is the output of a ReLU; gx is then zeroed where
 * relu_x <= 0, i.e. the ReLU backward of the layer below is fused into this epilogue; with accumulate (ABI v7) the old value is
 * added first and the TOTAL is masked (the last contribution to a fan-out point applies the mask to all of them). */
/* wino_w (nullable, mrcnn_conv2d_winograd_w_bytes() bytes; only where that is > 0): the Winograd path reads gy ONCE
 * and leaves the filter-gradient GEMM's operand (A dy A^T) there for the mrcnn_conv2d_bwd_filter_f32 call of the same
 * layer; with wino_w, gbias (nullable, Cout) receives the bias gradient (column sums of gy; added to when
 * gbias_accumulate != 0) from the same read. */
size_t mrcnn_conv2d_winograd_w_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mrcnn_conv2d_bwd_data_f32(const float *gy, const float *w, float *gx, const float *relu_x, int N, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad, int accumulate, float *wino_w,
                              float *gbias, int gbias_accumulate, void *ws, size_t ws_bytes, void *stream);
size_t mrcnn_conv2d_bwd_filter_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                               int stride, int pad);
int mrcnn_conv2d_bwd_filter_f32(const float *x, const float *gy, float *gw, float *gbias, int N, int H,
                                int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                                const float *wino_v, const float *wino_w,
                                void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backbone / head layer kernels (nn.hip), NHWC fp32, "(P, C)" = P pixels x C channels, C % 4 == 0.
 * Replace the cuDNN / CuPy kernels behind ResNet50Layers' BatchNormalization + ReLU + residual add,
 * F.max_pooling_2d(ksize=2) (cover_all), F.unpooling_2d(ksize=2, outsize) + add
 * (chainer_maskrcnn/model/extractor/feature_pyramid_network.py:48-66), the ReLUs and the
 * Deconvolution2D data movement of chainer_maskrcnn/model/head/fpn_roi_mask_head.py:65-83, and the
 * optimizer of train.py:107-109.
 * ---------------------------------------------------------------------------------------- */
size_t mrcnn_bn_workspace_bytes(int P, int C);
/* Training-mode BN: batch statistics over P, y = gamma*(x-mean)*invstd + beta (+ residual) (ReLU if
 * relu != 0); save_mean / save_invstd (C) are kept for backward; running_* (nullable) are updated with
 * Chainer's rule (decay, unbiased variance). */
int mrcnn_bn_train_fwd_f32(const float *x, const float *gamma, const float *beta, const float *residual,
                           float *y, float *save_mean, float *save_invstd, float *running_mean,
                           float *running_var, int P, int C, float eps, float decay, int relu, void *ws,
                           size_t ws_bytes, void *stream);
/* dz = relu ? gy*(y>0) : gy;  gx = BN backward of dz;  gres (nullable) = dz (gradient of the residual
 * input);  ggamma, gbeta (C) overwritten.  y may be NULL when relu != 0, the forward had no residual and beta is
 * given: the mask is then recomputed from x as gamma*(x-mean)*invstd + beta > 0 with the forward's exact expression
 * (bitwise the same mask, one HBM stream less).  beta is otherwise unused (nullable). */
int mrcnn_bn_train_fwd_stats_f32(const float *x, const float *part, int rows, const float *gamma, const float *beta,
                                 const float *residual, float *y, float *save_mean, float *save_invstd, float *running_mean,
                                 float *running_var, int P, int C, float eps, float decay, int relu, void *stream);
int mrcnn_bn_train_bwd_f32(const float *gy, const float *x, const float *y, const float *gamma, const float *beta,
                           const float *save_mean, const float *save_invstd, float *gx, float *gres,
                           float *ggamma, float *gbeta, int P, int C, int relu, void *ws, size_t ws_bytes,
                           void *stream);
/* (ABI v10) Two training-mode BatchNorms that meet in one residual sum - the main branch and the projection shortcut of a ResNet bottleneck
 * (extractor/feature_pyramid_network.py:48-66; Chainer's BottleneckA): y = relu(BN_a(xa) + BN_b(xb)).  Statistics of each layer as in the
 * single-layer entry points (part_x / rows_x: the partial rows its convolution left behind, mrcnn_conv2d_fwd_bnstats_f32; NULL / 0 = a
 * statistics pass over the tensor through ws of mrcnn_bn_workspace_bytes()), then ONE apply kernel: the shortcut's BatchNorm output is never
 * written.  The backward pair: dz = y ? gy * (y > 0) : gy (y NULL = gy arrives masked); gxa / gxb = BatchNorm backward of dz through each
 * layer (ws of mrcnn_bn_pair_workspace_bytes()); the masked gradient is never written either.  Same bits as the layer-by-layer sequence
 * mrcnn_bn_train_fwd(_stats)_f32(xb -> r), mrcnn_bn_train_fwd(_stats)_f32(xa, residual r, relu) and its two mrcnn_bn_train_bwd_f32 calls:
 * 5 passes over a (P, C) tensor fewer per projection block and step. */
/* (ABI v10) conv -> BatchNorm -> ReLU -> conv 3x3 without the normalised activation (the middle of a ResNet bottleneck,
 * extractor/feature_pyramid_network.py:48-66): mrcnn_bn_train_stats_f32 finishes the statistics of the first convolution's output (partial rows
 * from its epilogue, or NULL / 0 = a statistics pass through ws) and updates the running statistics - no apply kernel; the consumer's Winograd
 * input transform applies gamma * ((x - mean) * invstd) + beta and the ReLU to every tap as it loads it (taps outside the image stay zero):
 * mrcnn_conv2d_fwd_inbn_f32 (bn_part nullable: also the statistics partials of ITS output, as mrcnn_conv2d_fwd_bnstats_f32) and
 * mrcnn_conv2d_bwd_filter_inbn_f32 (wino_v as mrcnn_conv2d_bwd_filter_f32).  Same bits as the materialised sequence; 2 passes over the
 * (P, C) activation and one launch fewer.  Only where mrcnn_conv2d_inbn_ok() != 0 (3x3 / stride 1 / pad 1 geometries whose forward and
 * filter-gradient passes both take the Winograd path under the process settings in force); the entry points refuse other geometries. */
int mrcnn_bn_train_stats_f32(const float *x, const float *part, int rows, float *save_mean, float *save_invstd, float *running_mean,
                             float *running_var, int P, int C, float eps, float decay, void *ws, size_t ws_bytes, void *stream);
int mrcnn_conv2d_inbn_ok(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mrcnn_conv2d_fwd_inbn_f32(const float *x, const float *in_gamma, const float *in_beta, const float *in_mean, const float *in_invstd,
                              const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                              float *bn_part, float *wino_v, void *ws, size_t ws_bytes, void *stream);
int mrcnn_conv2d_bwd_filter_inbn_f32(const float *x, const float *in_gamma, const float *in_beta, const float *in_mean,
                                     const float *in_invstd, const float *gy, float *gw, int N, int H, int W, int Cin, int Cout, int KH,
                                     int KW, int stride, int pad, int accumulate, const float *wino_v, void *ws, size_t ws_bytes,
                                     void *stream);
size_t mrcnn_bn_pair_workspace_bytes(int P, int C);
int mrcnn_bn_train_fwd_pair_f32(const float *xa, const float *part_a, int rows_a, const float *gamma_a, const float *beta_a,
                                float *mean_a, float *invstd_a, float *run_mean_a, float *run_var_a, const float *xb,
                                const float *part_b, int rows_b, const float *gamma_b, const float *beta_b, float *mean_b,
                                float *invstd_b, float *run_mean_b, float *run_var_b, float *y, int P, int C, float eps, float decay,
                                void *ws, size_t ws_bytes, void *stream);
int mrcnn_bn_train_bwd_pair_f32(const float *gy, const float *y, const float *xa, const float *xb, const float *gamma_a,
                                const float *mean_a, const float *invstd_a, const float *gamma_b, const float *mean_b,
                                const float *invstd_b, float *gxa, float *gxb, float *ggamma_a, float *gbeta_a, float *ggamma_b,
                                float *gbeta_b, int P, int C, void *ws, size_t ws_bytes, void *stream);
/* Inference-mode BN with the running statistics (chainer.config.train == False, maskrcnn.py:171-172). */
int mrcnn_bn_infer_fwd_f32(const float *x, const float *gamma, const float *beta, const float *mean,
                           const float *var, const float *residual, float *y, int P, int C, float eps, int relu,
                           void *stream);
int mrcnn_relu_bwd_f32(const float *gy, const float *y, float *gx, size_t n, void *stream);
int mrcnn_add_f32(const float *a, const float *b, float *out, size_t n, void *stream);
int mrcnn_maxpool2x2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);
int mrcnn_maxpool2x2_bwd_f32(const float *x, const float *gy, float *gx, int N, int H, int W, int C, void *stream);
/* Legacy variants (forward only): F.max_pooling_2d(ksize=3, stride=2) with cover_all
 * (chainer_maskrcnn/model/extractor/c4_backbone.py:20; Ho = (H-3+1)/2 + 1), global average pooling of x (R,P,C) -> (R,C)
 * (model/head/resnet_roi_mask_head.py:65) and a stand-alone ReLU (x == y allowed). */
int mrcnn_maxpool3x3s2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);
int mrcnn_global_avg_pool_fwd_f32(const float *x, float *y, int R, int P, int C, void *stream);
int mrcnn_relu_fwd_f32(const float *x, float *y, size_t n, void *stream);
/* out (N,H,W,C) = nearest-2x(top (N,Ht,Wt,C)) cropped + lat;  backward: gtop (+)= 2x2 block sums of gout. */
int mrcnn_upsample2x_add_fwd_f32(const float *top, const float *lat, float *out, int N, int H, int W, int Ht,
                                 int Wt, int C, void *stream);
int mrcnn_upsample2x_bwd_f32(const float *gout, float *gtop, int N, int H, int W, int Ht, int Wt, int C,
                             int accumulate, void *stream);
/* Backward of the lattice subsampling x[:, ::s, ::s] (strided 1x1 convolutions): gx (N,H,W,C).  relu_x (nullable, ABI v7; shape of
 * gx): the lattice positions are zeroed where relu_x <= 0 after the accumulation - the ReLU backward of the layer that produced the
 * subsampled tensor, applied where its gradient is written last (positions off the lattice keep gx's contents). */
int mrcnn_subsample_bwd_f32(const float *gsub, float *gx, int N, int H, int W, int C, int stride, int accumulate,
                            const float *relu_x, void *stream);
/* 2x2/2 deconvolution data movement: (N,H,W,[2][2][C]) <-> (N,2H,2W,C); inverse != 0 is the backward. */
int mrcnn_pixel_shuffle2x_f32(const float *src, const float *bias, float *dst, int N, int H, int W, int C, int inverse,
                              void *stream);      /* bias (C, nullable) is added in the forward direction */
/* Algebraic merge of the mask / keypoint branch's last two layers: the reference has no non-linearity between the
 * 2x2/2 deconvolution and the final 1x1 convolution (head/fpn_roi_mask_head.py:83, fpn_roi_keypoint_head.py:93), so
 * conv2(deconv1(x)) is one 2x2/2 deconvolution to K2 channels (4x fewer MACs at 28x28, no (R,28,28,C) intermediate).
 *   wd (4*C, Cin) rows (a*2+b)*C + o;  bd (C);  w2 (K2, ld2 >= C) (columns >= C are channel padding and are left
 *   untouched in gw2);  b2 (K2, nullable)
 *   fwd: wm (4*K2, Cin) rows (a*2+b)*K2 + k = W2 * Wd[ab];  bm (K2) = b2 + W2 * bd
 *   bwd: G (4*K2, Cin), gb4 (4*K2) = filter / bias gradient of the merged layer  ->  gwd, gbd, gw2, gb2 (nullable),
 *        all overwritten; exact gradients of the un-merged parameters, fixed summation order. */
int mrcnn_deconv_merge_fwd_f32(const float *wd, const float *bd, const float *w2, const float *b2, float *wm,
                               float *bm, int C, int Cin, int K2, int ld2, void *stream);
int mrcnn_deconv_merge_bwd_f32(const float *G, const float *gb4, const float *wd, const float *bd,
                               const float *w2, float *gwd, float *gbd, float *gw2, float *gb2, int C,
                               int Cin, int K2, int ld2, void *stream);
/* Bilinear x2 with corner alignment (Chainer F.resize_images, chainer_maskrcnn/model/head/fpn_roi_keypoint_head.py:
 * 80-81,109): x (N,H,W,C) -> y (N,2H,2W,C); bwd is the exact adjoint (owner-computes, no atomics). */
int mrcnn_bilinear2x_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);
int mrcnn_bilinear2x_bwd_f32(const float *gy, float *gx, int N, int H, int W, int C, void *stream);
/* x (N,3,H,W) NCHW -> y (N,H,W,4) NHWC with a zero 4th channel (the image layer's operand layout). */
int mrcnn_image_nchw3_to_nhwc4_f32(const float *x, float *y, int N, int H, int W, void *stream);
/* Device side of the training Transform (train.py:21-37; chainer_maskrcnn/dataset/transforms.py is the host form):
 * the raw uint8 image (H,W,3) / instance masks (G,H,W) are resized into planes of a zero-initialised padded batch tensor
 * (dst_h x dst_w per plane; rows < oh and columns < ow are written).  cv2.resize INTER_LINEAR / INTER_NEAREST
 * coordinate rules; the image is divided by `div` (255 = MaskRCNN.prepare, maskrcnn.py:274). */
int mrcnn_image_resize_u8_f32(const uint8_t *src, int H, int W, float *dst, int oh, int ow, int dst_h, int dst_w,
                              float div, void *stream);
int mrcnn_mask_resize_nearest_u8(const uint8_t *src, int G, int H, int W, uint8_t *dst, int oh, int ow, int dst_h,
                                 int dst_w, void *stream);
/* MaskRCNN.prepare (chainer_maskrcnn/model/maskrcnn.py:261-276): the float32 CHW image (C,H,W) resized with cv2's
 * INTER_LINEAR float rule (what chainercv.transforms.resize calls) into planes of dst_h x dst_w, then divided by div. */
int mrcnn_image_resize_f32(const float *src, int C, int H, int W, float *dst, int oh, int ow, int dst_h, int dst_w,
                           float div, void *stream);
/* n uint32 sampler keys from a counter-based hash of (seed, index). */
int mrcnn_random_keys_u32(uint32_t *out, size_t n, unsigned long long seed, void *stream);
/* Same with the seed in device memory (state[0]); the call also advances the state, so a replayed HIP graph draws
 * fresh keys every time. */
int mrcnn_random_keys_dev_u32(uint32_t *out, size_t n, unsigned long long *state, void *stream);
/* g += wd*p; v = momentum*v - lr*g; p += v  over a flat parameter buffer (train.py:107-109). */
int mrcnn_sgd_momentum_wd_f32(float *p, const float *g, float *v, size_t n, float lr, float momentum,
                              float weight_decay, void *stream);

/* ------------------------------------------------------------------------------------------
 * (ABI v9) One whole ResNet bottleneck per call.  Replaces the Python-issued chain of Chainer's ResNet50Layers building block
 * (BottleneckA / BottleneckB: conv1x1 -> BN -> ReLU -> conv3x3 -> BN -> ReLU -> conv1x1 -> BN (+ shortcut, itself conv1x1 -> BN in
 * the first block of a stage) -> ReLU) as the reference runs it in training mode, extractor/feature_pyramid_network.py:22,48-66, and its
 * backward - the host side of the updater loop, train.py:117-132.  The call ENQUEUES exactly the launches the separate entry points
 * above would (mrcnn_conv2d_fwd_bnstats_f32 / mrcnn_conv2d_fwd_f32, mrcnn_bn_train_fwd_stats_f32 / mrcnn_bn_train_fwd_f32,
 * mrcnn_bn_train_bwd_f32, mrcnn_conv2d_bwd_filter_f32, mrcnn_conv2d_bwd_data_f32, mrcnn_subsample_bwd_f32, mrcnn_add_f32), in the same
 * order with the same operands: the results are the same bits; what changes is the host cost (one foreign call and two or three
 * allocations per block instead of ~25 calls and ~20 allocations).
 *
 * Descriptor (host memory, read during the call only).  Channel counts are the padded ones of the tensors (multiples of 32).
 *   x (N,H,W,cin) NHWC;  conv1: 1x1 stride `stride` cin->mid;  conv2: 3x3 pad 1 mid->mid;  conv3: 1x1 mid->cout;
 *   project != 0: shortcut conv4 1x1 stride `stride` cin->cout + bn4, else identity (cin == cout, stride 1).
 *   w[i] filters (Cout,KH,KW,Cin); gamma/beta/run_mean/run_var of bn1..bn4; gw/ggamma/gbeta: where the backward writes the gradients.
 *   fwd_split: -1 = the process setting (mrcnn_conv2d_set_split_operands); 0..3 = that arithmetic for the FORWARD convolutions of this
 *   call (the setting is restored before return).
 * Forward: mrcnn_bottleneck_fwd_plan fills `plan` (offsets of the saved tensors inside the caller's arena: pre-BN / post-BN
 * activations, kept Winograd input transforms, statistics partials, saved mean / invstd) and returns the arena and scratch sizes for
 * the CURRENT convolution settings; mrcnn_bottleneck_fwd_f32 follows that plan.  The arena is what the backward needs: keep it, with the
 * plan, until mrcnn_bottleneck_bwd_f32 has been enqueued.  y (N,Ho,Wo,cout) is the block's output.
 * Backward: gy = gradient of y.  gy_masked != 0: gy already carries the block's output ReLU mask (its producer applied it), the shortcut
 * gradient is gy itself; else g_r (N,Ho,Wo,cout, required) receives the masked gradient.  The block's input gradient:
 *   identity shortcut: accumulated into gx_acc when given (gx_acc = g_r + gx_acc first), else into g_r / gy IN PLACE (as the Python chain);
 *   projection: accumulated into gx_acc when given, else written to gx_new (N,H,W,cin, required then).
 * mask_gx != 0: that gradient is zeroed where x <= 0 (x is a ReLU output: the ReLU backward of the block before).
 * side_stream (nullable): the four filter gradients are enqueued there, each behind an event recorded on `stream` when its operands are
 * complete (the library keeps a small pool of timing-disabled events per device for this; no device memory); NULL = everything on
 * `stream`.  ws_side is the side stream's own scratch.  The caller joins the side stream when it needs the filter gradients.
 * ---------------------------------------------------------------------------------------- */
typedef struct mrcnn_bottleneck {
    int32_t N, H, W;
    int32_t cin, mid, cout;
    int32_t stride, project;
    int32_t fwd_split;
    float eps, decay;
    const float *w[4];
    const float *gamma[4];
    const float *beta[4];
    float *run_mean[4];
    float *run_var[4];
    float *gw[4];
    float *ggamma[4];
    float *gbeta[4];
} mrcnn_bottleneck_t;

enum { MRCNN_BN_H1 = 0, MRCNN_BN_A1, MRCNN_BN_H2, MRCNN_BN_A2, MRCNN_BN_H3, MRCNN_BN_H4, MRCNN_BN_R,
       MRCNN_BN_V = 7 /* +0..3 */, MRCNN_BN_PART = 11 /* +0..3 */, MRCNN_BN_MEAN = 15 /* +0..3 */, MRCNN_BN_INVSTD = 19 /* +0..3 */,
       MRCNN_BN_SLOTS = 23 };

typedef struct mrcnn_bottleneck_plan {
    uint64_t arena_bytes;               /* forward arena */
    uint64_t ws_bytes;                  /* forward scratch on `stream` */
    uint64_t off[MRCNN_BN_SLOTS];       /* byte offsets into the arena (256-B aligned); unused slots 0 */
    uint64_t v_bytes[4];                /* kept Winograd input transform of conv1..4 (0: none) */
    int32_t part_rows[4];               /* statistics partial rows of conv1..4 (0: BatchNorm runs its own statistics pass) */
} mrcnn_bottleneck_plan_t;

/* measurement: 0 = the composite calls materialise bn1's output as before (A/B of mrcnn_conv2d_fwd_inbn_f32); default 1 */
int mrcnn_debug_bottleneck_inbn(int on);
int mrcnn_bottleneck_fwd_plan(const mrcnn_bottleneck_t *b, mrcnn_bottleneck_plan_t *plan);
int mrcnn_bottleneck_fwd_f32(const mrcnn_bottleneck_t *b, const mrcnn_bottleneck_plan_t *plan, const float *x, float *y,
                             void *arena, size_t arena_bytes, void *ws, size_t ws_bytes, void *stream);
/* sizes3 (host, 3 x size_t): backward arena (the intermediate gradients), scratch on `stream`, scratch on `side_stream` */
int mrcnn_bottleneck_bwd_sizes(const mrcnn_bottleneck_t *b, size_t *sizes3);
int mrcnn_bottleneck_bwd_f32(const mrcnn_bottleneck_t *b, const mrcnn_bottleneck_plan_t *plan, const float *x, const float *y,
                             const void *fwd_arena, const float *gy, int gy_masked, float *g_r, float *gx_acc, float *gx_new,
                             int mask_gx, void *arena, size_t arena_bytes, void *ws_main, size_t ws_main_bytes, void *ws_side,
                             size_t ws_side_bytes, void *stream, void *side_stream);

/* ------------------------------------------------------------------------------------------
 * Losses (loss.hip).  Replace F.softmax_cross_entropy, _fast_rcnn_loc_loss/_smooth_l1_loss and
 * calc_mask_loss (chainer_maskrcnn/model/fpn_maskrcnn_train_chain.py:83-85,100-106; train.py:50-58;
 * train_keypoints.py:21-27).  loss_out is 2 floats on the device: [0] = loss, [1] = normaliser.
 * gx (nullable) receives d loss / d logits.
 * ---------------------------------------------------------------------------------------- */
size_t mrcnn_loss_workspace_bytes(void);
/* softmax_ce: element (r, j) of the logical (M, K) logits lives at (r / A) * gs + (r % A) * rs + j * es floats (gx: ggs, grs, ges).
 * Rows that are CHANNELS of an NHWC (G, K, C) tensor (rs = 1, es = C, gs = K * C, A <= C, C a multiple of 4 that divides 1024, K >= 64:
 * the keypoint loss of train_keypoints.py:21-27, 17 heat maps in 32 padded channels) take a coalesced kernel - one workgroup per group,
 * float4 loads over the (position, channel) plane, loss and gradient from one launch - which writes EVERY element of gx (zeros in the
 * channels >= A and in the ignored rows); on the other paths gx receives the K (Kfill) columns of its M rows only. */
/* 1 when mrcnn_softmax_ce_f32 called with these maps writes EVERY element of gx itself (its channel-interleaved path: the keypoint loss),
 * so that the caller need not zero-fill gx; 0 = the strided path writes only the K columns of the M rows.  The library's own dispatch
 * predicate, exported (one source of truth). */
int mrcnn_softmax_ce_fills_gx(int M, int K, int A, long long gs, long long rs, long long es, long long ggs, long long grs, long long ges,
                              int Kfill);
int mrcnn_softmax_ce_f32(const float *x, int A, long long gs, long long rs, long long es, const int32_t *t,
                         int M, int K, int ignore_label, float *loss_out, float *gx, long long ggs,
                         long long grs, long long ges, int Kfill, void *ws, size_t ws_bytes, void *stream);
int mrcnn_smooth_l1_f32(const float *x, int ldx, const float *t, const int32_t *label, int M, float sigma,
                        float *loss_out, float *gx, int ldg, int gfill, void *ws, size_t ws_bytes, void *stream);
int mrcnn_mask_bce_f32(const float *x, const int32_t *gt, const int32_t *label, int Rm, int HW, int Cm,
                       float *loss_out, float *gx, void *ws, size_t ws_bytes, void *stream);
/* Building blocks of a USER-SUPPLIED mask_loss_fun (the reference passes a plain Python function: train.py:50-58,98;
 * train_keypoints.py:21-27): the host layer wraps them as autograd functions (chainer_maskrcnn/functions/loss.py) so
 * the bodies of those functions run on device tensors.
 *   sigmoid_ce      chainer.functions.sigmoid_cross_entropy(x, t): normalize=True, ignore label -1, mean; n elements
 *   select_channel  roi_cls_mask[xp.arange(R), idx] on an NCHW tensor (R,C,HW): y (R,HW); negative idx wraps like NumPy;
 *                   backward != 0: src = gy (R,HW), dst = gx (R,C,HW) fully written (zeros elsewhere)
 *   nhwc_nchw       (R,HW,Cp) NHWC with Cp >= C padded channels -> (R,C,HW) NCHW; inverse != 0: back, padding zeroed
 *   scale_by_dev    x *= scale[0] with the scalar in device memory (upstream gradient of a loss output) */
int mrcnn_sigmoid_ce_f32(const float *x, const int32_t *t, long long n, float *loss_out, float *gx, void *ws,
                         size_t ws_bytes, void *stream);
int mrcnn_select_channel_f32(const float *src, const int32_t *idx, int R, int C, int HW, float *dst, int backward,
                             void *stream);
int mrcnn_nhwc_nchw_f32(const float *src, float *dst, int R, int HW, int Cp, int C, int inverse, void *stream);
int mrcnn_scale_by_dev_f32(float *x, size_t n, const float *scale, void *stream);
/* out[0] = sum of n (loss, normaliser) pairs' losses: the un-weighted total of fpn_maskrcnn_train_chain.py:106 */
int mrcnn_loss_total_f32(const float *losses, int n, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * RPN proposal path (rpn.hip).  Replaces the transposes/concats and the ChainerCV ProposalCreator +
 * non_maximum_suppression + map_rois_to_fpn_levels calls of
 * chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:16-31,133-164.
 *   head (N,HW,Cp): channels [0,4A) loc, [4A,6A) score;  locs (N,Atot,4);  scores (N,Atot,2)
 *   rois (N*n_post,4) yx, zero padded;  roi_indices (N*n_post) = image or -1;  levels f32;  n_rois (N)
 *   dbg_* (nullable): anchor index of each pre-NMS box in sort order (N*n_pre), NMS keep list
 *   (N*n_post, indices into the sort order), pre-NMS count (N).
 *   per_image (nullable, device, (N,3) f32 = h, w, min_size * scale): a padded batch of images of different sizes - every
 *   image's boxes are clipped to its own size and filtered with its own scaled min_size, as the reference (batch 1 per
 *   process, rpn/...:156-164: img_size and scale of THAT image) does; null = img_h / img_w / min_size for all.
 * ---------------------------------------------------------------------------------------- */
int mrcnn_rpn_pack_f32(const float *head, int N, int HW, int Cp, int A, float *locs, float *scores, int a_off,
                       int Atot, void *stream);
int mrcnn_rpn_unpack_grad_f32(const float *glocs, const float *gscores, int N, int HW, int Cp, int A,
                              float *ghead, int a_off, int Atot, void *stream);
/* (ABI v10) The same for ALL pyramid levels in one launch each way (heads / gheads: L host-side device pointers, HWs: positions per level;
 * level l's anchors follow level l-1's in the concatenated axis, sum(HWs) * A == Atot): the concat of rpn/...:143-152 and its backward. */
int mrcnn_rpn_pack_levels_f32(const float *const *heads, const int *HWs, int L, int N, int Cp, int A, float *locs, float *scores,
                              int Atot, void *stream);
int mrcnn_rpn_unpack_grad_levels_f32(const float *glocs, const float *gscores, float *const *gheads, const int *HWs, int L, int N,
                                     int Cp, int A, int Atot, void *stream);
size_t mrcnn_rpn_proposals_workspace_bytes(int N, int A, int n_pre, int n_post);
int mrcnn_rpn_proposals_f32(const float *locs, const float *scores, const float *anchors, int N, int A,
                            float img_h, float img_w, float min_size, const float *per_image, int n_pre, int n_post, float nms_thresh,
                            float *rois, int32_t *roi_indices, float *levels, int32_t *n_rois,
                            int32_t *dbg_sorted_anchor, int32_t *dbg_keep, int32_t *dbg_n_pre, void *ws,
                            size_t ws_bytes, void *stream);
size_t mrcnn_nms_workspace_bytes(int n);
int mrcnn_nms_f32(const float *boxes, int n, float thresh, int max_keep, int32_t *keep, int32_t *n_keep,
                  void *ws, size_t ws_bytes, void *stream);
int mrcnn_map_rois_to_fpn_levels_f32(const float *rois, int R, int k_min, int k_max, float *levels, void *stream);
/* out (M,2) = softmax over the 2 class scores of in (M,2): ChainerCV's single-level RegionProposalNetwork (the 'c4'
 * backbone, chainer_maskrcnn/model/maskrcnn.py:60-69) ranks its proposals by the foreground PROBABILITY. */
int mrcnn_softmax2_f32(const float *in, float *out, size_t M, void *stream);

/* ------------------------------------------------------------------------------------------
 * Target creators (targets.hip).  Replace chainer_maskrcnn/utils/proposal_target_creator.py:26-137
 * (host NumPy + cv2) and ChainerCV's AnchorTargetCreator (model/fpn_maskrcnn_train_chain.py:81-82).
 * Per image i the sampler owns rows [i*n_sample, (i+1)*n_sample): positives first, then negatives,
 * then padding (label -1).  keys: uint32 random numbers, (N, roi_cap + gt_cap) / (N, A).
 * proposal_target, reference-order mode: pos_order / neg_order (both or neither; (N, n_sample) int32) replace the keys -
 * row j of the positives is the candidate of RANK pos_order[j] among the foreground candidates in ascending index order,
 * which is np.random.choice(pos_index, size, replace=False) = pos_index[permutation(len)[:size]]
 * (utils/proposal_target_creator.py:63-78) when the host draws the permutation; n_cand (nullable, (N,2)) returns the
 * candidate-set sizes the host needs for that draw.
 * ---------------------------------------------------------------------------------------- */
int mrcnn_proposal_target_f32(const float *rois, const float *roi_levels, const int32_t *n_rois, int roi_cap,
                              const float *gt_boxes, const int32_t *gt_labels, const int32_t *n_gt, int gt_cap,
                              const uint32_t *keys, int N, int n_sample, int n_pos_max, float pos_iou_thresh,
                              float neg_iou_thresh_hi, float neg_iou_thresh_lo, const float *loc_mean4,
                              const float *loc_std4, float *sample_roi, float *rois_xy5, int32_t *sample_levels,
                              float *gt_roi_loc, int32_t *gt_roi_label, int32_t *gt_assign, int32_t *sample_src,
                              int32_t *n_pos, int32_t *n_sampled, const int32_t *pos_order, const int32_t *neg_order,
                              int32_t *n_cand, void *stream);
int mrcnn_mask_target_u8(const unsigned char *masks, int N, int gt_cap, int H, int W, const float *sample_roi,
                         const int32_t *gt_assign, const int32_t *n_pos, int n_sample, int pos_cap, int mask_size,
                         int32_t *gt_roi_mask, void *stream);
/* inplace_quirk != 0 reproduces the reference's in-place mutation of the gt keypoints (a gt assigned to several
 * positives is transformed again from its already-transformed coordinates, in sample order). */
int mrcnn_keypoint_target_f32(const float *keypoints, int N, int gt_cap, int K, const float *sample_roi,
                              const int32_t *gt_assign, const int32_t *n_pos, int n_sample, int pos_cap,
                              int mask_size, int inplace_quirk, int32_t *gt_roi_kp, void *stream);
/* n_gt[i] = number of labels[i, :] >= 0: the per-image gt counts of a padded batch (Chainer concat_examples pads with
 * -1; dataset/loader.py does the same), valid rows first.  Used when the caller passes no counts (train.py:117-125). */
int mrcnn_count_valid_labels_i32(const int32_t *labels, int N, int G, int32_t *n_gt, void *stream);
/* per_image_hw (nullable, device, (N,2) f32): each image's own (h, w) for the "anchor inside the image" test of a padded
 * batch; null = img_h / img_w for all images. */
size_t mrcnn_anchor_target_workspace_bytes(int N, int A);
int mrcnn_anchor_target_f32(const float *anchors, int A, const float *gt_boxes, const int32_t *n_gt, int gt_cap,
                            int N, float img_h, float img_w, const float *per_image_hw, const uint32_t *keys, int n_sample,
                            float pos_iou_thresh, float neg_iou_thresh, float pos_ratio, int do_sample,
                            float *gt_rpn_loc, int32_t *gt_rpn_label, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Inference post-processing (predict.hip).  Replace the host code of
 * chainer_maskrcnn/model/maskrcnn.py:178-210 (decode + softmax), :278-312 (_suppress: per-class score filter +
 * ChainerCV NMS) and :231-246 (sigmoid, channel pick, cv2.resize to the box, threshold, paste).
 *   rois (R,4) yx in network-input pixels; box_out (R,ld): columns [0,n_class) scores, [loc0,loc0+4) class-agnostic loc
 *   cls_bbox (R,4) yx in original-image pixels; prob (R,n_class)
 *   keep_idx (n_class,R) int32: RoI indices kept for class l in selection (descending score) order; keep_cnt (n_class)
 *   mask_logits (D,S,S,Cm) NHWC; label (D) 0-based class = mask channel; bbox (D,4); out (D,H,W) uint8 {0,1}
 * ---------------------------------------------------------------------------------------- */
int mrcnn_detect_decode_f32(const float *rois, int R, const float *box_out, int ld, int n_class, int loc0,
                            float scale, const float *loc_mean4, const float *loc_std4, float size_h, float size_w,
                            float *cls_bbox, float *prob, void *stream);
int mrcnn_class_nms_f32(const float *cls_bbox, const float *prob, int R, int n_class, int l_begin, int l_end,
                        float score_thresh, float nms_thresh, int32_t *keep_idx, int32_t *keep_cnt, void *stream);
int mrcnn_mask_paste_f32(const float *mask_logits, int D, int S, int Cm, const int32_t *label, const float *bbox,
                         int H, int W, unsigned char *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MRCNN_HIP_H */

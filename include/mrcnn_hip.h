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
 *     null stream); no call synchronises the device.
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

#define MRCNN_ABI_VERSION 1

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

/* Backward (adjoint scatter).  gx is fully overwritten (callee zero-fills cells no RoI touches).
 * Fast path (NHWC, C%4==0, PH,PW<=16, sampling_ratio>0): owner-computes tiles, no atomics,
 * bit-reproducible.  Other shapes: memset + atomic scatter. */
int mrcnn_roi_align_bwd_f32(const float *gy, int layout, int N, int C, int H, int W,
                            const float *rois, int R, int PH, int PW, float spatial_scale,
                            int sampling_ratio, float *gx, void *stream);

/* Multi-level (FPN) batched variants: ONE launch for all RoIs over all pyramid levels.
 * Replace the per-RoI Python loops of chainer_maskrcnn/model/head/fpn_roi_mask_head.py:59-61,
 * 75-77,93-94 (and fpn_roi_keypoint_head.py:63-68,85-86,102-103).
 *   xs / gxs        host arrays of L device base pointers (level l: (N,C,Hs[l],Ws[l]) NHWC)
 *   Hs, Ws, scales  host arrays of length L  (scales[l] = spatial_scales[l])
 *   levels          (R,) int32 device, level of each RoI (already clipped to [0,L))
 *   rois            (R,5) f32 device (batch_idx,x1,y1,x2,y2)
 *   y / gy          (R,PH,PW,C) f32
 * Only MRCNN_LAYOUT_NHWC with C%4==0 is supported. */
int mrcnn_roi_align_fpn_fwd_f32(const float *const *xs, const int *Hs, const int *Ws,
                                const float *scales, int L, int N, int C, const float *rois,
                                const int32_t *levels, int R, int PH, int PW,
                                int sampling_ratio, float *y, void *stream);
int mrcnn_roi_align_fpn_bwd_f32(const float *gy, float *const *gxs, const int *Hs, const int *Ws,
                                const float *scales, int L, int N, int C, const float *rois,
                                const int32_t *levels, int R, int PH, int PW,
                                int sampling_ratio, void *stream);

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
 *   Ho = (H + 2*pad - KH)/stride + 1.   Cin and Cout must be multiples of 32 (the host layer
 *   zero-pads channel counts such as 3, 4, 18, 80, 81).  bias (Cout) may be NULL; relu != 0
 *   fuses max(.,0) into the forward epilogue.  Linear layers are 1x1 convolutions with
 *   H = W = 1; the 2x2/2 deconvolution is a 1x1 convolution to 4*Cout channels + a host view.
 * bwd_data supports stride 1 only.  bwd_filter accumulates over pixels with a deterministic
 * split-K (slabs in the caller's workspace, fixed summation order); gbias may be NULL.
 * ---------------------------------------------------------------------------------------- */
int mrcnn_conv2d_fwd_f32(const float *x, const float *w, const float *bias, float *y, int N, int H,
                         int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                         void *stream);
int mrcnn_conv2d_bwd_data_f32(const float *gy, const float *w, float *gx, int N, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad, void *stream);
size_t mrcnn_conv2d_bwd_filter_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                               int stride, int pad);
int mrcnn_conv2d_bwd_filter_f32(const float *x, const float *gy, float *gw, float *gbias, int N, int H,
                                int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void *ws,
                                size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MRCNN_HIP_H */

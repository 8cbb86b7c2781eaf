// Inference post-processing on the device for gfx950: box decode + softmax, per-class
// score filter + NMS, mask paste.  Replaces the host NumPy / OpenCV code of
//   chainer_maskrcnn/model/maskrcnn.py:178-210  (un-scale, loc2bbox, clip, softmax, D2H)
//   chainer_maskrcnn/model/maskrcnn.py:278-312  (_suppress: per class, prob > score_thresh, ChainerCV NMS 0.3)
//   chainer_maskrcnn/model/maskrcnn.py:231-246  (sigmoid, channel pick, cv2.resize to the box, threshold, paste)
// Float arithmetic follows oracle/predict.py operation for operation (FP contraction off).  Latency-bound kernels
// (<= 300 RoIs, <= 80 classes).
#include "common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

// One thread per RoI: class-agnostic loc -> box in original-image pixels; softmax over the n_class scores.
__global__ __launch_bounds__(256) void k_detect_decode(const float *__restrict__ rois, int R, const float *__restrict__ box_out,
                                                       int ld, int n_class, int loc0, float scale, float4 mean, float4 stdv,
                                                       float size_h, float size_w, float *__restrict__ cls_bbox,
                                                       float *__restrict__ prob) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const float4 rr = *reinterpret_cast<const float4 *>(rois + (size_t)i * 4);
    const float4 r = make_float4(rr.x / scale, rr.y / scale, rr.z / scale, rr.w / scale);         // roi = rois / scale (:178)
    const float *o = box_out + (size_t)i * ld;
    const float dy = o[loc0] * stdv.x + mean.x, dx = o[loc0 + 1] * stdv.y + mean.y;                // (:191-195)
    const float dh = o[loc0 + 2] * stdv.z + mean.z, dw = o[loc0 + 3] * stdv.w + mean.w;
    const float h = r.z - r.x, w = r.w - r.y;                                                      // loc2bbox (:196)
    const float cy = r.x + 0.5f * h, cx = r.y + 0.5f * w;
    const float ncy = dy * h + cy, ncx = dx * w + cx;
    const float nh = expf(dh) * h, nw = expf(dw) * w;
    float y1 = ncy - 0.5f * nh, x1 = ncx - 0.5f * nw, y2 = ncy + 0.5f * nh, x2 = ncx + 0.5f * nw;
    y1 = fmaxf(fminf(y1, size_h), 0.f); y2 = fmaxf(fminf(y2, size_h), 0.f);                       // clip (:202-203)
    x1 = fmaxf(fminf(x1, size_w), 0.f); x2 = fmaxf(fminf(x2, size_w), 0.f);
    *reinterpret_cast<float4 *>(cls_bbox + (size_t)i * 4) = make_float4(y1, x1, y2, x2);
    float m = -INFINITY;
    for (int c = 0; c < n_class; ++c) m = fmaxf(m, o[c]);
    float s = 0.f;
    for (int c = 0; c < n_class; ++c) s += expf(o[c] - m);
    for (int c = 0; c < n_class; ++c) prob[(size_t)i * n_class + c] = expf(o[c] - m) / s;         // F.softmax (:205)
}

__device__ __forceinline__ unsigned orderable(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// One workgroup per class l in [l_begin, l_end): candidates prob[:, l] > thresh, sorted (score desc, index desc - the
// oracle's pin of argsort()[::-1]), greedy NMS.  Output: keep_idx[l][0..cnt) = RoI indices in selection order.
constexpr int CN_CAP = 512;
__global__ __launch_bounds__(256) void k_class_nms(const float *__restrict__ cls_bbox, const float *__restrict__ prob, int R,
                                                   int n_class, int l_begin, float score_thresh, float nms_thresh,
                                                   int32_t *__restrict__ keep_idx, int32_t *__restrict__ keep_cnt) {
    __shared__ u64 skey[CN_CAP];
    __shared__ float4 sbox[CN_CAP];
    __shared__ u64 smask[CN_CAP][CN_CAP / 64];
    __shared__ int s_n;
    const int l = l_begin + blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < CN_CAP; i += 256) {
        u64 k = 0ull;
        if (i < R) {
            const float p = prob[(size_t)i * n_class + l];
            if (p > score_thresh) k = (1ull << 63) | ((u64)orderable(p) << 31) | (u64)i;
        }
        skey[i] = k;
    }
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= CN_CAP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < CN_CAP; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u64 a = skey[i], b = skey[ixj];
                    const bool up = (i & k) == 0;
                    if ((a < b) == up) { skey[i] = b; skey[ixj] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) {
        int n = 0;
        while (n < CN_CAP && skey[n] != 0ull) ++n;
        s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    for (int i = tid; i < n; i += 256) sbox[i] = *reinterpret_cast<const float4 *>(cls_bbox + (size_t)(skey[i] & 0x7FFFFFFFull) * 4);
    __syncthreads();
    const int nw = (n + 63) / 64;
    for (int t = tid; t < n * nw; t += 256) {
        const int i = t / nw, wd = t % nw;
        const float4 b = sbox[i];
        const float area_i = (b.z - b.x) * (b.w - b.y);
        u64 bits = 0ull;
        for (int jj = 0; jj < 64; ++jj) {
            const int j = wd * 64 + jj;
            if (j >= n || j <= i) continue;
            const float4 c = sbox[j];
            const float top = fmaxf(b.x, c.x), left = fmaxf(b.y, c.y), bottom = fminf(b.z, c.z), right = fminf(b.w, c.w);
            const float hgt = fmaxf(bottom - top, 0.f), wid = fmaxf(right - left, 0.f);
            const float ai = hgt * wid;
            const float iou = ai / ((area_i + (c.z - c.x) * (c.w - c.y)) - ai);
            if (iou >= nms_thresh) bits |= 1ull << jj;
        }
        smask[i][wd] = bits;
    }
    __syncthreads();
    if (tid == 0) {           // n <= 512: the sequential sweep is a few microseconds
        u64 rem[CN_CAP / 64];
        for (int w = 0; w < CN_CAP / 64; ++w) rem[w] = 0ull;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            if ((rem[i >> 6] >> (i & 63)) & 1ull) continue;
            keep_idx[(size_t)l * R + cnt++] = (int)(skey[i] & 0x7FFFFFFFull);
            for (int w = 0; w < nw; ++w) rem[w] |= smask[i][w];
        }
        keep_cnt[l] = cnt;
    }
}

// Mask paste (maskrcnn.py:231-246): m = sigmoid(logit[d, :, :, label]); cv2.resize(m, (w, h)) float bilinear (half-pixel
// centres, edge clamp); *255 -> uint8 (truncate) -> > 127; pasted at (int(y1), int(x1)), clipped to the image.
__global__ __launch_bounds__(256) void k_mask_paste(const float *__restrict__ logits, int S, int Cm, const int32_t *__restrict__ label,
                                                    const float *__restrict__ bbox, int H, int W, unsigned char *__restrict__ out) {
    const int d = blockIdx.y;
    const float4 b = *reinterpret_cast<const float4 *>(bbox + (size_t)d * 4);
    const int mw = (int)(b.w - b.y), mh = (int)(b.z - b.x);
    const int s0 = (int)b.x, t0 = (int)b.y;
    const int ch = label[d];
    unsigned char *o = out + (size_t)d * H * W;
    const float *lg = logits + (size_t)d * S * S * Cm + ch;
    const double sy = mh > 0 ? 1.0 / ((double)mh / (double)S) : 0.0, sx = mw > 0 ? 1.0 / ((double)mw / (double)S) : 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
        const int y = p / W, x = p % W;
        const int dy = y - s0, dx = x - t0;
        unsigned char v = 0;
        if (dy >= 0 && dy < mh && dx >= 0 && dx < mw) {
            float fy = (float)(((double)dy + 0.5) * sy - 0.5), fx = (float)(((double)dx + 0.5) * sx - 0.5);
            int iy = (int)floorf(fy), ix = (int)floorf(fx);
            fy -= (float)iy; fx -= (float)ix;
            if (iy < 0) { fy = 0.f; iy = 0; }
            if (iy >= S - 1) { fy = 0.f; iy = S - 1; }
            if (ix < 0) { fx = 0.f; ix = 0; }
            if (ix >= S - 1) { fx = 0.f; ix = S - 1; }
            const int iy1 = min(iy + 1, S - 1), ix1 = min(ix + 1, S - 1);
            auto sg = [&](int yy, int xx) { return 1.0f / (1.0f + expf(-lg[((size_t)yy * S + xx) * Cm])); };
            const float r0 = sg(iy, ix) * (1.0f - fx) + sg(iy, ix1) * fx;
            const float r1 = sg(iy1, ix) * (1.0f - fx) + sg(iy1, ix1) * fx;
            const float m = r0 * (1.0f - fy) + r1 * fy;
            const int q = (int)(m * 255.0f);
            v = (unsigned char)((q & 0xFF) > 127 ? 1 : 0);
        }
        o[p] = v;
    }
}

}  // namespace

extern "C" int mrcnn_detect_decode_f32(const float *rois, int R, const float *box_out, int ld, int n_class, int loc0,
                                       float scale, const float *loc_mean4, const float *loc_std4, float size_h,
                                       float size_w, float *cls_bbox, float *prob, void *stream) {
    if (R == 0) return 0;
    if (!rois || !box_out || !loc_mean4 || !loc_std4 || !cls_bbox || !prob || R < 0 || n_class <= 0 || ld < loc0 + 4 || scale <= 0.f)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "detect_decode: bad arguments");
    hipLaunchKernelGGL(k_detect_decode, dim3(mrcnn::cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, rois, R, box_out, ld, n_class,
                       loc0, scale, make_float4(loc_mean4[0], loc_mean4[1], loc_mean4[2], loc_mean4[3]),
                       make_float4(loc_std4[0], loc_std4[1], loc_std4[2], loc_std4[3]), size_h, size_w, cls_bbox, prob);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_class_nms_f32(const float *cls_bbox, const float *prob, int R, int n_class, int l_begin, int l_end,
                                   float score_thresh, float nms_thresh, int32_t *keep_idx, int32_t *keep_cnt, void *stream) {
    if (!cls_bbox || !prob || !keep_idx || !keep_cnt || R <= 0 || n_class <= 0 || l_begin < 0 || l_end > n_class || l_begin > l_end)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "class_nms: bad arguments");
    if (R > CN_CAP) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "class_nms: %d RoIs > %d", R, CN_CAP);
    MRCNN_HIP_TRY(hipMemsetAsync(keep_cnt, 0, sizeof(int32_t) * n_class, (hipStream_t)stream));
    if (l_end > l_begin) {
        hipLaunchKernelGGL(k_class_nms, dim3(l_end - l_begin), dim3(256), 0, (hipStream_t)stream, cls_bbox, prob, R, n_class, l_begin,
                           score_thresh, nms_thresh, keep_idx, keep_cnt);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mrcnn_mask_paste_f32(const float *mask_logits, int D, int S, int Cm, const int32_t *label, const float *bbox,
                                    int H, int W, unsigned char *out, void *stream) {
    if (D == 0) return 0;
    if (!mask_logits || !label || !bbox || !out || D < 0 || S <= 0 || Cm <= 0 || H <= 0 || W <= 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "mask_paste: bad arguments");
    hipLaunchKernelGGL(k_mask_paste, dim3(std::min(mrcnn::cdiv((long long)H * W, 256), 1024), D), dim3(256), 0, (hipStream_t)stream,
                       mask_logits, S, Cm, label, bbox, H, W, out);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

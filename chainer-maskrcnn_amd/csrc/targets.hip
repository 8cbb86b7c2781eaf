// Training-target creation on the device for gfx950 (no host round trip, no OpenCV):
//   mrcnn_proposal_target_f32  <- ProposalTargetCreator.__call__ (chainer_maskrcnn/utils/proposal_target_creator.py:26-137)
//   mrcnn_mask_target_u8       <- the per-positive crop + cv2.resize loop (:96-103)
//   mrcnn_keypoint_target_f32  <- the keypoint branch (:105-127)
//   mrcnn_anchor_target_f32    <- ChainerCV AnchorTargetCreator (model/fpn_maskrcnn_train_chain.py:21,81-82)
// Float arithmetic follows oracle/boxes.py / oracle/targets.py operation for operation (FP
// contraction off); integer outputs (labels, assignments, mask targets) are bit-exact.
// Random subsets are drawn as "the k candidates with the smallest (key, index)" for caller-provided
// uint32 keys (uniform keys == np.random.choice(replace=False) in distribution; oracle:
// proposal_targets_from_keys / anchor_targets_from_keys); selected rows keep ascending index order.
// All kernels are latency-bound (a few thousand boxes) except anchor_target (A = 261,888 anchors
// x G boxes, HBM-trivial).
#include "common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;
constexpr float kEps = 1.1920929e-07f;   // np.finfo(np.float32).eps

__device__ __forceinline__ float box_iou(const float4 a, const float4 b) {
    // ChainerCV bbox_iou: area_i = prod(br - tl) * all(tl < br); iou = area_i / (area_a + area_b - area_i)
    const float tly = fmaxf(a.x, b.x), tlx = fmaxf(a.y, b.y);
    const float bry = fminf(a.z, b.z), brx = fminf(a.w, b.w);
    float area_i = (bry - tly) * (brx - tlx);
    area_i = area_i * ((tly < bry && tlx < brx) ? 1.0f : 0.0f);
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    return area_i / ((area_a + area_b) - area_i);
}

__device__ __forceinline__ float4 bbox2loc(const float4 s, const float4 d) {
    float h = s.z - s.x, w = s.w - s.y;
    const float cy = s.x + 0.5f * h, cx = s.y + 0.5f * w;
    const float bh = d.z - d.x, bw = d.w - d.y;
    const float bcy = d.x + 0.5f * bh, bcx = d.y + 0.5f * bw;
    h = fmaxf(h, kEps);
    w = fmaxf(w, kEps);
    return make_float4((bcy - cy) / h, (bcx - cx) / w, logf(bh / h), logf(bw / w));
}

__device__ __forceinline__ float fpn_level(float4 b) {
    const float area = (b.z - b.x) * (b.w - b.y);
    const float v = sqrtf(area) / 224.0f + 1e-6f;
    const float t = floorf(4.0f + (float)log2((double)v));
    return fminf(fmaxf(t, 0.f), 4.f);
}

// In-LDS bitonic sort (ascending) of n = power of two keys by the whole block.
__device__ void bitonic_sort(u64 *s, int n) {
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u64 a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

constexpr int PT_THREADS = 1024;
constexpr int PT_CAP = 4096;       // max candidates (proposals + gt boxes) per image

// One workgroup per image.
__global__ __launch_bounds__(PT_THREADS) void k_proposal_target(
    const float *__restrict__ rois, const float *__restrict__ roi_levels, const int32_t *__restrict__ n_rois, int roi_cap,
    const float *__restrict__ gt_boxes, const int32_t *__restrict__ gt_labels, const int32_t *__restrict__ n_gt, int gt_cap,
    const uint32_t *__restrict__ keys, int n_sample, int n_pos_max, float pos_thresh, float neg_hi, float neg_lo,
    float4 mean, float4 stdv, float *__restrict__ sample_roi, float *__restrict__ rois_xy5,
    int32_t *__restrict__ sample_levels, float *__restrict__ gt_roi_loc, int32_t *__restrict__ gt_roi_label,
    int32_t *__restrict__ gt_assign, int32_t *__restrict__ sample_src, int32_t *__restrict__ n_pos_out,
    int32_t *__restrict__ n_sample_out, const int32_t *__restrict__ pos_order, const int32_t *__restrict__ neg_order,
    int32_t *__restrict__ n_cand) {
    __shared__ u64 skey[PT_CAP];
    __shared__ float smax[PT_CAP];
    __shared__ short sarg[PT_CAP];
    __shared__ int ssel[512];
    __shared__ int scount[2];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int nr = min(n_rois[img], roi_cap), G = min(n_gt[img], gt_cap);
    const int nc = min(nr + G, PT_CAP);
    const float *rb = rois + (size_t)img * roi_cap * 4;
    const float *gb = gt_boxes + (size_t)img * gt_cap * 4;
    const uint32_t *kk = keys + (size_t)img * (roi_cap + gt_cap);
    auto cand = [&](int c) -> float4 {
        return c < nr ? *reinterpret_cast<const float4 *>(rb + (size_t)c * 4)
                      : *reinterpret_cast<const float4 *>(gb + (size_t)(c - nr) * 4);
    };
    if (tid < 2) scount[tid] = 0;
    int np2 = 64;                      // the sorts below run over the next power of two >= the candidate count, not PT_CAP
    while (np2 < nc) np2 <<= 1;
    __syncthreads();
    // IoU max / first argmax
    for (int c = tid; c < nc; c += PT_THREADS) {
        const float4 b = cand(c);
        float best = -INFINITY;
        int arg = 0;
        for (int g = 0; g < G; ++g) {
            const float v = box_iou(b, *reinterpret_cast<const float4 *>(gb + (size_t)g * 4));
            if (g == 0 || v > best) { best = v; arg = g; }
        }
        smax[c] = G > 0 ? best : 0.f;
        sarg[c] = (short)arg;
    }
    __syncthreads();
    int n_pos = 0, n_tot = 0;
    for (int phase = 0; phase < 2; ++phase) {
        // candidate set of this phase -> composite keys (key, index), others = max
        int local = 0;
        for (int c = tid; c < np2; c += PT_THREADS) {
            bool in = false;
            if (c < nc) {            // an image without objects (G = 0, max IoU 0): every proposal is a background candidate
                const float m = smax[c];
                in = phase == 0 ? (G > 0 && m >= pos_thresh) : (m < neg_hi && m >= neg_lo);
            }
            // reference-order mode (pos_order / neg_order given): keys are ignored, the sort leaves the candidates in
            // ascending index order and row j takes the candidate of rank order[j] - np.random.choice's draw order
            // (proposal_target_creator.py:63-78) supplied by the host
            const int32_t *order = phase == 0 ? pos_order : neg_order;
            const uint32_t key = (c < nc && !order) ? kk[c < nr ? c : roi_cap + (c - nr)] : 0u;
            skey[c] = in ? (((u64)key << 32) | (u64)c) : ~0ull;
            local += in ? 1 : 0;
        }
        atomicAdd(&scount[phase], local);
        __syncthreads();
        const int avail = scount[phase];
        const int want = phase == 0 ? min(n_pos_max, avail) : min(n_sample - n_pos, avail);
        bitonic_sort(skey, np2);
        if (n_cand && tid == 0) n_cand[img * 2 + phase] = avail;
        const int32_t *order = phase == 0 ? pos_order : neg_order;
        if (order) {
            for (int j = tid; j < want; j += PT_THREADS) {
                const int rk = min(max(order[(size_t)img * n_sample + j], 0), max(avail - 1, 0));
                ssel[j] = (int)(skey[rk] & 0xFFFFFFFFull);
            }
        } else {
            // the `want` smallest, re-ordered by ascending candidate index (rank sort, want <= 512)
            for (int j = tid; j < want; j += PT_THREADS) {
                const int cj = (int)(skey[j] & 0xFFFFFFFFull);
                int rank = 0;
                for (int q = 0; q < want; ++q) rank += ((int)(skey[q] & 0xFFFFFFFFull) < cj) ? 1 : 0;
                ssel[rank] = cj;
            }
        }
        __syncthreads();
        // emit rows [n_tot, n_tot + want)
        for (int j = tid; j < want; j += PT_THREADS) {
            const int c = ssel[j];
            const int row = img * n_sample + n_tot + j;
            const float4 b = cand(c);
            const int g = sarg[c];
            const float4 gbx = *reinterpret_cast<const float4 *>(gb + (size_t)g * 4);
            *reinterpret_cast<float4 *>(sample_roi + (size_t)row * 4) = b;
            float *r5 = rois_xy5 + (size_t)row * 5;
            r5[0] = (float)img; r5[1] = b.y; r5[2] = b.x; r5[3] = b.w; r5[4] = b.z;
            sample_levels[row] = (int)(c < nr ? roi_levels[(size_t)img * roi_cap + c] : fpn_level(b));
            float4 loc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (G > 0) {             // (no gt: row 0 is padding, its zero size would give -inf targets)
                loc = bbox2loc(b, gbx);
                loc.x = (loc.x - mean.x) / stdv.x; loc.y = (loc.y - mean.y) / stdv.y;
                loc.z = (loc.z - mean.z) / stdv.z; loc.w = (loc.w - mean.w) / stdv.w;
            }
            *reinterpret_cast<float4 *>(gt_roi_loc + (size_t)row * 4) = loc;
            gt_roi_label[row] = phase == 0 ? gt_labels[(size_t)img * gt_cap + g] + 1 : 0;
            gt_assign[row] = G > 0 ? g : -1;
            sample_src[row] = c;
        }
        if (phase == 0) n_pos = want;
        n_tot += want;
        __syncthreads();
    }
    // padding rows
    for (int j = n_tot + tid; j < n_sample; j += PT_THREADS) {
        const int row = img * n_sample + j;
        *reinterpret_cast<float4 *>(sample_roi + (size_t)row * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        float *r5 = rois_xy5 + (size_t)row * 5;
        r5[0] = (float)img; r5[1] = 0.f; r5[2] = 0.f; r5[3] = 0.f; r5[4] = 0.f;
        sample_levels[row] = 0;
        *reinterpret_cast<float4 *>(gt_roi_loc + (size_t)row * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        gt_roi_label[row] = -1;
        gt_assign[row] = -1;
        sample_src[row] = -1;
    }
    if (tid == 0) { n_pos_out[img] = n_pos; n_sample_out[img] = n_tot; }
}

// ---- mask targets: crop to the int()-truncated RoI, OpenCV INTER_LINEAR uint8 fixed-point resize ----
struct Coef { int s; int a0, a1; };
__device__ __forceinline__ Coef cv_coef(int d, int ssize, int dsize) {
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    Coef c;
    c.s = s;
    c.a0 = (int)rintf((1.0f - f) * 2048.0f);
    c.a1 = (int)rintf(f * 2048.0f);
    return c;
}

// slots: image i, positive j (< n_pos[i]) -> row i*n_sample + j of the sampler outputs; output slot i*pos_cap + j.
__global__ __launch_bounds__(256) void k_mask_target(const unsigned char *__restrict__ masks, int gt_cap, int H, int W,
                                                     const float *__restrict__ sample_roi, const int32_t *__restrict__ gt_assign,
                                                     const int32_t *__restrict__ n_pos, int n_sample, int pos_cap, int msz,
                                                     int32_t *__restrict__ out) {
    const int slot = blockIdx.x, img = slot / pos_cap, j = slot % pos_cap;
    int32_t *o = out + (size_t)slot * msz * msz;
    if (j >= n_pos[img]) {
        for (int p = threadIdx.x; p < msz * msz; p += blockDim.x) o[p] = -1;
        return;
    }
    const int row = img * n_sample + j;
    const float4 b = *reinterpret_cast<const float4 *>(sample_roi + (size_t)row * 4);
    const int y0 = max((int)b.x, 0), y1 = min((int)b.z, H), x0 = max((int)b.y, 0), x1 = min((int)b.w, W);
    const int sh = y1 - y0, sw = x1 - x0;
    const unsigned char *src = masks + ((size_t)img * gt_cap + gt_assign[row]) * H * W;
    for (int p = threadIdx.x; p < msz * msz; p += blockDim.x) {
        if (sh <= 0 || sw <= 0) { o[p] = 0; continue; }      // the reference raises here (cv2.resize of an empty crop)
        const int dy = p / msz, dx = p % msz;
        const Coef cy = cv_coef(dy, sh, msz), cx = cv_coef(dx, sw, msz);
        const int ya = cy.s, yb = min(cy.s + 1, sh - 1), xa = cx.s, xb = min(cx.s + 1, sw - 1);
        const int p00 = src[(size_t)(y0 + ya) * W + x0 + xa], p01 = src[(size_t)(y0 + ya) * W + x0 + xb];
        const int p10 = src[(size_t)(y0 + yb) * W + x0 + xa], p11 = src[(size_t)(y0 + yb) * W + x0 + xb];
        const int S0 = p00 * cx.a0 + p01 * cx.a1, S1 = p10 * cx.a0 + p11 * cx.a1;
        int v = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        o[p] = v;
    }
}

// Keypoint targets (proposal_target_creator.py:105-127).  kps (N, gt_cap, K, 3) f32 (y, x, v).
// inplace_quirk == 0: every positive transforms its gt's ORIGINAL coordinates (the sensible reading).
// inplace_quirk != 0: the reference's behaviour - `kp = mask[idx]` is a view and `kp[:, :2] = ...`
// writes the transformed coordinates back into the gt array, so a gt assigned to several positives is transformed again
// from its already-transformed coordinates, in sample order.  Positive j replays the transforms of the earlier positives
// i < j with the same gt (<= 64 per image) - same float64 arithmetic and float32 stores as NumPy's.
__global__ __launch_bounds__(64) void k_keypoint_target(const float *__restrict__ kps, int gt_cap, int K,
                                                        const float *__restrict__ sample_roi,
                                                        const int32_t *__restrict__ gt_assign, const int32_t *__restrict__ n_pos,
                                                        int n_sample, int pos_cap, int msz, int inplace_quirk,
                                                        int32_t *__restrict__ out) {
    const int slot = blockIdx.x, img = slot / pos_cap, j = slot % pos_cap;
    int32_t *o = out + (size_t)slot * K;
    const bool live = j < n_pos[img];
    const int row = img * n_sample + j;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int lab = -1;
        if (live) {
            const int g = gt_assign[row];
            const float *kp = kps + (((size_t)img * gt_cap + g) * K + k) * 3;
            float fy = kp[0], fx = kp[1];
            for (int i = inplace_quirk ? 0 : j; i <= j; ++i) {
                const int ri = img * n_sample + i;
                if (i != j && gt_assign[ri] != g) continue;
                const float4 b = *reinterpret_cast<const float4 *>(sample_roi + (size_t)ri * 4);
                const int y0 = (int)b.x, x0 = (int)b.y, y1 = (int)b.z, x1 = (int)b.w;
                // numpy: (kp - [y0,x0]) / [max(y1-y0,1), max(x1-x0,1)] * mask_size in float64, stored to float32
                fy = (float)(((double)fy - (double)y0) / (double)max(y1 - y0, 1) * (double)msz);
                fx = (float)(((double)fx - (double)x0) / (double)max(x1 - x0, 1) * (double)msz);
            }
            const int y = (int)fy, x = (int)fx, v = (int)kp[2];
            if (v == 2 && 0 <= y && y < msz && 0 <= x && x < msz) lab = y * msz + x;
        }
        o[k] = lab;
    }
}

// n_gt[i] = number of rows of image i with label >= 0 (valid rows are packed first; padding rows carry -1).
__global__ __launch_bounds__(64) void k_count_valid_labels(const int32_t *__restrict__ labels, int G, int32_t *__restrict__ n_gt) {
    int c = 0;
    for (int g = threadIdx.x; g < G; g += 64) c += labels[(size_t)blockIdx.x * G + g] >= 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (threadIdx.x == 0) n_gt[blockIdx.x] = c;
}

// ---- anchor targets ------------------------------------------------------------------------------
constexpr int AT_GCAP = 256;

__device__ __forceinline__ bool anchor_inside(const float4 a, float img_h, float img_w) {
    return a.x >= 0.f && a.y >= 0.f && a.z <= img_h && a.w <= img_w;
}

__global__ __launch_bounds__(256) void k_at_iou(const float *__restrict__ anchors, int A, const float *__restrict__ gt_boxes,
                                                const int32_t *__restrict__ n_gt, int gt_cap, float img_h, float img_w,
                                                const float *__restrict__ per_image_hw, float *__restrict__ max_iou,
                                                int32_t *__restrict__ argmax, int *__restrict__ gt_max_bits) {
    __shared__ float4 sg[AT_GCAP];
    const int img = blockIdx.y;
    if (per_image_hw) { img_h = per_image_hw[2 * img]; img_w = per_image_hw[2 * img + 1]; }     // ragged batch: own size
    const int G = min(min(n_gt[img], gt_cap), AT_GCAP);
    for (int g = threadIdx.x; g < G; g += 256) sg[g] = *reinterpret_cast<const float4 *>(gt_boxes + ((size_t)img * gt_cap + g) * 4);
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool live = a < A;
    const float4 an = live ? *reinterpret_cast<const float4 *>(anchors + (size_t)a * 4) : make_float4(-1.f, -1.f, -1.f, -1.f);
    float best = 0.f;
    int arg = 0;
    const bool in = live && anchor_inside(an, img_h, img_w);
    for (int g = 0; g < G; ++g) {
        float v = 0.f;
        if (in) {
            v = box_iou(an, sg[g]);
            if (g == 0 || v > best) { best = v; arg = g; }
        }
        // per-gt maximum over the inside anchors: wave reduction first, one atomic per wave (IoU >= 0: int order == float order)
        int vb = in ? __float_as_int(v) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vb = max(vb, __shfl_xor(vb, o));
        if ((threadIdx.x & 63) == 0 && vb > 0) atomicMax(&gt_max_bits[(size_t)img * AT_GCAP + g], vb);
    }
    if (!live) return;
    max_iou[(size_t)img * A + a] = best;
    argmax[(size_t)img * A + a] = in ? arg : -1;
}

__global__ __launch_bounds__(256) void k_at_label(const float *__restrict__ anchors, int A, const float *__restrict__ gt_boxes,
                                                  const int32_t *__restrict__ n_gt, int gt_cap, const float *__restrict__ max_iou,
                                                  const int32_t *__restrict__ argmax, const int *__restrict__ gt_max_bits,
                                                  const uint32_t *__restrict__ keys, float pos_thresh, float neg_thresh,
                                                  int32_t *__restrict__ label, u64 *__restrict__ skeys, int *__restrict__ counts) {
    __shared__ float4 sg[AT_GCAP];
    __shared__ float sgm[AT_GCAP];
    const int img = blockIdx.y;
    const int G = min(min(n_gt[img], gt_cap), AT_GCAP);
    for (int g = threadIdx.x; g < G; g += 256) {
        sg[g] = *reinterpret_cast<const float4 *>(gt_boxes + ((size_t)img * gt_cap + g) * 4);
        sgm[g] = __int_as_float(gt_max_bits[(size_t)img * AT_GCAP + g]);
    }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    int lab = -1;
    if (a < A && argmax[(size_t)img * A + a] >= 0) {
        const float m = max_iou[(size_t)img * A + a];
        if (m < neg_thresh) lab = 0;
        const float4 an = *reinterpret_cast<const float4 *>(anchors + (size_t)a * 4);
        for (int g = 0; g < G; ++g)
            if (box_iou(an, sg[g]) == sgm[g]) lab = 1;
        if (m >= pos_thresh) lab = 1;
    }
    if (a < A) {
        label[(size_t)img * A + a] = lab;
        const u64 k = (u64)keys[(size_t)img * A + a];
        skeys[(size_t)img * A + a] = lab < 0 ? ~0ull : (((u64)(lab == 0) << 63) | (k << 31) | (u64)a);
    }
    // counts[img*2 + {0: pos, 1: neg}]
    const unsigned long long bp = __ballot(lab == 1), bn = __ballot(lab == 0);
    if ((threadIdx.x & 63) == 0) {
        if (bp) atomicAdd(&counts[img * 2], __popcll(bp));
        if (bn) atomicAdd(&counts[img * 2 + 1], __popcll(bn));
    }
}

// Sub-sampling (AnchorTargetCreator: at most n_pos_max positives, n_sample in total): the kept anchors of a class are
// the ones with the smallest (random key, index).  With the composite keys of k_at_label (positives sort before negatives,
// unlabelled anchors last) those are the keys <= the keep_p-th smallest key (positives) / <= the (np + keep_n)-th smallest
// key (negatives): two radix selects (sort.hip) instead of a sort of all A keys.
__global__ void k_at_ranks(const int *__restrict__ counts, int N, int n_sample, int n_pos_max, unsigned *__restrict__ kreq) {
    const int img = blockIdx.x * 64 + threadIdx.x;
    if (img >= N) return;
    const int np = counts[img * 2], nn = counts[img * 2 + 1];
    const int keep_p = min(np, n_pos_max), keep_n = min(nn, n_sample - keep_p);
    kreq[img * 2] = (unsigned)max(keep_p, 1);                 // rank 0 is not a rank: the class is then skipped below
    kreq[img * 2 + 1] = (unsigned)max(np + keep_n, 1);
}

__global__ __launch_bounds__(256) void k_at_disable(const u64 *__restrict__ skeys, int A, const int *__restrict__ counts,
                                                    const u64 *__restrict__ kth, int n_sample, int n_pos_max,
                                                    int32_t *__restrict__ label) {
    const int img = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const int lab = label[(size_t)img * A + a];
    if (lab < 0) return;
    const int np = counts[img * 2], nn = counts[img * 2 + 1];
    const int keep_p = min(np, n_pos_max), keep_n = min(nn, n_sample - keep_p);
    const u64 key = skeys[(size_t)img * A + a];
    const bool drop = lab == 1 ? (keep_p <= 0 || key > kth[img * 2]) : (keep_n <= 0 || key > kth[img * 2 + 1]);
    if (drop) label[(size_t)img * A + a] = -1;
}

__global__ __launch_bounds__(256) void k_at_loc(const float *__restrict__ anchors, int A, const float *__restrict__ gt_boxes,
                                                const int32_t *__restrict__ n_gt, int gt_cap, const int32_t *__restrict__ argmax,
                                                float *__restrict__ loc) {
    const int img = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const int g = argmax[(size_t)img * A + a];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    // an image without objects (n_gt = 0) has no box to regress to: its inside anchors carry argmax 0 = a padding row, whose
    // zero size would turn into -inf targets (and 0 x inf = NaN in the loss of anchors that are not even positive)
    if (g >= 0 && g < n_gt[img])
        o = bbox2loc(*reinterpret_cast<const float4 *>(anchors + (size_t)a * 4),
                     *reinterpret_cast<const float4 *>(gt_boxes + ((size_t)img * gt_cap + g) * 4));
    *reinterpret_cast<float4 *>(loc + ((size_t)img * A + a) * 4) = o;
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
struct ATLayout { size_t max_iou, argmax, gtmax, counts, skeys, kreq, kth, tmp, tmp_bytes, total; };
ATLayout at_layout(int N, int A) {
    ATLayout L{};
    size_t o = 0;
    L.max_iou = o; o += al((size_t)N * A * 4);
    L.argmax = o; o += al((size_t)N * A * 4);
    L.gtmax = o; o += al((size_t)N * AT_GCAP * 4);
    L.counts = o; o += al((size_t)N * 2 * 4);
    L.skeys = o; o += al((size_t)N * A * 8);
    L.kreq = o; o += al((size_t)N * 2 * 4);
    L.kth = o; o += al((size_t)N * 2 * 8);
    L.tmp_bytes = mrcnn::select_ws_bytes(N, 64);
    L.tmp = o; o += al(L.tmp_bytes);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int mrcnn_proposal_target_f32(const float *rois, const float *roi_levels, const int32_t *n_rois, int roi_cap,
                                         const float *gt_boxes, const int32_t *gt_labels, const int32_t *n_gt,
                                         int gt_cap, const uint32_t *keys, int N, int n_sample, int n_pos_max,
                                         float pos_iou_thresh, float neg_iou_thresh_hi, float neg_iou_thresh_lo,
                                         const float *loc_mean4, const float *loc_std4, float *sample_roi,
                                         float *rois_xy5, int32_t *sample_levels, float *gt_roi_loc,
                                         int32_t *gt_roi_label, int32_t *gt_assign, int32_t *sample_src,
                                         int32_t *n_pos, int32_t *n_sampled, const int32_t *pos_order,
                                         const int32_t *neg_order, int32_t *n_cand, void *stream) {
    if ((pos_order == nullptr) != (neg_order == nullptr))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "proposal_target: pos_order and neg_order go together");
    if (!rois || !roi_levels || !n_rois || !gt_boxes || !gt_labels || !n_gt || (!keys && !pos_order) || !loc_mean4 || !loc_std4 ||
        !sample_roi || !rois_xy5 || !sample_levels || !gt_roi_loc || !gt_roi_label || !gt_assign || !sample_src ||
        !n_pos || !n_sampled)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "proposal_target: null pointer");
    if (N <= 0 || roi_cap <= 0 || gt_cap <= 0 || n_sample <= 0 || n_sample > 512 || n_pos_max < 0 || n_pos_max > n_sample)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "proposal_target: bad sizes (n_sample <= 512)");
    if (roi_cap + gt_cap > PT_CAP)
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "proposal_target: %d proposals + %d gt boxes > %d", roi_cap, gt_cap, PT_CAP);
    const float4 mean = make_float4(loc_mean4[0], loc_mean4[1], loc_mean4[2], loc_mean4[3]);
    const float4 stdv = make_float4(loc_std4[0], loc_std4[1], loc_std4[2], loc_std4[3]);
    hipLaunchKernelGGL(k_proposal_target, dim3(N), dim3(PT_THREADS), 0, (hipStream_t)stream, rois, roi_levels, n_rois,
                       roi_cap, gt_boxes, gt_labels, n_gt, gt_cap, keys, n_sample, n_pos_max, pos_iou_thresh,
                       neg_iou_thresh_hi, neg_iou_thresh_lo, mean, stdv, sample_roi, rois_xy5, sample_levels, gt_roi_loc,
                       gt_roi_label, gt_assign, sample_src, n_pos, n_sampled, pos_order, neg_order, n_cand);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_mask_target_u8(const unsigned char *masks, int N, int gt_cap, int H, int W, const float *sample_roi,
                                    const int32_t *gt_assign, const int32_t *n_pos, int n_sample, int pos_cap,
                                    int mask_size, int32_t *gt_roi_mask, void *stream) {
    if (!masks || !sample_roi || !gt_assign || !n_pos || !gt_roi_mask || N <= 0 || gt_cap <= 0 || H <= 0 || W <= 0 ||
        n_sample <= 0 || pos_cap <= 0 || pos_cap > n_sample || mask_size <= 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "mask_target: bad arguments");
    hipLaunchKernelGGL(k_mask_target, dim3(N * pos_cap), dim3(256), 0, (hipStream_t)stream, masks, gt_cap, H, W, sample_roi,
                       gt_assign, n_pos, n_sample, pos_cap, mask_size, gt_roi_mask);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_keypoint_target_f32(const float *keypoints, int N, int gt_cap, int K, const float *sample_roi,
                                         const int32_t *gt_assign, const int32_t *n_pos, int n_sample, int pos_cap,
                                         int mask_size, int inplace_quirk, int32_t *gt_roi_kp, void *stream) {
    if (!keypoints || !sample_roi || !gt_assign || !n_pos || !gt_roi_kp || N <= 0 || gt_cap <= 0 || K <= 0 ||
        n_sample <= 0 || pos_cap <= 0 || pos_cap > n_sample || mask_size <= 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "keypoint_target: bad arguments");
    hipLaunchKernelGGL(k_keypoint_target, dim3(N * pos_cap), dim3(64), 0, (hipStream_t)stream, keypoints, gt_cap, K,
                       sample_roi, gt_assign, n_pos, n_sample, pos_cap, mask_size, inplace_quirk, gt_roi_kp);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_count_valid_labels_i32(const int32_t *labels, int N, int G, int32_t *n_gt, void *stream) {
    if (!labels || !n_gt || N <= 0 || G <= 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "count_valid_labels: bad arguments");
    hipLaunchKernelGGL(k_count_valid_labels, dim3(N), dim3(64), 0, (hipStream_t)stream, labels, G, n_gt);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t mrcnn_anchor_target_workspace_bytes(int N, int A) {
    if (N <= 0 || A <= 0) return 0;
    return at_layout(N, A).total;
}

extern "C" int mrcnn_anchor_target_f32(const float *anchors, int A, const float *gt_boxes, const int32_t *n_gt, int gt_cap,
                                       int N, float img_h, float img_w, const float *per_image_hw, const uint32_t *keys,
                                       int n_sample, float pos_iou_thresh, float neg_iou_thresh, float pos_ratio, int do_sample,
                                       float *gt_rpn_loc, int32_t *gt_rpn_label, void *ws, size_t ws_bytes,
                                       void *stream) {
    if (!anchors || !gt_boxes || !n_gt || !gt_rpn_loc || !gt_rpn_label || !ws || (do_sample && !keys))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "anchor_target: null pointer");
    if (N <= 0 || A <= 0 || gt_cap <= 0 || gt_cap > AT_GCAP || n_sample <= 0 || A >= (1 << 30))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "anchor_target: bad sizes (gt_cap <= %d)", AT_GCAP);
    const ATLayout L = at_layout(N, A);
    if (ws_bytes < L.total) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "anchor_target: workspace %zu < %zu", ws_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)ws;
    float *max_iou = (float *)(w + L.max_iou);
    int32_t *argmax = (int32_t *)(w + L.argmax);
    int *gtmax = (int *)(w + L.gtmax), *counts = (int *)(w + L.counts);
    u64 *skeys = (u64 *)(w + L.skeys), *kth = (u64 *)(w + L.kth);
    unsigned *kreq = (unsigned *)(w + L.kreq);
    MRCNN_HIP_TRY(hipMemsetAsync(w + L.gtmax, 0, (L.counts - L.gtmax) + al((size_t)N * 2 * 4), st));
    const dim3 grid(mrcnn::cdiv(A, 256), N);
    hipLaunchKernelGGL(k_at_iou, grid, dim3(256), 0, st, anchors, A, gt_boxes, n_gt, gt_cap, img_h, img_w, per_image_hw, max_iou, argmax, gtmax);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_at_label, grid, dim3(256), 0, st, anchors, A, gt_boxes, n_gt, gt_cap, max_iou, argmax, gtmax,
                       do_sample ? keys : (const uint32_t *)max_iou, pos_iou_thresh, neg_iou_thresh, gt_rpn_label, skeys, counts);
    MRCNN_LAUNCH_CHECK();
    if (do_sample) {
        const int n_pos_max = (int)(pos_ratio * n_sample);
        hipLaunchKernelGGL(k_at_ranks, dim3(mrcnn::cdiv(N, 64)), dim3(64), 0, st, counts, N, n_sample, n_pos_max, kreq);
        MRCNN_LAUNCH_CHECK();
        // kth[img][0] = the keep_p-th smallest key, kth[img][1] = the (np + keep_n)-th smallest key (the two selects share
        // the workspace: the second starts after the first has finished on this stream)
        for (int c = 0; c < 2; ++c)
            if (int e = mrcnn::select_kth(skeys, N, (size_t)A, 64, false, kreq + c, 2, kth + c, 2, w + L.tmp, st)) return e;
        hipLaunchKernelGGL(k_at_disable, grid, dim3(256), 0, st, skeys, A, counts, kth, n_sample, n_pos_max, gt_rpn_label);
        MRCNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_at_loc, grid, dim3(256), 0, st, anchors, A, gt_boxes, n_gt, gt_cap, argmax, gt_rpn_loc);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// The five training losses of the Mask R-CNN step, forward + gradient, for gfx950.
//
// Replaces (chainer_maskrcnn/model/fpn_maskrcnn_train_chain.py:83-85,100-106; train.py:50-58;
// train_keypoints.py:21-27) the Chainer / ChainerCV functions
//   F.softmax_cross_entropy (normalize=True, ignore_label=-1)  -> mrcnn_softmax_ce_f32
//   _fast_rcnn_loc_loss / _smooth_l1_loss                        -> mrcnn_smooth_l1_f32
//   calc_mask_loss = channel select + F.sigmoid_cross_entropy    -> mrcnn_mask_bce_f32
// Each entry point writes the scalar loss to device memory (no host sync) and the gradient of
// the loss w.r.t. the logits (d loss = 1), already normalised.  Reductions are two-stage with a
// fixed order (bit-reproducible).  All are HBM/latency-bound: bytes = 2 x logits (+ targets).
#include "common.h"
#include <algorithm>

namespace {

constexpr int NT = 256;
constexpr int MAXB = 1024;     // partial slots
__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

struct RowMap {  // element (r, j) of a logically (M, K) matrix lives at (r/A)*gs + (r%A)*rs + j*es
    int A;
    long long gs, rs, es;
};
__device__ __forceinline__ size_t rm_off(const RowMap &m, int r, int j) {
    return (size_t)((long long)(r / m.A) * m.gs + (long long)(r % m.A) * m.rs + (long long)j * m.es);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// block-level deterministic sum of (a, b) -> part[blockIdx.x*2 + {0,1}]
__device__ __forceinline__ void block_partial(float a, float b, float *part) {
    __shared__ float sa[NT / 64], sb[NT / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.f, y = 0.f;
        for (int k = 0; k < NT / 64; ++k) { x += sa[k]; y += sb[k]; }
        part[blockIdx.x * 2] = x;
        part[blockIdx.x * 2 + 1] = y;
    }
}

// out[0] = sum(a)/max(sum(b),1), out[1] = max(sum(b),1): one wave, lane l adds partials l, l+64, ... in double,
// then a fixed xor-shuffle tree (bit-reproducible).
__global__ __launch_bounds__(64) void k_finalize(const float *__restrict__ part, int nb, float *__restrict__ out) {
    // (r6) four (sum, count) pairs in flight per lane, four accumulator pairs combined in fixed order: the loop was one dependent load per
    // iteration - up to 64 round trips, 13 us per call on the loss tail of the forward pass; still bit-reproducible)
    const float2 *p2 = reinterpret_cast<const float2 *>(part);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int k = threadIdx.x;
    for (; k + 192 < nb; k += 256) {
        const float2 v0 = p2[k], v1 = p2[k + 64], v2 = p2[k + 128], v3 = p2[k + 192];
        a0 += (double)v0.x; b0 += (double)v0.y; a1 += (double)v1.x; b1 += (double)v1.y;
        a2 += (double)v2.x; b2 += (double)v2.y; a3 += (double)v3.x; b3 += (double)v3.y;
    }
    for (; k < nb; k += 64) { const float2 v = p2[k]; a0 += (double)v.x; b0 += (double)v.y; }
    double a = (a0 + a1) + (a2 + a3), b = (b0 + b1) + (b2 + b3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (threadIdx.x == 0) {
        if (b < 1.0) b = 1.0;
        out[0] = (float)(a / b);
        out[1] = (float)b;
    }
}

// ---- softmax cross entropy -------------------------------------------------------------------
// Small K (<= 8): one thread per row.
template <int PHASE>   // 0: loss partials, 1: gradient
__global__ __launch_bounds__(NT) void k_sce_small(const float *__restrict__ x, RowMap xm, const int32_t *__restrict__ t,
                                                  int M, int K, int ignore, float *__restrict__ part,
                                                  const float *__restrict__ norm, float *__restrict__ gx, RowMap gm) {
    float ls = 0.f, cnt = 0.f;
    const float inv = PHASE ? 1.0f / norm[1] : 0.f;
    for (int r = blockIdx.x * NT + threadIdx.x; r < M; r += gridDim.x * NT) {
        const int tt = t[r];
        const bool valid = tt != ignore;
        float v[8], m = -INFINITY;
        for (int j = 0; j < K; ++j) { v[j] = x[rm_off(xm, r, j)]; m = fmaxf(m, v[j]); }
        float s = 0.f;
        for (int j = 0; j < K; ++j) s += expf(v[j] - m);
        const float lse = logf(s);
        if (PHASE == 0) {
            if (valid) { ls += -((v[tt] - m) - lse); cnt += 1.f; }
        } else {
            for (int j = 0; j < K; ++j) {
                float g = 0.f;
                if (valid) g = (expf((v[j] - m) - lse) - (j == tt ? 1.f : 0.f)) * inv;
                gx[rm_off(gm, r, j)] = g;
            }
        }
    }
    if (PHASE == 0) block_partial(ls, cnt, part);
}

// General K: one wave per row.
template <int PHASE>
__global__ __launch_bounds__(NT) void k_sce_wave(const float *__restrict__ x, RowMap xm, const int32_t *__restrict__ t,
                                                 int M, int K, int Kfill, int ignore, float *__restrict__ part,
                                                 const float *__restrict__ norm, float *__restrict__ gx, RowMap gm) {
    const int lane = threadIdx.x & 63;
    float ls = 0.f, cnt = 0.f;
    const float inv = PHASE ? 1.0f / norm[1] : 0.f;
    for (int r = (blockIdx.x * NT + threadIdx.x) >> 6; r < M; r += (gridDim.x * NT) >> 6) {
        const int tt = t[r];
        const bool valid = tt != ignore;
        float m = -INFINITY;
        for (int j = lane; j < K; j += 64) m = fmaxf(m, x[rm_off(xm, r, j)]);
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < K; j += 64) s += expf(x[rm_off(xm, r, j)] - m);
        s = wave_sum(s);
        const float lse = logf(s);
        if (PHASE == 0) {
            if (valid && lane == 0) { ls += -((x[rm_off(xm, r, tt)] - m) - lse); cnt += 1.f; }
        } else {
            for (int j = lane; j < Kfill; j += 64) {
                float g = 0.f;
                if (valid && j < K) g = (expf((x[rm_off(xm, r, j)] - m) - lse) - (j == tt ? 1.f : 0.f)) * inv;
                gx[rm_off(gm, r, j)] = g;
            }
        }
    }
    if (PHASE == 0) block_partial(ls, cnt, part);
}

// Channel-interleaved rows: logits (G, K positions, C channels) NHWC, logical row r = (g, a) with a < A <= C, class j = position
// (the keypoint loss: 17 heat maps of 56 x 56 per RoI in 32 padded channels).  The one-wave-per-row kernel above reads such a row
// with a stride of C floats - one cache line per lane - and ran at 0.8 TB/s.  Here ONE workgroup owns a g: float4 loads over the
// contiguous (position, channel) plane, every thread keeps the running maximum and sum of its four channels over its positions
// (online softmax: the sum is rescaled when the maximum moves), the 256 / (C/4) threads of a channel quad are combined through
// LDS, and - when the gradient is wanted - a second pass over the plane writes it, zeros in the padded channels and the ignored
// rows included.  The normaliser (number of valid rows) is counted from t by every workgroup itself, so loss partials and
// gradient come from ONE launch.
template <bool GRAD>
__global__ __launch_bounds__(NT) void k_sce_chan(const float *__restrict__ x, const int32_t *__restrict__ t, int G, int A, int K, int C,
                                                 int ignore, float *__restrict__ part, float *__restrict__ gx) {
    __shared__ float s_m[NT][4], s_s[NT][4];
    __shared__ float s_max[256], s_lse[256];          // per channel (C <= 256)
    __shared__ float s_cnt[NT / 64];
    const int Q = C >> 2, q = threadIdx.x % Q, p0 = threadIdx.x / Q, PP = NT / Q;
    float inv = 0.f;
    if (GRAD) {                                       // valid rows of the whole call (M = G * A labels, L2-resident)
        float c = 0.f;
        for (int i = threadIdx.x; i < G * A; i += NT) c += (t[i] != ignore) ? 1.f : 0.f;
        c = wave_sum(c);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
        __syncthreads();
        float tot = 0.f;
        for (int k = 0; k < NT / 64; ++k) tot += s_cnt[k];
        inv = 1.0f / fmaxf(tot, 1.0f);
    }
    float ls = 0.f, cnt = 0.f;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const float *xg = x + (size_t)g * K * C + q * 4;
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = p0; p < K; p += PP) {
            const float4 v4 = ldg4(xg + (size_t)p * C);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (v[c] > m[c]) { s[c] *= expf(m[c] - v[c]); m[c] = v[c]; }      // rare after the first positions
                s[c] += expf(v[c] - m[c]);
            }
        }
        __syncthreads();                              // (previous g's readers of s_max / s_lse are done)
#pragma unroll
        for (int c = 0; c < 4; ++c) { s_m[threadIdx.x][c] = m[c]; s_s[threadIdx.x][c] = s[c]; }
        __syncthreads();
        if (threadIdx.x < C) {                        // thread = channel: combine the PP partial (max, sum) pairs in order
            const int cq = threadIdx.x >> 2, cc = threadIdx.x & 3;
            float M_ = -INFINITY;
            for (int k = 0; k < PP; ++k) M_ = fmaxf(M_, s_m[k * Q + cq][cc]);
            float S_ = 0.f;
            for (int k = 0; k < PP; ++k) {
                const float mk = s_m[k * Q + cq][cc];
                if (mk > -INFINITY) S_ += s_s[k * Q + cq][cc] * expf(mk - M_);
            }
            s_max[threadIdx.x] = M_;
            s_lse[threadIdx.x] = logf(S_);
            if (threadIdx.x < A) {
                const int tt = t[g * A + threadIdx.x];
                if (tt != ignore) {
                    ls += -((x[(size_t)g * K * C + (size_t)tt * C + threadIdx.x] - M_) - s_lse[threadIdx.x]);
                    cnt += 1.f;
                }
            }
        }
        __syncthreads();
        if (GRAD) {
            int tt[4];
            float mm[4], ll[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = q * 4 + c;
                tt[c] = ch < A ? t[g * A + ch] : ignore;
                mm[c] = s_max[ch]; ll[c] = s_lse[ch];
            }
            float *gg = gx + (size_t)g * K * C + q * 4;
            for (int p = p0; p < K; p += PP) {
                const float4 v4 = ldg4(xg + (size_t)p * C);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool valid = (q * 4 + c < A) && tt[c] != ignore;
                    o[c] = valid ? (expf((v[c] - mm[c]) - ll[c]) - (p == tt[c] ? 1.f : 0.f)) * inv : 0.f;
                }
                *reinterpret_cast<float4 *>(gg + (size_t)p * C) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    block_partial(ls, cnt, part);
}

// ---- smooth L1 (Fast R-CNN loc loss) -----------------------------------------------------------
template <int PHASE>
__global__ __launch_bounds__(NT) void k_sl1(const float *__restrict__ x, int ldx, const float *__restrict__ t,
                                            const int32_t *__restrict__ label, int M, float sigma2,
                                            float *__restrict__ part, const float *__restrict__ norm,
                                            float *__restrict__ gx, int ldg, int gfill) {
    float ls = 0.f, cnt = 0.f;
    const float inv = PHASE ? 1.0f / norm[1] : 0.f;
    for (int r = blockIdx.x * NT + threadIdx.x; r < M; r += gridDim.x * NT) {
        const int lb = label[r];
        const float w = lb > 0 ? 1.f : 0.f;
        if (PHASE == 0 && lb >= 0) cnt += 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = w * (x[(size_t)r * ldx + j] - t[(size_t)r * 4 + j]);
            const float ad = fabsf(d);
            const bool quad = ad < 1.0f / sigma2;
            if (PHASE == 0) ls += quad ? (sigma2 * 0.5f) * d * d : ad - 0.5f / sigma2;
            else {
                const float gd = quad ? sigma2 * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                gx[(size_t)r * ldg + j] = gd * w * inv;
            }
        }
        if (PHASE == 1)
            for (int j = 4; j < gfill; ++j) gx[(size_t)r * ldg + j] = 0.f;
    }
    if (PHASE == 0) block_partial(ls, cnt, part);
}

// ---- mask loss: channel label-1 of each row with label > 0, BCE with logits -------------------
template <int PHASE>
__global__ __launch_bounds__(NT) void k_mask_bce(const float *__restrict__ x, const int32_t *__restrict__ gt,
                                                 const int32_t *__restrict__ label, int Rm, int HW, int Cm,
                                                 float *__restrict__ part, const float *__restrict__ norm,
                                                 float *__restrict__ gx) {
    float ls = 0.f, cnt = 0.f;
    const float inv = PHASE ? 1.0f / norm[1] : 0.f;
    const long long total = (long long)Rm * HW;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int r = (int)(i / HW);
        const int lb = label[r];
        if (lb <= 0) continue;
        const int tt = gt[i];
        if (tt == -1) continue;
        const size_t o = (size_t)i * Cm + (lb - 1);
        const float v = x[o], tf = (float)tt;
        if (PHASE == 0) {
            ls += -(v * (tf - (v >= 0.f ? 1.f : 0.f)) - log1pf(expf(-fabsf(v))));
            cnt += 1.f;
        } else {
            gx[o] = (1.0f / (1.0f + expf(-v)) - tf) * inv;
        }
    }
    if (PHASE == 0) block_partial(ls, cnt, part);
}

// ---- generic (user-supplied mask_loss_fun) building blocks -------------------------------------
// F.sigmoid_cross_entropy(x, t) with normalize=True, ignore label -1, mean reduction (train.py:57-58).
template <int PHASE>
__global__ __launch_bounds__(NT) void k_sigmoid_ce(const float *__restrict__ x, const int32_t *__restrict__ t, long long n,
                                                   float *__restrict__ part, const float *__restrict__ norm,
                                                   float *__restrict__ gx) {
    float ls = 0.f, cnt = 0.f;
    const float inv = PHASE ? 1.0f / norm[1] : 0.f;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int tt = t[i];
        const float v = x[i], tf = (float)tt;
        if (PHASE == 0) {
            if (tt != -1) {
                ls += -(v * (tf - (v >= 0.f ? 1.f : 0.f)) - log1pf(expf(-fabsf(v))));
                cnt += 1.f;
            }
        } else {
            gx[i] = tt != -1 ? (1.0f / (1.0f + expf(-v)) - tf) * inv : 0.f;
        }
    }
    if (PHASE == 0) block_partial(ls, cnt, part);
}

// roi_cls_mask[arange(R), idx] of train.py:55-56 on an NCHW tensor: y[r, p] = x[r, idx[r], p]; negative indices wrap
// like NumPy's (the background rows carry label 0 - 1 = -1).  BWD: gx zero-filled, gx[r, idx[r], p] = gy[r, p].
template <int BWD>
__global__ __launch_bounds__(NT) void k_select_channel(const float *__restrict__ src, const int32_t *__restrict__ idx, int R, int C,
                                                       int HW, float *__restrict__ dst) {
    const long long total = BWD ? (long long)R * C * HW : (long long)R * HW;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        if (BWD) {
            const int p = (int)(i % HW);
            const long long rc = i / HW;
            const int c = (int)(rc % C), r = (int)(rc / C);
            int k = idx[r];
            k = k < 0 ? k + C : k;
            dst[i] = c == k ? src[(size_t)r * HW + p] : 0.f;
        } else {
            const int p = (int)(i % HW), r = (int)(i / HW);
            int k = idx[r];
            k = k < 0 ? k + C : k;
            dst[i] = src[((size_t)r * C + k) * HW + p];
        }
    }
}

// (R, HW, Cp) NHWC (Cp >= C: padded channels) <-> (R, C, HW) NCHW through a 32x32 LDS tile (both sides coalesced).
// inverse != 0: NCHW -> NHWC with the padding channels zero-filled.
__global__ __launch_bounds__(256) void k_nhwc_nchw(const float *__restrict__ src, float *__restrict__ dst, int HW, int Cp, int C,
                                                   int inverse) {
    __shared__ float tile[32][33];
    const int r = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const float *s = src + (size_t)r * (inverse ? (size_t)C * HW : (size_t)HW * Cp);
    float *d = dst + (size_t)r * (inverse ? (size_t)HW * Cp : (size_t)C * HW);
    if (!inverse) {
        for (int j = ty; j < 32; j += 8) {       // read pixel p0+j, channels c0+tx
            const int p = p0 + j, c = c0 + tx;
            tile[j][tx] = (p < HW && c < C) ? s[(size_t)p * Cp + c] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {       // write channel c0+j, pixels p0+tx
            const int c = c0 + j, p = p0 + tx;
            if (c < C && p < HW) d[(size_t)c * HW + p] = tile[tx][j];
        }
    } else {
        for (int j = ty; j < 32; j += 8) {       // read channel c0+j, pixels p0+tx
            const int c = c0 + j, p = p0 + tx;
            tile[j][tx] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {       // write pixel p0+j, channels c0+tx (padding channels = 0)
            const int p = p0 + j, c = c0 + tx;
            if (p < HW && c < Cp) d[(size_t)p * Cp + c] = tile[tx][j];
        }
    }
}

// x[i] *= scale[0] (scale in device memory): the upstream gradient of a loss whose gradient was formed for d loss = 1.
__global__ __launch_bounds__(NT) void k_scale_dev(float *__restrict__ x, size_t n, const float *__restrict__ scale) {
    const float s = scale[0];
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) x[i] *= s;
}

// out[0] = sum_i losses[2*i]   (the un-weighted sum of fpn_maskrcnn_train_chain.py:106)
__global__ void k_loss_total(const float *__restrict__ losses, int n, float *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += losses[2 * i];
    out[0] = s;
}

int grid_for(long long work) { return (int)std::max(1ll, std::min<long long>((work + NT - 1) / NT, MAXB)); }

}  // namespace

extern "C" size_t mrcnn_loss_workspace_bytes(void) { return (size_t)MAXB * 2 * sizeof(float) + 16; }

// x: logically (M,K); element (r,j) at (r/A)*gs + (r%A)*rs + j*es floats.  gx uses the same map shape with its own
// strides; columns K..Kfill-1 of gx are zero-filled (es must be 1 for Kfill > K).  loss_out[0] = loss, [1] = count.
// The ONE statement of when mrcnn_softmax_ce_f32 takes the channel-interleaved kernel (k_sce_chan), which writes EVERY element of gx
// (zeros in padded channels and ignored rows): used by the dispatch below and exported as mrcnn_softmax_ce_fills_gx, so that a caller
// that skips zero-filling gx can never disagree with the callee.
static bool sce_chan_path(int M, int K, int A, long long gs, long long rs, long long es, bool has_gx, long long ggs, long long grs,
                          long long ges, int Kfill) {
    if (Kfill < K) Kfill = K;
    return rs == 1 && es >= 4 && es <= 256 && (es % 4) == 0 && (NT % (es / 4)) == 0 && A <= es && gs == (long long)K * es &&
           (!has_gx || (grs == 1 && ges == es && ggs == gs)) && Kfill == K && M > 0 && (M % A) == 0 && K >= 64;
}
extern "C" int mrcnn_softmax_ce_fills_gx(int M, int K, int A, long long gs, long long rs, long long es, long long ggs, long long grs,
                                         long long ges, int Kfill) {
    return sce_chan_path(M, K, A, gs, rs, es, true, ggs, grs, ges, Kfill) ? 1 : 0;
}

extern "C" int mrcnn_softmax_ce_f32(const float *x, int A, long long gs, long long rs, long long es, const int32_t *t,
                                    int M, int K, int ignore_label, float *loss_out, float *gx, long long ggs,
                                    long long grs, long long ges, int Kfill, void *ws, size_t ws_bytes, void *stream) {
    if ((M > 0 && (!x || !t)) || !loss_out || !ws || M < 0 || K <= 0 || A <= 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "softmax_ce: bad arguments");
    if (ws_bytes < mrcnn_loss_workspace_bytes()) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "softmax_ce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)ws;
    const RowMap xm{A, gs, rs, es}, gm{A, ggs, grs, ges};
    if (Kfill < K) Kfill = K;
    // channel-interleaved rows (the keypoint loss): one workgroup per group of A rows, coalesced (k_sce_chan)
    const bool chan = sce_chan_path(M, K, A, gs, rs, es, gx != nullptr, ggs, grs, ges, Kfill);
    if (chan) {
        const int G = M / A, nbc = std::min(G, MAXB);
        if (gx) hipLaunchKernelGGL(k_sce_chan<true>, dim3(nbc), dim3(NT), 0, st, x, t, G, A, K, (int)es, ignore_label, part, gx);
        else hipLaunchKernelGGL(k_sce_chan<false>, dim3(nbc), dim3(NT), 0, st, x, t, G, A, K, (int)es, ignore_label, part, gx);
        MRCNN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, part, nbc, loss_out);
        MRCNN_LAUNCH_CHECK();
        return 0;
    }
    const bool small = K <= 8 && Kfill == K;
    const int nb = small ? grid_for(M) : grid_for((long long)M * 64);
    if (small) hipLaunchKernelGGL(k_sce_small<0>, dim3(nb), dim3(NT), 0, st, x, xm, t, M, K, ignore_label, part, nullptr, nullptr, gm);
    else hipLaunchKernelGGL(k_sce_wave<0>, dim3(nb), dim3(NT), 0, st, x, xm, t, M, K, Kfill, ignore_label, part, nullptr, nullptr, gm);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, part, nb, loss_out);
    MRCNN_LAUNCH_CHECK();
    if (gx && M > 0) {
        if (small) hipLaunchKernelGGL(k_sce_small<1>, dim3(nb), dim3(NT), 0, st, x, xm, t, M, K, ignore_label, part, loss_out, gx, gm);
        else hipLaunchKernelGGL(k_sce_wave<1>, dim3(nb), dim3(NT), 0, st, x, xm, t, M, K, Kfill, ignore_label, part, loss_out, gx, gm);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}

// x (M, ldx>=4) predictions, t (M,4) targets, label (M,): weight 1 where label>0, normaliser #(label>=0).
extern "C" int mrcnn_smooth_l1_f32(const float *x, int ldx, const float *t, const int32_t *label, int M, float sigma,
                                   float *loss_out, float *gx, int ldg, int gfill, void *ws, size_t ws_bytes,
                                   void *stream) {
    if ((M > 0 && (!x || !t || !label)) || !loss_out || !ws || M < 0 || ldx < 4)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "smooth_l1: bad arguments");
    if (ws_bytes < mrcnn_loss_workspace_bytes()) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "smooth_l1: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)ws;
    const int nb = grid_for(M);
    const float s2 = sigma * sigma;
    hipLaunchKernelGGL(k_sl1<0>, dim3(nb), dim3(NT), 0, st, x, ldx, t, label, M, s2, part, nullptr, nullptr, 0, 0);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, part, nb, loss_out);
    MRCNN_LAUNCH_CHECK();
    if (gx && M > 0) {
        hipLaunchKernelGGL(k_sl1<1>, dim3(nb), dim3(NT), 0, st, x, ldx, t, label, M, s2, part, loss_out, gx, ldg, gfill);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}

// x (Rm, HW, Cm) NHWC mask logits, gt (Rm, HW) int32 in {0,1,-1}, label (Rm,) int32 (class 1..; <=0 = row unused).
// gx (same shape as x) is zero-filled by the callee and receives the gradient at channel label-1.
extern "C" int mrcnn_mask_bce_f32(const float *x, const int32_t *gt, const int32_t *label, int Rm, int HW, int Cm,
                                  float *loss_out, float *gx, void *ws, size_t ws_bytes, void *stream) {
    if ((Rm > 0 && (!x || !gt || !label)) || !loss_out || !ws || Rm < 0 || HW <= 0 || Cm <= 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "mask_bce: bad arguments");
    if (ws_bytes < mrcnn_loss_workspace_bytes()) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "mask_bce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)ws;
    const int nb = grid_for((long long)Rm * HW);
    hipLaunchKernelGGL(k_mask_bce<0>, dim3(nb), dim3(NT), 0, st, x, gt, label, Rm, HW, Cm, part, nullptr, nullptr);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, part, nb, loss_out);
    MRCNN_LAUNCH_CHECK();
    if (gx && Rm > 0) {
        MRCNN_HIP_TRY(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)Rm * HW * Cm, st));
        hipLaunchKernelGGL(k_mask_bce<1>, dim3(nb), dim3(NT), 0, st, x, gt, label, Rm, HW, Cm, part, loss_out, gx);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mrcnn_sigmoid_ce_f32(const float *x, const int32_t *t, long long n, float *loss_out, float *gx, void *ws,
                                    size_t ws_bytes, void *stream) {
    if ((n > 0 && (!x || !t)) || !loss_out || !ws || n < 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "sigmoid_ce: bad arguments");
    if (ws_bytes < mrcnn_loss_workspace_bytes()) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "sigmoid_ce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)ws;
    const int nb = grid_for(n);
    hipLaunchKernelGGL(k_sigmoid_ce<0>, dim3(nb), dim3(NT), 0, st, x, t, n, part, nullptr, nullptr);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, part, nb, loss_out);
    MRCNN_LAUNCH_CHECK();
    if (gx && n > 0) {
        hipLaunchKernelGGL(k_sigmoid_ce<1>, dim3(nb), dim3(NT), 0, st, x, t, n, part, loss_out, gx);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mrcnn_select_channel_f32(const float *src, const int32_t *idx, int R, int C, int HW, float *dst, int backward,
                                        void *stream) {
    if (R < 0 || C <= 0 || HW <= 0 || (R > 0 && (!src || !idx || !dst)))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "select_channel: bad arguments");
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (backward) hipLaunchKernelGGL(k_select_channel<1>, dim3(grid_for((long long)R * C * HW)), dim3(NT), 0, st, src, idx, R, C, HW, dst);
    else hipLaunchKernelGGL(k_select_channel<0>, dim3(grid_for((long long)R * HW)), dim3(NT), 0, st, src, idx, R, C, HW, dst);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_nhwc_nchw_f32(const float *src, float *dst, int R, int HW, int Cp, int C, int inverse, void *stream) {
    if (R < 0 || HW <= 0 || C <= 0 || Cp < C || R > 65535 || (R > 0 && (!src || !dst)))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "nhwc_nchw: bad arguments (R <= 65535)");
    if (R == 0) return 0;
    hipLaunchKernelGGL(k_nhwc_nchw, dim3(mrcnn::cdiv(HW, 32), mrcnn::cdiv(inverse ? Cp : C, 32), R), dim3(256), 0,
                       (hipStream_t)stream, src, dst, HW, Cp, C, inverse);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_scale_by_dev_f32(float *x, size_t n, const float *scale, void *stream) {
    if (!scale || (n > 0 && !x)) return mrcnn::fail_arg(MRCNN_E_INVALID, "scale_by_dev: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scale_dev, dim3(grid_for((long long)n)), dim3(NT), 0, (hipStream_t)stream, x, n, scale);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// losses: n pairs (loss, normaliser) as written by the entry points above; out[0] = sum of the n losses.
extern "C" int mrcnn_loss_total_f32(const float *losses, int n, float *out, void *stream) {
    if (!losses || !out || n <= 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "loss_total: bad arguments");
    hipLaunchKernelGGL(k_loss_total, dim3(1), dim3(64), 0, (hipStream_t)stream, losses, n, out);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

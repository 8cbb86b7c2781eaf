// ROIAlign forward (bilinear gather) and backward (owner-computes scatter) for gfx950.
//
// Replaces chainer_maskrcnn.functions.roi_align.roi_align_2d (un-vendored submodule, called
// through chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7) and the per-RoI Python loops of
// chainer_maskrcnn/model/head/fpn_roi_mask_head.py:59-61,75-77.
//
// Layout: channel-innermost (NHWC).  One wavefront (64 lanes) covers 256 channels as one
// float4 per lane, so every bilinear tap and every output row is one coalesced 1-KiB
// segment.  The coordinate arithmetic is compiled with FP contraction OFF and follows
// oracle/roi_align.py operation for operation, so sample indices and weights are bit-exact.
//
// Backward design (see DESIGN.md "roi_align_bwd"): no global atomics.  The gradient map is cut
// into 8x8-cell tiles; one workgroup owns a tile and every wave owns whole cells, so gx is
// written exactly once, coalesced, and sums are bit-reproducible.  Because bilinear weights
// are separable, a RoI's contribution to cell (Y,X) is
//     sum_ph sum_pw Wy[Y][ph] * Wx[X][pw] * gy[r,ph,pw,:] / (gh*gw)
// where Wy[Y][ph] is the summed weight that bin-row ph's samples put on map row Y.  The
// workgroup builds these tiny per-(RoI,tile) tables in LDS (lane-parallel geometry), then
// each wave walks the non-zero (ph,pw) pairs of its cell with wave-uniform control flow
// (ballot + readlane), streaming gy rows through L1/L2.
#include "common.h"
#include <algorithm>
#include <type_traits>

#pragma clang fp contract(off)

#include "roi_align_common.h"

using namespace mrcnn_roi;

namespace {

// ------------------------------------------------------------------------------------------
// Forward, NHWC: one wave per (roi, ph, pw) bin; lane = 4 channels.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_roi_align_fwd_nhwc(Levels lv, const float *__restrict__ rois,
                                                            const int32_t *__restrict__ levels, int R,
                                                            int N, int C, int PH, int PW, int sr,
                                                            float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long long bin_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int bins = PH * PW;
    if (bin_id >= (long long)R * bins) return;
    const int r = (int)(bin_id / bins);
    const int b = (int)(bin_id - (long long)r * bins);
    const int ph = b / PW, pw = b - ph * PW;
    int l = levels ? levels[r] : 0;
    l = min(max(l, 0), lv.L - 1);
    const int H = lv.H[l], W = lv.W[l];
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, lv.scale[l], PH, PW, sr);
    const int C4 = C >> 2;
    float *yo = y + ((size_t)r * bins + b) * C;
    const bool bad = g.n < 0 || g.n >= N;
    const float *xb = lv.x[l] + (size_t)(bad ? 0 : g.n) * H * W * C;
    const float cnt = (float)(g.gh * g.gw);
    for (int c4 = lane; c4 < C4; c4 += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!bad) {
            for (int iy = 0; iy < g.gh; ++iy) {
                const Samp sy = axis_sample(g.y1f, g.bh, ph, iy, g.gh, H);
                if (sy.lo < 0) continue;
                const float *rl = xb + (size_t)sy.lo * W * C + c4 * 4;
                const float *rh = xb + (size_t)sy.hi * W * C + c4 * 4;
                for (int ix = 0; ix < g.gw; ++ix) {
                    const Samp sx = axis_sample(g.x1f, g.bw, pw, ix, g.gw, W);
                    if (sx.lo < 0) continue;
                    const float w1 = sy.wl * sx.wl, w2 = sy.wl * sx.wh;
                    const float w3 = sy.wh * sx.wl, w4 = sy.wh * sx.wh;
                    const float4 f1 = ld4(rl + (size_t)sx.lo * C), f2 = ld4(rl + (size_t)sx.hi * C);
                    const float4 f3 = ld4(rh + (size_t)sx.lo * C), f4 = ld4(rh + (size_t)sx.hi * C);
                    // ((w1*f1 + w2*f2) + w3*f3) + w4*f4, no contraction: matches the oracle bit for bit
                    acc.x += ((w1 * f1.x + w2 * f2.x) + w3 * f3.x) + w4 * f4.x;
                    acc.y += ((w1 * f1.y + w2 * f2.y) + w3 * f3.y) + w4 * f4.y;
                    acc.z += ((w1 * f1.z + w2 * f2.z) + w3 * f3.z) + w4 * f4.z;
                    acc.w += ((w1 * f1.w + w2 * f2.w) + w3 * f3.w) + w4 * f4.w;
                }
            }
        }
        float4 o = make_float4(acc.x / cnt, acc.y / cnt, acc.z / cnt, acc.w / cnt);
        *reinterpret_cast<float4 *>(yo + c4 * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------
// Forward: MAP ORDER.  The caller's RoIs arrive in sampling order (random in space), so the taps of co-resident waves are
// spread over the whole map and a line is fetched again by other XCDs' L2s (3.4x the map on configs[1]).  This kernel
// computes a permutation that walks the RoIs by (level, image, 64-row band of the box centre, centre column):
// perm[rank] = RoI index (measured on configs[1], ranking + forward: caller order 33.3 us; 8 / 16 / 32 / 64 / 128-row bands
// 32.5 / 31.6 / 29.5 / 28.5 / 28.6 us; Z order of 4..32-cell blocks 29.1-29.7; columns only 29.8; rows only 32.5).  The forward kernel processes RoI perm[i] at position i and writes output row perm[i], so the
// result is the same bits in the same place.  Rank by counting (every thread compares its key with all R keys in LDS,
// ties by index): R <= MAP_ORDER_MAX_R, sixteen lanes per RoI, any number of workgroups.
// ------------------------------------------------------------------------------------------
constexpr int MAP_ORDER_MAX_R = 8192;
int g_fwd_map_order = 1;
__global__ __launch_bounds__(256) void k_roi_map_order(Levels lv, const float *__restrict__ rois, const int32_t *__restrict__ levels,
                                                       int R, int32_t *__restrict__ perm) {
    extern __shared__ unsigned mo_keys[];
    for (int r = threadIdx.x; r < R; r += 256) {
        int l = levels ? levels[r] : 0;
        l = min(max(l, 0), lv.L - 1);
        const float *roi = rois + (size_t)r * 5;
        const float s = lv.scale[l];
        const int n = min(max((int)roi[0], 0), 63);
        const float cy = 0.5f * (roi[2] + roi[4]) * s, cx = 0.5f * (roi[1] + roi[3]) * s;
        const int iy = min(max((int)cy, 0), min(lv.H[l], 4096) - 1), ix = min(max((int)cx, 0), min(lv.W[l], 4096) - 1);
        const unsigned pos = ((unsigned)min(iy >> 6, 255) << 12) | (unsigned)ix;
        mo_keys[r] = ((unsigned)l << 26) | ((unsigned)n << 20) | pos;
    }
    __syncthreads();
    // 16 lanes per RoI: lane q counts the keys j = q, q + 16, ... that sort before the RoI's (ties: lower index first)
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
    const int rc = min(r, R - 1);
    const unsigned k = mo_keys[rc];
    int cnt = 0;
#pragma unroll 8
    for (int j = q; j < R; j += 16) {
        const unsigned kj = mo_keys[j];
        cnt += (kj < k || (kj == k && j < rc)) ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 1); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 4); cnt += __shfl_xor(cnt, 8);
    if (r < R && q == 0) perm[cnt] = r;
}

// ------------------------------------------------------------------------------------------
// Forward, NHWC, one wave per (roi, ph) ROW of bins (the default for sampling grids with PW * grid <= 64).
// The per-bin kernel above spends ~550 VALU instructions per bin, most of them the coordinate arithmetic that all 64
// lanes repeat; it is VALU-bound at a quarter of the HBM rate.  Here the geometry is done once per row: lane t computes
// the x sample (pw = t / gw, ix = t % gw) with the same axis_sample() (bit-identical indices and weights), the y samples
// are wave-uniform, and the tap loop reads cell offsets and weights from SGPRs (v_readlane): a tap is one buffer load with
// a scalar offset + the bit-exact ((w1 f1 + w2 f2) + w3 f3) + w4 f4 update.  Workgroup -> row order is XCD-banded so
// that the rows of one RoI (which share their taps' cache lines) run on one XCD's L2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_roi_align_fwd_rows(Levels lv, const float *__restrict__ rois,
                                                            const int32_t *__restrict__ levels, int R, int N, int C, int PH, int PW,
                                                            int sr, float *__restrict__ y, int chunk,
                                                            const int32_t *__restrict__ perm) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x;
    const int wg = (bid & 7) * chunk + (bid >> 3);
    if ((int)(bid >> 3) >= chunk) return;
    const int pos_id = wg * 4 + wave;                  // (position in processing order, ph)
    if (pos_id >= R * PH) return;
    int r, ph;
    divmod_u24(pos_id, PH, r, ph);
    if (perm) r = __builtin_amdgcn_readfirstlane(perm[r]);      // map order: process RoI perm[i], write ITS rows
    const int row_id = r * PH + ph;
    int l = levels ? levels[r] : 0;
    l = __builtin_amdgcn_readfirstlane(min(max(l, 0), lv.L - 1));
    const int H = lv.H[l], W = lv.W[l];
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, lv.scale[l], PH, PW, sr);
    const int gh = __builtin_amdgcn_readfirstlane(g.gh), gw = __builtin_amdgcn_readfirstlane(g.gw);
    const int gn = __builtin_amdgcn_readfirstlane(g.n);
    const bool bad = gn < 0 || gn >= N;
    // x samples: lane t = pw * gw + ix (host guarantees PW * gw <= 64 for fixed grids; adaptive grids are checked here)
    const int tpw = lane / max(gw, 1), tix = lane - tpw * max(gw, 1);
    const Samp sxl = axis_sample(g.x1f, g.bw, min(tpw, PW - 1), tix, gw, W);
    const int cnt_i = gh * gw;
    const float cnt = (float)cnt_i;
    const bool pow2 = (cnt_i & (cnt_i - 1)) == 0;
    const float inv = 1.0f / cnt;                      // exact when cnt is a power of two: x * inv == x / cnt bit for bit
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)(lv.x[l] + (size_t)(bad ? 0 : gn) * H * W * C), 0,
                                                        (unsigned)((size_t)H * W * C * 4), 0x00020000);
    float *yo = y + (size_t)row_id * PW * C;
    const unsigned OOB = 0xFFFFFFFFu;
    if (gh == 2 && gw == 2) {
        // The common 2x2 grid, branch-free and software-pipelined: the 16 taps of bin pw+1 are in flight while bin pw is
        // reduced (measured 33.4 us on configs[1] against 36.2 us single-buffered at twice the occupancy).  A void sample has weight 0 and an out-of-range offset (the load returns 0): adding its exact +0 is
        // the same as skipping it, so the result stays bit-identical to the reference loop.
        const Samp sy0 = axis_sample(g.y1f, g.bh, ph, 0, 2, H), sy1 = axis_sample(g.y1f, g.bh, ph, 1, 2, H);
        int yl[2] = {__builtin_amdgcn_readfirstlane(sy0.lo), __builtin_amdgcn_readfirstlane(sy1.lo)};
        int yh[2] = {__builtin_amdgcn_readfirstlane(sy0.hi), __builtin_amdgcn_readfirstlane(sy1.hi)};
        const float wyl[2] = {sgpr_f(sy0.wl), sgpr_f(sy1.wl)}, wyh[2] = {sgpr_f(sy0.wh), sgpr_f(sy1.wh)};
        const unsigned cell = (unsigned)C * 4u;
        for (int cb = 0; cb < C; cb += 256) {
            const bool act = cb + lane * 4 < C;
            const unsigned vlane = (act && !bad) ? (unsigned)(cb + lane * 4) * 4u : OOB;
            float4 tapA[16], tapB[16];
            auto issue = [&](int pw, float4 (&tap)[16]) {
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const int t = min(pw, PW - 1) * 2 + ix;
                        const int xlo = __builtin_amdgcn_readlane(sxl.lo, t), xhi = __builtin_amdgcn_readlane(sxl.hi, t);
                        const bool ok = xlo >= 0 && yl[iy] >= 0 && pw < PW;
                        const unsigned vo = ok ? vlane : OOB;
                        const unsigned rlo = (unsigned)(max(yl[iy], 0) * W) * cell, rhi = (unsigned)(max(yh[iy], 0) * W) * cell;
                        const unsigned clo = (unsigned)max(xlo, 0) * cell, chi = (unsigned)max(xhi, 0) * cell;
                        const int b = (iy * 2 + ix) * 4;
                        { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, rlo + clo, 0); __builtin_memcpy(&tap[b + 0], &v, 16); }
                        { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, rlo + chi, 0); __builtin_memcpy(&tap[b + 1], &v, 16); }
                        { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, rhi + clo, 0); __builtin_memcpy(&tap[b + 2], &v, 16); }
                        { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, rhi + chi, 0); __builtin_memcpy(&tap[b + 3], &v, 16); }
                    }
            };
            auto reduce = [&](int pw, const float4 (&tap)[16]) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const int t = pw * 2 + ix;
                        const float wxl = readlane_f(sxl.wl, t), wxh = readlane_f(sxl.wh, t);
                        const float w1 = wyl[iy] * wxl, w2 = wyl[iy] * wxh, w3 = wyh[iy] * wxl, w4 = wyh[iy] * wxh;
                        const int b = (iy * 2 + ix) * 4;
                        const float4 f1 = tap[b], f2 = tap[b + 1], f3 = tap[b + 2], f4 = tap[b + 3];
                        acc.x += ((w1 * f1.x + w2 * f2.x) + w3 * f3.x) + w4 * f4.x;
                        acc.y += ((w1 * f1.y + w2 * f2.y) + w3 * f3.y) + w4 * f4.y;
                        acc.z += ((w1 * f1.z + w2 * f2.z) + w3 * f3.z) + w4 * f4.z;
                        acc.w += ((w1 * f1.w + w2 * f2.w) + w3 * f3.w) + w4 * f4.w;
                    }
                const float4 o = make_float4(acc.x * 0.25f, acc.y * 0.25f, acc.z * 0.25f, acc.w * 0.25f);    // == acc / 4 exactly
                if (act) *reinterpret_cast<float4 *>(yo + (size_t)pw * C + cb + lane * 4) = o;
            };
            issue(0, tapA);
#pragma nounroll
            for (int pw = 0; pw < PW; pw += 2) {
                issue(pw + 1, tapB);
                reduce(pw, tapA);
                if (pw + 1 < PW) {
                    issue(pw + 2, tapA);
                    reduce(pw + 1, tapB);
                }
            }
        }
        return;
    }
    for (int cb = 0; cb < C; cb += 256) {
        const bool act = cb + lane * 4 < C;
        const unsigned vlane = (act && !bad) ? (unsigned)(cb + lane * 4) * 4u : OOB;
        for (int pw = 0; pw < PW; ++pw) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int iy = 0; iy < gh; ++iy) {
                const Samp sy = axis_sample(g.y1f, g.bh, ph, iy, gh, H);
                const int ylo = __builtin_amdgcn_readfirstlane(sy.lo), yhi = __builtin_amdgcn_readfirstlane(sy.hi);
                if (ylo < 0) continue;
                const float wyl = sgpr_f(sy.wl), wyh = sgpr_f(sy.wh);
                const unsigned rlo = (unsigned)(ylo * W) * (unsigned)C * 4u, rhi = (unsigned)(yhi * W) * (unsigned)C * 4u;
                for (int ix = 0; ix < gw; ++ix) {
                    const int t = pw * gw + ix;
                    const int xlo = __builtin_amdgcn_readlane(sxl.lo, t), xhi = __builtin_amdgcn_readlane(sxl.hi, t);
                    if (xlo < 0) continue;
                    const float wxl = readlane_f(sxl.wl, t), wxh = readlane_f(sxl.wh, t);
                    const unsigned clo = (unsigned)xlo * (unsigned)C * 4u, chi = (unsigned)xhi * (unsigned)C * 4u;
                    float4 f1, f2, f3, f4;
                    { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vlane, rlo + clo, 0); __builtin_memcpy(&f1, &v, 16); }
                    { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vlane, rlo + chi, 0); __builtin_memcpy(&f2, &v, 16); }
                    { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vlane, rhi + clo, 0); __builtin_memcpy(&f3, &v, 16); }
                    { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vlane, rhi + chi, 0); __builtin_memcpy(&f4, &v, 16); }
                    const float w1 = wyl * wxl, w2 = wyl * wxh, w3 = wyh * wxl, w4 = wyh * wxh;
                    // ((w1*f1 + w2*f2) + w3*f3) + w4*f4, no contraction: matches the oracle bit for bit
                    acc.x += ((w1 * f1.x + w2 * f2.x) + w3 * f3.x) + w4 * f4.x;
                    acc.y += ((w1 * f1.y + w2 * f2.y) + w3 * f3.y) + w4 * f4.y;
                    acc.z += ((w1 * f1.z + w2 * f2.z) + w3 * f3.z) + w4 * f4.z;
                    acc.w += ((w1 * f1.w + w2 * f2.w) + w3 * f3.w) + w4 * f4.w;
                }
            }
            float4 o;
            if (pow2) o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
            else o = make_float4(acc.x / cnt, acc.y / cnt, acc.z / cnt, acc.w / cnt);
            if (act) *reinterpret_cast<float4 *>(yo + (size_t)pw * C + cb + lane * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward, NHWC, owner-computes tiles.
// ------------------------------------------------------------------------------------------
constexpr int SLOTS = 32;         // RoIs whose weight tables are resident in LDS per round
constexpr int LISTCAP = 512;      // RoIs scanned per segment
constexpr int QCAP_MAX = 256;     // per-wave queue capacity (template QC): 64 entries for 7x7 pooling, 256 for 14x14
                                  // (a full window is rebuilt per pass; entry = (gy row, 4 row weights, 4 col weights))


template <int QC>
struct PatchQueue {     // one per wave, in LDS
    float4 wy[QC];      // weights of the entry's bin row on the patch's 4 map rows (already / count)
    float4 wx[QC];      // weights of the entry's bin column on the patch's 4 map columns
    int row[QC];        // gy row index (r*PH + ph)*PW + pw
};

// acc[i][k] += wy[i] * wx[k] * g for the 4x4 patch.  Deliberately branch-free: conditional updates of
// the 64 accumulator registers make hipcc copy them around every branch.
__device__ __forceinline__ void apply_entry(float4 (&acc)[PT][PT], const float4 wy, const float4 wx,
                                            const float4 g) {
    const float4 t0 = make_float4(wx.x * g.x, wx.x * g.y, wx.x * g.z, wx.x * g.w);
    const float4 t1 = make_float4(wx.y * g.x, wx.y * g.y, wx.y * g.z, wx.y * g.w);
    const float4 t2 = make_float4(wx.z * g.x, wx.z * g.y, wx.z * g.z, wx.z * g.w);
    const float4 t3 = make_float4(wx.w * g.x, wx.w * g.y, wx.w * g.z, wx.w * g.w);
    const float wyv[PT] = {wy.x, wy.y, wy.z, wy.w};
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        MRCNN_FMA4(acc[i][0], wyv[i], t0) MRCNN_FMA4(acc[i][1], wyv[i], t1)
        MRCNN_FMA4(acc[i][2], wyv[i], t2) MRCNN_FMA4(acc[i][3], wyv[i], t3)
    }
}

// Stream the wave's queue: every entry is one coalesced 1-KiB gy row (lane = 4 channels) applied to
// up to 16 cells.  Entries are handled two at a time and the next two loads are issued before the
// current two are consumed.  Entries past n get weight 0 on a live row, so the loop is branch-free
// with respect to the accumulators.  The empty asm statements only stop hipcc from hoisting every
// LDS weight read to the top of the loop (register pressure).
template <int QC>
__device__ __forceinline__ void drain_queue(const PatchQueue<QC> &q, int n, const float *__restrict__ gyl, int C,
                                            float4 (&acc)[PT][PT]) {
    if (n <= 0) return;
    const size_t Cs = (size_t)C;
    const int r0 = q.row[0];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a0 = ld4(gyl + (size_t)r0 * Cs);
    float4 a1 = ld4(gyl + (size_t)(1 < n ? q.row[1] : r0) * Cs);
#pragma nounroll
    for (int j = 0; j < n; j += 2) {
        const float4 b0 = ld4(gyl + (size_t)(j + 2 < n ? q.row[j + 2] : r0) * Cs);
        const float4 b1 = ld4(gyl + (size_t)(j + 3 < n ? q.row[j + 3] : r0) * Cs);
        apply_entry(acc, q.wy[j], q.wx[j], a0);
        asm volatile("" ::: "memory");
        apply_entry(acc, j + 1 < n ? q.wy[j + 1] : z, j + 1 < n ? q.wx[j + 1] : z, a1);
        asm volatile("" ::: "memory");
        a0 = b0;
        a1 = b1;
    }
}

template <int PBT>   // bins per axis held in LDS: 8 (7x7 pooling) or 16 (14x14)
__global__ __launch_bounds__(BWD_THREADS, 3) void k_roi_align_bwd_nhwc(Levels lv, const float *__restrict__ gy,
                                                                       const float *__restrict__ rois,
                                                                       const int32_t *__restrict__ levels,
                                                                       int R, int N, int C, int PH, int PW,
                                                                       int sr, int chunk, int accumulate) {
    // sW[slot][axis][bin][tile row/col]: summed weight that bin `bin` of the RoI in `slot` puts on
    // map row ty0+row (axis 0) / map column tx0+row (axis 1).  4 consecutive rows = one 16-B read.
    constexpr int QCAP = PBT == 8 ? 64 : QCAP_MAX;
    __shared__ __attribute__((aligned(16))) float sW[SLOTS][2][PBT][TH];
    __shared__ __attribute__((aligned(16))) PatchQueue<QCAP> sQ[BWD_WAVES];
    __shared__ float4 sGeom[LISTCAP];        // (x1f, y1f, bw, bh) of listed RoIs
    __shared__ int sList[LISTCAP];
    __shared__ int sMask[SLOTS];             // bits 0-7: rows with weight, bits 8-15: cols
    __shared__ int sWaveCnt[BWD_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so XCD k
    // takes the contiguous band [k*chunk, (k+1)*chunk) of the row-major tile list: neighbouring
    // tiles - which read the same RoIs' gy rows - hit the same 4-MiB L2.  Speed only.
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || tile_id >= lv.tile_begin[lv.L]) return;
    int l = 0;
    while (l + 1 < lv.L && tile_id >= lv.tile_begin[l + 1]) ++l;
    int t = tile_id - lv.tile_begin[l];
    const int nsplit = lv.split[l];
    const int zsplit = t % nsplit;               // adjacent workgroups = the splits of one tile (same gy rows, same L2)
    t /= nsplit;
    const int per_img = lv.tiles_x[l] * lv.tiles_y[l];
    const int n = t / per_img;
    t -= n * per_img;
    const int ty0 = (t / lv.tiles_x[l]) * TH, tx0 = (t % lv.tiles_x[l]) * TW;
    const int H = lv.H[l], W = lv.W[l];
    const float scale = lv.scale[l];
    float *gxb = (nsplit > 1 ? lv.slab[l] + (size_t)zsplit * N * H * W * C : lv.gx[l]) + (size_t)n * H * W * C;
    if (nsplit > 1) accumulate = 0;              // partial maps are always written whole
    const float inv_cnt = 1.0f / (float)(sr * sr);
    PatchQueue<QCAP> &q = sQ[wave];
    const int cy0 = (wave >> 1) * PT, cx0 = (wave & 1) * PT;      // this wave's patch inside the tile
    const int nrow = min(PT, H - (ty0 + cy0)), ncol = min(PT, W - (tx0 + cx0));   // may be <= 0

    int round = 0;
    for (int seg = 0; seg == 0 || seg < R; seg += LISTCAP) {
        const int seg_end = min(R, seg + LISTCAP);
        // ---- phase 0: ordered list (+ geometry) of the RoIs of this (level, image) whose
        //      footprint may touch the tile.
        int n_list = 0;
        for (int base = seg; base < seg_end; base += BWD_THREADS) {
            bool f = false;
            float4 geo = make_float4(0.f, 0.f, 0.f, 0.f);
            const int i = base + tid;
            if (i < seg_end) {
                int li = levels ? levels[i] : 0;
                li = min(max(li, 0), lv.L - 1);
                const float *roi = rois + (size_t)i * 5;
                if (li == l && (int)roi[0] == n && (nsplit == 1 || i % nsplit == zsplit)) {
                    const RoiGeom g = roi_geom(roi, scale, PH, PW, sr);
                    f = (g.y1f - 2.0f < (float)(ty0 + TH)) && (g.y1f + g.rh + 2.0f > (float)ty0) &&
                        (g.x1f - 2.0f < (float)(tx0 + TW)) && (g.x1f + g.rw + 2.0f > (float)tx0);
                    geo = make_float4(g.x1f, g.y1f, g.bw, g.bh);
                }
            }
            const unsigned long long bal = __ballot(f);
            if (lane == 0) sWaveCnt[wave] = __popcll(bal);
            __syncthreads();
            int off = n_list;
            for (int w = 0; w < BWD_WAVES; ++w) {
                if (w < wave) off += sWaveCnt[w];
                n_list += sWaveCnt[w];
            }
            if (f) {
                const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
                sList[pos] = i;
                sGeom[pos] = geo;
            }
            __syncthreads();
        }

        for (int s0 = 0; s0 < n_list || round == 0; s0 += SLOTS) {
            const int nslots = max(0, min(SLOTS, n_list - s0));
            // ---- phase 1: per-(slot, axis, bin, row) summed weights, one thread each
            if (tid < SLOTS) sMask[tid] = 0;
            __syncthreads();
            // one thread per (slot, axis, bin): the bin's `sr` samples are evaluated once and their weights summed
            // into the bin's 8 tile rows (the thread owns that LDS row: no atomics, fixed order)
            for (int task = tid; task < nslots * 2 * PBT; task += BWD_THREADS) {
                const int slot = task / (2 * PBT), qq = task % (2 * PBT);
                const int axis = qq / PBT, bin = qq % PBT;
                const float4 ge = sGeom[s0 + slot];
                const int P = axis ? PW : PH, size = axis ? W : H, t0 = axis ? tx0 : ty0;
                const float start = axis ? ge.x : ge.y, bsz = axis ? ge.z : ge.w;
                float *tab = &sW[slot][axis][bin][0];
#pragma unroll
                for (int j = 0; j < TH; ++j) tab[j] = 0.0f;
                int bits = 0;
                if (bin < P) {
                    for (int i2 = 0; i2 < sr; ++i2) {
                        const Samp sp = axis_sample(start, bsz, bin, i2, sr, size);
                        const unsigned dl = (unsigned)(sp.lo - t0), dh = (unsigned)(sp.hi - t0);
                        if (sp.lo >= 0 && dl < (unsigned)TH) tab[dl] += sp.wl;
                        if (sp.hi >= 0 && dh < (unsigned)TH) tab[dh] += sp.wh;
                    }
#pragma unroll
                    for (int j = 0; j < TH; ++j) bits |= (tab[j] != 0.0f) ? (1 << j) : 0;
                }
                if (bits) atomicOr(&sMask[slot], bits << (axis * 8));
            }
            __syncthreads();

            // ---- phase 2: wave = one 4x4 patch.  2a (lane-parallel queue build): lane = bin (ph,pw)
            //      of a listed RoI; bins with weight on the patch's rows AND columns append
            //      (gy row, 4 row weights, 4 column weights) at a ballot-derived rank => deterministic
            //      order.  2b: drain_queue.  Queue overflow => windowed passes (gx read-modify-write
            //      by its owner wave).
            if (nrow > 0 && ncol > 0) {
                const int m = lane < nslots ? sMask[lane] : 0;
                const unsigned rel = (unsigned)__ballot(((m >> cy0) & 0xF) && ((m >> (8 + cx0)) & 0xF));
                const int roi_of_lane = lane < nslots ? sList[s0 + lane] : 0;
                float *dst = gxb + ((size_t)(ty0 + cy0) * W + tx0 + cx0) * C;
                int pass = 0, cnt;
                do {
                    cnt = 0;      // wave-uniform: entries seen so far
                    const int win = pass * QCAP;
                    for (unsigned rm = rel; rm; rm &= rm - 1) {
                        const int slot = __builtin_ctz(rm);
                        const int r = __builtin_amdgcn_readlane(roi_of_lane, slot);
#pragma unroll
                        for (int ch = 0; ch < (PBT * PBT) / 64; ++ch) {
                            const int ph = (PBT == 8) ? (lane >> 3) : (ch * 4 + (lane >> 4));
                            const int pw = (PBT == 8) ? (lane & 7) : (lane & 15);
                            float4 wy = *reinterpret_cast<const float4 *>(&sW[slot][0][ph][cy0]);
                            const float4 wx = *reinterpret_cast<const float4 *>(&sW[slot][1][pw][cx0]);
                            const bool nzl = (wy.x != 0.f || wy.y != 0.f || wy.z != 0.f || wy.w != 0.f) &&
                                             (wx.x != 0.f || wx.y != 0.f || wx.z != 0.f || wx.w != 0.f);
                            const unsigned long long bal = __ballot(nzl);
                            const int idx = cnt + __popcll(bal & ((1ull << lane) - 1ull)) - win;
                            if (nzl && (unsigned)idx < (unsigned)QCAP) {
                                wy.x *= inv_cnt; wy.y *= inv_cnt; wy.z *= inv_cnt; wy.w *= inv_cnt;
                                q.wy[idx] = wy;
                                q.wx[idx] = wx;
                                q.row[idx] = (r * PH + ph) * PW + pw;
                            }
                            cnt += __popcll(bal);
                        }
                    }
                    const bool first = (round == 0 && pass == 0) && !accumulate;
                    const int nq = max(0, min(QCAP, cnt - win));
#pragma nounroll
                    for (int cb = 0; cb < C; cb += CCH) {
                        const bool act = cb + lane * 4 < C;          // C < 256 (or a tail): idle lanes
                        const int lo = act ? cb + lane * 4 : 0;
                        float4 acc[PT][PT];
#pragma unroll
                        for (int i = 0; i < PT; ++i)
#pragma unroll
                            for (int k = 0; k < PT; ++k) {
                                acc[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (!first && act && i < nrow && k < ncol) acc[i][k] = ld4(dst + ((size_t)i * W + k) * C + lo);
                            }
                        drain_queue(q, nq, gy + lo, C, acc);
                        // Opaque zero: keeps hipcc from materialising the 16 store addresses (32
                        // VGPRs) before the drain loop and holding them live across it.
                        int opq;
                        asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
                        float *dst2 = dst + lo + opq;
#pragma unroll
                        for (int i = 0; i < PT; ++i)
#pragma unroll
                            for (int k = 0; k < PT; ++k)
                                if (act && i < nrow && k < ncol)
                                    *reinterpret_cast<float4 *>(dst2 + ((size_t)i * W + k) * C) = acc[i][k];
                    }
                    ++pass;
                } while (cnt > pass * QCAP);
            }
            ++round;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward, NHWC, owner-computes PATCHES with fully independent waves (the default; the tile kernel above is kept as
// variant 1 for A/B).  One wave owns a 4x4 patch of gradient-map cells x 256 channels for the whole kernel: 64
// accumulator VGPRs, no workgroup barrier anywhere, every phase is private to the wave (wave-private LDS regions), so
// the 16 waves of a CU are in different phases at any time and the memory pipe never idles behind a barrier:
//   1. scan: lanes test 64 RoIs at a time against the patch (4 x 64 RoIs of loads in flight), candidates are compacted
//      in ascending RoI order into the wave's LDS list (geometry + index);
//   2. per candidate, lanes = (axis, bin) compute the summed sample weights of that bin on the patch's 4 rows / 4 columns
//      ONCE (bilinear weights are separable) with the forward's axis_sample() (bit-identical indices), ballots give the
//      non-empty bins; lanes = (ph, pw) then append (gy row, wy[4], wx[4]) entries for the bin pairs that touch the patch;
//   3. drain: entries are wave-uniform, so a row of the patch whose weight is zero is skipped with a scalar branch
//      (a bin touches ~2 of the 4 rows); gy rows stream with 2*DEPTH loads in flight;
//   4. the 16 cells are written once at the end (zeros where nothing lands; += old value when accumulating).
// Sums are in fixed (RoI, ph, pw) order: bit-reproducible.
// ------------------------------------------------------------------------------------------
constexpr int W2_DEPTH = 2;       // 2 * W2_DEPTH gy rows in flight per wave during a drain
constexpr int W2_SEG = 512;       // RoIs per segment: their boxes and bin geometry sit in the workgroup's LDS table
constexpr int PRIO_CAND[3] = {10, 14, 18};      // candidate counts from which a wave runs at priority 1 / 2 / 3

template <int QC>
struct WaveQueue {
    float4 wy[QC], wx[QC];
    int row[QC];
};

template <int PBT>
struct WaveLds {
    static constexpr int QC = PBT == 8 ? 112 : 96;      // the queue is drained when it cannot take another 64-entry unit
    WaveQueue<QC> q;
    unsigned short idx[W2_SEG];                // candidates of the segment (positions in the segment table), ascending
    float4 tab[(64 / (2 * PBT))][2][PBT];      // one table pass: (slot, axis, bin) -> 4 weights
};

// gy rows are fetched with buffer loads: the row offset is wave-uniform (an SGPR soffset), the per-lane part (lane * 16 B)
// is one constant VGPR, so the 2*DEPTH loads in flight cost no address registers.  The queue itself is read from LDS
// ONCE per 64 entries with lane = entry (9 VGPRs); entry j's row and weights then come from v_readlane - no LDS round
// trip inside the entry loop.
__device__ __forceinline__ float readlane_fs(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <int QC, int DEPTH>
__device__ __forceinline__ void drain_wave_queue(const WaveQueue<QC> &q, int n, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned row_bytes,
                                                 float4 (&acc)[PT][PT], int lane) {
#pragma nounroll
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int m = min(64, n - e0);
        const int qi = min(e0 + lane, QC - 1);
        const int qrow = q.row[qi];
        const float4 qwy = q.wy[qi], qwx = q.wx[qi];
        auto ldrow = [&](int j) -> float4 {
            const unsigned so = (unsigned)__builtin_amdgcn_readlane(qrow, j < m ? j : 0) * row_bytes;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
            float4 f;
            __builtin_memcpy(&f, &v, 16);
            return f;
        };
        float4 buf[2 * DEPTH];
        // The first loads must ISSUE in slot order: slot d is consumed with 2*DEPTH-1 younger loads outstanding (vmcnt counts
        // in issue order).  Left alone the scheduler issued them in reverse, the waitcnt pass then had to assume slot 0 is the
        // youngest and put s_waitcnt vmcnt(0) in front of EVERY entry - no load of the wave ever overlapped its own FMAs.
#pragma unroll
        for (int d = 0; d < 2 * DEPTH; ++d) {
            buf[d] = ldrow(d);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma nounroll
        for (int j = 0; j < m; j += 2 * DEPTH) {
#pragma unroll
            for (int d = 0; d < 2 * DEPTH; ++d) {
                if (j + d < m) {
                    const int e = j + d;
                    // weights are >= 0, so "== 0.0f" is an integer test of the SGPR (scalar ALU, not a VALU compare)
                    const int wyb[PT] = {__builtin_amdgcn_readlane(__float_as_int(qwy.x), e), __builtin_amdgcn_readlane(__float_as_int(qwy.y), e),
                                         __builtin_amdgcn_readlane(__float_as_int(qwy.z), e), __builtin_amdgcn_readlane(__float_as_int(qwy.w), e)};
                    const float wx[PT] = {readlane_fs(qwx.x, e), readlane_fs(qwx.y, e), readlane_fs(qwx.z, e), readlane_fs(qwx.w, e)};
#pragma unroll
                    for (int i = 0; i < PT; ++i) {
                        if (wyb[i] != 0) {               // wave-uniform: scalar compare + branch
                            const float wyi = __int_as_float(wyb[i]);
                            // (gy * wy) * wx: the row weight is applied once to the gy row (2 packed multiplies), each of the
                            // 4 cells then takes 2 packed FMAs with its scalar column weight - 10 VALU per row, not 4 x (1 + 2) + 1
                            const float4 t = make_float4(buf[d].x * wyi, buf[d].y * wyi, buf[d].z * wyi, buf[d].w * wyi);
#pragma unroll
                            for (int k = 0; k < PT; ++k) { MRCNN_FMA4(acc[i][k], wx[k], t) }
                        }
                    }
                }
                buf[d] = ldrow(j + d + 2 * DEPTH);          // refill this slot for the next round
                asm volatile("" ::: "memory");
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// (r5) The backward in two kernels: ENTRY LISTS built ahead of time, a lean streaming backward.
// The wave kernel below spends two thirds of a wave's life deriving WHICH (RoI, bin) pairs land on its patch (RoI loads, segment table,
// scan, table passes, queue appends: ~17 k of the median 31 k cycles) and the kernel ends with its slowest wave; none of that depends
// on gy.  In a training step the RoIs of the backward are known a whole head earlier (fpn_roi_mask_head.py:59-61,75-77 run the
// forward with them), so the geometry half runs THEN, beside the forward, as its own launch:
//   MODE 1 (plan builder): the wave kernel's own code path up to the queue - a full queue is written to a node of the caller's plan
//          buffer instead of being drained (first node of a patch at a fixed place, further ones from a pool by one integer atomic per
//          flush; the nodes of a patch are chained in flush order = (RoI, ph, pw) order).  A tile whose patch finds the pool empty is
//          FLAGGED in the plan;
//   k_roi_align_bwd_lean: a patch is TWO waves (128 channels each, lane = 2 channels: 32 accumulator registers leave room for 16 gy
//          rows in flight - the fused kernel's wave has 4 - and a heavy patch's rows are fetched by two waves at once): node arrays are
//          read with lane = entry (coalesced), entry j's row offset and weights come from v_readlane, cells are written once.  Entry
//          order, weights and every FMA are the fused path's: identical bits.
//   MODE 2 (behind the lean kernel, same stream, unless the caller has VERIFIED the plan - mrcnn_roi_align_bwd_plan_status): the fused
//          path for the flagged tiles only, or for every tile when the plan's header does not hold (foreign buffer, other geometry);
//          every other workgroup leaves at its first instruction - but dispatching ~850 of them costs 3.5 - 5 us, hence the verified
//          form.  (Tried instead: that launch on a helper stream - the fork / join cost more than the launch; a slow path inside the lean
//          kernel - its registers spilled the hot loop, +4 us.)
// The plan buffer is the caller's and must stay unmodified between the builder and the backward.
// ------------------------------------------------------------------------------------------
constexpr int NP_MAGIC = 0x4E504C4E;
enum { NP_MAGIC_I = 0, NP_OVERFLOW, NP_EXTRA, NP_UNITS, NP_CAP, NP_R, NP_PH, NP_PW, NP_SR, NP_N, NP_QC, NP_L, NP_NODES_OFF, NP_HW /* 2 per level */,
       NP_HDR_INTS = 64 };
int g_lean_dbg = 0;
unsigned long long *g_lean_stamps = nullptr;      // measurement (mrcnn_debug_roi_align_lean_stamps)
int g_lean_variant = 8;           // gy rows in flight per wave of the lean kernel / waves per SIMD / threads per workgroup: 0 = 10 / 8 / 512, 1 = 16 / 7 / 512,
                                  // 2 = 8 / 8 / 512, 8 = 8 / 8 / 256 (default), 9 = 8 / 8 / 128 (mrcnn_debug_roi_align_lean_variant)
constexpr int LEAN_CH = 128;      // channels per wave there: lane = 2 channels

template <int QC>
struct PlanNode {
    int count, next, pad[14];        // 64-byte header: groups of four entries' weights are 64-byte aligned (one s_load_dwordx16 each)
    WaveQueue<QC> q;
};
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// a wave-uniform read through the scalar cache (the plan was written by an earlier launch: constant for this kernel)
template <typename T> __device__ __forceinline__ T scalar_load(const void *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}
// (a stride that is not a multiple of 2 KiB - 4352 instead of the 7 x 7 node's 4096 - was tried in round 5 against HBM channel conflicts on the
// first touch of the 3400 lists: no change, 19.9 against 19.8 us)
template <int QC> constexpr size_t plan_node_stride() { return (sizeof(PlanNode<QC>) + 255) / 256 * 256; }
// [header][tile flags, rounded to 64 ints][nodes]
inline size_t plan_nodes_off_ints(int total_tiles) { return NP_HDR_INTS + (size_t)(total_tiles + 63) / 64 * 64; }

// (every field the builder wrote is compared: the level count and each level's H x W as well - two pyramids can have equal tile counts -
// and the builder's node capacity against what THIS call's plan_bytes can hold, so that no `next` index points past the caller's buffer)
__device__ __forceinline__ bool plan_header_holds(const int *__restrict__ nplan, const Levels &lv, int total_units, int R, int N, int PH, int PW,
                                                  int sr, int QC, int nodes_off, int node_cap) {
    bool ok = nplan[NP_MAGIC_I] == NP_MAGIC && nplan[NP_UNITS] == total_units && nplan[NP_R] == R && nplan[NP_PH] == PH &&
              nplan[NP_PW] == PW && nplan[NP_SR] == sr && nplan[NP_N] == N && nplan[NP_QC] == QC && nplan[NP_NODES_OFF] == nodes_off &&
              nplan[NP_L] == lv.L && nplan[NP_CAP] <= node_cap;
    for (int q = 0; q < lv.L; ++q) ok = ok && nplan[NP_HW + 2 * q] == lv.H[q] && nplan[NP_HW + 2 * q + 1] == lv.W[q];
    return __builtin_amdgcn_readfirstlane((int)ok) != 0;
}

// STAMP: diagnostic build only (mrcnn_debug_roi_align_bwd_stamps): s_memtime at the phase boundaries of every wave goes
// to `stamps` (12 x u64 per wave: start, scan done [last], build done [accumulated build cycles in slot 2], drain cycles
// [slot 3], before stores, end, s_memrealtime at start, at end); nothing is computed from them.
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int PBT, int DEPTH, bool STAMP = false, int MODE = 0>
__global__ __launch_bounds__(BWD_THREADS, 4) void k_roi_align_bwd_waves(Levels lv, const float *__restrict__ gy,
                                                                        const float *__restrict__ rois,
                                                                        const int32_t *__restrict__ levels, int R, int N, int C,
                                                                        int PH, int PW, int sr, int chunk, int accumulate,
                                                                        unsigned long long *__restrict__ stamps = nullptr,
                                                                        int nodes_off = 0, int node_cap = 0,       // MODE 1 / 2: offset of the plan's nodes (ints), node capacity
                                                                        int *__restrict__ nplan = nullptr) {
    unsigned long long st0 = 0, st_scan = 0, st_drain = 0, st_tmp = 0, st_rt0 = 0, st_pre = 0, st_tab = 0, st_cnt = 0;
    if (STAMP) { st0 = stamp_now(); st_rt0 = __builtin_amdgcn_s_memrealtime(); }
    using LDS = WaveLds<PBT>;
    constexpr int QC = LDS::QC;
    constexpr int SPP = 64 / (2 * PBT);          // RoIs (slots) per table pass: 4 (7x7 pooling) or 2 (14x14)
    constexpr int CH = (PBT * PBT) / 64;         // 64-bin chunks per RoI: 1 (7x7) or 4 (14x14)
    __shared__ __attribute__((aligned(16))) LDS lds_all[BWD_WAVES];
    // Segment table, shared by the 4 waves (the only cooperative part of the kernel: each wave fills a quarter, one
    // barrier): box = (y1f, y1f + rh + 1, x1f, x1f + rw + 1), empty for RoIs of another level / image / split;
    // geo = (x1f, y1f, bw, bh).  The geometry (two correctly-rounded divisions per RoI) is computed once per workgroup.
    // Every sample coordinate c of a RoI lies strictly inside (y1f, y1f + rh) (by >= rh / (2 PH grid), far more than
    // float rounding), it touches cells floor(c) and floor(c) + 1 (c < 0 clamps to cell 0), so the RoI can only touch
    // cell rows k with y1f - 1 < k <= y1f + rh + 1: a patch [py0, py0 + 4) is a candidate iff y1f < py0 + 4 and
    // y1f + rh + 1 >= py0.  Conservative (the exact weights decide), and half the candidates of a +-2 margin.
    __shared__ float4 sBox[W2_SEG], sGeo[W2_SEG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    LDS &lds = lds_all[wave];
    // The RoI rows of the first segment do not depend on the tile: their loads go out first and fly while the tile is
    // decoded from the kernel arguments (a chain of dependent scalar loads, ~2 us when every workgroup starts at once).
    const auto rs_roi = __builtin_amdgcn_make_buffer_rsrc((void *)rois, 0, (unsigned)R * 20u, 0x00020000);
    const auto rs_lvl = __builtin_amdgcn_make_buffer_rsrc((void *)levels, 0, levels ? (unsigned)R * 4u : 0u, 0x00020000);
    constexpr int FILL = W2_SEG / BWD_THREADS;
    float rv0[FILL][5];
    int lv0[FILL];
    auto load_rois = [&](int seg, float (&rv)[FILL][5], int (&lvv)[FILL]) {
#pragma unroll
        for (int h2 = 0; h2 < FILL; ++h2) {
            const int i = seg + tid + h2 * BWD_THREADS;
            const unsigned o = i < R ? (unsigned)i : 0x0FFFFFFFu;        // * 20 / * 4 stays out of range -> 0
            lvv[h2] = __builtin_amdgcn_raw_buffer_load_b32(rs_lvl, o * 4u, 0, 0);
#pragma unroll
            for (int k = 0; k < 5; ++k) rv[h2][k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_roi, o * 20u + 4u * k, 0, 0));
        }
    };
    if (MODE != 2) load_rois(0, rv0, lv0);          // (MODE 2: most workgroups leave before they need a RoI)
    int l = 0, nsplit = 1, zsplit = 0, n, py0, px0, tile_id = 0;
    {
        tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);                 // XCD-banded tile order (speed only)
        if ((int)(blockIdx.x >> 3) >= chunk || tile_id >= lv.tile_begin[lv.L]) return;      // whole workgroup
        while (l + 1 < lv.L && tile_id >= lv.tile_begin[l + 1]) ++l;
        int t = tile_id - lv.tile_begin[l];
        nsplit = lv.split[l];
        int tyi, txi;
        if (nsplit > 1) divmod_u24(t, nsplit, t, zsplit);
        divmod_u24(t, lv.tiles_x[l] * lv.tiles_y[l], n, t);
        divmod_u24(t, lv.tiles_x[l], tyi, txi);
        py0 = tyi * TH + (wave >> 1) * PT; px0 = txi * TW + (wave & 1) * PT;
    }
    const int H = lv.H[l], W = lv.W[l];
    const int nrow = min(PT, H - py0), ncol = min(PT, W - px0);
    const bool live = nrow > 0 && ncol > 0;      // a wave whose patch lies outside the map only helps to fill the table
    const float scale = lv.scale[l];
    float *gxb = (nsplit > 1 ? lv.slab[l] + (size_t)zsplit * N * H * W * C : lv.gx[l]) + (size_t)n * H * W * C;
    if (nsplit > 1) accumulate = 0;
    const float inv_cnt = 1.0f / (float)(sr * sr);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // All global traffic goes through buffer descriptors with 32-bit offsets (the host guarantees every tensor of this
    // launch is < 4 GiB): wave-uniform parts ride in the SGPR soffset, a false predicate turns into an out-of-range
    // offset (the load returns 0) - no divergent branch and no 64-bit address arithmetic in the kernel.
    const auto rs_gy = __builtin_amdgcn_make_buffer_rsrc((void *)gy, 0, (unsigned)((size_t)R * PH * PW * C * 4), 0x00020000);
    const auto rs_gx = __builtin_amdgcn_make_buffer_rsrc((void *)gxb, 0, (unsigned)((size_t)H * W * C * 4), 0x00020000);
    const unsigned row_bytes = (unsigned)C * 4u;
    const unsigned OOB = 0xFFFFFFFFu;
    const unsigned patch_off = (unsigned)(((size_t)py0 * W + px0) * C * 4);
    const float fy0 = (float)py0, fy1 = (float)(py0 + PT), fx0 = (float)px0, fx1 = (float)(px0 + PT);

    // the 16 cells of the patch: written once (zeros where nothing landed), or added to the map when accumulating
    auto store_patch = [&](float4 (&acc)[PT][PT], bool touched, bool act, unsigned vlane) {
        if (accumulate) {
            // gx += acc for the patches that received an entry, with no-return float atomics: this wave is the ONLY writer of
            // its cells (owner-computes), so the result is the single IEEE addition old + acc whatever the hardware's
            // order - and neither a second set of 16 float4 registers (the 64 accumulators fill the budget) nor a
            // load -> add -> store round trip is needed.
            // Idle lanes (C < 256 or a channel tail) are masked off by a real branch: the out-of-range-offset trick of the
            // loads and stores is NOT safe for atomics (an out-of-range buffer atomic raised a hardware exception).
            if (touched && act) {
#pragma unroll
                for (int i = 0; i < PT; ++i)
#pragma unroll
                    for (int k = 0; k < PT; ++k)
                        if (i < nrow && k < ncol) {      // wave-uniform
                            const unsigned so = patch_off + (unsigned)((i * W + k) * C) * 4u;
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][k].x, rs_gx, vlane, so, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][k].y, rs_gx, vlane + 4u, so, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][k].z, rs_gx, vlane + 8u, so, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][k].w, rs_gx, vlane + 12u, so, 0);
                        }
            }
        } else
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int k = 0; k < PT; ++k)
                if (i < nrow && k < ncol)                // wave-uniform
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const __attribute__((ext_vector_type(4))) unsigned *>(&acc[i][k]), rs_gx,
                                                           vlane, patch_off + (unsigned)((i * W + k) * C) * 4u, 2);       // non-temporal: written once, read by a later kernel
    };
    // ---- entry-list plan (see PlanNode): a patch's first node sits at its launch-order slot, further ones come from the pool behind
    const int total_units = lv.tile_begin[lv.L] * BWD_WAVES;
    const int unit = tile_id * BWD_WAVES + wave;
    constexpr size_t NSTRIDE = plan_node_stride<LDS::QC>();
    if (MODE == 2) {
        // behind the lean kernel: only the tiles the plan builder flagged (pool exhausted) - or all of them when the header does not hold
        const bool holds = plan_header_holds(nplan, lv, total_units, R, N, PH, PW, sr, (int)LDS::QC, nodes_off, node_cap);
        if (holds && __builtin_amdgcn_readfirstlane(nplan[NP_HDR_INTS + tile_id]) == 0) return;
        load_rois(0, rv0, lv0);
    }
    // MODE 1: a full queue goes to a plan node instead of being drained
    int cur_node = -1;
    auto flush_to_plan = [&](int &qn_) {
        int id = unit;
        if (cur_node >= 0) {
            int got = 0;
            if (lane == 0) got = atomicAdd(&nplan[NP_EXTRA], 1);
            id = total_units + __builtin_amdgcn_readfirstlane(got);
            if (id >= node_cap) {                  // pool exhausted: this tile takes the fused path (MODE 2)
                if (lane == 0) { nplan[NP_HDR_INTS + tile_id] = 1; atomicAdd(&nplan[NP_OVERFLOW], 1); }
                id = -1;
            }
        }
        if (id >= 0) {
            char *nb = reinterpret_cast<char *>(nplan + nodes_off);
            PlanNode<LDS::QC> *pn = reinterpret_cast<PlanNode<LDS::QC> *>(nb + (size_t)id * NSTRIDE);
            for (int i = lane; i < qn_; i += 64) {
                pn->q.wy[i] = lds.q.wy[i];
                pn->q.wx[i] = lds.q.wx[i];
                pn->q.row[i] = lds.q.row[i];
            }
            if (lane == 0) {
                pn->count = qn_;
                pn->next = -1;
                if (cur_node >= 0) reinterpret_cast<PlanNode<LDS::QC> *>(nb + (size_t)cur_node * NSTRIDE)->next = id;
            }
            cur_node = id;
        }
        qn_ = 0;
    };
    if (MODE == 1 && tile_id == 0 && tid == 0) {
        nplan[NP_UNITS] = total_units; nplan[NP_CAP] = node_cap; nplan[NP_R] = R; nplan[NP_PH] = PH; nplan[NP_PW] = PW; nplan[NP_SR] = sr;
        nplan[NP_N] = N; nplan[NP_QC] = (int)LDS::QC; nplan[NP_L] = lv.L; nplan[NP_NODES_OFF] = nodes_off;
        for (int q = 0; q < lv.L; ++q) { nplan[NP_HW + 2 * q] = lv.H[q]; nplan[NP_HW + 2 * q + 1] = lv.W[q]; }
        nplan[NP_MAGIC_I] = NP_MAGIC;
    }

    auto fill_table = [&](int seg, float (&rv)[FILL][5], int (&lvv)[FILL]) {
#pragma unroll
        for (int h2 = 0; h2 < FILL; ++h2) {
            const int e = tid + h2 * BWD_THREADS, i = seg + e;
            const RoiGeom g = roi_geom(rv[h2], scale, PH, PW, sr);
            bool mine = i < R && min(max(lvv[h2], 0), lv.L - 1) == l && g.n == n;
            if (nsplit > 1) mine = mine && (i % nsplit == zsplit);       // wave-uniform condition: no division otherwise
            sBox[e] = mine ? make_float4(g.y1f, g.y1f + g.rh + 1.0f, g.x1f, g.x1f + g.rw + 1.0f)
                           : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
            sGeo[e] = make_float4(g.x1f, g.y1f, g.bw, g.bh);
        }
    };
    if (STAMP) { st_tmp = stamp_now(); st_pre = st_tmp - st0; }
    fill_table(0, rv0, lv0);
#pragma nounroll
    for (int cb = 0; cb < (MODE == 1 ? 1 : C); cb += CCH) {          // (the plan does not depend on the channels)
        const bool act = cb + lane * 4 < C;
        const unsigned vlane = act ? (unsigned)(cb + lane * 4) * 4u : OOB;        // byte offset of this lane's 4 channels
        float4 acc[PT][PT];
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int k = 0; k < PT; ++k) acc[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        // accumulate: the old values are read at the END, and only by patches that received an entry - most patches of the
        // fine levels are touched by no RoI at all (a step's RoIs sit on the coarse levels) and then cost no traffic
        bool touched = false;
        int qn = 0;
#pragma nounroll
        for (int seg = 0; seg < R; seg += W2_SEG) {
            // ---- segment table: thread tid fills entries tid and tid + 256
            if (STAMP && !(seg == 0 && cb == 0)) st_tmp = stamp_now();
            if (seg > 0) __syncthreads();                // every wave is done with the previous segment's table
            if (!(seg == 0 && cb == 0)) {                // (the first table was filled before the accumulators existed)
                float rv1[FILL][5];
                int lv1[FILL];
                load_rois(seg, rv1, lv1);
                fill_table(seg, rv1, lv1);
            }
            __syncthreads();
            if (!live) continue;
            // ---- scan the table: 4 compares per (lane, RoI); candidates compacted in ascending RoI order
            int nlist = 0;
            const int seg_n = min(W2_SEG, R - seg);
            // four 64-RoI groups per round: their box reads are issued together (one LDS round trip per round, not the two
            // dependent ones per group a short-circuit `&&` chain compiles to), the tests are bitwise ANDs of four compares
#pragma unroll
            for (int g4 = 0; g4 < W2_SEG / 64; g4 += 4) {
                if (g4 * 64 >= seg_n) break;
                float4 bx[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) bx[u] = sBox[(g4 + u) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g2 = g4 + u;
                    const bool f = (g2 * 64 < seg_n) & (bx[u].x < fy1) & (bx[u].y >= fy0) & (bx[u].z < fx1) & (bx[u].w >= fx0);
                    const unsigned long long bal = __ballot(f);
                    if (f) lds.idx[nlist + __popcll(bal & lt_mask)] = (unsigned short)(g2 * 64 + lane);
                    nlist += __popcll(bal);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (STAMP) st_scan += stamp_now() - st_tmp;
            // Every wave of the launch is resident at once and the kernel ends with its slowest wave; a wave's work grows with
            // its candidate count (median 12, maximum 23 on configs[1]).  VALU issue is arbitrated by priority, then age: the
            // heavy waves are raised so that they run at their own pace from the start and the light waves of the SIMD fill
            // the slots they leave, instead of everybody sharing evenly and the heavy ones running on alone at the end.
            if (seg == 0 && cb == 0) {
                if (nlist >= PRIO_CAND[2]) __builtin_amdgcn_s_setprio(3);
                else if (nlist >= PRIO_CAND[1]) __builtin_amdgcn_s_setprio(2);
                else if (nlist >= PRIO_CAND[0]) __builtin_amdgcn_s_setprio(1);
            }
            // ---- list units (RoI, 64-bin chunk): table pass at the start of every batch of SPP RoIs, then queue entries;
            //      the queue is drained whenever it cannot take another unit (accumulators stay in registers)
            unsigned long long nzb = 0;
            // table pass for the batch of SPP RoIs starting at list item li: lane = (slot, axis, bin) -> the summed sample weights
            // of that bin on the patch's 4 rows / 4 columns; nzb = ballot of the bins with a non-zero weight
            auto table_pass = [&](int li) {
                unsigned long long st_t0 = 0;
                if (STAMP) { st_t0 = stamp_now(); st_cnt += 1 << 16; }
                const int slot = lane / (2 * PBT), axis = (lane / PBT) & 1, bin = lane % PBT;
                const bool tv = li + slot < nlist && bin < (axis ? PW : PH);
                const float4 ge = sGeo[lds.idx[min(li + slot, nlist - 1)]];
                float w[PT] = {0.f, 0.f, 0.f, 0.f};
                const int size = axis ? W : H, t0 = axis ? px0 : py0;
                const float start = axis ? ge.x : ge.y, bsz = axis ? ge.z : ge.w;
                for (int i2 = 0; i2 < sr; ++i2) {
                    const Samp sp = axis_sample(start, bsz, bin, i2, sr, size);
                    const int dl = sp.lo - t0, dh = sp.hi - t0;
#pragma unroll
                    for (int j = 0; j < PT; ++j) {
                        w[j] += (tv && sp.lo >= 0 && dl == j) ? sp.wl : 0.0f;
                        w[j] += (tv && sp.hi >= 0 && dh == j) ? sp.wh : 0.0f;
                    }
                }
                const float sc = axis == 0 ? inv_cnt : 1.0f;
#pragma unroll
                for (int j = 0; j < PT; ++j) w[j] *= sc;
                lds.tab[slot][axis][bin] = make_float4(w[0], w[1], w[2], w[3]);
                nzb = __ballot(w[0] != 0.f || w[1] != 0.f || w[2] != 0.f || w[3] != 0.f);
                __builtin_amdgcn_wave_barrier();
                if (STAMP) st_tab += stamp_now() - st_t0;
            };
            auto drain_now = [&]() {
                __builtin_amdgcn_wave_barrier();
                if (MODE == 1) {
                    flush_to_plan(qn);
                    __builtin_amdgcn_wave_barrier();
                    return;
                }
                if (STAMP) st_tmp = stamp_now();
                drain_wave_queue<QC, DEPTH>(lds.q, qn, rs_gy, vlane, row_bytes, acc, lane);
                if (STAMP) st_drain += stamp_now() - st_tmp;
                __builtin_amdgcn_wave_barrier();
                qn = 0;
            };
            if constexpr (PBT == 8) {
                // 7x7 pooling: a batch of SPP = 4 candidates per table pass; their queue entries are appended TOGETHER - the
                // entry counts follow from the ballots (scalar), so all LDS reads of the batch (bin weights, RoI indices) are
                // issued at once and the appends cost one LDS round trip per batch instead of three dependent ones per candidate
                const int ph = lane >> 3, pw = lane & 7;
#pragma nounroll
                for (int li = 0; li < nlist; li += SPP) {
                    if (STAMP) st_cnt += min(SPP, nlist - li);
                    table_pass(li);
                    if (nzb == 0ull) continue;
                    // two candidates per round (their 4 weight vectors = 16 VGPRs; all four would spill next to the 64
                    // accumulators); a pair adds at most 98 entries, so it always fits an empty queue
#pragma nounroll
                    for (int h = 0; h < SPP; h += 2) {
                        const unsigned pairbits = (unsigned)(nzb >> (h * 2 * PBT));
                        const unsigned ym0 = pairbits & 0xFFu, xm0 = (pairbits >> 8) & 0xFFu, ym1 = (pairbits >> 16) & 0xFFu, xm1 = pairbits >> 24;
                        const int c0 = __popc(ym0) * __popc(xm0), c1 = __popc(ym1) * __popc(xm1);
                        if (c0 + c1 == 0) continue;
                        touched = true;
                        if (qn + c0 + c1 > QC) drain_now();
                        const float4 wy0 = lds.tab[h][0][ph], wx0 = lds.tab[h][1][pw], wy1 = lds.tab[h + 1][0][ph], wx1 = lds.tab[h + 1][1][pw];
                        const int r0 = seg + (int)lds.idx[min(li + h, nlist - 1)], r1 = seg + (int)lds.idx[min(li + h + 1, nlist - 1)];
                        const bool in0 = ((ym0 >> ph) & 1u) & ((xm0 >> pw) & 1u), in1 = ((ym1 >> ph) & 1u) & ((xm1 >> pw) & 1u);
                        const unsigned long long b0 = __ballot(in0), b1 = __ballot(in1);
                        if (in0) {
                            const int pos = qn + __popcll(b0 & lt_mask);
                            lds.q.wy[pos] = wy0;
                            lds.q.wx[pos] = wx0;
                            lds.q.row[pos] = (r0 * PH + ph) * PW + pw;
                        }
                        if (in1) {
                            const int pos = qn + c0 + __popcll(b1 & lt_mask);
                            lds.q.wy[pos] = wy1;
                            lds.q.wx[pos] = wx1;
                            lds.q.row[pos] = (r1 * PH + ph) * PW + pw;
                        }
                        qn += c0 + c1;
                    }
                }
            } else {
            // 14x14 pooling: list units (RoI, 64-bin chunk): table pass at the start of every batch of SPP RoIs, then queue
            // entries; the queue is drained whenever it cannot take another unit (accumulators stay in registers)
            const int units = nlist * CH;
#pragma nounroll
            for (int u = 0; u < units; ++u) {
                if (qn + 64 > QC) drain_now();
                const int li = u / CH, ch = u - li * CH;
                const int sl = li % SPP;
                if (STAMP) st_cnt += 1;
                if (ch == 0 && sl == 0) table_pass(li);
                // queue pass of unit (list item li, chunk ch): lane = (ph, pw)
                const unsigned ym = (unsigned)(nzb >> (sl * 2 * PBT)) & ((1u << PBT) - 1u);
                const unsigned xm = (unsigned)(nzb >> (sl * 2 * PBT + PBT)) & ((1u << PBT) - 1u);
                if (ym && xm) {
                    const int r = seg + __builtin_amdgcn_readfirstlane((int)lds.idx[li]);
                    const int ph = ch * 4 + (lane >> 4);
                    const int pw = lane & 15;
                    const bool in = ((ym >> ph) & 1u) && ((xm >> pw) & 1u);
                    const unsigned long long bal = __ballot(in);
                    if (in) {
                        const int pos = qn + __popcll(bal & lt_mask);
                        lds.q.wy[pos] = lds.tab[sl][0][ph];
                        lds.q.wx[pos] = lds.tab[sl][1][pw];
                        lds.q.row[pos] = (r * PH + ph) * PW + pw;
                    }
                    qn += __popcll(bal);
                    touched = touched || bal != 0ull;
                }
            }
            }
        }
        // (a wave without a patch must still reach the barrier at the end of the channel pass: skipping it desynchronised the workgroup's
        // barriers from the second channel pass on - C > 256 on a map with ragged tiles, found by the planned-vs-fused test of round 5)
        do {
        if (!live) break;
        __builtin_amdgcn_wave_barrier();
        if (MODE == 1) {                                 // the last (possibly empty) node of the patch
            if (qn > 0 || cur_node < 0) flush_to_plan(qn);
            break;
        }
        if (STAMP) st_tmp = stamp_now();
        drain_wave_queue<QC, DEPTH>(lds.q, qn, rs_gy, vlane, row_bytes, acc, lane);
        if (STAMP) st_drain += stamp_now() - st_tmp;
        unsigned long long st_store = 0;
        if (STAMP) st_store = stamp_now();
        store_patch(acc, touched, act, vlane);
        if (STAMP && stamps && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long st_end = stamp_now();
            unsigned long long *o = stamps + ((size_t)blockIdx.x * BWD_WAVES + wave) * 12;
            o[8] = st_pre; o[9] = st_tab; o[10] = st_cnt; o[11] = (unsigned long long)qn;
            o[0] = st0; o[1] = st_scan; o[2] = st_store - st0 - st_scan - st_drain; o[3] = st_drain; o[4] = st_store; o[5] = st_end;
            o[6] = st_rt0; o[7] = __builtin_amdgcn_s_memrealtime();
        }
        } while (0);
        if (cb + CCH < C) __syncthreads();               // the next channel block refills the segment table
    }
}

// ------------------------------------------------------------------------------------------
// The lean backward (see the PlanNode comment): workgroup = one 8 x 8 tile = 4 patches x 2 channel halves (8 waves).
// ------------------------------------------------------------------------------------------
// WPB waves per workgroup: 8 = one 8 x 8 tile (4 patches x 2 channel halves), 4 = half a tile, 2 = one patch.  Every wave of the launch is
// resident at once and the kernel is bound by the VALU of its busiest SIMD (tools/roi_lean_stamps.py: SIMD time = 215 ticks x its entries,
// correlation 0.92): the workgroup is the grain at which the dispatcher spreads the entries over the CUs.
template <int PBT, int DEPTH, int OCC, int WPB>
__global__ __launch_bounds__(WPB * 64, OCC) void k_roi_align_bwd_lean(Levels lv, const float *__restrict__ gy, int R, int N, int C, int PH, int PW,
                                                                         int sr, int chunk, int accumulate, const int *__restrict__ nplan,
                                                                         int nodes_off, int node_cap, int dbg, unsigned long long *__restrict__ stamps) {
    static_assert(DEPTH % 2 == 0, "rows are consumed in pairs of entries");
    // measurement (mrcnn_debug_roi_align_lean_stamps; null otherwise): 10 x u64 per wave - s_memtime at entry, when the first node's
    // loads have arrived, at the end of the entry loop, when the stores have been acknowledged; entries; HW_ID | XCC_ID << 32;
    // s_memrealtime (100 MHz, one counter for the chip: s_memtime is not) at entry and at the end; s_memtime when the kernel arguments / the
    // node's scalar loads are back
    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
    if (stamps) { st0 = stamp_now(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int QC = WaveLds<PBT>::QC;
    static_assert(QC % 2 == 0, "weight pairs");
    constexpr size_t NSTRIDE = plan_node_stride<QC>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int SUBS = 8 / WPB;                                            // workgroups per tile
    const int bj = (int)(blockIdx.x >> 3), sub = bj % SUBS;
    const int tile_id = (blockIdx.x & 7) * chunk + bj / SUBS;               // XCD-banded tile order, as the wave kernel
    const int total = lv.tile_begin[lv.L];
    if (bj / SUBS >= chunk || tile_id >= total) return;
    // (pairing the tile's heaviest patch with its lightest on one SIMD pair was tried: the four extra counts cost a dependent round trip,
    // +1.3 us on configs[1])
    const char *nb = reinterpret_cast<const char *>(nplan + nodes_off);
    const int pslot = sub * (WPB / 2) + (wave >> 1), half = wave & 1;
    const int unit = tile_id * BWD_WAVES + pslot;
    // the plan's loads go out first: header, tile flag, the first node's count / next / row indices (lane = entry) - one round trip
    const PlanNode<QC> *pn = reinterpret_cast<const PlanNode<QC> *>(nb + (size_t)unit * NSTRIDE);
    const int ql = min(lane, QC - 1);
    unsigned long long stA = 0, stB = 0;
    if (stamps) stA = stamp_now();               // (waits for the kernel arguments: the node's address is known here)
    int qrow = pn->q.row[ql];
    // (at the start of a launch the scalar loads are back 2 - 6 us before this vector load - tools/roi_lean_stamps.py: kernel arguments 0.4 us,
    // scalar loads +1.3 us, the row indices +2.3 us, p90 +6 us - but sending the first DEPTH gy rows out on scalar-loaded indices changes
    // nothing, 18.6 against 18.5 us: it is the FIRST vector access of a wave that is slow, whichever it is)
    int cnt = scalar_load<int>(&pn->count), next = scalar_load<int>(&pn->next);
    const bool holds = plan_header_holds(nplan, lv, total * BWD_WAVES, R, N, PH, PW, sr, QC, nodes_off, node_cap);
    const int flagged = scalar_load<int>(nplan + NP_HDR_INTS + tile_id);
    if (stamps) { asm volatile("" :: "s"(cnt), "s"(next), "s"(flagged)); stB = stamp_now(); }      // (the scalar loads of the node and the header are back)
    // tile -> level, image, patch
    int l = 0, nsplit = 1, zsplit = 0, n;
    while (l + 1 < lv.L && tile_id >= lv.tile_begin[l + 1]) ++l;
    int t = tile_id - lv.tile_begin[l];
    nsplit = lv.split[l];
    int tyi, txi;
    if (nsplit > 1) divmod_u24(t, nsplit, t, zsplit);
    divmod_u24(t, lv.tiles_x[l] * lv.tiles_y[l], n, t);
    divmod_u24(t, lv.tiles_x[l], tyi, txi);
    const int py0 = tyi * TH + (pslot >> 1) * PT, px0 = txi * TW + (pslot & 1) * PT;
    const int H = lv.H[l], W = lv.W[l];
    const int nrow = min(PT, H - py0), ncol = min(PT, W - px0);
    if (nrow <= 0 || ncol <= 0) return;
    // a tile the plan builder flagged (its pool was exhausted), or a plan whose header does not hold: the slow path below derives the
    // entries itself, one at a time, in the fused kernel's order and arithmetic
    if (!holds || flagged != 0) return;           // MODE 2 of the wave kernel computes this tile
    float *gxb = (nsplit > 1 ? lv.slab[l] + (size_t)zsplit * N * H * W * C : lv.gx[l]) + (size_t)n * H * W * C;
    if (nsplit > 1) accumulate = 0;
    const auto rs_gy = __builtin_amdgcn_make_buffer_rsrc((void *)gy, 0, (unsigned)((size_t)R * PH * PW * C * 4), 0x00020000);
    const auto rs_gx = __builtin_amdgcn_make_buffer_rsrc((void *)gxb, 0, (unsigned)((size_t)H * W * C * 4), 0x00020000);
    const unsigned row_bytes = (unsigned)C * 4u, OOB = 0xFFFFFFFFu;
    const unsigned patch_off = (unsigned)(((size_t)py0 * W + px0) * C * 4);
    // every wave of the launch is resident at once and the kernel ends with its slowest wave: long lists run at raised priority
    if (cnt >= 64 || next >= 0) __builtin_amdgcn_s_setprio(3);
    else if (cnt >= 40) __builtin_amdgcn_s_setprio(2);
    else if (cnt >= 24) __builtin_amdgcn_s_setprio(1);
#pragma nounroll
    for (int cb = 0; cb < C; cb += 2 * LEAN_CH) {
        const int ch = cb + half * LEAN_CH + lane * 2;
        const bool act = ch < C;
        const unsigned vlane = act ? (unsigned)ch * 4u : OOB;
        const unsigned vload = (dbg & 1) ? OOB : vlane;               // (measurement: no gy traffic - out-of-range loads return zeros)
        float2 acc[PT][PT];
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int k = 0; k < PT; ++k) acc[i][k] = make_float2(0.f, 0.f);
        int total_n = 0;
        const PlanNode<QC> *cur = pn;
        if (stamps && cb == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            st1 = stamp_now();
        }
        if (cb > 0) {                      // a further channel pass walks the list again from its first node
            qrow = pn->q.row[ql];
            cnt = scalar_load<int>(&pn->count); next = scalar_load<int>(&pn->next);
        }
#pragma nounroll
        while (true) {
            total_n += cnt;
#pragma nounroll
            for (int e0 = 0; e0 < cnt; e0 += 64) {
                const int m = min(64, cnt - e0);
                if (e0 > 0) qrow = cur->q.row[min(e0 + lane, QC - 1)];
                // lanes past the list's end name row 0 (one v_cndmask per 64 entries): the prefetch slots beyond the end then need no select
                // per entry - only the clamp of the lane index
                qrow = lane < m ? qrow : 0;
                // (every load below is issued unconditionally and in slot order: the waitcnt pass can then count - vmcnt(DEPTH - 1) in front of
                // an entry - instead of falling back to vmcnt(0) at a control-flow merge, which serialises the wave on memory latency)
                auto ldrow = [&](int j) -> float2 {
                    const unsigned so = (unsigned)__builtin_amdgcn_readlane(qrow, min(j, 63)) * row_bytes;
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_gy, vload, so, 0);
                    float2 f;
                    __builtin_memcpy(&f, &v, 8);
                    return f;
                };
                // the weights of two entries per scalar load (wy[e], wy[e+1] / wx[e], wx[e+1]: 32 bytes each), one pair ahead; the row
                // weights travel as integers: "weight != 0" is then a scalar integer compare + branch (weights are >= 0), not a VALU class test
                // (no clamp of the pair's index: the furthest pair a list of QC entries asks for starts DEPTH entries past its end - inside the
                // node's next array, never used)
                static_assert(sizeof(float4) * (DEPTH + 2) <= sizeof(int) * QC, "the weight prefetch past a full list stays inside the node");
                auto ldw = [&](int j, i32x8 &wy, f32x8 &wx) {
                    wy = scalar_load<i32x8>(&cur->q.wy[e0 + j]);
                    wx = scalar_load<f32x8>(&cur->q.wx[e0 + j]);
                };
                float2 buf[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {          // issue in slot order (see drain_wave_queue)
                    buf[d] = ldrow(d);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // two pairs of entries' weights in SGPRs: the pair in use and the next one on its way (scalar loads return out of order, so
                // a consumer waits for everything issued before it: the next pair's loads go out right after that wait, a whole pair ahead)
                i32x8 wy, wyn;
                f32x8 wx, wxn;
                ldw(0, wy, wx);
                // (also measured and dropped: the row indices like the weights, eight per scalar load one group ahead, as byte offsets made by the
                // builder - 32 instead of 36 instructions per entry, 14.5 instead of 15.3 us without gy / gx traffic, but 19.1 - 19.8 instead of 18.3 us
                // with it: the prefetch loads then queue behind the scalar loads' waits)
                // (round 5, late, measured and dropped - tools/roi_plan_bench.py, every form bit-identical: the weights like the row indices, lane = entry
                // and eight v_readlane per entry: 27.6 us against 19.8 - a v_readlane costs as much VALU time as two packed FMAs; one wave per patch
                // with four channels per lane and the weights broadcast out of LDS: 23.0 us - the launch lasts as long as the wave of its longest
                // list, and a wave issues one instruction per ~6 ticks whatever it is: 60 instructions per entry instead of 37; whole groups of
                // DEPTH entries without the per-entry guard: the same 38 instructions per entry - the guard was never the cost - plus spills)
#pragma nounroll
                for (int j = 0; j < m; j += DEPTH) {
#pragma unroll
                    for (int g = 0; g < DEPTH / 2; ++g) {
                        ldw(j + 2 * g + 2, wyn, wxn);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int d = 2 * g + u;
                            if (j + d < m) {
                                // all four rows, all four columns, no branch on a zero weight: fma(wx, gy * 0, acc) leaves acc as it is (finite
                                // gy), and the 20 packed instructions cost less than the taken branches that skipped half of them
#pragma unroll
                                for (int i = 0; i < PT; ++i) {
                                    const float wyi = __int_as_float(wy[4 * u + i]);
                                    const float2 tt = make_float2(buf[d].x * wyi, buf[d].y * wyi);      // (gy * wy) * wx, as the fused path
#pragma unroll
                                    for (int k = 0; k < PT; ++k) {
                                        acc[i][k].x = fmaf(wx[4 * u + k], tt.x, acc[i][k].x);
                                        acc[i][k].y = fmaf(wx[4 * u + k], tt.y, acc[i][k].y);
                                    }
                                }
                            }
                            buf[d] = ldrow(j + d + DEPTH);
                            asm volatile("" ::: "memory");
                        }
                        wy = wyn; wx = wxn;
                    }
                }
            }
            if (next < 0) break;
            cur = reinterpret_cast<const PlanNode<QC> *>(nb + (size_t)next * NSTRIDE);
            qrow = cur->q.row[ql];
            cnt = scalar_load<int>(&cur->count);
            next = scalar_load<int>(&cur->next);
        }
        if (stamps && cb == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st2 = stamp_now();
        }
        if (dbg & 2) {                       // measurement: no gx traffic (one store keeps the sums alive)
            float2 sum = make_float2(0.f, 0.f);
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int k = 0; k < PT; ++k) { sum.x += acc[i][k].x; sum.y += acc[i][k].y; }
            if (sum.x == 12345.678f) __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned *>(&sum), rs_gx, vlane, patch_off, 0);
        } else if (accumulate) {
            // (owner-computes: this wave is the only writer of its cells and channels - see the wave kernel's store)
            if (total_n > 0 && act) {
#pragma unroll
                for (int i = 0; i < PT; ++i)
#pragma unroll
                    for (int k = 0; k < PT; ++k)
                        if (i < nrow && k < ncol) {
                            const unsigned so = patch_off + (unsigned)((i * W + k) * C) * 4u;
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][k].x, rs_gx, vlane, so, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][k].y, rs_gx, vlane + 4u, so, 0);
                        }
            }
        } else
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int k = 0; k < PT; ++k)
                if (i < nrow && k < ncol) {
                    // written once, read by a later kernel: non-temporal (does not displace the gy rows other waves re-read from the L2)
                    if (dbg & 16)
                        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned *>(&acc[i][k]), rs_gx, vlane,
                                                              patch_off + (unsigned)((i * W + k) * C) * 4u, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned *>(&acc[i][k]), rs_gx, vlane,
                                                              patch_off + (unsigned)((i * W + k) * C) * 4u, 2);
                }
        if (stamps && cb == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long st3 = stamp_now();
            if (lane == 0) {
                unsigned long long *o = stamps + ((size_t)tile_id * 8 + sub * WPB + wave) * 10;
                o[6] = rt0; o[7] = __builtin_amdgcn_s_memrealtime(); o[8] = stA; o[9] = stB;
                o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = (unsigned long long)total_n;
                o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic (any layout through strides, any pooled size / sampling): reference-layout fallback.
// ------------------------------------------------------------------------------------------
struct Strides4 {
    long long n, c, h, w;
};

__global__ __launch_bounds__(256) void k_roi_align_fwd_generic(const float *__restrict__ x, Strides4 xs,
                                                               const float *__restrict__ rois, int R, int N,
                                                               int C, int H, int W, int PH, int PW,
                                                               float scale, int sr, float *__restrict__ y,
                                                               Strides4 ys, int c_fast) {
    const long long total = (long long)R * C * PH * PW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int r, c, ph, pw;
    long long q = idx;
    if (c_fast) {  // NHWC order: c fastest
        c = q % C; q /= C; pw = q % PW; q /= PW; ph = q % PH; r = (int)(q / PH);
    } else {       // NCHW order: pw fastest
        pw = q % PW; q /= PW; ph = q % PH; q /= PH; c = q % C; r = (int)(q / C);
    }
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, PH, PW, sr);
    float acc = 0.0f;
    if (g.n >= 0 && g.n < N) {
        const float *f = x + g.n * xs.n + c * xs.c;
        for (int iy = 0; iy < g.gh; ++iy) {
            const Samp sy = axis_sample(g.y1f, g.bh, ph, iy, g.gh, H);
            if (sy.lo < 0) continue;
            for (int ix = 0; ix < g.gw; ++ix) {
                const Samp sx = axis_sample(g.x1f, g.bw, pw, ix, g.gw, W);
                if (sx.lo < 0) continue;
                const float w1 = sy.wl * sx.wl, w2 = sy.wl * sx.wh, w3 = sy.wh * sx.wl, w4 = sy.wh * sx.wh;
                acc += ((w1 * f[sy.lo * xs.h + sx.lo * xs.w] + w2 * f[sy.lo * xs.h + sx.hi * xs.w]) +
                        w3 * f[sy.hi * xs.h + sx.lo * xs.w]) + w4 * f[sy.hi * xs.h + sx.hi * xs.w];
            }
        }
    }
    y[r * ys.n + c * ys.c + ph * ys.h + pw * ys.w] = acc / (float)(g.gh * g.gw);
}

__global__ __launch_bounds__(256) void k_roi_align_bwd_generic(const float *__restrict__ gy, Strides4 ys,
                                                               const float *__restrict__ rois, int R, int N,
                                                               int C, int H, int W, int PH, int PW,
                                                               float scale, int sr, float *__restrict__ gx,
                                                               Strides4 xs, int c_fast) {
    const long long total = (long long)R * C * PH * PW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int r, c, ph, pw;
    long long q = idx;
    if (c_fast) {
        c = q % C; q /= C; pw = q % PW; q /= PW; ph = q % PH; r = (int)(q / PH);
    } else {
        pw = q % PW; q /= PW; ph = q % PH; q /= PH; c = q % C; r = (int)(q / C);
    }
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, PH, PW, sr);
    if (g.n < 0 || g.n >= N) return;
    const float gv = gy[r * ys.n + c * ys.c + ph * ys.h + pw * ys.w];
    const float cnt = (float)(g.gh * g.gw);
    float *f = gx + g.n * xs.n + c * xs.c;
    for (int iy = 0; iy < g.gh; ++iy) {
        const Samp sy = axis_sample(g.y1f, g.bh, ph, iy, g.gh, H);
        if (sy.lo < 0) continue;
        for (int ix = 0; ix < g.gw; ++ix) {
            const Samp sx = axis_sample(g.x1f, g.bw, pw, ix, g.gw, W);
            if (sx.lo < 0) continue;
            unsafeAtomicAdd(f + sy.lo * xs.h + sx.lo * xs.w, (gv * (sy.wl * sx.wl)) / cnt);
            unsafeAtomicAdd(f + sy.lo * xs.h + sx.hi * xs.w, (gv * (sy.wl * sx.wh)) / cnt);
            unsafeAtomicAdd(f + sy.hi * xs.h + sx.lo * xs.w, (gv * (sy.wh * sx.wl)) / cnt);
            unsafeAtomicAdd(f + sy.hi * xs.h + sx.hi * xs.w, (gv * (sy.wh * sx.wh)) / cnt);
        }
    }
}

__global__ void k_roi_align_sample_tables(const float *__restrict__ rois, int R, int H, int W, int PH, int PW,
                                          float scale, int sr, int smax, int32_t *cnt, int32_t *idx,
                                          float *wgt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * 2 * smax) return;
    const int k = i % smax, axis = (i / smax) & 1, r = i / (2 * smax);
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, PH, PW, sr);
    const int grid = axis ? g.gw : g.gh, P = axis ? PW : PH;
    const int total = P * grid;
    if (k == 0) cnt[r * 2 + axis] = total;
    int lo = -2, hi = -2;
    float wl = 0.f, wh = 0.f;
    if (k < total) {
        const Samp s = axis_sample(axis ? g.x1f : g.y1f, axis ? g.bw : g.bh, k / grid, k % grid, grid,
                                   axis ? W : H);
        lo = s.lo; hi = s.hi; wl = s.wl; wh = s.wh;
    }
    idx[(size_t)i * 2] = lo;
    idx[(size_t)i * 2 + 1] = hi;
    wgt[(size_t)i * 2] = wl;
    wgt[(size_t)i * 2 + 1] = wh;
}

Strides4 strides_of(int layout, int C, int H, int W) {
    Strides4 s;
    if (layout == MRCNN_LAYOUT_NHWC) {
        s.n = (long long)H * W * C; s.c = 1; s.h = (long long)W * C; s.w = C;
    } else {
        s.n = (long long)C * H * W; s.c = (long long)H * W; s.h = W; s.w = 1;
    }
    return s;
}

// `map` = the feature-map side pointer (always required); `pooled` = the (R,...) side (may be null when R==0)
int check_common(const void *pooled, const void *rois, const void *map, int layout, int N, int C, int H, int W,
                 int R, int PH, int PW, int sr) {
    if (layout != MRCNN_LAYOUT_NCHW && layout != MRCNN_LAYOUT_NHWC)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align: unknown layout %d", layout);
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || R < 0 || sr < 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align: bad sizes N=%d C=%d H=%d W=%d R=%d PH=%d PW=%d sr=%d",
                               N, C, H, W, R, PH, PW, sr);
    if (!map || (R > 0 && (!rois || !pooled))) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align: null pointer");
    return 0;
}

bool fast_bwd_ok(int C, int PH, int PW, int sr, int R) {
    return (C % 4) == 0 && PH <= PB && PW <= PB && sr > 0 && (long long)R * PH * PW < (1ll << 31);
}

// out = (accumulate ? out : 0) + sum_z slab[z]   (n4 float4 per map, z order)
__global__ __launch_bounds__(256) void k_sum_level_slabs(const float *__restrict__ slab, float *__restrict__ out, size_t n4,
                                                         int nsplit, int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = accumulate ? ld4(out + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= nsplit; z += 4) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ld4(slab + ((size_t)(z + j) * n4 + i) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    for (; z < nsplit; ++z) {
        const float4 v = ld4(slab + ((size_t)z * n4 + i) * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + i * 4) = a;
}

size_t slab_ws_bytes(const int *Hs, const int *Ws, int L, int N, int C) {
    size_t b = 0;
    for (int l = 0; l < L; ++l) {
        const int sp = level_split(Hs[l], Ws[l], N);
        if (sp > 1) b += (size_t)sp * N * Hs[l] * Ws[l] * C * sizeof(float);
    }
    return (b + 255) / 256 * 256;
}
// workspace of one backward call: the RoI-split slabs of the coarse levels
size_t bwd_ws_bytes(const int *Hs, const int *Ws, int L, int N, int C, int R, int PH, int PW, int sr) {
    return slab_ws_bytes(Hs, Ws, L, N, C);
}

size_t fwd_ws_bytes(int R) { return R > 0 ? ((size_t)R * sizeof(int32_t) + 255) / 256 * 256 : 0; }
// Forward: rows kernel when the x samples of a row fit one wave (fixed sampling grid, PW * grid <= 64) and every level is
// within 32-bit buffer offsets; else the one-wave-per-bin kernel.
void launch_fwd(Levels &lv, const float *rois, const int32_t *levels, int R, int N, int C, int PH, int PW, int sr, float *y,
                hipStream_t st, void *ws = nullptr, size_t ws_bytes = 0) {
    bool rows_ok = sr > 0 && PW * sr <= 64 && (long long)R * PH < (1 << 24);
    for (int l = 0; l < lv.L; ++l) rows_ok = rows_ok && (unsigned long long)lv.H[l] * lv.W[l] * C * 4ull < (1ull << 32);
    if (rows_ok) {
        const int wgs = mrcnn::cdiv((long long)R * PH, 4), chunk = mrcnn::cdiv(wgs, 8);
        // map order (needs R ints of workspace; worth a second launch from a few waves per CU on)
        int32_t *perm = nullptr;
        if (g_fwd_map_order && ws && ws_bytes >= fwd_ws_bytes(R) && R >= 128 && R <= MAP_ORDER_MAX_R) {
            perm = reinterpret_cast<int32_t *>(ws);
            hipLaunchKernelGGL(k_roi_map_order, dim3(mrcnn::cdiv(R, 16)), dim3(256), (size_t)R * sizeof(unsigned), st, lv, rois, levels, R, perm);
        }
        hipLaunchKernelGGL(k_roi_align_fwd_rows, dim3(chunk * 8), dim3(256), 0, st, lv, rois, levels, R, N, C, PH, PW, sr, y, chunk, perm);
    } else {
        const long long waves = (long long)R * PH * PW;
        hipLaunchKernelGGL(k_roi_align_fwd_nhwc, dim3(mrcnn::cdiv(waves, 4)), dim3(256), 0, st, lv, rois, levels, R, N, C, PH, PW, sr, y);
    }
}

// 2 = independent waves that derive the geometry themselves (default), 1 = the barrier-synchronised tile kernel (the fallback for
// tensors >= 4 GiB; selectable so that the tests reach it on small inputs)
int g_bwd_variant = 2;

// entry-list plan buffer: [256-byte header][tile flags][nodes]; nodes = one per patch slot of the launch + a pool for the patches that
// flush more than once
size_t nplan_nodes(int total_tiles, int R, int PH, int PW) { return (size_t)total_tiles * BWD_WAVES * 2 + (size_t)R * PH * PW / 4 + 1024; }
size_t nplan_stride(int PH, int PW) { return (PH <= 8 && PW <= 8) ? plan_node_stride<WaveLds<8>::QC>() : plan_node_stride<WaveLds<16>::QC>(); }
int count_tiles(const int *Hs, const int *Ws, int L, int N, bool split) {
    int total = 0;
    for (int l = 0; l < L; ++l) total += mrcnn::cdiv(Ws[l], TW) * mrcnn::cdiv(Hs[l], TH) * N * (split ? level_split(Hs[l], Ws[l], N) : 1);
    return total;
}
size_t nplan_bytes(const int *Hs, const int *Ws, int L, int N, int R, int PH, int PW, bool split) {
    const int tiles = count_tiles(Hs, Ws, L, N, split);
    return plan_nodes_off_ints(tiles) * sizeof(int) + nplan_nodes(tiles, R, PH, PW) * nplan_stride(PH, PW);
}

// nplan_mode 0: backward (fused; with `nplan` the lean path when the plan holds); 1: build the entry-list plan of these RoIs into `nplan`
// (no gy / gx needed); split: in mode 1 whether the backward call will have its slab workspace (the coarse levels' RoI split)
int launch_bwd_tiles(Levels &lv, int N, const float *gy, const float *rois, const int32_t *levels, int R,
                     int C, int PH, int PW, int sr, int accumulate, void *ws, size_t ws_bytes, hipStream_t st,
                     int *nplan = nullptr, size_t nplan_size = 0, int nplan_mode = 0, bool plan_split = false, bool plan_verified = false) {
    int total = 0;
    const size_t need = slab_ws_bytes(lv.H, lv.W, lv.L, N, C);
    const bool can_split = nplan_mode == 1 ? (plan_split && R > 0) : (ws && ws_bytes >= need && R > 0);
    float *wp = (float *)ws;
    for (int l = 0; l < lv.L; ++l) {
        lv.tiles_x[l] = mrcnn::cdiv(lv.W[l], TW);
        lv.tiles_y[l] = mrcnn::cdiv(lv.H[l], TH);
        lv.split[l] = can_split ? level_split(lv.H[l], lv.W[l], N) : 1;
        lv.slab[l] = nullptr;
        if (lv.split[l] > 1) {
            lv.slab[l] = wp;
            wp += (size_t)lv.split[l] * N * lv.H[l] * lv.W[l] * C;
        }
        lv.tile_begin[l] = total;
        total += lv.tiles_x[l] * lv.tiles_y[l] * N * lv.split[l];
    }
    lv.tile_begin[lv.L] = total;
    const int chunk = mrcnn::cdiv(total, 8);
    bool waves_ok = g_bwd_variant >= 2 && (unsigned long long)R * PH * PW * C * 4ull < (1ull << 32) && R < (1 << 27) && total < (1 << 24);
    for (int l = 0; l < lv.L; ++l) waves_ok = waves_ok && (unsigned long long)lv.H[l] * lv.W[l] * C * 4ull < (1ull << 32);
    const size_t nodes_off = plan_nodes_off_ints(total);
    if (nplan_mode == 1) {
        // plan builder: needs the wave kernel's preconditions and a buffer of mrcnn_roi_align_fpn_bwd_plan_bytes(); otherwise the header
        // stays without its magic and the backward's MODE 2 launch computes every tile
        if (!nplan || nplan_size < NP_HDR_INTS * sizeof(int)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "roi_align_bwd_plan: plan buffer too small");
        const bool room = nplan_size >= (nodes_off + 64) * sizeof(int);
        MRCNN_HIP_TRY(hipMemsetAsync(nplan, 0, (room ? nodes_off : (size_t)NP_HDR_INTS) * sizeof(int), st));
        const size_t cap = room ? (nplan_size - nodes_off * sizeof(int)) / nplan_stride(PH, PW) : 0;
        if (!waves_ok || R == 0 || cap < (size_t)total * BWD_WAVES || cap >= (1u << 30)) return 0;
        if (PH <= 8 && PW <= 8)
            hipLaunchKernelGGL((k_roi_align_bwd_waves<8, W2_DEPTH, false, 1>), dim3(chunk * 8), dim3(BWD_THREADS), 0, st, lv, (const float *)nullptr, rois, levels,
                               R, N, C, PH, PW, sr, chunk, 0, (unsigned long long *)nullptr, (int)nodes_off, (int)cap, nplan);
        else
            hipLaunchKernelGGL((k_roi_align_bwd_waves<16, W2_DEPTH, false, 1>), dim3(chunk * 8), dim3(BWD_THREADS), 0, st, lv, (const float *)nullptr, rois, levels,
                               R, N, C, PH, PW, sr, chunk, 0, (unsigned long long *)nullptr, (int)nodes_off, (int)cap, nplan);
        MRCNN_LAUNCH_CHECK();
        return 0;
    }
    // a plan buffer that cannot even hold the patch slots of this launch was never filled by the builder: plain fused backward
    if (nplan && nplan_size < nodes_off * sizeof(int) + (size_t)total * BWD_WAVES * nplan_stride(PH, PW)) nplan = nullptr;
    if (!waves_ok || R == 0) nplan = nullptr;
    if (nplan) {
        // the lean kernel along the entry lists, then (unless the caller verified the plan) the wave kernel for whatever the plan could not hold;
        // have_cap: the nodes THIS call's buffer can hold - a plan built into a larger buffer does not validate (its chains could leave this one)
        const int have_cap = (int)std::min<size_t>((nplan_size - nodes_off * sizeof(int)) / nplan_stride(PH, PW), (size_t)1 << 30);
        auto lean = [&](auto kern, int wpb) {
            hipLaunchKernelGGL(kern, dim3(chunk * 8 * (8 / wpb)), dim3(wpb * 64), 0, st, lv, gy, R, N, C, PH, PW, sr, chunk, accumulate, (const int *)nplan, (int)nodes_off,
                               have_cap, g_lean_dbg, g_lean_stamps);
        };
        const bool small = PH <= 8 && PW <= 8;
        switch (g_lean_variant) {
        case 0: if (small) lean(k_roi_align_bwd_lean<8, 10, 8, 8>, 8); else lean(k_roi_align_bwd_lean<16, 10, 8, 8>, 8); break;
        case 1: if (small) lean(k_roi_align_bwd_lean<8, 16, 7, 8>, 8); else lean(k_roi_align_bwd_lean<16, 16, 7, 8>, 8); break;
        case 2: if (small) lean(k_roi_align_bwd_lean<8, 8, 8, 8>, 8); else lean(k_roi_align_bwd_lean<16, 8, 8, 8>, 8); break;
        case 9: if (small) lean(k_roi_align_bwd_lean<8, 8, 8, 2>, 2); else lean(k_roi_align_bwd_lean<16, 8, 8, 2>, 2); break;
        default: if (small) lean(k_roi_align_bwd_lean<8, 8, 8, 4>, 4); else lean(k_roi_align_bwd_lean<16, 8, 8, 4>, 4); break;       // 8
        }
        MRCNN_LAUNCH_CHECK();
        if (plan_verified) {}
        else if (small)
            hipLaunchKernelGGL((k_roi_align_bwd_waves<8, W2_DEPTH, false, 2>), dim3(chunk * 8), dim3(BWD_THREADS), 0, st, lv, gy, rois, levels, R, N, C,
                               PH, PW, sr, chunk, accumulate, (unsigned long long *)nullptr, (int)nodes_off, have_cap, nplan);
        else
            hipLaunchKernelGGL((k_roi_align_bwd_waves<16, W2_DEPTH, false, 2>), dim3(chunk * 8), dim3(BWD_THREADS), 0, st, lv, gy, rois, levels, R, N, C,
                               PH, PW, sr, chunk, accumulate, (unsigned long long *)nullptr, (int)nodes_off, have_cap, nplan);
    } else if (waves_ok) {
        const dim3 grid(chunk * 8);
        if (PH <= 8 && PW <= 8)
            hipLaunchKernelGGL((k_roi_align_bwd_waves<8, W2_DEPTH>), grid, dim3(BWD_THREADS), 0, st, lv, gy, rois, levels, R, N, C,
                               PH, PW, sr, chunk, accumulate);
        else
            hipLaunchKernelGGL((k_roi_align_bwd_waves<16, W2_DEPTH>), grid, dim3(BWD_THREADS), 0, st, lv, gy, rois, levels, R, N, C,
                               PH, PW, sr, chunk, accumulate);
    }
    else if (PH <= 8 && PW <= 8)
        hipLaunchKernelGGL(k_roi_align_bwd_nhwc<8>, dim3(chunk * 8), dim3(BWD_THREADS), 0, st, lv, gy, rois,
                           levels, R, N, C, PH, PW, sr, chunk, accumulate);
    else
        hipLaunchKernelGGL(k_roi_align_bwd_nhwc<16>, dim3(chunk * 8), dim3(BWD_THREADS), 0, st, lv, gy, rois,
                           levels, R, N, C, PH, PW, sr, chunk, accumulate);
    MRCNN_LAUNCH_CHECK();
    for (int l = 0; l < lv.L; ++l)
        if (lv.split[l] > 1) {
            const size_t n4 = (size_t)N * lv.H[l] * lv.W[l] * C / 4;
            hipLaunchKernelGGL(k_sum_level_slabs, dim3((unsigned)mrcnn::cdiv(n4, 256)), dim3(256), 0, st, lv.slab[l], lv.gx[l], n4,
                               lv.split[l], accumulate);
            MRCNN_LAUNCH_CHECK();
        }
    return 0;
}

}  // namespace

extern "C" int mrcnn_roi_align_fwd_ws_f32(const float *x, int layout, int N, int C, int H, int W,
                                          const float *rois, int R, int PH, int PW, float spatial_scale,
                                          int sampling_ratio, float *y, void *ws, size_t ws_bytes, void *stream) {
    if (int e = check_common(y, rois, x, layout, N, C, H, W, R, PH, PW, sampling_ratio)) return e;
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (layout == MRCNN_LAYOUT_NHWC && (C % 4) == 0) {
        Levels lv{};
        lv.L = 1; lv.x[0] = x; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
        launch_fwd(lv, rois, nullptr, R, N, C, PH, PW, sampling_ratio, y, st, ws, ws_bytes);
    } else {
        const long long total = (long long)R * C * PH * PW;
        hipLaunchKernelGGL(k_roi_align_fwd_generic, dim3(mrcnn::cdiv(total, 256)), dim3(256), 0, st, x,
                           strides_of(layout, C, H, W), rois, R, N, C, H, W, PH, PW, spatial_scale,
                           sampling_ratio, y, strides_of(layout, C, PH, PW), layout == MRCNN_LAYOUT_NHWC);
    }
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_roi_align_fwd_f32(const float *x, int layout, int N, int C, int H, int W,
                                       const float *rois, int R, int PH, int PW, float spatial_scale,
                                       int sampling_ratio, float *y, void *stream) {
    return mrcnn_roi_align_fwd_ws_f32(x, layout, N, C, H, W, rois, R, PH, PW, spatial_scale, sampling_ratio, y, nullptr, 0, stream);
}

extern "C" size_t mrcnn_roi_align_fwd_workspace_bytes(int R) { return fwd_ws_bytes(R); }

extern "C" int mrcnn_roi_align_set_fwd_map_order(int on) {
    g_fwd_map_order = on ? 1 : 0;
    return 0;
}

extern "C" int mrcnn_roi_align_bwd_ws_f32(const float *gy, int layout, int N, int C, int H, int W,
                                          const float *rois, int R, int PH, int PW, float spatial_scale,
                                          int sampling_ratio, float *gx, void *ws, size_t ws_bytes, void *stream) {
    if (int e = check_common(gy, rois, gx, layout, N, C, H, W, R, PH, PW, sampling_ratio)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (layout == MRCNN_LAYOUT_NHWC && fast_bwd_ok(C, PH, PW, sampling_ratio, R)) {
        Levels lv{};
        lv.L = 1; lv.gx[0] = gx; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
        return launch_bwd_tiles(lv, N, gy, rois, nullptr, R, C, PH, PW, sampling_ratio, 0, ws, ws_bytes, st);
    }
    MRCNN_HIP_TRY(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)N * C * H * W, st));
    if (R == 0) return 0;
    const long long total = (long long)R * C * PH * PW;
    hipLaunchKernelGGL(k_roi_align_bwd_generic, dim3(mrcnn::cdiv(total, 256)), dim3(256), 0, st, gy,
                       strides_of(layout, C, PH, PW), rois, R, N, C, H, W, PH, PW, spatial_scale,
                       sampling_ratio, gx, strides_of(layout, C, H, W), layout == MRCNN_LAYOUT_NHWC);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_roi_align_bwd_f32(const float *gy, int layout, int N, int C, int H, int W,
                                       const float *rois, int R, int PH, int PW, float spatial_scale,
                                       int sampling_ratio, float *gx, void *stream) {
    return mrcnn_roi_align_bwd_ws_f32(gy, layout, N, C, H, W, rois, R, PH, PW, spatial_scale, sampling_ratio, gx, nullptr, 0, stream);
}

static int fill_levels(Levels &lv, const float *const *xs, float *const *gxs, const int *Hs, const int *Ws,
                       const float *scales, int L) {
    if (L <= 0 || L > MRCNN_MAX_LEVELS) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn: L=%d not in [1,%d]", L, MRCNN_MAX_LEVELS);
    if (!Hs || !Ws || !scales) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn: null host array");
    lv.L = L;
    for (int l = 0; l < L; ++l) {
        if (Hs[l] <= 0 || Ws[l] <= 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn: bad level %d shape", l);
        if (xs) { if (!xs[l]) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn: null level pointer"); lv.x[l] = xs[l]; }
        if (gxs) { if (!gxs[l]) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn: null level pointer"); lv.gx[l] = gxs[l]; }
        lv.H[l] = Hs[l]; lv.W[l] = Ws[l]; lv.scale[l] = scales[l];
    }
    return 0;
}

extern "C" int mrcnn_roi_align_fpn_fwd_ws_f32(const float *const *xs, const int *Hs, const int *Ws,
                                              const float *scales, int L, int N, int C, const float *rois,
                                              const int32_t *levels, int R, int PH, int PW,
                                              int sampling_ratio, float *y, void *ws, size_t ws_bytes, void *stream) {
    if (!xs || !y || (R > 0 && (!rois || !levels))) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_fwd: null pointer");
    if (N <= 0 || C <= 0 || (C % 4) || PH <= 0 || PW <= 0 || R < 0 || sampling_ratio < 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_fwd: bad sizes (C must be a multiple of 4)");
    Levels lv{};
    if (int e = fill_levels(lv, xs, nullptr, Hs, Ws, scales, L)) return e;
    if (R == 0) return 0;
    launch_fwd(lv, rois, levels, R, N, C, PH, PW, sampling_ratio, y, (hipStream_t)stream, ws, ws_bytes);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_roi_align_fpn_fwd_f32(const float *const *xs, const int *Hs, const int *Ws,
                                           const float *scales, int L, int N, int C, const float *rois,
                                           const int32_t *levels, int R, int PH, int PW,
                                           int sampling_ratio, float *y, void *stream) {
    return mrcnn_roi_align_fpn_fwd_ws_f32(xs, Hs, Ws, scales, L, N, C, rois, levels, R, PH, PW, sampling_ratio, y, nullptr, 0, stream);
}

extern "C" int mrcnn_roi_align_fpn_bwd_f32(const float *gy, float *const *gxs, const int *Hs, const int *Ws,
                                           const float *scales, int L, int N, int C, const float *rois,
                                           const int32_t *levels, int R, int PH, int PW,
                                           int sampling_ratio, int accumulate, void *ws, size_t ws_bytes, void *stream) {
    if (!gxs || (R > 0 && (!rois || !levels || !gy))) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_bwd: null pointer");
    if (N <= 0 || C <= 0 || PH <= 0 || PW <= 0 || R < 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_bwd: bad sizes");
    if (!fast_bwd_ok(C, PH, PW, sampling_ratio, R))
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "roi_align_fpn_bwd: needs C%%4==0, PH,PW<=%d, sampling_ratio>0", PB);
    Levels lv{};
    if (int e = fill_levels(lv, nullptr, gxs, Hs, Ws, scales, L)) return e;
    return launch_bwd_tiles(lv, N, gy, rois, levels, R, C, PH, PW, sampling_ratio, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

// ---- (ABI v9) entry-list plan of the backward, built ahead of time from the RoIs alone; a backward that follows it
extern "C" size_t mrcnn_roi_align_fpn_bwd_plan_bytes(const int *Hs, const int *Ws, int L, int N, int R, int PH, int PW, int split_levels) {
    if (!Hs || !Ws || L <= 0 || L > MRCNN_MAX_LEVELS || N <= 0 || R <= 0 || PH <= 0 || PW <= 0 || PH > PB || PW > PB) return 0;
    for (int l = 0; l < L; ++l) if (Hs[l] <= 0 || Ws[l] <= 0) return 0;
    return nplan_bytes(Hs, Ws, L, N, R, PH, PW, split_levels != 0);
}
extern "C" int mrcnn_roi_align_fpn_bwd_plan_f32(const int *Hs, const int *Ws, const float *scales, int L, int N, int C, const float *rois,
                                                const int32_t *levels, int R, int PH, int PW, int sampling_ratio, int split_levels,
                                                void *plan, size_t plan_bytes, void *stream) {
    if (!plan || (R > 0 && (!rois || (L > 1 && !levels)))) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_bwd_plan: null pointer");
    if (N <= 0 || C <= 0 || PH <= 0 || PW <= 0 || R < 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_bwd_plan: bad sizes");
    if (!fast_bwd_ok(C, PH, PW, sampling_ratio, R))
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "roi_align_fpn_bwd_plan: needs C%%4==0, PH,PW<=%d, sampling_ratio>0", PB);
    Levels lv{};
    if (int e = fill_levels(lv, nullptr, nullptr, Hs, Ws, scales, L)) return e;
    return launch_bwd_tiles(lv, N, nullptr, rois, levels, R, C, PH, PW, sampling_ratio, 0, nullptr, 0, (hipStream_t)stream, (int *)plan,
                            plan_bytes, 1, split_levels != 0);
}
extern "C" int mrcnn_roi_align_fpn_bwd_planned_f32(const float *gy, float *const *gxs, const int *Hs, const int *Ws, const float *scales, int L,
                                                   int N, int C, const float *rois, const int32_t *levels, int R, int PH, int PW,
                                                   int sampling_ratio, int accumulate, void *ws, size_t ws_bytes, void *plan,
                                                   size_t plan_bytes, int plan_verified, void *stream) {
    if (!gxs || (R > 0 && (!rois || (L > 1 && !levels) || !gy))) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_bwd_planned: null pointer");
    if (N <= 0 || C <= 0 || PH <= 0 || PW <= 0 || R < 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_bwd_planned: bad sizes");
    if (!fast_bwd_ok(C, PH, PW, sampling_ratio, R))
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "roi_align_fpn_bwd_planned: needs C%%4==0, PH,PW<=%d, sampling_ratio>0", PB);
    Levels lv{};
    if (int e = fill_levels(lv, nullptr, gxs, Hs, Ws, scales, L)) return e;
    return launch_bwd_tiles(lv, N, gy, rois, levels, R, C, PH, PW, sampling_ratio, accumulate, ws, ws_bytes, (hipStream_t)stream, (int *)plan,
                            plan ? plan_bytes : 0, 0, false, plan_verified != 0);
}
// status3 (host): [0] 1 = the header carries the builder's magic, [1] tiles the builder flagged (pool exhausted), [2] pool nodes used.
// The ONE entry point of this library that waits for the device (it copies the header back and synchronises `stream`).
extern "C" int mrcnn_roi_align_bwd_plan_status(const void *plan, size_t plan_bytes, int *status3, void *stream) {
    if (!plan || !status3 || plan_bytes < NP_HDR_INTS * sizeof(int)) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_bwd_plan_status: null pointer or short buffer");
    int hdr[NP_HDR_INTS];
    MRCNN_HIP_TRY(hipMemcpyAsync(hdr, plan, sizeof hdr, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MRCNN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    status3[0] = hdr[NP_MAGIC_I] == NP_MAGIC ? 1 : 0;
    status3[1] = hdr[NP_OVERFLOW];
    status3[2] = hdr[NP_EXTRA];
    return 0;
}

extern "C" size_t mrcnn_roi_align_fpn_bwd_workspace_bytes(const int *Hs, const int *Ws, int L, int N, int C, int R, int PH, int PW,
                                                          int sampling_ratio) {
    if (!Hs || !Ws || L <= 0 || L > MRCNN_MAX_LEVELS || N <= 0 || C <= 0) return 0;
    return bwd_ws_bytes(Hs, Ws, L, N, C, R, PH, PW, sampling_ratio);
}

extern "C" size_t mrcnn_roi_align_bwd_workspace_bytes(int N, int C, int H, int W, int R, int PH, int PW, int sampling_ratio) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return bwd_ws_bytes(&H, &W, 1, N, C, R, PH, PW, sampling_ratio);
}

// Diagnostic: configs-style single-level backward with phase stamps (see k_roi_align_bwd_waves<.., STAMP>); stamps =
// (grid workgroups x 4 waves x 8) u64, grid = 8 * ceil(tiles / 8).  Not part of the product path.
extern "C" int mrcnn_debug_roi_align_bwd_stamps(const float *gy, int N, int C, int H, int W, const float *rois, int R, int PH,
                                                int PW, float spatial_scale, int sampling_ratio, float *gx,
                                                unsigned long long *stamps, void *stream) {
    if (!gy || !rois || !gx || !stamps || PH > 8 || PW > 8 || !fast_bwd_ok(C, PH, PW, sampling_ratio, R))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_roi_align_bwd_stamps: bad arguments (7x7-class pooling only)");
    Levels lv{};
    lv.L = 1; lv.gx[0] = gx; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
    lv.tiles_x[0] = mrcnn::cdiv(W, TW); lv.tiles_y[0] = mrcnn::cdiv(H, TH); lv.split[0] = 1; lv.tile_begin[0] = 0;
    const int total = lv.tiles_x[0] * lv.tiles_y[0] * N;
    lv.tile_begin[1] = total;
    const int chunk = mrcnn::cdiv(total, 8);
    hipLaunchKernelGGL((k_roi_align_bwd_waves<8, W2_DEPTH, true>), dim3(chunk * 8), dim3(BWD_THREADS), 0, (hipStream_t)stream, lv, gy, rois,
                       (const int32_t *)nullptr, R, N, C, PH, PW, sampling_ratio, chunk, 0, stamps);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// Measurement: where does the hardware put the workgroups of a launch shaped like k_roi_align_bwd_waves (256 threads, every block
// resident at once)?  out[b] = HW_ID | XCC_ID << 32 of block b's first wave, out[nblocks + b] = its s_memrealtime at start; every block
// then spins for `spin_us` so that the whole grid is co-resident like the real kernel's.
__global__ __launch_bounds__(BWD_THREADS, 4) void k_dispatch_census(unsigned long long *out, int nblocks, int spin_ticks) {
    __shared__ float pad[WaveLds<8>::QC > 0 ? 1 : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        out[nblocks + blockIdx.x] = t0;
    }
    pad[0] = 0.f;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int mrcnn_debug_dispatch_census(unsigned long long *out, int nblocks, int spin_us, void *stream) {
    if (!out || nblocks <= 0 || spin_us < 0 || spin_us > 1000) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_dispatch_census: bad arguments");
    hipLaunchKernelGGL(k_dispatch_census, dim3(nblocks), dim3(BWD_THREADS), 0, (hipStream_t)stream, out, nblocks, spin_us * 100);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_debug_roi_align_lean_stamps(unsigned long long *stamps) {     // 10 x u64 per wave (8 per 8 x 8 tile) of the next planned calls; null = off
    g_lean_stamps = stamps;
    return 0;
}

extern "C" int mrcnn_debug_roi_align_lean_variant(int v) {          // measurement: rows in flight / occupancy of the lean backward
    if (v < 0 || ((v & 0xff) > 2 && (v & 0xff) != 8 && (v & 0xff) != 9)) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_roi_align_lean_variant: 0, 1, 2, 8 or 9 (+ 256 x measurement bits)");
    g_lean_variant = v & 0xff;
    g_lean_dbg = v >> 8;         // 1: no gy loads, 2: no gx stores (results are wrong with either), 16: plain instead of non-temporal stores
    return 0;
}

extern "C" int mrcnn_roi_align_set_bwd_variant(int variant) {
    if (variant < 1 || variant > 2) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_set_bwd_variant: 1 or 2");
    g_bwd_variant = variant;
    return 0;
}

extern "C" int mrcnn_roi_align_sample_tables(const float *rois, int R, int H, int W, int PH, int PW,
                                             float spatial_scale, int sampling_ratio, int smax,
                                             int32_t *cnt, int32_t *idx, float *wgt, void *stream) {
    if (R < 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || smax <= 0 || sampling_ratio < 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_sample_tables: bad sizes");
    if (R == 0) return 0;
    if (!rois || !cnt || !idx || !wgt) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_sample_tables: null pointer");
    const int total = R * 2 * smax;
    hipLaunchKernelGGL(k_roi_align_sample_tables, dim3(mrcnn::cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, rois, R, H, W, PH, PW, spatial_scale, sampling_ratio, smax, cnt,
                       idx, wgt);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

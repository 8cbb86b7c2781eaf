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

#pragma clang fp contract(off)

namespace {

struct RoiGeom {
    float x1f, y1f, bw, bh, rw, rh;
    int gh, gw, n;
};

__device__ __forceinline__ RoiGeom roi_geom(const float *roi, float s, int PH, int PW, int sr) {
    RoiGeom g;
    g.n = (int)roi[0];
    g.x1f = roi[1] * s;
    g.y1f = roi[2] * s;
    float x2f = roi[3] * s, y2f = roi[4] * s;
    g.rw = fmaxf(x2f - g.x1f, 1.0f);
    g.rh = fmaxf(y2f - g.y1f, 1.0f);
    g.bw = g.rw / (float)PW;
    g.bh = g.rh / (float)PH;
    if (sr > 0) {
        g.gh = g.gw = sr;
    } else {
        g.gh = (int)ceilf(g.rh / (float)PH);
        g.gw = (int)ceilf(g.rw / (float)PW);
    }
    return g;
}

struct Samp {
    int lo, hi;      // corner cells, -1 when the sample is void
    float wl, wh;    // weight of lo / hi cell (hy / ly in the Caffe2 formula)
};

// One axis of one sample: c = (start + p*bin) + ((i+0.5)*bin)/grid, in exactly this order.
__device__ __forceinline__ Samp axis_sample(float start, float bin, int p, int i, int grid, int size) {
    float c = (start + (float)p * bin) + (((float)i + 0.5f) * bin) / (float)grid;
    bool valid = !(c < -1.0f || c > (float)size);
    c = fmaxf(c, 0.0f);
    int lo = (int)c, hi;
    if (lo >= size - 1) {
        lo = hi = size - 1;
        c = (float)lo;
    } else {
        hi = lo + 1;
    }
    Samp s;
    s.wh = c - (float)lo;
    s.wl = 1.0f - s.wh;
    if (!valid) {
        s.lo = s.hi = -1;
        s.wl = s.wh = 0.0f;
    } else {
        s.lo = lo;
        s.hi = hi;
    }
    return s;
}

struct Levels {
    const float *x[MRCNN_MAX_LEVELS];
    float *gx[MRCNN_MAX_LEVELS];
    int H[MRCNN_MAX_LEVELS], W[MRCNN_MAX_LEVELS];
    float scale[MRCNN_MAX_LEVELS];
    int tile_begin[MRCNN_MAX_LEVELS + 1];
    int tiles_x[MRCNN_MAX_LEVELS], tiles_y[MRCNN_MAX_LEVELS];
    // backward RoI split: on a coarse level (few tiles, many RoIs - the reference maps most RoIs to the coarsest levels)
    // each tile is computed by split[l] workgroups, workgroup z taking the RoIs with index % split == z and writing a
    // partial map to slab[l] + z * (N*H*W*C); k_sum_level_slabs adds the partial maps in z order (deterministic).
    int split[MRCNN_MAX_LEVELS];
    float *slab[MRCNN_MAX_LEVELS];
    int L;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

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
// Backward, NHWC, owner-computes tiles.
// ------------------------------------------------------------------------------------------
constexpr int TH = 8, TW = 8;     // tile of gradient-map cells owned by one workgroup
constexpr int PT = 4;             // a wave owns a PT x PT patch of cells (accumulators in registers)
constexpr int SLOTS = 32;         // RoIs whose weight tables are resident in LDS per round
constexpr int PB = 16;            // max pooled bins per axis on this path (7 and 14 in the model)
constexpr int BWD_THREADS = 256;  // 4 waves = 2 x 2 patches
constexpr int BWD_WAVES = BWD_THREADS / 64;
constexpr int LISTCAP = 512;      // RoIs scanned per segment
constexpr int QCAP_MAX = 256;     // per-wave queue capacity (template QC): 64 entries for 7x7 pooling, 256 for 14x14
                                  // (a full window is rebuilt per pass; entry = (gy row, 4 row weights, 4 col weights))
constexpr int CCH = 256;          // channels per pass: lane = 4 channels

#define MRCNN_FMA4(A, c, g)                \
    A.x = fmaf(c, g.x, A.x); A.y = fmaf(c, g.y, A.y); \
    A.z = fmaf(c, g.z, A.z); A.w = fmaf(c, g.w, A.w);

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

// RoI split of a level: enough workgroups to occupy the chip when the level has few tiles.
int level_split(int H, int W, int N) {
    const int tiles = mrcnn::cdiv(W, TW) * mrcnn::cdiv(H, TH) * N;
    if (tiles >= 256) return 1;
    return std::min(32, mrcnn::cdiv(512, tiles));
}
size_t bwd_ws_bytes(const int *Hs, const int *Ws, int L, int N, int C) {
    size_t b = 0;
    for (int l = 0; l < L; ++l) {
        const int sp = level_split(Hs[l], Ws[l], N);
        if (sp > 1) b += (size_t)sp * N * Hs[l] * Ws[l] * C * sizeof(float);
    }
    return b;
}

int launch_bwd_tiles(Levels &lv, int N, const float *gy, const float *rois, const int32_t *levels, int R,
                     int C, int PH, int PW, int sr, int accumulate, void *ws, size_t ws_bytes, hipStream_t st) {
    int total = 0;
    size_t need = 0;
    for (int l = 0; l < lv.L; ++l) need += level_split(lv.H[l], lv.W[l], N) > 1
        ? (size_t)level_split(lv.H[l], lv.W[l], N) * N * lv.H[l] * lv.W[l] * C * sizeof(float) : 0;
    const bool can_split = ws && ws_bytes >= need && R > 0;
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
    if (PH <= 8 && PW <= 8)
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

extern "C" int mrcnn_roi_align_fwd_f32(const float *x, int layout, int N, int C, int H, int W,
                                       const float *rois, int R, int PH, int PW, float spatial_scale,
                                       int sampling_ratio, float *y, void *stream) {
    if (int e = check_common(y, rois, x, layout, N, C, H, W, R, PH, PW, sampling_ratio)) return e;
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (layout == MRCNN_LAYOUT_NHWC && (C % 4) == 0) {
        Levels lv{};
        lv.L = 1; lv.x[0] = x; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
        const long long waves = (long long)R * PH * PW;
        hipLaunchKernelGGL(k_roi_align_fwd_nhwc, dim3(mrcnn::cdiv(waves, 4)), dim3(256), 0, st, lv, rois,
                           (const int32_t *)nullptr, R, N, C, PH, PW, sampling_ratio, y);
    } else {
        const long long total = (long long)R * C * PH * PW;
        hipLaunchKernelGGL(k_roi_align_fwd_generic, dim3(mrcnn::cdiv(total, 256)), dim3(256), 0, st, x,
                           strides_of(layout, C, H, W), rois, R, N, C, H, W, PH, PW, spatial_scale,
                           sampling_ratio, y, strides_of(layout, C, PH, PW), layout == MRCNN_LAYOUT_NHWC);
    }
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_roi_align_bwd_f32(const float *gy, int layout, int N, int C, int H, int W,
                                       const float *rois, int R, int PH, int PW, float spatial_scale,
                                       int sampling_ratio, float *gx, void *stream) {
    if (int e = check_common(gy, rois, gx, layout, N, C, H, W, R, PH, PW, sampling_ratio)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (layout == MRCNN_LAYOUT_NHWC && fast_bwd_ok(C, PH, PW, sampling_ratio, R)) {
        Levels lv{};
        lv.L = 1; lv.gx[0] = gx; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
        return launch_bwd_tiles(lv, N, gy, rois, nullptr, R, C, PH, PW, sampling_ratio, 0, nullptr, 0, st);
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

extern "C" int mrcnn_roi_align_fpn_fwd_f32(const float *const *xs, const int *Hs, const int *Ws,
                                           const float *scales, int L, int N, int C, const float *rois,
                                           const int32_t *levels, int R, int PH, int PW,
                                           int sampling_ratio, float *y, void *stream) {
    if (!xs || !y || (R > 0 && (!rois || !levels))) return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_fwd: null pointer");
    if (N <= 0 || C <= 0 || (C % 4) || PH <= 0 || PW <= 0 || R < 0 || sampling_ratio < 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "roi_align_fpn_fwd: bad sizes (C must be a multiple of 4)");
    Levels lv{};
    if (int e = fill_levels(lv, xs, nullptr, Hs, Ws, scales, L)) return e;
    if (R == 0) return 0;
    const long long waves = (long long)R * PH * PW;
    hipLaunchKernelGGL(k_roi_align_fwd_nhwc, dim3(mrcnn::cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, lv,
                       rois, levels, R, N, C, PH, PW, sampling_ratio, y);
    MRCNN_LAUNCH_CHECK();
    return 0;
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

extern "C" size_t mrcnn_roi_align_fpn_bwd_workspace_bytes(const int *Hs, const int *Ws, int L, int N, int C) {
    if (!Hs || !Ws || L <= 0 || L > MRCNN_MAX_LEVELS || N <= 0 || C <= 0) return 0;
    return bwd_ws_bytes(Hs, Ws, L, N, C);
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

// Shared device helpers of the ROIAlign kernels (roi_align.hip: forward, the backward variants 1 / 2, the entry-list plan and the lean
// backward).  The coordinate arithmetic is compiled with FP contraction OFF and follows oracle/roi_align.py
// operation for operation, so sample indices and weights are bit-exact in every kernel that uses axis_sample().
#pragma once
#include "common.h"
#include <algorithm>

namespace mrcnn_roi {


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
    // x / 2.0f == x * 0.5f bit for bit (barring subnormals): the common sampling ratio avoids the correctly-rounded divide
    const float t = ((float)i + 0.5f) * bin;
    float c = (start + (float)p * bin) + (grid == 2 ? t * 0.5f : t / (float)grid);
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

// q = v / d, r = v % d for 0 <= v < 2^24 with inv = 1.0f / d: float estimate + one correction each way (exact); the
// tile decode of a workgroup would otherwise spend ~100 instructions in four 32-bit integer divisions.
__device__ __forceinline__ void divmod_u24(int v, int d, int &q, int &r) {
    q = (int)((float)v * (1.0f / (float)d));
    r = v - q * d;
    if (r >= d) { r -= d; ++q; }
    if (r < 0) { r += d; --q; }
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float sgpr_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}


constexpr int TH = 8, TW = 8;     // tile of gradient-map cells owned by one workgroup (4 waves = 2 x 2 patches)
constexpr int PT = 4;             // a wave owns a PT x PT patch of cells (accumulators in registers)
constexpr int PB = 16;            // max pooled bins per axis on the fast backward paths (7 and 14 in the model)
constexpr int BWD_THREADS = 256;  // 4 waves = 2 x 2 patches
constexpr int BWD_WAVES = BWD_THREADS / 64;
constexpr int CCH = 256;          // channels per pass: lane = 4 channels

#define MRCNN_FMA4(A, c, g)                \
    A.x = fmaf(c, g.x, A.x); A.y = fmaf(c, g.y, A.y); \
    A.z = fmaf(c, g.z, A.z); A.w = fmaf(c, g.w, A.w);

// RoI split of a level: enough workgroups to occupy the chip when the level has few tiles.
inline int level_split(int H, int W, int N) {
    const int tiles = mrcnn::cdiv(W, TW) * mrcnn::cdiv(H, TH) * N;
    if (tiles >= 256) return 1;
    return std::min(32, mrcnn::cdiv(512, tiles));
}


}  // namespace mrcnn_roi

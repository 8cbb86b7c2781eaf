// ROIAlign backward, variant 3: table-driven owner-computes scatter for gfx950 (no global float atomics on the data path).
//
// Replaces the backward of chainer_maskrcnn.functions.roi_align.roi_align_2d (un-vendored submodule, called through
// chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7) and of the per-RoI loops of
// chainer_maskrcnn/model/head/fpn_roi_mask_head.py:59-61,75-77 - same results as variant 2 (roi_align.hip), bit for bit.
//
// Why a third variant.  In variant 2 every wave (one 4 x 4-cell patch x 256 channels) re-derives what it needs from the raw
// RoI list: 512 RoI rows loaded and turned into boxes per workgroup, a 512-RoI scan per wave, axis_sample() arithmetic per
// (candidate, bin, sample) per wave, ballot-compacted queue entries in LDS - ~1500 of a wave's ~2200 VALU instructions and
// 15 k of its 38 k cycles before the first gradient row is touched; and with 127 VGPRs it can keep only 4 gy rows in flight,
// so the drain of the most loaded patch (the kernel's critical path: all waves are resident at once) runs at memory latency / 4.
// Here the geometry is computed ONCE per RoI by a small first kernel and the patch kernel is a lean gather:
//
//   k_bwd3_tables   one wave per (RoI, axis).  Lane = pooled bin: its `sr` samples via the forward's axis_sample() (bit-identical
//                   indices and weights).  For every block of 4 map rows (columns) the RoI touches, the bins with weight
//                   on it form a contiguous run; the wave emits  rd[block - b0] = (first bin, run length, offset)  and one
//                   float4 per bin of the run = the summed sample weights on the block's 4 rows (already / sr^2 on the y
//                   axis), plus a 16-byte header (block range per axis, level, image).  Bounded: <= 2 sr P float4 per axis.
//   k_bwd3_patches  one wave per patch, no workgroup barrier.  (1) lane-parallel scan of the headers: exact block-range test,
//                   ballot-compacted candidate list (ascending RoI order); (2) lane = candidate: its two run descriptors,
//                   entry count = ny * nx, wave prefix sum; (3) lane = entry: binary search of its candidate, (ph, pw), the
//                   gy row index and the two weight vectors (two 16-byte loads) - these nine registers ARE the queue, no LDS
//                   round trip; (4) drain with v_readlane as in variant 2, but 8 gy rows in flight (the freed registers).
//
// Entry order (RoI, ph, pw) and the arithmetic of every weight and FMA are those of variant 2, hence identical bits.
#include "common.h"

#pragma clang fp contract(off)

#include "roi_align_common.h"

using namespace mrcnn_roi;

namespace {

constexpr int B3_SEG = 512;      // RoIs scanned per segment (candidate positions are 16-bit)
constexpr int B3_SRMAX = 4;      // sampling ratios served (the reference uses 2)
constexpr int B3_NBUF = 8;       // gy rows in flight per wave
// Measurement knobs (mrcnn_debug_roi_align_bwd3_knobs): extra dynamic LDS per workgroup (caps the resident workgroups per CU, so
// that part of the grid is dispatched as earlier workgroups finish - dynamic load balance) and s_setprio for heavy waves.
int g_b3_pad_lds = 0, g_b3_prio = 0;

struct B3Layout {
    size_t hdr, rd, tw, total;
    int maxb, capt;
};

B3Layout b3_layout(const int *Hs, const int *Ws, int L, int R, int PH, int PW, int sr) {
    B3Layout o{};
    int maxb = 1;
    for (int l = 0; l < L; ++l) maxb = std::max(maxb, std::max(mrcnn::cdiv(Hs[l], PT), mrcnn::cdiv(Ws[l], PT)));
    o.maxb = maxb;
    o.capt = 2 * sr * std::max(PH, PW);          // a sample's two taps touch at most two blocks
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t p = 0;
    o.hdr = p; p += al((size_t)R * 16);
    o.rd = p; p += al((size_t)R * 2 * maxb * 4);
    o.tw = p; p += al((size_t)R * 2 * o.capt * 16);
    o.total = p;
    return o;
}

bool b3_ok(const int *Hs, const int *Ws, int L, int R, int PH, int PW, int sr) {
    if (sr < 1 || sr > B3_SRMAX || PH > PB || PW > PB || R < 1 || R >= (1 << 22)) return false;
    const B3Layout o = b3_layout(Hs, Ws, L, R, PH, PW, sr);
    // 32-bit byte offsets into the three tables, 15-bit block indices, 16-bit run offsets
    return o.maxb < (1 << 15) && o.capt < (1 << 16) && (unsigned long long)R * 2 * o.maxb * 4 < (1ull << 32) &&
           (unsigned long long)R * 2 * o.capt * 16 < (1ull << 32);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- kernel 1: per-RoI tables -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bwd3_tables(Levels lv, const float *__restrict__ rois, const int32_t *__restrict__ levels,
                                                     int R, int N, int PH, int PW, int sr, int *__restrict__ hdr,
                                                     unsigned *__restrict__ rd, float4 *__restrict__ tw, int maxb, int capt) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (int)(threadIdx.x >> 6);        // (RoI, axis): axis 0 = y (rows), 1 = x (columns)
    if (wid >= 2 * R) return;
    const int r = wid >> 1, axis = wid & 1;
    int l = levels ? levels[r] : 0;
    l = __builtin_amdgcn_readfirstlane(min(max(l, 0), lv.L - 1));
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, lv.scale[l], PH, PW, sr);
    const bool bad = g.n < 0 || g.n >= N;
    const int P = axis ? PW : PH, size = axis ? lv.W[l] : lv.H[l];
    const float start = axis ? g.x1f : g.y1f, bsz = axis ? g.bw : g.bh;
    const bool tv = lane < P && !bad;
    Samp sp[B3_SRMAX];
    int cmin = 0x7fffffff, cmax = -1;
#pragma unroll
    for (int i = 0; i < B3_SRMAX; ++i) {
        sp[i].lo = sp[i].hi = -1; sp[i].wl = sp[i].wh = 0.0f;
        if (i < sr) {
            sp[i] = axis_sample(start, bsz, min(lane, P - 1), i, sr, size);
            if (tv && sp[i].lo >= 0) { cmin = min(cmin, sp[i].lo); cmax = max(cmax, sp[i].hi); }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        cmin = min(cmin, __shfl_xor(cmin, o, 64));
        cmax = max(cmax, __shfl_xor(cmax, o, 64));
    }
    int b0 = cmin >> 2, b1 = cmax >> 2;
    if (cmax < 0) { b0 = 1; b1 = 0; }            // nothing lands on the map (or a bad image index): an empty block range
    b0 = __builtin_amdgcn_readfirstlane(b0); b1 = __builtin_amdgcn_readfirstlane(b1);
    unsigned *rdo = rd + (size_t)wid * maxb;
    float4 *two = tw + (size_t)wid * capt;
    const float sc = axis == 0 ? 1.0f / (float)(sr * sr) : 1.0f;
    int cursor = 0;
    for (int b = b0; b <= b1; ++b) {
        // the summed sample weights of this lane's bin on the block's 4 rows / columns: the expression (and the order of
        // its additions) of variant 2's table pass
        float w[PT] = {0.f, 0.f, 0.f, 0.f};
        const int t0 = b * PT;
#pragma unroll
        for (int i = 0; i < B3_SRMAX; ++i)
            if (i < sr) {
                const int dl = sp[i].lo - t0, dh = sp[i].hi - t0;
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    w[j] += (tv && sp[i].lo >= 0 && dl == j) ? sp[i].wl : 0.0f;
                    w[j] += (tv && sp[i].hi >= 0 && dh == j) ? sp[i].wh : 0.0f;
                }
            }
#pragma unroll
        for (int j = 0; j < PT; ++j) w[j] *= sc;
        const unsigned long long bal = __ballot(w[0] != 0.f || w[1] != 0.f || w[2] != 0.f || w[3] != 0.f);
        unsigned d = 0;
        if (bal) {
            const int first = __builtin_ctzll(bal), last = 63 - __builtin_clzll(bal), cnt = last - first + 1;
            if (cursor + cnt <= capt) {          // (always: a sample's taps touch at most two blocks)
                d = (unsigned)first | ((unsigned)cnt << 8) | ((unsigned)cursor << 16);
                if (lane >= first && lane <= last) two[cursor + lane - first] = make_float4(w[0], w[1], w[2], w[3]);
                cursor += cnt;
            }
        }
        if (lane == 0) rdo[b - b0] = d;
    }
    if (lane == 0) {
        int *h = hdr + (size_t)r * 4;
        h[axis] = (b0 & 0xffff) | (b1 << 16);
        if (axis == 0) h[2] = bad ? -1 : (l | (g.n << 8));
    }
}

// ---- kernel 2: one wave per patch ---------------------------------------------------------------------------------------
struct B3Wave {
    unsigned cand[B3_SEG][2];        // (position in the segment | jy << 16, jx): candidates of the segment, ascending
    int rec[64][4];                  // per candidate of a batch: first entry, RoI, ph0 | pw0 << 8 | nx << 16, offy | offx << 16
};

__device__ __forceinline__ float readlane_fs(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// 64 entries held one per lane (qrow, qwy, qwx) applied to the 16 accumulators: NBUF gy rows in flight.
template <int NBUF>
__device__ __forceinline__ void b3_drain(int m, int qrow, const float4 qwy, const float4 qwx, __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                         unsigned row_bytes, float4 (&acc)[PT][PT]) {
    auto ldrow = [&](int j) -> float4 {
        const unsigned so = (unsigned)__builtin_amdgcn_readlane(qrow, j < m ? j : 0) * row_bytes;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
        float4 f;
        __builtin_memcpy(&f, &v, 16);
        return f;
    };
    float4 buf[NBUF];
    // the first loads issue in slot order (vmcnt counts in issue order; see drain_wave_queue of variant 2)
#pragma unroll
    for (int d = 0; d < NBUF; ++d) {
        buf[d] = ldrow(d);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma nounroll
    for (int j = 0; j < m; j += NBUF) {
#pragma unroll
        for (int d = 0; d < NBUF; ++d) {
            if (j + d < m) {
                const int e = j + d;
                const int wyb[PT] = {__builtin_amdgcn_readlane(__float_as_int(qwy.x), e), __builtin_amdgcn_readlane(__float_as_int(qwy.y), e),
                                     __builtin_amdgcn_readlane(__float_as_int(qwy.z), e), __builtin_amdgcn_readlane(__float_as_int(qwy.w), e)};
                const float wx[PT] = {readlane_fs(qwx.x, e), readlane_fs(qwx.y, e), readlane_fs(qwx.z, e), readlane_fs(qwx.w, e)};
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    if (wyb[i] != 0) {               // wave-uniform (weights are >= 0: an integer test of the SGPR)
                        const float wyi = __int_as_float(wyb[i]);
                        const float4 t = make_float4(buf[d].x * wyi, buf[d].y * wyi, buf[d].z * wyi, buf[d].w * wyi);
#pragma unroll
                        for (int k = 0; k < PT; ++k) { MRCNN_FMA4(acc[i][k], wx[k], t) }
                    }
                }
            }
            buf[d] = ldrow(j + d + NBUF);            // refill this slot for the next round
            asm volatile("" ::: "memory");
        }
    }
}

// STAMP: diagnostic build (mrcnn_debug_roi_align_bwd3_stamps): s_memtime per phase of every wave, 8 x u64 per wave:
// start, scan cycles, descriptor cycles, entry-generation cycles, drain cycles, store start, end, entries.
__device__ __forceinline__ unsigned long long b3_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <bool STAMP>
__global__ __launch_bounds__(BWD_THREADS, 4) void k_bwd3_patches(Levels lv, const float *__restrict__ gy, int R, int N, int C, int PH, int PW,
                                                                 int chunk, int accumulate, const int *__restrict__ hdr,
                                                                 const unsigned *__restrict__ rd, const float4 *__restrict__ tw,
                                                                 int maxb, int capt, unsigned long long *__restrict__ stamps, int prio) {
    extern __shared__ float b3_pad[];            // occupancy pad only (never touched)
    unsigned long long s0 = 0, s_scan = 0, s_desc = 0, s_gen = 0, s_drain = 0, s_t = 0, s_ent = 0;
    if (STAMP) s0 = b3_now();
    __shared__ __attribute__((aligned(16))) B3Wave lds_all[BWD_WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    B3Wave &lds = lds_all[wave];
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);       // XCD-banded tile order (speed only)
    if ((int)(blockIdx.x >> 3) >= chunk || tile_id >= lv.tile_begin[lv.L]) return;
    int l = 0;
    while (l + 1 < lv.L && tile_id >= lv.tile_begin[l + 1]) ++l;
    int t = tile_id - lv.tile_begin[l];
    const int nsplit = lv.split[l];
    int zsplit = 0, n, tyi, txi;
    if (nsplit > 1) divmod_u24(t, nsplit, t, zsplit);
    divmod_u24(t, lv.tiles_x[l] * lv.tiles_y[l], n, t);
    divmod_u24(t, lv.tiles_x[l], tyi, txi);
    const int by = tyi * 2 + (wave >> 1), bx = txi * 2 + (wave & 1);         // this wave's block of 4 rows / 4 columns
    const int py0 = by * PT, px0 = bx * PT;
    const int H = lv.H[l], W = lv.W[l];
    const int nrow = min(PT, H - py0), ncol = min(PT, W - px0);
    if (nrow <= 0 || ncol <= 0) return;          // (no workgroup barrier in this kernel: a wave may leave on its own)
    float *gxb = (nsplit > 1 ? lv.slab[l] + (size_t)zsplit * N * H * W * C : lv.gx[l]) + (size_t)n * H * W * C;
    if (nsplit > 1) accumulate = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const auto rs_gy = __builtin_amdgcn_make_buffer_rsrc((void *)gy, 0, (unsigned)((size_t)R * PH * PW * C * 4), 0x00020000);
    const auto rs_gx = __builtin_amdgcn_make_buffer_rsrc((void *)gxb, 0, (unsigned)((size_t)H * W * C * 4), 0x00020000);
    const auto rs_hdr = __builtin_amdgcn_make_buffer_rsrc((void *)hdr, 0, (unsigned)R * 16u, 0x00020000);
    const auto rs_rd = __builtin_amdgcn_make_buffer_rsrc((void *)rd, 0, (unsigned)((size_t)R * 2 * maxb * 4), 0x00020000);
    const auto rs_tw = __builtin_amdgcn_make_buffer_rsrc((void *)tw, 0, (unsigned)((size_t)R * 2 * capt * 16), 0x00020000);
    const unsigned row_bytes = (unsigned)C * 4u;
    const unsigned OOB = 0xFFFFFFFFu;
    const unsigned patch_off = (unsigned)(((size_t)py0 * W + px0) * C * 4);
    const int key = l | (n << 8);
#pragma nounroll
    for (int cb = 0; cb < C; cb += CCH) {
        const bool act = cb + lane * 4 < C;
        const unsigned vlane = act ? (unsigned)(cb + lane * 4) * 4u : OOB;
        float4 acc[PT][PT];
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int k = 0; k < PT; ++k) acc[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        bool touched = false;
#pragma nounroll
        for (int seg = 0; seg < R; seg += B3_SEG) {
            // ---- (1) scan the headers of the segment: eight 16-byte loads per lane, issued together
            if (STAMP) s_t = b3_now();
            const int seg_n = min(B3_SEG, R - seg);
            u32x4 hv[B3_SEG / 64];
#pragma unroll
            for (int g = 0; g < B3_SEG / 64; ++g) {
                const int i = seg + g * 64 + lane;
                hv[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_hdr, (g * 64 + lane) < seg_n ? (unsigned)i * 16u : OOB, 0, 0);
            }
            int nlist = 0;
#pragma unroll
            for (int g = 0; g < B3_SEG / 64; ++g) {
                if (g * 64 >= seg_n) break;
                const int hy = (int)hv[g].x, hx = (int)hv[g].y;
                const int b0y = hy & 0xffff, b1y = hy >> 16, b0x = hx & 0xffff, b1x = hx >> 16;
                bool f = ((g * 64 + lane) < seg_n) & ((int)hv[g].z == key) & (by >= b0y) & (by <= b1y) & (bx >= b0x) & (bx <= b1x);
                if (nsplit > 1) f = f && ((seg + g * 64 + lane) % nsplit == zsplit);
                const unsigned long long bal = __ballot(f);
                if (f) {
                    const int pos = nlist + __popcll(bal & lt_mask);
                    lds.cand[pos][0] = (unsigned)(g * 64 + lane) | ((unsigned)(by - b0y) << 16);
                    lds.cand[pos][1] = (unsigned)(bx - b0x);
                }
                nlist += __popcll(bal);
            }
            __builtin_amdgcn_wave_barrier();
            if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s_scan += b3_now() - s_t; }
            // ---- (2) batches of 64 candidates: run descriptors, entry counts, prefix sum
#pragma nounroll
            for (int c0 = 0; c0 < nlist; c0 += 64) {
                if (STAMP) s_t = b3_now();
                const bool cv = c0 + lane < nlist;
                const int ci = min(c0 + lane, nlist - 1);
                const unsigned ca = lds.cand[ci][0], jx = lds.cand[ci][1];
                const int ri = seg + (int)(ca & 0xffffu);
                const unsigned jy = ca >> 16;
                const unsigned rdy = __builtin_amdgcn_raw_buffer_load_b32(rs_rd, cv ? (unsigned)((ri * 2 + 0) * maxb + (int)jy) * 4u : OOB, 0, 0);
                const unsigned rdx = __builtin_amdgcn_raw_buffer_load_b32(rs_rd, cv ? (unsigned)((ri * 2 + 1) * maxb + (int)jx) * 4u : OOB, 0, 0);
                const int ny = (int)((rdy >> 8) & 0xffu), nx = (int)((rdx >> 8) & 0xffu);
                const int cnt = cv ? ny * nx : 0;
                int end = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(end, o, 64);
                    if (lane >= o) end += up;
                }
                const int T = __builtin_amdgcn_readlane(end, 63);
                if (T == 0) continue;
                touched = true;
                if (prio && c0 == 0 && seg == 0) {           // heavy waves run at their own pace, the light ones fill in
                    if (T >= 56) __builtin_amdgcn_s_setprio(3);
                    else if (T >= 40) __builtin_amdgcn_s_setprio(2);
                    else if (T >= 28) __builtin_amdgcn_s_setprio(1);
                }
                __builtin_amdgcn_wave_barrier();                 // the previous batch's records have been read
                lds.rec[lane][0] = end - cnt;
                lds.rec[lane][1] = ri;
                lds.rec[lane][2] = (int)((rdy & 0xffu) | ((rdx & 0xffu) << 8) | ((unsigned)nx << 16));
                lds.rec[lane][3] = (int)((rdy >> 16) | ((rdx >> 16) << 16));
                __builtin_amdgcn_wave_barrier();
                if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s_desc += b3_now() - s_t; s_ent += T; }
                // ---- (3) + (4) chunks of 64 entries: lane = entry
#pragma nounroll
                for (int e0 = 0; e0 < T; e0 += 64) {
                    if (STAMP) s_t = b3_now();
                    const int e = e0 + lane;
                    const bool ev = e < T;
                    int pos = 0;                 // the last candidate whose first entry is <= e (zero-count candidates share
                                                 // their successor's first entry and are never the last)
#pragma unroll
                    for (int s = 32; s >= 1; s >>= 1)
                        if (pos + s < 64 && lds.rec[pos + s][0] <= e) pos += s;
                    const int first = lds.rec[pos][0], rr = lds.rec[pos][1], pk = lds.rec[pos][2];
                    const unsigned offs = (unsigned)lds.rec[pos][3];
                    const int k = ev ? e - first : 0;
                    const int nxe = max((pk >> 16) & 0xff, 1);
                    const int iy = (int)(((float)k + 0.5f) / (float)nxe);        // k / nxe for small k (the quotient sits >= 1/32 from an integer)
                    const int ix = k - iy * nxe;
                    const int ph = (pk & 0xff) + iy, pw = ((pk >> 8) & 0xff) + ix;
                    const int qrow = ev ? (rr * PH + ph) * PW + pw : 0;
                    const unsigned oy = (unsigned)((rr * 2 + 0) * capt) + (offs & 0xffffu) + (unsigned)iy;
                    const unsigned ox = (unsigned)((rr * 2 + 1) * capt) + (offs >> 16) + (unsigned)ix;
                    const u32x4 vy = __builtin_amdgcn_raw_buffer_load_b128(rs_tw, ev ? oy * 16u : OOB, 0, 0);
                    const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rs_tw, ev ? ox * 16u : OOB, 0, 0);
                    const float4 qwy = make_float4(__uint_as_float(vy.x), __uint_as_float(vy.y), __uint_as_float(vy.z), __uint_as_float(vy.w));
                    const float4 qwx = make_float4(__uint_as_float(vx.x), __uint_as_float(vx.y), __uint_as_float(vx.z), __uint_as_float(vx.w));
                    if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t1 = b3_now(); s_gen += t1 - s_t; s_t = t1; }
                    b3_drain<B3_NBUF>(min(64, T - e0), qrow, qwy, qwx, rs_gy, vlane, row_bytes, acc);
                    if (STAMP) s_drain += b3_now() - s_t;
                }
            }
            __builtin_amdgcn_wave_barrier();             // the next segment overwrites the candidate list
        }
        unsigned long long s_store = 0;
        if (STAMP) s_store = b3_now();
        if (accumulate) {
            // gx += acc for the patches that received an entry, with no-return float atomics: this wave is the ONLY writer of
            // its cells, so the result is the single IEEE addition old + acc (see variant 2); idle lanes are branched around
            // (an out-of-range buffer atomic raises a hardware exception)
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
        } else {
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int k = 0; k < PT; ++k)
                    if (i < nrow && k < ncol)                // wave-uniform
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&acc[i][k]), rs_gx, vlane,
                                                               patch_off + (unsigned)((i * W + k) * C) * 4u, 0);
        }
        if (STAMP && stamps && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long *o = stamps + ((size_t)blockIdx.x * BWD_WAVES + wave) * 8;
            o[0] = s0; o[1] = s_scan; o[2] = s_desc; o[3] = s_gen; o[4] = s_drain; o[5] = s_store; o[6] = b3_now(); o[7] = s_ent;
        }
    }
}

}  // namespace

extern "C" int mrcnn_debug_roi_align_bwd3_knobs(int pad_lds_bytes, int prio) {
    if (pad_lds_bytes < 0 || pad_lds_bytes > 120 * 1024) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_roi_align_bwd3_knobs: pad in [0, 120 KiB]");
    g_b3_pad_lds = pad_lds_bytes; g_b3_prio = prio;
    return 0;
}

namespace mrcnn_roi {

size_t bwd3_ws_bytes(const int *Hs, const int *Ws, int L, int R, int PH, int PW, int sr) {
    if (!Hs || !Ws || L <= 0 || L > MRCNN_MAX_LEVELS || !b3_ok(Hs, Ws, L, R, PH, PW, sr)) return 0;
    return b3_layout(Hs, Ws, L, R, PH, PW, sr).total;
}

int launch_bwd3(const Levels &lv, int total_tiles, int N, const float *gy, const float *rois, const int32_t *levels, int R, int C,
                int PH, int PW, int sr, int accumulate, void *ws, size_t ws_bytes, hipStream_t st, unsigned long long *stamps) {
    if (!b3_ok(lv.H, lv.W, lv.L, R, PH, PW, sr)) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "roi_align backward variant 3: geometry not served");
    const B3Layout o = b3_layout(lv.H, lv.W, lv.L, R, PH, PW, sr);
    if (!ws || ws_bytes < o.total) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "roi_align backward variant 3: workspace %zu < %zu", ws_bytes, o.total);
    char *base = (char *)ws;
    int *hdr = (int *)(base + o.hdr);
    unsigned *rd = (unsigned *)(base + o.rd);
    float4 *tw = (float4 *)(base + o.tw);
    hipLaunchKernelGGL(k_bwd3_tables, dim3(mrcnn::cdiv(2 * (long long)R, 4)), dim3(256), 0, st, lv, rois, levels, R, N, PH, PW, sr, hdr, rd, tw,
                       o.maxb, o.capt);
    MRCNN_LAUNCH_CHECK();
    const int chunk = mrcnn::cdiv(total_tiles, 8);
    if (stamps)
        hipLaunchKernelGGL(k_bwd3_patches<true>, dim3(chunk * 8), dim3(BWD_THREADS), g_b3_pad_lds, st, lv, gy, R, N, C, PH, PW, chunk, accumulate,
                           (const int *)hdr, (const unsigned *)rd, (const float4 *)tw, o.maxb, o.capt, stamps, g_b3_prio);
    else
        hipLaunchKernelGGL(k_bwd3_patches<false>, dim3(chunk * 8), dim3(BWD_THREADS), g_b3_pad_lds, st, lv, gy, R, N, C, PH, PW, chunk, accumulate,
                           (const int *)hdr, (const unsigned *)rd, (const float4 *)tw, o.maxb, o.capt, (unsigned long long *)nullptr, g_b3_prio);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace mrcnn_roi

// RPN proposal path on the device for gfx950: head-output packing, anchor decode + clip + min-size
// filter, top-n_pre selection by score, greedy NMS (bit-mask kernel + in-kernel sequential reduce),
// FPN level assignment.  No device->host copy anywhere on the path.
//
// Replaces, per image (chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:133-164):
//   the transpose/reshape/concat copies (:133-152)                  -> mrcnn_rpn_pack_f32 / _unpack_grad_
//   ChainerCV ProposalCreator.__call__ (third-party; in-tree mirror utils/proposal_creator.py:108-169)
//     loc2bbox, clip, min-size filter, argsort()[::-1][:n_pre], non_maximum_suppression, [:n_post]
//                                                                    -> mrcnn_rpn_proposals_f32
//   map_rois_to_fpn_levels (:16-31)                                  -> mrcnn_map_rois_to_fpn_levels_f32
//
// Float arithmetic is compiled with contraction off and follows oracle/boxes.py and
// oracle/proposal.py operation for operation.  Sort order is the oracle's pin: (score desc, anchor
// index desc).  NMS keep lists are bit-exact given identical boxes.  All kernels are latency/HBM
// bound; sizes at config 3: A = 261,888 anchors, n_pre = 12,000 (18 MB of bit masks), n_post = 2,000.
#include "common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

// ---- pack / unpack the RPN head output ---------------------------------------------------------
// head (N, HW, Cp) NHWC with channels [0,4A) = loc (a*4+k), [4A,6A) = score (a*2+c).
// locs (N, Atot, 4), scores (N, Atot, 2); this level's anchors start at a_off; anchor index within
// the level = pos*A + a  (position-major, anchor-minor: multilevel_region_proposal_network.py:133-141).
__global__ __launch_bounds__(256) void k_rpn_pack(const float *__restrict__ head, int N, int HW, int Cp, int A,
                                                  float *__restrict__ locs, float *__restrict__ scores, int a_off,
                                                  int Atot) {
    const long long total = (long long)N * HW * A;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int a = (int)(i % A);
        const long long q = i / A;
        const int pos = (int)(q % HW), n = (int)(q / HW);
        const float *h = head + ((size_t)n * HW + pos) * Cp;
        const float4 l = *reinterpret_cast<const float4 *>(h + a * 4);
        const float2 s = *reinterpret_cast<const float2 *>(h + 4 * A + a * 2);
        const size_t o = (size_t)n * Atot + a_off + (size_t)pos * A + a;
        *reinterpret_cast<float4 *>(locs + o * 4) = l;
        *reinterpret_cast<float2 *>(scores + o * 2) = s;
    }
}

__global__ __launch_bounds__(256) void k_rpn_unpack(const float *__restrict__ glocs, const float *__restrict__ gscores,
                                                    int N, int HW, int Cp, int A, float *__restrict__ ghead, int a_off,
                                                    int Atot) {
    const long long total = (long long)N * HW * (Cp / 2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c2 = (int)(i % (Cp / 2));
        const long long q = i / (Cp / 2);
        const int pos = (int)(q % HW), n = (int)(q / HW);
        const int c = c2 * 2;
        float2 v = make_float2(0.f, 0.f);
        const size_t base = (size_t)n * Atot + a_off + (size_t)pos * A;
        if (c < 4 * A) v = *reinterpret_cast<const float2 *>(glocs + (base + c / 4) * 4 + (c & 3));
        else if (c < 6 * A) v = *reinterpret_cast<const float2 *>(gscores + (base + (c - 4 * A) / 2) * 2);
        *reinterpret_cast<float2 *>(ghead + ((size_t)n * HW + pos) * Cp + c) = v;
    }
}

// (r6) All pyramid levels in ONE launch each way: the five per-level launches were five dependent 6-us kernels on the forward pass's
// critical stream.  Level l's HW positions start at pos0[l] of the concatenated position axis; its anchors at pos0[l] * A.
constexpr int RPN_MAX_LEVELS = 8;
struct RpnLevels { const float *head[RPN_MAX_LEVELS]; float *ghead[RPN_MAX_LEVELS]; int pos0[RPN_MAX_LEVELS + 1]; int L; };
__global__ __launch_bounds__(256) void k_rpn_pack_levels(RpnLevels lv, int N, int Cp, int A, float *__restrict__ locs, float *__restrict__ scores,
                                                         int Atot) {
    const int PT = lv.pos0[lv.L];                               // positions of all levels
    const long long total = (long long)N * PT * A;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int a = (int)(i % A);
        const long long q = i / A;
        const int gp = (int)(q % PT), n = (int)(q / PT);
        int l = 0;
        while (l + 1 < lv.L && gp >= lv.pos0[l + 1]) ++l;
        const int pos = gp - lv.pos0[l], HW = lv.pos0[l + 1] - lv.pos0[l];
        const float *h = lv.head[l] + ((size_t)n * HW + pos) * Cp;
        const float4 v = *reinterpret_cast<const float4 *>(h + a * 4);
        const float2 sc = *reinterpret_cast<const float2 *>(h + 4 * A + a * 2);
        const size_t o = (size_t)n * Atot + (size_t)gp * A + a;
        *reinterpret_cast<float4 *>(locs + o * 4) = v;
        *reinterpret_cast<float2 *>(scores + o * 2) = sc;
    }
}
__global__ __launch_bounds__(256) void k_rpn_unpack_levels(RpnLevels lv, const float *__restrict__ glocs, const float *__restrict__ gscores, int N,
                                                           int Cp, int A, int Atot) {
    const int PT = lv.pos0[lv.L];
    const long long total = (long long)N * PT * (Cp / 2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c2 = (int)(i % (Cp / 2));
        const long long q = i / (Cp / 2);
        const int gp = (int)(q % PT), n = (int)(q / PT);
        int l = 0;
        while (l + 1 < lv.L && gp >= lv.pos0[l + 1]) ++l;
        const int pos = gp - lv.pos0[l], HW = lv.pos0[l + 1] - lv.pos0[l];
        const int c = c2 * 2;
        float2 v = make_float2(0.f, 0.f);
        const size_t base = (size_t)n * Atot + (size_t)gp * A;
        if (c < 4 * A) v = *reinterpret_cast<const float2 *>(glocs + (base + c / 4) * 4 + (c & 3));
        else if (c < 6 * A) v = *reinterpret_cast<const float2 *>(gscores + (base + (c - 4 * A) / 2) * 2);
        *reinterpret_cast<float2 *>(lv.ghead[l] + ((size_t)n * HW + pos) * Cp + c) = v;
    }
}

// ---- decode + clip + filter + sort key ---------------------------------------------------------
__device__ __forceinline__ unsigned orderable(float f) {     // monotone float -> uint
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_decode(const float *__restrict__ locs, const float *__restrict__ scores,
                                                const float *__restrict__ anchors, int N, int A, float img_h, float img_w,
                                                float min_size, const float *__restrict__ per_image, float *__restrict__ boxes,
                                                u64 *__restrict__ keys, int ib, int batched, mrcnn::TopkInit ti, int32_t *__restrict__ n_valid) {
    // (r6) the words the rest of the chain needs initialised (radix-select histograms, compaction counters, k per image, the valid
    // counters of the gather): four tiny fill launches on a latency-bound chain before
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ti.nzero; i += gridDim.x * 256) ti.zero[i] = 0u;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < ti.ncount; i += 256) ti.count[i] = 0u;
        for (int i = threadIdx.x; i < ti.nk; i += 256) ti.kreq[i] = ti.kval;
        if (n_valid) for (int i = threadIdx.x; i < N; i += 256) n_valid[i] = 0;
    }
    const long long total = (long long)N * A;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int a = (int)(i % A);
        if (per_image) {            // ragged batch: every image is clipped to its OWN size and filtered with its own min_size * scale
            const int n = (int)(i / A);
            img_h = per_image[3 * n]; img_w = per_image[3 * n + 1]; min_size = per_image[3 * n + 2];
        }
        const float4 an = *reinterpret_cast<const float4 *>(anchors + (size_t)a * 4);
        const float4 l = *reinterpret_cast<const float4 *>(locs + (size_t)i * 4);
        const float h = an.z - an.x, w = an.w - an.y;
        const float cy = an.x + 0.5f * h, cx = an.y + 0.5f * w;
        const float ncy = l.x * h + cy, ncx = l.y * w + cx;
        const float nh = expf(l.z) * h, nw = expf(l.w) * w;
        float y1 = ncy - 0.5f * nh, x1 = ncx - 0.5f * nw, y2 = ncy + 0.5f * nh, x2 = ncx + 0.5f * nw;
        y1 = fmaxf(fminf(y1, img_h), 0.f); y2 = fmaxf(fminf(y2, img_h), 0.f);
        x1 = fmaxf(fminf(x1, img_w), 0.f); x2 = fmaxf(fminf(x2, img_w), 0.f);
        *reinterpret_cast<float4 *>(boxes + (size_t)i * 4) = make_float4(y1, x1, y2, x2);
        const bool ok = (y2 - y1 >= min_size) && (x2 - x1 >= min_size);
        const float sc = scores[(size_t)i * 2 + 1];
        // [image N-1-n] | valid | orderable score (32) | anchor index (ib bits): ONE descending sort of all images =>
        // image n's keys land in segment n, ordered (score desc, index desc); invalid keys close their segment
        const u64 img = batched ? (u64)(N - 1 - (int)(i / A)) << (ib + 33) : 0ull;
        keys[i] = img | (ok ? ((1ull << (ib + 32)) | ((u64)orderable(sc) << ib) | (u64)a) : 0ull);
    }
}

// Gather the first n_pre sorted boxes; n_valid[n] = min(n_pre, #valid).
__global__ __launch_bounds__(256) void k_gather_sorted(const u64 *__restrict__ keys_sorted, const float *__restrict__ boxes,
                                                       int A, int n_pre, float *__restrict__ sboxes,
                                                       int32_t *__restrict__ sidx, int32_t *__restrict__ n_valid, int ib,
                                                       int ks_stride) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pre || i >= A) return;
    const u64 *ks = keys_sorted + (size_t)n * ks_stride;
    const u64 vbit = 1ull << (ib + 32);
    const u64 k = ks[i];
    if (!(k & vbit)) return;
    const int a = (int)(k & ((1ull << ib) - 1ull));
    *reinterpret_cast<float4 *>(sboxes + ((size_t)n * n_pre + i) * 4) =
        *reinterpret_cast<const float4 *>(boxes + ((size_t)n * A + a) * 4);
    sidx[(size_t)n * n_pre + i] = a;
    if (i + 1 == n_pre || i + 1 == A || !(ks[i + 1] & vbit)) n_valid[n] = i + 1;
}

// ---- NMS ---------------------------------------------------------------------------------------
// mask[(n*n_pre + i)*nblk + c] bit j: box i suppresses box c*64+j (j-th box of column block c), only
// for c*64+j > i.  IoU exactly as ChainerCV's devIoU / oracle.boxes.nms.
__global__ __launch_bounds__(64) void k_nms_mask(const float *__restrict__ sboxes, const int32_t *__restrict__ n_valid,
                                                 int n_pre, int nblk, float thresh, u64 *__restrict__ mask) {
    const int img = blockIdx.z;
    const int n = n_valid[img];
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
    __shared__ float4 cbox[64];
    const float *bx = sboxes + (size_t)img * n_pre * 4;
    const int cj = cb * 64 + threadIdx.x;
    cbox[threadIdx.x] = cj < n ? *reinterpret_cast<const float4 *>(bx + (size_t)cj * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 b = *reinterpret_cast<const float4 *>(bx + (size_t)i * 4);
    const float area_i = (b.z - b.x) * (b.w - b.y);
    u64 bits = 0ull;
    const int jmax = min(64, n - cb * 64);
    for (int j = 0; j < jmax; ++j) {
        const float4 c = cbox[j];
        const float top = fmaxf(b.x, c.x), left = fmaxf(b.y, c.y);
        const float bottom = fminf(b.z, c.z), right = fminf(b.w, c.w);
        const float hgt = fmaxf(bottom - top, 0.f), wid = fmaxf(right - left, 0.f);
        const float ai = hgt * wid;
        const float area_j = (c.z - c.x) * (c.w - c.y);
        const float u = (area_i + area_j) - ai;
        // iou >= thresh with the correctly-rounded division of the reference is decided WITHOUT the division (a dozen
        // instructions) whenever ai is outside a 4e-7 band around thresh * u: fl(ai / u) >= thresh <=> ai / u >= thresh (1 - 6e-8 ..),
        // and thresh * u is known to 6e-8.  Inside the band - and for u <= 0, where the sign rules of the division decide -
        // the wave falls back to the exact expression (rare: a whole-wave branch).
        const float p = thresh * u;
        const bool sure_yes = u > 0.f && ai > p * 1.0000004f;
        const bool sure_no = u > 0.f && ai < p * 0.9999996f;
        bool hit = sure_yes;
        if (__any(!sure_yes && !sure_no)) hit = (ai / u) >= thresh;
        if (hit && cb * 64 + j > i) bits |= 1ull << j;
    }
    mask[((size_t)img * n_pre + i) * nblk + cb] = bits;
}

// One 1024-thread workgroup per image walks the boxes in order.  The walk is a dependency chain (a box is kept iff no
// earlier KEPT box overlaps it), so what matters is the latency of one link.  Round 1 paid a memory round trip and two
// barriers per 64-box chunk (~5 us); here wave 0 - the resolver - holds, for the NMS_SC chunks of a super-chunk, the mask
// words of the next NMS_LA column blocks in registers (lane = box): it resolves the chunks back to back with scalar bit
// operations on the diagonal words, ORs the rows of the boxes it keeps into the following words with wave reductions
// (no memory access on the chain), and publishes the kept list.  The other 15 waves then OR those boxes' rows into the
// removed set for the words BEYOND the resolver's look-ahead while the resolver is already on the next super-chunk: one
// barrier per super-chunk, the memory latency off the critical path.  The removed set is maintained for a WINDOW of
// NMS_WIN words ahead only (the walk stops at n_post kept boxes, with few suppressions after n_post/64 chunks); when the
// walk approaches the window's end every box kept so far is applied to the next window's words first.
constexpr int NMS_RED_THREADS = 1024;
constexpr int NMS_WIN = 48;
constexpr int NMS_SC = 4;           // chunks per super-chunk
constexpr int NMS_LA = 8;           // column blocks the resolver holds per box: its super-chunk's own NMS_SC and the next NMS_SC

__device__ __forceinline__ u64 wave_or_u64(u64 v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo |= (unsigned)__shfl_xor((int)lo, o, 64);
        hi |= (unsigned)__shfl_xor((int)hi, o, 64);
    }
    return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)lo);
}

__global__ __launch_bounds__(NMS_RED_THREADS) void k_nms_reduce(const u64 *__restrict__ mask, const int32_t *__restrict__ n_valid,
                                                                int n_pre, int nblk, int n_post, int32_t *__restrict__ keep,
                                                                int32_t *__restrict__ n_keep) {
    constexpr int MAXW = 256;      // up to 16384 boxes
    constexpr int HELPERS = NMS_RED_THREADS - 64;
    __shared__ u64 rem[MAXW];
    __shared__ int s_list[2][NMS_SC * 64];
    __shared__ int s_cnt[2];
    __shared__ int s_kept;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = n_valid[img];
    const u64 *mk = mask + (size_t)img * n_pre * nblk;
    int32_t *kp = keep + (size_t)img * n_post;
    for (int w = tid; w < MAXW; w += NMS_RED_THREADS) rem[w] = 0ull;
    if (tid == 0) { s_kept = 0; s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    const int nb = (n + 63) / 64;
    int wend = min(nb, NMS_WIN);      // rem[] receives the helpers' contributions for words < wend
    // resolver registers: X[j][w] = row of box (g + j) * 64 + lane, column block g + w (only w >= j is ever used: the mask
    // kernel writes the upper triangle)
    u64 X[NMS_SC][NMS_LA];
    auto load_x = [&](int g) {
#pragma unroll
        for (int j = 0; j < NMS_SC; ++j)
#pragma unroll
            for (int w = 0; w < NMS_LA; ++w) {
                const int box = (g + j) * 64 + lane, word = g + w;
                X[j][w] = (w >= j && box < n && word < nb) ? mk[(size_t)box * nblk + word] : 0ull;
            }
    };
    if (wave == 0) load_x(0);
    int kept_before = 0;
    for (int g = 0, s = 0; g < nb; g += NMS_SC, ++s) {
        const int buf = s & 1;
        if (wend < nb && g + NMS_SC + NMS_LA > wend) {      // block-uniform: open the next window
            const int wnew = min(nb, wend + NMS_WIN), Wn = wnew - wend, total = kept_before * Wn;
            for (int idx = tid; idx < total; idx += 4 * NMS_RED_THREADS) {
                u64 v[4];
                int w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int id = idx + j * NMS_RED_THREADS;
                    const bool ok = id < total;
                    const int q = ok ? id / Wn : 0;
                    w[j] = wend + (ok ? id - q * Wn : 0);
                    v[j] = ok ? mk[(size_t)kp[q] * nblk + w[j]] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j]) atomicOr(&rem[w[j]], v[j]);
            }
            __syncthreads();
            wend = wnew;
        }
        if (wave == 0) {
            // ---- resolver: NMS_SC chunks back to back, everything in registers / LDS words that are already final
            u64 racc[NMS_LA];
#pragma unroll
            for (int w = 0; w < NMS_LA; ++w) racc[w] = 0ull;
            int kept = kept_before, k = 0;
#pragma unroll
            for (int j = 0; j < NMS_SC; ++j) {
                const int c = g + j;
                if (c < nb && kept < n_post) {              // wave-uniform
                    const int cnt = min(64, n - c * 64);
                    const u64 remc = rem[c];                 // wave-uniform value: make that provable (scalar loop below)
                    u64 alive = ~((((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(remc >> 32)) << 32) |
                                   (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)remc)) | racc[j]);
                    if (cnt < 64) alive &= (1ull << cnt) - 1ull;
                    const unsigned dlo = (unsigned)X[j][j], dhi = (unsigned)(X[j][j] >> 32);
                    // Only a box whose diagonal word is non-zero can change the fate of a later box of the chunk, and in
                    // score order such boxes are rare: the dependency chain visits those alone (ascending), every other
                    // live box is kept without a step of its own.
                    const u64 nz = __ballot((dlo | dhi) != 0u);
                    u64 pend = alive & nz;
                    while (pend) {
                        const int b = __builtin_ctzll(pend);
                        const u64 Db = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                                       (u64)(unsigned)__builtin_amdgcn_readlane((int)dlo, b);
                        alive &= ~Db;
                        pend &= ~Db;
                        pend &= ~(1ull << b);
                    }
                    // the boxes still alive are kept, in index order, up to n_post in total
                    const int rank = __popcll(alive & ((1ull << lane) - 1ull));
                    const bool take = ((alive >> lane) & 1ull) && kept + rank < n_post;
                    const u64 keptmask = __ballot(take);
                    if (take) {
                        kp[kept + rank] = c * 64 + lane;
                        s_list[buf][k + rank] = c * 64 + lane;
                    }
                    const int nk = __popcll(keptmask);
                    kept += nk; k += nk;
                    const bool mine = (keptmask >> lane) & 1ull;
#pragma unroll
                    for (int w = j + 1; w < NMS_LA; ++w) racc[w] |= wave_or_u64(mine ? X[j][w] : 0ull);
                }
            }
            // the look-ahead words beyond this super-chunk: merged into the removed set (the helpers may be adding to the
            // same words for older boxes: atomics)
            if (lane == 0) {
#pragma unroll
                for (int w = NMS_SC; w < NMS_LA; ++w)
                    if (g + w < nb && racc[w]) atomicOr(&rem[g + w], racc[w]);
                s_cnt[buf] = k;
                s_kept = kept;
            }
            if (g + NMS_SC < nb && kept < n_post) load_x(g + NMS_SC);      // in flight across the barrier
        } else if (s > 0) {
            // ---- helpers: the boxes kept in the PREVIOUS super-chunk, words beyond the resolver's look-ahead
            const int pb = buf ^ 1, K = s_cnt[pb], w0 = (g - NMS_SC) + NMS_LA, Wd = wend - w0;
            if (K > 0 && Wd > 0) {
                const int total = K * Wd;
                for (int idx = tid - 64; idx < total; idx += 8 * HELPERS) {
                    u64 v[8];
                    int w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int id = idx + j * HELPERS;
                        const bool ok = id < total;
                        const int q = ok ? id / Wd : 0;
                        w[j] = w0 + (ok ? id - q * Wd : 0);
                        v[j] = ok ? mk[(size_t)s_list[pb][q] * nblk + w[j]] : 0ull;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (v[j]) atomicOr(&rem[w[j]], v[j]);
                }
            }
        }
        __syncthreads();
        kept_before = s_kept;
        if (kept_before >= n_post) break;
    }
    if (tid == 0) n_keep[img] = s_kept;
}

__device__ __forceinline__ float fpn_level(float y1, float x1, float y2, float x2, float k_min, float k_max) {
    // multilevel_region_proposal_network.py:23-31: floor(4 + log2(sqrt(area)/224 + 1e-6)) clipped
    const float area = (y2 - y1) * (x2 - x1);
    const float s = sqrtf(area);
    const float v = s / 224.0f + 1e-6f;
    const float lg = (float)log2((double)v);
    const float t = floorf(4.0f + lg);
    return fminf(fmaxf(t, k_min), k_max);
}

// rois (N*n_post, 4) padded with zeros; roi_indices (N*n_post) i32 (-1 padding); levels f32.
__global__ __launch_bounds__(256) void k_emit_rois(const float *__restrict__ sboxes, const int32_t *__restrict__ keep,
                                                   const int32_t *__restrict__ n_keep, int n_pre, int n_post,
                                                   float *__restrict__ rois, int32_t *__restrict__ roi_idx,
                                                   float *__restrict__ levels) {
    const int img = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_post) return;
    const size_t o = (size_t)img * n_post + j;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    int idx = -1;
    float lv = 0.f;
    if (j < n_keep[img]) {
        b = *reinterpret_cast<const float4 *>(sboxes + ((size_t)img * n_pre + keep[o]) * 4);
        idx = img;
        lv = fpn_level(b.x, b.y, b.z, b.w, 0.f, 4.f);
    }
    *reinterpret_cast<float4 *>(rois + o * 4) = b;
    roi_idx[o] = idx;
    levels[o] = lv;
}

__global__ __launch_bounds__(256) void k_levels(const float *__restrict__ rois, int R, float k_min, float k_max,
                                                float *__restrict__ levels) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const float4 b = *reinterpret_cast<const float4 *>(rois + (size_t)i * 4);
    levels[i] = fpn_level(b.x, b.y, b.z, b.w, k_min, k_max);
}

__global__ void k_set_i32(int32_t *p, int32_t v) { *p = v; }

struct PropLayout {
    size_t boxes, keys, keys_sorted, sboxes, sidx, n_valid, mask, keep, sort_tmp, sort_tmp_bytes, total;
    int nblk;
};
size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
PropLayout prop_layout(int N, int A, int n_pre, int n_post) {
    PropLayout L{};
    size_t o = 0;
    L.nblk = (n_pre + 63) / 64;
    L.boxes = o; o += al((size_t)N * A * 16);
    L.keys = o; o += al((size_t)N * A * 8);
    L.keys_sorted = o; o += al((size_t)N * (n_pre + 1) * 8);     // the n_pre best keys per image, descending (+ a zero)
    L.sboxes = o; o += al((size_t)N * n_pre * 16);
    L.sidx = o; o += al((size_t)N * n_pre * 4);
    L.n_valid = o; o += al((size_t)N * 4);
    L.mask = o; o += al((size_t)N * n_pre * L.nblk * 8);
    L.keep = o; o += al((size_t)N * n_post * 4);
    int ib = 1;
    while ((1ll << ib) < A) ++ib;
    L.sort_tmp_bytes = mrcnn::topk_ws_bytes(N, 33 + ib, n_pre);
    L.sort_tmp = o; o += al(L.sort_tmp_bytes);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int mrcnn_rpn_pack_f32(const float *head, int N, int HW, int Cp, int A, float *locs, float *scores,
                                  int a_off, int Atot, void *stream) {
    if (!head || !locs || !scores || N <= 0 || HW <= 0 || A <= 0 || Cp < 6 * A || (Cp % 4))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "rpn_pack: bad arguments");
    const long long total = (long long)N * HW * A;
    hipLaunchKernelGGL(k_rpn_pack, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, head, N, HW, Cp, A, locs, scores, a_off, Atot);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_rpn_unpack_grad_f32(const float *glocs, const float *gscores, int N, int HW, int Cp, int A,
                                         float *ghead, int a_off, int Atot, void *stream) {
    if (!ghead || !glocs || !gscores || N <= 0 || HW <= 0 || A <= 0 || Cp < 6 * A || (Cp % 4))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "rpn_unpack_grad: bad arguments");
    const long long total = (long long)N * HW * (Cp / 2);
    hipLaunchKernelGGL(k_rpn_unpack, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, glocs, gscores, N, HW, Cp, A, ghead, a_off, Atot);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

static int rpn_levels_of(RpnLevels &lv, const float *const *heads, float *const *gheads, const int *HWs, int L, int N, int Cp, int A, int Atot,
                         const char *who) {
    if (L <= 0 || L > RPN_MAX_LEVELS || !HWs || N <= 0 || A <= 0 || Cp < 6 * A || (Cp % 4)) return mrcnn::fail_arg(MRCNN_E_INVALID, "%s: bad arguments", who);
    lv = RpnLevels{};
    lv.L = L;
    long long p = 0;
    for (int l = 0; l < L; ++l) {
        if (HWs[l] <= 0 || (heads && !heads[l]) || (gheads && !gheads[l])) return mrcnn::fail_arg(MRCNN_E_INVALID, "%s: null level / empty level", who);
        if (heads) lv.head[l] = heads[l];
        if (gheads) lv.ghead[l] = gheads[l];
        lv.pos0[l] = (int)p;
        p += HWs[l];
    }
    lv.pos0[L] = (int)p;
    if (p * A != Atot || (long long)N * p * Cp >= (1ll << 31)) return mrcnn::fail_arg(MRCNN_E_INVALID, "%s: the levels' positions x A must equal Atot", who);
    return 0;
}
extern "C" int mrcnn_rpn_pack_levels_f32(const float *const *heads, const int *HWs, int L, int N, int Cp, int A, float *locs, float *scores,
                                         int Atot, void *stream) {
    RpnLevels lv;
    if (!heads || !locs || !scores) return mrcnn::fail_arg(MRCNN_E_INVALID, "rpn_pack_levels: null pointer");
    if (int e = rpn_levels_of(lv, heads, nullptr, HWs, L, N, Cp, A, Atot, "rpn_pack_levels")) return e;
    const long long total = (long long)N * lv.pos0[L] * A;
    hipLaunchKernelGGL(k_rpn_pack_levels, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, lv, N, Cp, A,
                       locs, scores, Atot);
    MRCNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int mrcnn_rpn_unpack_grad_levels_f32(const float *glocs, const float *gscores, float *const *gheads, const int *HWs, int L, int N,
                                                int Cp, int A, int Atot, void *stream) {
    RpnLevels lv;
    if (!gheads || !glocs || !gscores) return mrcnn::fail_arg(MRCNN_E_INVALID, "rpn_unpack_grad_levels: null pointer");
    if (int e = rpn_levels_of(lv, nullptr, gheads, HWs, L, N, Cp, A, Atot, "rpn_unpack_grad_levels")) return e;
    const long long total = (long long)N * lv.pos0[L] * (Cp / 2);
    hipLaunchKernelGGL(k_rpn_unpack_levels, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, lv, glocs,
                       gscores, N, Cp, A, Atot);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t mrcnn_rpn_proposals_workspace_bytes(int N, int A, int n_pre, int n_post) {
    if (N <= 0 || A <= 0 || n_pre <= 0 || n_post <= 0) return 0;
    return prop_layout(N, A, std::min(n_pre, A), n_post).total;
}

extern "C" int mrcnn_rpn_proposals_f32(const float *locs, const float *scores, const float *anchors, int N, int A,
                                       float img_h, float img_w, float min_size, const float *per_image, int n_pre,
                                       int n_post, float nms_thresh, float *rois, int32_t *roi_indices, float *levels,
                                       int32_t *n_rois, int32_t *dbg_sorted_anchor, int32_t *dbg_keep,
                                       int32_t *dbg_n_pre, void *ws, size_t ws_bytes, void *stream) {
    if (!locs || !scores || !anchors || !rois || !roi_indices || !levels || !n_rois || !ws)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "rpn_proposals: null pointer");
    if (N <= 0 || A <= 0 || n_pre <= 0 || n_post <= 0 || A >= (1 << 30))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "rpn_proposals: bad sizes");
    n_pre = std::min(n_pre, A);
    if (n_pre > 16384) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "rpn_proposals: n_pre %d > 16384", n_pre);
    const PropLayout L = prop_layout(N, A, n_pre, n_post);
    if (ws_bytes < L.total) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "rpn_proposals: workspace %zu < %zu", ws_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)ws;
    float *boxes = (float *)(w + L.boxes);
    u64 *keys = (u64 *)(w + L.keys), *keys_sorted = (u64 *)(w + L.keys_sorted);
    float *sboxes = (float *)(w + L.sboxes);
    int32_t *sidx = (int32_t *)(w + L.sidx), *n_valid = (int32_t *)(w + L.n_valid), *keep = (int32_t *)(w + L.keep);
    u64 *mask = (u64 *)(w + L.mask);
    const long long total = (long long)N * A;
    int ib = 1;
    while ((1ll << ib) < A) ++ib;
    hipLaunchKernelGGL(k_decode, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st, locs, scores,
                       anchors, N, A, img_h, img_w, min_size, per_image, boxes, keys, ib, 0,
                       mrcnn::topk_init_of(N, 33 + ib, n_pre, w + L.sort_tmp), n_valid);
    MRCNN_LAUNCH_CHECK();
    // argsort()[::-1][:n_pre] per image: radix select of the n_pre-th key + compaction + bitonic sort in LDS (sort.hip); its workspace
    // words and n_valid were initialised by k_decode
    if (int e = mrcnn::top_k_sorted(keys, N, (size_t)A, 33 + ib, 1ull << (ib + 32), n_pre, keys_sorted, (size_t)n_pre + 1, w + L.sort_tmp, st, true))
        return e;
    hipLaunchKernelGGL(k_gather_sorted, dim3(mrcnn::cdiv(n_pre, 256), N), dim3(256), 0, st, keys_sorted, boxes, A, n_pre,
                       sboxes, sidx, n_valid, ib, n_pre + 1);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nms_mask, dim3(L.nblk, L.nblk, N), dim3(64), 0, st, sboxes, n_valid, n_pre, L.nblk, nms_thresh, mask);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nms_reduce, dim3(N), dim3(NMS_RED_THREADS), 0, st, mask, n_valid, n_pre, L.nblk, n_post, keep, n_rois);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_emit_rois, dim3(mrcnn::cdiv(n_post, 256), N), dim3(256), 0, st, sboxes, keep, n_rois, n_pre, n_post,
                       rois, roi_indices, levels);
    MRCNN_LAUNCH_CHECK();
    if (dbg_sorted_anchor) MRCNN_HIP_TRY(hipMemcpyAsync(dbg_sorted_anchor, sidx, sizeof(int32_t) * (size_t)N * n_pre, hipMemcpyDeviceToDevice, st));
    if (dbg_keep) MRCNN_HIP_TRY(hipMemcpyAsync(dbg_keep, keep, sizeof(int32_t) * (size_t)N * n_post, hipMemcpyDeviceToDevice, st));
    if (dbg_n_pre) MRCNN_HIP_TRY(hipMemcpyAsync(dbg_n_pre, n_valid, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));
    return 0;
}

// Stand-alone greedy NMS on boxes already in priority order (ChainerCV non_maximum_suppression without
// scores; maskrcnn.py:300 passes score-sorted boxes through the same routine).  boxes (n,4) yx.
extern "C" size_t mrcnn_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    const size_t nblk = (n + 63) / 64;
    return al((size_t)n * nblk * 8) + 256;
}
extern "C" int mrcnn_nms_f32(const float *boxes, int n, float thresh, int max_keep, int32_t *keep, int32_t *n_keep,
                             void *ws, size_t ws_bytes, void *stream) {
    if (!keep || !n_keep || !ws || n < 0 || max_keep <= 0 || (n > 0 && !boxes))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "nms: bad arguments");
    if (n > 16384) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "nms: n %d > 16384", n);
    if (ws_bytes < mrcnn_nms_workspace_bytes(n)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "nms: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = std::max(1, (n + 63) / 64);
    int32_t *n_valid = (int32_t *)ws;
    u64 *mask = (u64 *)((char *)ws + 256);
    hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, st, n_valid, (int32_t)n);
    MRCNN_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(k_nms_mask, dim3(nblk, nblk, 1), dim3(64), 0, st, boxes, n_valid, n, nblk, thresh, mask);
        MRCNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(NMS_RED_THREADS), 0, st, mask, n_valid, std::max(n, 1), nblk, max_keep, keep, n_keep);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// ChainerCV's single-level RegionProposalNetwork ranks proposals by the SOFTMAX foreground probability of the two
// class scores (the multi-level RPN of this repo's reference uses the raw foreground logit): out (M,2) = softmax(in (M,2)).
namespace {
__global__ __launch_bounds__(256) void k_softmax2(const float *__restrict__ in, float *__restrict__ out, size_t M) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float a = in[2 * i], b = in[2 * i + 1];
    const float m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    const float s = ea + eb;
    out[2 * i] = ea / s;
    out[2 * i + 1] = eb / s;
}
}  // namespace

extern "C" int mrcnn_softmax2_f32(const float *in, float *out, size_t M, void *stream) {
    if (M > 0 && (!in || !out)) return mrcnn::fail_arg(MRCNN_E_INVALID, "softmax2: null pointer");
    if (M == 0) return 0;
    hipLaunchKernelGGL(k_softmax2, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, M);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_map_rois_to_fpn_levels_f32(const float *rois, int R, int k_min, int k_max, float *levels,
                                                void *stream) {
    if (R == 0) return 0;
    if (!rois || !levels || R < 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "map_rois_to_fpn_levels: bad arguments");
    hipLaunchKernelGGL(k_levels, dim3(mrcnn::cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, rois, R, (float)k_min,
                       (float)k_max, levels);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

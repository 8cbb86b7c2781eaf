// Selection and sorting of 64-bit keys for the proposal path and the samplers - hand-written for gfx950, no library.
//
//   select_kth   per segment (image), the k-th largest (or smallest) of `seg_len` unsigned 64-bit keys, k read from device
//                memory: most-significant-digit radix select, 11 bits per level.  One launch per level: every
//                workgroup first resolves the previous level from that level's 2048-bin histogram (redundantly - 8 KB of
//                L2-resident counters - instead of a separate scan launch), then histograms the next digit of the keys
//                that still match the prefix (LDS counters, one integer atomicAdd per non-empty bin and workgroup).
//                Integer atomics only: the result does not depend on arrival order.
//   top_k_sorted the proposal path's "argsort()[::-1][:n_pre]" (utils/proposal_creator.py:144-148): select the n_pre-th
//                largest composite key, compact the keys >= it (at most n_pre, keys are unique), sort them descending in
//                LDS (bitonic, one workgroup per image, <= 16384 keys = 128 KB of the CU's 160 KB).
// Replaces the device radix sort of ALL 261,888 keys per image that round 1 called from rocPRIM: 5 + 2 launches that
// read the keys instead of ~16 that read and write them.
#include "common.h"
#include <algorithm>

namespace mrcnn {
namespace {

typedef unsigned long long u64;
constexpr int RS_BITS = 11, RS_BINS = 1 << RS_BITS;      // digit width
constexpr int RS_THREADS = 256;

struct SelState {          // per segment, in the workspace
    u64 prefix;            // digits resolved so far (in place, lower bits zero)
    unsigned k;            // rank still to find inside the matching keys (1-based)
    unsigned pad;
};

// Resolve one level from its histogram: digit d with (keys with a better digit) < k <= (those + hist[d]); `descending`
// walks the bins from the top.  All threads of the block return the same (d, k').
__device__ __forceinline__ void resolve_level(const unsigned *__restrict__ hist, unsigned k, bool descending, unsigned *sscan, int &d_out,
                                              unsigned &k_out) {
    const int t = threadIdx.x;
    constexpr int PER = RS_BINS / RS_THREADS;           // 8 consecutive bins (in walk order) per thread
    unsigned c[PER], tot = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int w = t * PER + j;                        // position in walk order
        c[j] = hist[descending ? RS_BINS - 1 - w : w];
        tot += c[j];
    }
    sscan[t] = tot;
    __syncthreads();
    // exclusive prefix of the per-thread totals (256 values: serial per thread over LDS is fine at this size)
    unsigned before = 0;
    for (int i = 0; i < t; ++i) before += sscan[i];
    __shared__ int s_d;
    __shared__ unsigned s_k;
    if (t == 0) { s_d = descending ? 0 : RS_BINS - 1; s_k = 1; }      // k beyond the population: the extreme digit
    __syncthreads();
    unsigned run = before;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (k > run && k <= run + c[j]) {
            const int w = t * PER + j;
            s_d = descending ? RS_BINS - 1 - w : w;
            s_k = k - run;
        }
        run += c[j];
    }
    __syncthreads();
    d_out = s_d;
    k_out = s_k;
    __syncthreads();
}

// level l examines bits [top - 11(l+1), top - 11 l) of the key, top = number of significant key bits rounded up to a
// multiple of 11 (keys are < 2^top).
__global__ __launch_bounds__(RS_THREADS) void k_select_level(const u64 *__restrict__ keys, size_t seg_len, int level, int top,
                                                             bool descending, const unsigned *__restrict__ kreq, int kreq_stride,
                                                             SelState *__restrict__ state, unsigned *__restrict__ hist) {
    __shared__ unsigned sh[RS_BINS];
    __shared__ unsigned sscan[RS_THREADS];
    const int seg = blockIdx.y, nseg = gridDim.y, t = threadIdx.x;
    u64 prefix = 0;
    unsigned k = kreq[(size_t)seg * kreq_stride];
    if (level > 0) {
        // double-buffered records: level l reads half (l & 1), written by level l-1, and writes half ((l & 1) ^ 1)
        const SelState st = state[(size_t)nseg * (level & 1) + seg];
        int d;
        resolve_level(hist + ((size_t)(level - 1) * nseg + seg) * RS_BINS, st.k, descending, sscan, d, k);
        prefix = st.prefix | ((u64)d << (top - RS_BITS * level));
    }
    __syncthreads();
    if (blockIdx.x == 0 && t == 0)            // record for the next level (every block of the segment computes the same)
        state[(size_t)nseg * ((level & 1) ^ 1) + seg] = SelState{prefix, k, 0};
    for (int i = t; i < RS_BINS; i += RS_THREADS) sh[i] = 0;
    __syncthreads();
    const int shift = top - RS_BITS * (level + 1);
    const u64 hi_mask = level == 0 ? 0ull : ~0ull << (top - RS_BITS * level);
    const u64 *ks = keys + (size_t)seg * seg_len;
    for (size_t i = (size_t)blockIdx.x * RS_THREADS + t; i < seg_len; i += (size_t)gridDim.x * RS_THREADS) {
        const u64 key = ks[i];
        if ((key & hi_mask) == prefix) atomicAdd(&sh[(unsigned)(key >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    unsigned *gh = hist + ((size_t)level * nseg + seg) * RS_BINS;
    for (int i = t; i < RS_BINS; i += RS_THREADS)
        if (sh[i]) atomicAdd(&gh[i], sh[i]);
}

// Last step: resolve the final level -> kth[seg] = the selected key.
__global__ __launch_bounds__(RS_THREADS) void k_select_final(int levels, int top, bool descending, const SelState *__restrict__ state,
                                                             const unsigned *__restrict__ hist, u64 *__restrict__ kth, int kth_stride) {
    __shared__ unsigned sscan[RS_THREADS];
    const int seg = blockIdx.x, nseg = gridDim.x;
    const SelState st = state[(size_t)nseg * (levels & 1) + seg];
    int d;
    unsigned k;
    resolve_level(hist + ((size_t)(levels - 1) * nseg + seg) * RS_BINS, st.k, descending, sscan, d, k);
    if (threadIdx.x == 0) kth[(size_t)seg * kth_stride] = st.prefix | ((u64)d << (top - RS_BITS * levels));
}

// keys >= kth[seg] that carry `valid_bit` are appended (order arbitrary) to cand[seg][...]; count[seg] = how many.
// A workgroup takes tiles of RS_THREADS * 8 keys: all 8 loads of a thread are issued first, the survivors of the tile are
// counted through ballots + LDS, and ONE returning atomic per tile reserves their slots (a returning atomic per wave and
// iteration serialised the kernel on the latency of the counter: 43 us for 2 x 261,888 keys).
__global__ __launch_bounds__(RS_THREADS) void k_compact_ge(const u64 *__restrict__ keys, size_t seg_len, const u64 *__restrict__ kth,
                                                           u64 valid_bit, int cap, u64 *__restrict__ cand, unsigned *__restrict__ count) {
    constexpr int PER = 8, NW = RS_THREADS / 64;
    __shared__ unsigned s_wave[NW][PER];
    __shared__ unsigned s_base;
    const int seg = blockIdx.y;
    const u64 T = kth[seg];
    const u64 *ks = keys + (size_t)seg * seg_len;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile = (size_t)RS_THREADS * PER;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < seg_len; t0 += (size_t)gridDim.x * tile) {
        u64 key[PER];
        bool take[PER];
        unsigned before[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const size_t i = t0 + (size_t)j * RS_THREADS + threadIdx.x;
            key[j] = i < seg_len ? ks[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            take[j] = key[j] >= T && (key[j] & valid_bit);
            const u64 bal = __ballot(take[j]);
            before[j] = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[wave][j] = (unsigned)__popcll(bal);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int w = 0; w < NW; ++w)
                for (int j = 0; j < PER; ++j) { const unsigned c = s_wave[w][j]; s_wave[w][j] = tot; tot += c; }
            s_base = tot ? atomicAdd(&count[seg], tot) : 0u;
        }
        __syncthreads();
        const unsigned base = s_base;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const unsigned pos = base + s_wave[wave][j] + before[j];
            if (take[j] && pos < (unsigned)cap) cand[(size_t)seg * cap + pos] = key[j];
        }
        __syncthreads();
    }
}

// One workgroup per segment: bitonic sort (descending) of the compacted keys; out[seg][0 .. seg_out) = sorted keys, zero
// beyond the count.  NP = power of two >= cap, NP = 1024 * E: thread t holds E keys in REGISTERS; a compare-exchange step
// of distance 2^p is done inside a thread when index bit p is one of the thread's LB = log2(E) "local" bits, and the keys
// change owners through LDS only when the next steps need other local bits ("layout" = which LB index bits are local: a
// stage of the network walks p = s-1 .. 0 in groups of LB bits).  For NP = 16384: 32 exchanges through LDS instead of the
// 105 barrier-separated LDS passes of the textbook form (measured 247 us -> see DESIGN.md).  LDS index i is padded by one
// slot every 32 (8-byte slots: the blocked layout would otherwise put a wave on two banks).
// SLICE: blockIdx.y = slice; the workgroup sorts candidates [slice * NP, (slice + 1) * NP) of its segment and writes all NP
// keys to out[(seg * gridDim.y + slice) * NP ..] (runs for k_merge_runs).
template <int NP, bool SLICE = false>
__global__ __launch_bounds__(1024) void k_sort_desc_lds(const u64 *__restrict__ cand, const unsigned *__restrict__ count, int cap,
                                                        u64 *__restrict__ out, size_t out_stride, int seg_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *s = reinterpret_cast<u64 *>(smem_raw);
    constexpr int T = NP >= 1024 ? 1024 : NP;           // active threads
    constexpr int E = NP / T;                            // keys per thread (1, 2, 4, 8 or 16)
    constexpr int LB = E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : 4;
    constexpr int NBITS = __builtin_ctz(NP);
    const int seg = blockIdx.x, t = threadIdx.x;
    const int base = SLICE ? (int)blockIdx.y * NP : 0;
    const int n = min((int)count[seg], cap) - base;
    auto phys = [](int i) { return i + (i >> 5); };
    // index of local element e of thread t when the local bits are [b, b + LB)
    auto index_of = [](int t_, int e, int b) { return ((t_ >> b) << (b + LB)) | (e << b) | (t_ & ((1 << b) - 1)); };
    u64 v[E];
    const bool active = t < T;
    int b = 0;                                           // current layout
    if (active) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = index_of(t, e, 0);
            v[e] = i < n ? cand[(size_t)seg * cap + base + i] : 0ull;
        }
    }
    for (int sbit = 1; sbit <= NBITS; ++sbit) {          // stage: sorted runs of length 2^sbit
        int p = sbit - 1;
        while (p >= 0) {
            const int nb = LB == 0 ? p : min(max(p - (LB - 1), 0), NBITS - LB);      // layout whose local bits contain p
            if (LB == 0 || nb != b) {
                // ---- change owners through LDS (every key written at its index, read back in the new layout)
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int e = 0; e < E; ++e) s[phys(index_of(t, e, b))] = v[e];
                }
                __syncthreads();
                if (LB == 0) {
                    // one key per thread: the partner is read from LDS
                    if (active) {
                        const int i = t, ixj = i ^ (1 << p);
                        const u64 a = v[0], o = s[phys(ixj)];
                        const bool down = (i & (1 << sbit)) == 0;
                        const bool lower = i < ixj;
                        // descending run: the lower index keeps the larger key
                        const bool take_max = (lower == down);
                        v[0] = take_max ? (a > o ? a : o) : (a < o ? a : o);
                    }
                    --p;
                    continue;
                }
                b = nb;
                if (active) {
#pragma unroll
                    for (int e = 0; e < E; ++e) v[e] = s[phys(index_of(t, e, b))];
                }
            }
            // ---- all steps whose bit lies in the local window [b, b + LB) and is <= p
            const int lo = b;
#pragma unroll
            for (int q = LB - 1; q >= 0; --q) {
                const int bit = lo + q;
                if (bit > p || bit < 0) continue;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int e2 = e ^ (1 << q);
                    if (e2 > e) {
                        const int i = index_of(t, e, b);
                        const bool down = (i & (1 << sbit)) == 0 || sbit == NBITS;      // the last stage sorts the whole array descending
                        const u64 x = v[e], y = v[e2];
                        const bool sw = down ? (x < y) : (x > y);
                        v[e] = sw ? y : x;
                        v[e2] = sw ? x : y;
                    }
                }
            }
            p = lo - 1;
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int e = 0; e < E; ++e) s[phys(index_of(t, e, b))] = v[e];
    }
    __syncthreads();
    if (SLICE) {
        u64 *o = out + ((size_t)seg * gridDim.y + blockIdx.y) * NP;
        for (int i = t; i < NP; i += 1024) o[i] = s[phys(i)];
        return;
    }
    for (int i = t; i < seg_out; i += 1024) out[(size_t)seg * out_stride + i] = i < NP ? s[phys(i)] : 0ull;
}

// Merge of NR descending runs of RUN keys each (k_sort_desc_lds<RUN, true>) by RANK: a key's position in the merged order is
// its index in its own run + the number of keys of every other run that precede it - binary searches in LDS, where the
// workgroup holds all NR runs (NR * RUN * 8 bytes <= 128 KB).  Equal keys (only the zero padding) are ordered by run, then
// by index, so ranks are a permutation.  Workgroup (r, seg) places run r; 2 launches (slices, merge) sort 16,384 keys on 8
// CUs per segment in ~1/5 of the time ONE workgroup needs for the whole array.
constexpr int RUN = 2048;
template <int NR>
__global__ __launch_bounds__(1024) void k_merge_runs(const u64 *__restrict__ runs, u64 *__restrict__ out, size_t out_stride, int seg_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *s = reinterpret_cast<u64 *>(smem_raw);
    const int seg = blockIdx.x, r = blockIdx.y, t = threadIdx.x;
    const u64 *src = runs + (size_t)seg * NR * RUN;
    for (int i = t; i < NR * RUN; i += 1024) s[i] = src[i];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < RUN / 1024; ++e) {
        const int idx = t + e * 1024;
        const u64 x = s[r * RUN + idx];
        int rank = idx;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if (q == r) continue;
            // descending run q: number of keys > x (q after r) or >= x (q before r) = first index whose key fails the test
            const u64 *rq = s + q * RUN;
            int lo = 0, hi = RUN;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const u64 y = rq[mid];
                const bool before = q < r ? (y >= x) : (y > x);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < seg_out) out[(size_t)seg * out_stride + rank] = x;
    }
    if (r == 0 && t == 0 && seg_out > NR * RUN) out[(size_t)seg * out_stride + NR * RUN] = 0ull;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
int levels_for(int key_bits) { return (key_bits + RS_BITS - 1) / RS_BITS; }

}  // namespace

// Workspace of select_kth: histograms of every level + double-buffered state.
size_t select_ws_bytes(int nseg, int key_bits) {
    return al256((size_t)levels_for(key_bits) * nseg * RS_BINS * sizeof(unsigned)) + al256((size_t)2 * nseg * sizeof(SelState));
}

// kth[seg * kth_stride] = the kreq[seg * kreq_stride]-th largest (descending) / smallest key of segment seg.  keys < 2^key_bits.
int select_kth(const u64 *keys, int nseg, size_t seg_len, int key_bits, bool descending, const unsigned *kreq, int kreq_stride, u64 *kth,
               int kth_stride, void *ws, hipStream_t st, bool hist_is_zero) {
    const int levels = levels_for(key_bits), top = levels * RS_BITS;
    unsigned *hist = (unsigned *)ws;
    SelState *state = (SelState *)((char *)ws + al256((size_t)levels * nseg * RS_BINS * sizeof(unsigned)));
    if (!hist_is_zero) MRCNN_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)levels * nseg * RS_BINS * sizeof(unsigned), st));
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>((seg_len + RS_THREADS * 8 - 1) / (RS_THREADS * 8), 256));
    for (int l = 0; l < levels; ++l) {
        hipLaunchKernelGGL(k_select_level, dim3(blocks, nseg), dim3(RS_THREADS), 0, st, keys, seg_len, l, top, descending, kreq, kreq_stride,
                           state, hist);
        MRCNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_select_final, dim3(nseg), dim3(RS_THREADS), 0, st, levels, top, descending, state, hist, kth, kth_stride);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ void k_fill_u32(unsigned *p, unsigned v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace
void fill_u32(unsigned *p, unsigned v, int n, hipStream_t st) { hipLaunchKernelGGL(k_fill_u32, dim3((n + 255) / 256), dim3(256), 0, st, p, v, n); }

size_t topk_ws_bytes(int nseg, int key_bits, int cap) {
    int np = 64;
    while (np < cap) np <<= 1;
    return select_ws_bytes(nseg, key_bits) + al256((size_t)nseg * 8) + al256((size_t)nseg * 4) + al256((size_t)nseg * cap * 8) + al256((size_t)nseg * 4) +
           al256(np > RUN ? (size_t)nseg * np * 8 : 0);         // sorted runs of the two-launch sort
}

// out[seg][0 .. cap) = the (up to) cap largest keys of segment seg that carry valid_bit, in descending order, zero-filled
// beyond their number.  cap <= 16384.  keys are unique (composite keys), < 2^key_bits.
// The words top_k_sorted needs initialised before its first kernel: a caller whose own previous kernel on the stream can write them
// (TopkInit from topk_init_of; that kernel zeroes [zero, zero + nzero) and [count, count + ncount) and stores kval to kreq[0 .. nk)) passes
// prepared = true and saves three tiny launches on a latency-bound chain.
TopkInit topk_init_of(int nseg, int key_bits, int cap, void *ws) {
    char *w = (char *)ws;
    TopkInit t;
    t.zero = (unsigned *)w;
    t.nzero = (int)((size_t)levels_for(key_bits) * nseg * RS_BINS);
    size_t o = select_ws_bytes(nseg, key_bits) + al256((size_t)nseg * 8);
    t.kreq = (unsigned *)(w + o); o += al256((size_t)nseg * 4);
    t.nk = nseg; t.kval = (unsigned)cap;
    o += al256((size_t)nseg * cap * 8);
    t.count = (unsigned *)(w + o); t.ncount = nseg;
    return t;
}
int top_k_sorted(const u64 *keys, int nseg, size_t seg_len, int key_bits, u64 valid_bit, int cap, u64 *out, size_t out_stride, void *ws,
                 hipStream_t st, bool prepared) {
    if (cap <= 0 || cap > 16384) {
        set_error("top_k_sorted: cap %d not in [1, 16384]", cap);
        return MRCNN_E_UNSUPPORTED;
    }
    char *w = (char *)ws;
    size_t o = select_ws_bytes(nseg, key_bits);
    u64 *kth = (u64 *)(w + o); o += al256((size_t)nseg * 8);
    unsigned *kreq = (unsigned *)(w + o); o += al256((size_t)nseg * 4);
    u64 *cand = (u64 *)(w + o); o += al256((size_t)nseg * cap * 8);
    unsigned *count = (unsigned *)(w + o); o += al256((size_t)nseg * 4);
    u64 *runs = (u64 *)(w + o);
    if (!prepared) {
        fill_u32(kreq, (unsigned)cap, nseg, st);              // k = cap for every segment
        MRCNN_HIP_TRY(hipMemsetAsync(count, 0, (size_t)nseg * 4, st));
    }
    if (int e = select_kth(keys, nseg, seg_len, key_bits, true, kreq, 1, kth, 1, ws, st, prepared)) return e;
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>((seg_len + RS_THREADS * 8 - 1) / (RS_THREADS * 8), 256));
    hipLaunchKernelGGL(k_compact_ge, dim3(blocks, nseg), dim3(RS_THREADS), 0, st, keys, seg_len, kth, valid_bit, cap, cand, count);
    MRCNN_LAUNCH_CHECK();
    const int seg_out = (int)std::min<size_t>(out_stride, (size_t)cap + 1);     // one zero after the keys when there is room
    int np = 64;
    while (np < cap) np <<= 1;
    if (np > RUN) {
        // more than one run: slices of RUN keys sorted by np / RUN workgroups per segment, then merged by rank
        const int nr = np / RUN;
        const size_t lds_a = (size_t)(RUN + RUN / 32 + 1) * 8, lds_b = (size_t)np * 8;
        MRCNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sort_desc_lds<RUN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
        hipLaunchKernelGGL((k_sort_desc_lds<RUN, true>), dim3(nseg, nr), dim3(1024), lds_a, st, cand, count, cap, runs, (size_t)0, 0);
        MRCNN_LAUNCH_CHECK();
#define MERGE_CASE(NRV)                                                                                                     \
    case NRV:                                                                                                               \
        MRCNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_merge_runs<NRV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b)); \
        hipLaunchKernelGGL((k_merge_runs<NRV>), dim3(nseg, NRV), dim3(1024), lds_b, st, runs, out, out_stride, seg_out);    \
        break;
        switch (nr) { MERGE_CASE(2) MERGE_CASE(4) MERGE_CASE(8) }
#undef MERGE_CASE
        MRCNN_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = (size_t)(np + np / 32 + 1) * 8;
#define SORT_CASE(NPV)                                                                                                      \
    case NPV:                                                                                                               \
        MRCNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sort_desc_lds<NPV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_sort_desc_lds<NPV>), dim3(nseg), dim3(1024), lds, st, cand, count, cap, out, out_stride, seg_out);    \
        break;
    switch (np) {
        SORT_CASE(64) SORT_CASE(128) SORT_CASE(256) SORT_CASE(512) SORT_CASE(1024) SORT_CASE(2048) SORT_CASE(4096) SORT_CASE(8192)
        SORT_CASE(16384)
    }
#undef SORT_CASE
    MRCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace mrcnn

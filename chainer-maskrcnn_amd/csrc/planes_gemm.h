// Float32-accurate GEMMs on PRE-SPLIT bf16 planes (included by conv.hip inside its anonymous namespace).
//
// The "bf16x6" arithmetic of conv.hip (k_conv_igemm<..., SPLIT = 3>): a float32 operand x is carried EXACTLY by three bf16 planes
// hi + mid + lo (3 x 8 significant bits, bf16 has float32's exponent range) and a product a*b is accumulated in float32 as
// lh + hl + mm + mh + hm + hh with six v_mfma_f32_32x32x16_bf16.  Here the planes are not made inside the GEMM (3 VALU
// instructions per element and plane, repeated by every workgroup that reads the element) but ONCE by the kernel that produces the
// operand - the Winograd input / gradient / filter transforms, HBM-bound kernels whose VALU is idle - and the GEMM is a pure
// LDS-DMA + MFMA kernel: no VGPR staging, no split, no ds_write, one barrier per 16-deep K step on a two-stage LDS ring.
//
// Plane layout "P16" of a (R rows, C channels) float32 matrix, C % 16 == 0: unsigned short [R][C/16][3][16] - the three planes
// (hi, mid, lo) of 16 consecutive channels are 96 contiguous bytes, a row is 6*C bytes.  One layout serves both GEMM kinds:
//   F ("forward kind", K = channels):   C[m][n] = sum_k A[m][k] B[n][k]      a K step = 96 contiguous bytes of every row
//   G ("filter-gradient kind", K = rows): C[m][n] = sum_t A[t][m] B[t][n]    a K step = 16 rows x the tile's channel range
// Batched like the Winograd GEMMs of conv.hip: F - row block [b*batch_rows, (b+1)*batch_rows) of A multiplies matrix b of B;
// G - batch b of both operands (batch_rows rows each) gives C[:, b, :] of a (split, M, nbatch, N) slab array.
//
// LDS images are written by buffer_load_dwordx4 ... lds (1 KiB per wave instruction, destination = wave-uniform base + lane * 16:
// lane-linear, so the bank swizzles below are applied to the per-lane SOURCE address and again on the read - cdna_hip_programming.md
// rule 21):
//   F: image [rows][6 chunks of 16 B] (96-B rows); chunk c = 2*plane + (k >> 3) sits at c ^ ((row >> 3) & 1): ds_read_b128 of one
//      plane / k half by 32 consecutive rows is conflict-free (16-lane groups of ds_read_b128: MI355X_MICROARCH.md, LDS).
//   G: image [16 t][NCH chunks] (NCH = 6 * tile channels / 16: 768-B or 384-B rows); logical chunk lc of row t sits at
//      lc ^ (sw(t & 3) << 2), sw(q) = q (768-B rows) or q >> 1 (384-B rows): the four t rows of a ds_read_b64_tr_b16 block land in
//      four different 64-B bank groups.
// Ring protocol (two stages): top of step s - every wave waits vmcnt(0) for ITS pieces of stage s (issued one whole step earlier),
// then s_barrier: all pieces of stage s have landed and every wave has consumed stage s-1 (its fragment reads were waited for by
// the MFMAs that precede the barrier in program order); only then are the pieces of stage s+1 issued into the buffer of s-1.

constexpr int PG_THREADS = 256;
constexpr int PG_BK = 16;

struct PlaneGemmP {
    const unsigned short *a, *b;
    float *c;
    int M, N, K;               // F: M = nbatch * batch_rows rows of A, N rows of each B matrix, K channels.  G: M, N channels of A, B; K = batch_rows
    int batch_rows, nbatch;
    int tiles_m, tiles_n, remap_n;
    int ksplit, kchunk;        // G: K splits, rows per split (multiple of PG_BK)
    unsigned bytes_a, bytes_b;
    int ldc;                   // F: N.  G: nbatch * N
    unsigned long long *stamps; // measurement (nullable): 4 x u64 per workgroup - s_memtime start, end; s_memrealtime start, end; [hw id | xcc id << 32] in slot 4
    int dbg;                   // measurement (mrcnn_debug_conv_parts bits 16..): 1 no MFMA, 2 every tile reads tile 0's operands (L2-resident), 4 no epilogue, 8 no loads
};

__device__ __forceinline__ void pg_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char *lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ unsigned long long pg_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define PG_BF(x) __builtin_bit_cast(bf16x8_t, x)
// six products of one 32x32x16 step, smallest first (the order of mma_step_split<..., 3>)
#define PG_MMA6(ACC, AH, AM, AL, BH, BM, BL)                                                              \
    do {                                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_BF(AL), PG_BF(BH), ACC, 0, 0, 0);                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_BF(AH), PG_BF(BL), ACC, 0, 0, 0);                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_BF(AM), PG_BF(BM), ACC, 0, 0, 0);                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_BF(AM), PG_BF(BH), ACC, 0, 0, 0);                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_BF(AH), PG_BF(BM), ACC, 0, 0, 0);                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_BF(AH), PG_BF(BH), ACC, 0, 0, 0);                \
    } while (0)

// Epilogue shared by both kinds: the wave's TM x TN accumulator tiles (row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31) are
// transposed through a per-wave 32 x 36 LDS tile and leave as float4 stores, 8 rows x 128 B per instruction.
template <int TM, int TN>
__device__ __forceinline__ void pg_store_tiles(f32x16 (&acc)[TM][TN], float *et, float *cbase, size_t ldc, int mrow0, int ncol0,
                                               int M, int N, int lane) {
    constexpr int EPI_LD = 36;
    const int r = lane & 31, h = lane >> 5;
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
            for (int e = 0; e < 16; ++e) et[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_LD + r] = acc[tm][tn][e];
            const int n = ncol0 + tn * 32 + ec;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = er + 8 * j;
                const int m = mrow0 + tm * 32 + row;
                const float4 v = *reinterpret_cast<const float4 *>(&et[row * EPI_LD + ec]);
                if (n < N && m < M) *reinterpret_cast<float4 *>(cbase + (size_t)m * ldc + n) = v;
            }
        }
}

// ---- F kind ---------------------------------------------------------------------------------------------------------------
template <int BM_, int BN_>
__global__ __launch_bounds__(PG_THREADS, 2) void k_pgemm_f(PlaneGemmP p) {
    constexpr int TM = BM_ / 64, TN = BN_ / 64;
    constexpr int ROWB = 96;                                  // bytes of one row of one stage: 3 planes x 16 k
    constexpr int A_STAGE = BM_ * ROWB, B_STAGE = BN_ * ROWB, STAGE = A_STAGE + B_STAGE;
    constexpr int NPA = A_STAGE / 1024, NPB = B_STAGE / 1024;          // 1-KiB pieces per stage
    constexpr int SA = (NPA + 3) / 4, SB = (NPB + 3) / 4;              // piece slots per wave
    constexpr int EPI_BYTES = 4 * 32 * 36 * 4;
    static_assert(STAGE >= EPI_BYTES, "the epilogue's transpose tiles reuse stage 0");
    // TWO LDS objects, one per ring stage: the waitcnt pass can then prove that the fragment reads of one stage do not alias the
    // LDS-DMA in flight into the other (with one array it puts s_waitcnt vmcnt(0) in front of the first ds_read of every step)
    __shared__ __attribute__((aligned(1024))) unsigned char smem0[STAGE];
    __shared__ __attribute__((aligned(1024))) unsigned char smem1[STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int vid = blockIdx.x;
    {   // bijective XCD remap: every XCD works on a contiguous range of tiles (the N tiles of an M tile share its A rows)
        const int q = p.remap_n >> 3, r = p.remap_n & 7, xcd = vid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) { st_t0 = pg_now(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int bx = vid / p.tiles_n, by = vid - bx * p.tiles_n;
    const int m0 = bx * BM_, n0 = by * BN_;
    const unsigned rowbytes = (unsigned)p.K * 6u;
    const int batch = m0 / p.batch_rows;
    const int m0l = (p.dbg & 2) ? 0 : m0;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    // per-lane source offsets of this wave's pieces (chunk f = 64 * piece + lane of the stage image: row f / 6, physical chunk f % 6)
    unsigned voffA[SA], voffB[SB];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int f = 64 * (wave + 4 * i) + lane, row = f / 6, pc = f - row * 6, lc = pc ^ ((row >> 3) & 1);
        voffA[i] = (m0 + row < p.M && row < BM_) ? (unsigned)(m0l + row) * rowbytes + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int f = 64 * (wave + 4 * i) + lane, row = f / 6, pc = f - row * 6, lc = pc ^ ((row >> 3) & 1);
        voffB[i] = (n0 + row < p.N && row < BN_) ? (unsigned)(batch * p.N + n0 + row) * rowbytes + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
    auto issue = [&](unsigned char *sa, unsigned soff) {
        if (p.dbg & 8) return;
        unsigned char *sb = sa + A_STAGE;
#pragma unroll
        for (int i = 0; i < SA; ++i)
            if (NPA % 4 == 0 || wave + 4 * i < NPA) pg_dma16(rsA, sa + (wave + 4 * i) * 1024, voffA[i], soff);
#pragma unroll
        for (int i = 0; i < SB; ++i)
            if (NPB % 4 == 0 || wave + 4 * i < NPB) pg_dma16(rsB, sb + (wave + 4 * i) * 1024, voffB[i], soff);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment addresses: row r = lane & 31 of a 32-row MFMA tile, k half h = lane >> 5; chunk (2 * plane + h) ^ ((r >> 3) & 1)
    const int r = lane & 31, h = lane >> 5;
    const unsigned fragA = (unsigned)(wm * (BM_ / 2) + r) * ROWB + (unsigned)((h ^ ((r >> 3) & 1)) * 16);
    const unsigned fragB = (unsigned)A_STAGE + (unsigned)(wn * (BN_ / 2) + r) * ROWB + (unsigned)((h ^ ((r >> 3) & 1)) * 16);
    auto compute = [&](const unsigned char *st) {
        uint4 a[TM][3], b[TN][3];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[t][pl] = *reinterpret_cast<const uint4 *>(st + fragA + t * 32 * ROWB + pl * 32);
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[t][pl] = *reinterpret_cast<const uint4 *>(st + fragB + t * 32 * ROWB + pl * 32);
        if (p.dbg & 1) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(a[t][pl].x), "v"(a[t][pl].y), "v"(a[t][pl].z), "v"(a[t][pl].w));
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(b[t][pl].x), "v"(b[t][pl].y), "v"(b[t][pl].z), "v"(b[t][pl].w));
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
    };

    const int nsteps = p.K / PG_BK;          // even: K % 32 == 0
    unsigned soff = 0;
    // PG_SYNC: my pieces of the next stage have landed; after the barrier everybody's have, and everybody has read the stage before
#define PG_SYNC()                                              \
    do {                                                       \
        __builtin_amdgcn_sched_barrier(0);                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)
    issue(smem0, soff);
    // (the last pair of steps is peeled: an issue inside a conditional makes the waitcnt pass wait vmcnt(0) at the join)
    for (int s = 0; s + 2 < nsteps; s += 2) {
        PG_SYNC();
        issue(smem1, soff + ROWB);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem0);
        PG_SYNC();
        soff += 2 * ROWB;
        issue(smem0, soff);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem1);
    }
    PG_SYNC();
    issue(smem1, soff + ROWB);
    __builtin_amdgcn_sched_barrier(0);
    compute(smem0);
    PG_SYNC();
    compute(smem1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // the ring is free: the epilogue's transpose tiles reuse stage 0
    if (!(p.dbg & 4))
    pg_store_tiles<TM, TN>(acc, reinterpret_cast<float *>(smem0) + wave * (32 * 36), p.c, (size_t)p.ldc, m0 + wm * (BM_ / 2),
                           n0 + wn * (BN_ / 2), p.M, p.N, lane);
    if (p.stamps && tid == 0) {
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 5;
        o[0] = st_t0; o[1] = pg_now(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

// ---- F kind, A operand as float32 rows -----------------------------------------------------------------------------------
// The Winograd GEMMs of this network (K, N <= 512) are HBM-bound at the bf16 MFMA rate, so bytes decide: the big operand A (the
// transformed activations, streamed once from HBM) stays float32 (4 B per element instead of 6) and is split into its three planes
// when a wave reads its fragment from LDS (8 floats per lane and 16-deep step: ~44 VALU instructions per fragment, in the shadow of
// the wave's 24 MFMAs); the small, L2-resident operand B (the transformed filters) comes pre-split.  A image: [rows][4 chunks of
// 16 B] (64-B rows: one K step of a row), chunk c of row r at c ^ ((r >> 2) & 3): conflict-free ds_read_b128 by 32 consecutive rows.
template <int BM_, int BN_>
__global__ __launch_bounds__(PG_THREADS, 2) void k_pgemm_fa(PlaneGemmP p) {
    constexpr int TM = BM_ / 64, TN = BN_ / 64;
    constexpr int ROWA = 64, ROWB = 96;
    constexpr int A_STAGE = BM_ * ROWA, B_STAGE = BN_ * ROWB, STAGE = A_STAGE + B_STAGE;
    constexpr int NPA = A_STAGE / 1024, NPB = B_STAGE / 1024;
    constexpr int SA = (NPA + 3) / 4, SB = (NPB + 3) / 4;
    constexpr int EPI_BYTES = 4 * 32 * 36 * 4;
    constexpr int SM0 = STAGE > EPI_BYTES ? STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem0[SM0];
    __shared__ __attribute__((aligned(1024))) unsigned char smem1[STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int vid = blockIdx.x;
    {
        const int q = p.remap_n >> 3, r = p.remap_n & 7, xcd = vid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) { st_t0 = pg_now(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int bx = vid / p.tiles_n, by = vid - bx * p.tiles_n;
    const int m0 = bx * BM_, n0 = by * BN_;
    const unsigned rowbytesA = (unsigned)p.K * 4u, rowbytesB = (unsigned)p.K * 6u;
    const int batch = m0 / p.batch_rows;
    const int m0l = (p.dbg & 2) ? 0 : m0;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    unsigned voffA[SA], voffB[SB];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int f = 64 * (wave + 4 * i) + lane, row = f >> 2, pc = f & 3, lc = pc ^ ((row >> 2) & 3);
        voffA[i] = (m0 + row < p.M && row < BM_) ? (unsigned)(m0l + row) * rowbytesA + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int f = 64 * (wave + 4 * i) + lane, row = f / 6, pc = f - row * 6, lc = pc ^ ((row >> 3) & 1);
        voffB[i] = (n0 + row < p.N && row < BN_) ? (unsigned)(batch * p.N + n0 + row) * rowbytesB + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
    auto issue = [&](unsigned char *sa, unsigned soffA, unsigned soffB) {
        if (p.dbg & 8) return;
        unsigned char *sb = sa + A_STAGE;
#pragma unroll
        for (int i = 0; i < SA; ++i)
            if (NPA % 4 == 0 || wave + 4 * i < NPA) pg_dma16(rsA, sa + (wave + 4 * i) * 1024, voffA[i], soffA);
#pragma unroll
        for (int i = 0; i < SB; ++i)
            if (NPB % 4 == 0 || wave + 4 * i < NPB) pg_dma16(rsB, sb + (wave + 4 * i) * 1024, voffB[i], soffB);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int r = lane & 31, h = lane >> 5;
    const unsigned sA = (unsigned)((r >> 2) & 3);
    const unsigned fragA0 = (unsigned)(wm * (BM_ / 2) + r) * ROWA + (((unsigned)(2 * h) ^ sA) * 16u);
    const unsigned fragA1 = (unsigned)(wm * (BM_ / 2) + r) * ROWA + (((unsigned)(2 * h + 1) ^ sA) * 16u);
    const unsigned fragB = (unsigned)A_STAGE + (unsigned)(wn * (BN_ / 2) + r) * ROWB + (unsigned)((h ^ ((r >> 3) & 1)) * 16);
    auto compute = [&](const unsigned char *st) {
        uint4 a[TM][3], b[TN][3];
        float4 ra[TM][2];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            ra[t][0] = *reinterpret_cast<const float4 *>(st + fragA0 + t * 32 * ROWA);
            ra[t][1] = *reinterpret_cast<const float4 *>(st + fragA1 + t * 32 * ROWA);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[t][pl] = *reinterpret_cast<const uint4 *>(st + fragB + t * 32 * ROWB + pl * 32);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            uint2 h0, m0_, l0, h1, m1, l1;
            split3_bf16x4(ra[t][0], h0, m0_, l0);
            split3_bf16x4(ra[t][1], h1, m1, l1);
            a[t][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            a[t][1] = make_uint4(m0_.x, m0_.y, m1.x, m1.y);
            a[t][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        if (p.dbg & 1) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(a[t][pl].x), "v"(a[t][pl].y), "v"(a[t][pl].z), "v"(a[t][pl].w));
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(b[t][pl].x), "v"(b[t][pl].y), "v"(b[t][pl].z), "v"(b[t][pl].w));
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
    };

    const int nsteps = p.K / PG_BK;
    unsigned soffA = 0, soffB = 0;
    issue(smem0, soffA, soffB);
    for (int s = 0; s + 2 < nsteps; s += 2) {
        PG_SYNC();
        issue(smem1, soffA + ROWA, soffB + ROWB);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem0);
        PG_SYNC();
        soffA += 2 * ROWA; soffB += 2 * ROWB;
        issue(smem0, soffA, soffB);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem1);
    }
    PG_SYNC();
    issue(smem1, soffA + ROWA, soffB + ROWB);
    __builtin_amdgcn_sched_barrier(0);
    compute(smem0);
    PG_SYNC();
    compute(smem1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!(p.dbg & 4))
    pg_store_tiles<TM, TN>(acc, reinterpret_cast<float *>(smem0) + wave * (32 * 36), p.c, (size_t)p.ldc, m0 + wm * (BM_ / 2),
                           n0 + wn * (BN_ / 2), p.M, p.N, lane);
    if (p.stamps && tid == 0) {
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 5;
        o[0] = st_t0; o[1] = pg_now(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

// ---- F kind, big tile: 256 x 256 per 512-thread workgroup, A float32 rows, B planes in the "P16R4" layout -------------------------
// What the stamps of the 128 x 128 kernels above say (tools/pgemm_bench.py stamps, the mask-head GEMM 36 x 8192 x 256 x 256): an
// MFMA-dense loop holds 1.65 GHz on this chip, not 2.4 (MFMA floor of that GEMM 134 us, not 92); with one stage in flight per
// workgroup every K step exposes ~2 us of load latency, because 80 KB in flight per CU at 26 B per MFMA-bound cycle cover 1.9 us -
// a 128 x 128 tile is LATENCY-bound whatever the ring does inside 160 KB of LDS.  The lever is bytes per MFMA cycle: a 256 x 256
// tile moves 40 KB per 3072 cycles (13 B per cycle: A once from HBM, B half as often from L2), and a three-stage ring keeps 80 KB
// in flight = 6000 cycles of cover.  One workgroup per CU, 8 waves as 4 (M) x 2 (N), a wave owns 64 x 128 (2 x 4 MFMA tiles).
// "P16R4" layout of B (rows = output channels n): unsigned short [N/4][K/16][3 planes][4 rows][16 k] - every 128-byte line holds
// one plane of 4 rows of ONE K step, so a K step of the tile is fetched in whole lines and no line twice (in P16 a step touches
// 96 of a row's 128-byte line and the next step fetches it again: L2 -> L1 traffic x 1.5 .. 2, which is what saturated first).
// LDS stage: A [256 rows][64 B] (chunk c at c ^ ((row >> 2) & 3)) + B [3 planes][256 rows][32 B] (k half h at h ^ ((row >> 3) & 1)).
// Fragment reads are inline asm: the waitcnt pass cannot prove that a C++ read of one ring stage does not alias the LDS-DMA in
// flight into another and would drain the ring (s_waitcnt vmcnt(0)) in front of every step; counts are kept by hand.
typedef unsigned pg_u32x4 __attribute__((ext_vector_type(4)));
#define PG_DSR128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
constexpr int PGB_THREADS = 512, PGB_BM = 256, PGB_BN = 256;
constexpr int PGB_A_STAGE = PGB_BM * 64, PGB_B_PLANE = PGB_BN * 32, PGB_B_STAGE = 3 * PGB_B_PLANE, PGB_STAGE = PGB_A_STAGE + PGB_B_STAGE;
constexpr int PGB_NSTAGE = 3;

__global__ __launch_bounds__(PGB_THREADS, 2) void k_pgemm_big(PlaneGemmP p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PGB_NSTAGE * PGB_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int vid = blockIdx.x;
    {
        const int q = p.remap_n >> 3, r = p.remap_n & 7, xcd = vid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) { st_t0 = pg_now(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int bx = vid / p.tiles_n, by = vid - bx * p.tiles_n;
    const int m0 = bx * PGB_BM, n0 = by * PGB_BN;
    const int batch = m0 / p.batch_rows;
    const unsigned rowbytesA = (unsigned)p.K * 4u;
    const unsigned kblocks = (unsigned)p.K / 16u;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    // A pieces (16 rows x 64 B each): wave w loads pieces w and w + 8.  B pieces (one plane of 32 rows = 8 lines): w, w + 8, w + 16.
    unsigned voffA[2], voffB[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = 64 * (wave + 8 * i) + lane, row = f >> 2, pc = f & 3, lc = pc ^ ((row >> 2) & 3);
        voffA[i] = (m0 + row < p.M) ? (unsigned)(m0 + row) * rowbytesA + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = wave + 8 * i, pl = piece >> 3, rb = piece & 7;
        const int row = 32 * rb + (lane >> 1), half = (lane & 1) ^ ((row >> 3) & 1);
        const int n = n0 + row;
        voffB[i] = (n < p.N) ? ((unsigned)((batch * p.N + n) >> 2) * kblocks) * 384u + (unsigned)pl * 128u + (unsigned)(n & 3) * 32u + (unsigned)half * 16u
                             : 0xFFFFFFFFu;
    }
    auto issue = [&](unsigned buf, unsigned kstep) {          // buf: byte offset of the stage in smem
        unsigned char *sa = smem + buf, *sb = sa + PGB_A_STAGE;
        const unsigned soffA = kstep * 64u, soffB = kstep * 384u;
#pragma unroll
        for (int i = 0; i < 2; ++i) pg_dma16(rsA, sa + (wave + 8 * i) * 1024, voffA[i], soffA);
#pragma unroll
        for (int i = 0; i < 3; ++i) pg_dma16(rsB, sb + (wave + 8 * i) * 1024, voffB[i], soffB);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int r = lane & 31, h = lane >> 5;
    const unsigned sA = (unsigned)((r >> 2) & 3);
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned fragA0 = smem_base + (unsigned)(wm * 64 + r) * 64u + (((unsigned)(2 * h) ^ sA) * 16u);
    const unsigned fragA1 = smem_base + (unsigned)(wm * 64 + r) * 64u + (((unsigned)(2 * h + 1) ^ sA) * 16u);
    const unsigned fragB = smem_base + (unsigned)PGB_A_STAGE + (unsigned)(wn * 128 + r) * 32u + (unsigned)((h ^ ((r >> 3) & 1)) * 16);

    auto compute = [&](unsigned buf) {
        const unsigned a0 = fragA0 + buf, a1 = fragA1 + buf, bb = fragB + buf;
        pg_u32x4 ra[2][2], b[4][3];
        PG_DSR128(ra[0][0], a0, 0); PG_DSR128(ra[0][1], a1, 0);
        PG_DSR128(ra[1][0], a0, 32 * 64); PG_DSR128(ra[1][1], a1, 32 * 64);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            PG_DSR128(b[0][pl], bb, pl * PGB_B_PLANE);
            PG_DSR128(b[1][pl], bb, pl * PGB_B_PLANE + 32 * 32);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            PG_DSR128(b[2][pl], bb, pl * PGB_B_PLANE + 64 * 32);
            PG_DSR128(b[3][pl], bb, pl * PGB_B_PLANE + 96 * 32);
        }
        uint4 a[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint2 h0, m0_, l0, h1, m1, l1;
            split3_bf16x4(__builtin_bit_cast(float4, ra[t][0]), h0, m0_, l0);
            split3_bf16x4(__builtin_bit_cast(float4, ra[t][1]), h1, m1, l1);
            a[t][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            a[t][1] = make_uint4(m0_.x, m0_.y, m1.x, m1.y);
            a[t][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        if (!(p.dbg & 1)) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 1)) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 2; tn < 4; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(a[t][pl].x), "v"(a[t][pl].y), "v"(a[t][pl].z), "v"(a[t][pl].w));
        }
    };

    // ring: stages s and s + 1 in flight while stage s - 1 ... wait: at the top of step s the wave allows only the 5 pieces of stage
    // s + 1 to be outstanding (in-order vmcnt), the barrier then says stage s is complete and stage s - 1 consumed by everybody
    const int nsteps = (int)kblocks;
    unsigned b0 = 0, b1 = PGB_STAGE, b2 = 2 * PGB_STAGE;
    if (!(p.dbg & 8)) { issue(b0, 0); if (nsteps > 1) issue(b1, 1); }
    for (int s = 0; s < nsteps; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps && !(p.dbg & 8)) issue(b2, (unsigned)(s + 2));
        __builtin_amdgcn_sched_barrier(0);
        compute(b0);
        const unsigned t = b0; b0 = b1; b1 = b2; b2 = t;
    }
    // epilogue: straight from the accumulator layout - per register two rows of 32 consecutive floats (128-byte segments)
    if (!(p.dbg & 4)) {
        const int col = n0 + wn * 128 + (lane & 31);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float *dst = p.c + (size_t)m * p.ldc + col;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    if (m < p.M && col + tn * 32 < p.N) dst[tn * 32] = acc[tm][tn][e];
            }
    }
    if (p.stamps && tid == 0) {
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 5;
        o[0] = st_t0; o[1] = pg_now(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

// split3_bf16x4 with the subtractions kept scalar: beside another wave's MFMAs a v_pk_add_f32 costs ~13 cycles more than the two
// v_sub_f32 it replaces (MI355X_MICROARCH.md, cycle constants: packed f32 VALU is an anti-lever beside MFMAs); the empty asm
// statements keep hipcc's SLP vectoriser from re-packing them.
__device__ __forceinline__ void pg_split3x4(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
    const f32x2_t v01 = {v.x, v.y}, v23 = {v.z, v.w};
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2_t));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2_t));
    float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xffff0000u);
    float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xffff0000u);
    asm volatile("" : "+v"(r0)); asm volatile("" : "+v"(r1)); asm volatile("" : "+v"(r2)); asm volatile("" : "+v"(r3));
    const f32x2_t r01 = {r0, r1}, r23 = {r2, r3};
    const unsigned m01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2_t));
    const unsigned m23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2_t));
    float q0 = r0 - __uint_as_float(m01 << 16), q1 = r1 - __uint_as_float(m01 & 0xffff0000u);
    float q2 = r2 - __uint_as_float(m23 << 16), q3 = r3 - __uint_as_float(m23 & 0xffff0000u);
    asm volatile("" : "+v"(q0)); asm volatile("" : "+v"(q1)); asm volatile("" : "+v"(q2)); asm volatile("" : "+v"(q3));
    const f32x2_t q01 = {q0, q1}, q23 = {q2, q3};
    hi = make_uint2(h01, h23);
    mid = make_uint2(m01, m23);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(q01, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(q23, bf16x2_t)));
}

// ---- F kind, 256 x 256 tile, ping-pong halves --------------------------------------------------------------------------------
// k_pgemm_big with its two wave groups (waves 0-3: rows 0..127 of the tile, waves 4-7: rows 128..255; a SIMD hosts one wave of each)
// offset by half a K step: while one group runs the 48 MFMAs of step s ("Y"), the other issues its LDS-DMA pieces of stage s + 2,
// reads and splits its fragments of the next step ("X") - everything that is not an MFMA sits in the partner's MFMA shadow instead
// of between two barriers of its own (MI355X_MICROARCH.md, Two waves per SIMD).  Phase p = 2s: group 0 X(s), group 1 Y(s-1);
// phase 2s + 1: group 0 Y(s), group 1 X(s); one s_barrier between phases.  Stage s is read in phases 2s and 2s + 1 and its buffer
// refilled (stage s + 3) from phase 2s + 2 on; a wave checks ITS pieces of stage s + 1 (counted vmcnt) at the end of its X(s), the
// barriers up to the first read of that stage (phase 2s + 2) carry the rest.
__global__ __launch_bounds__(PGB_THREADS, 2) void k_pgemm_pp(PlaneGemmP p) {
    // PERSISTENT: workgroup b walks the tiles vid0 + j * gridDim.x; the ring runs on across tile boundaries (the first stages of the
    // next tile are in flight while the last steps of this one compute, and its stores drain under the next tile's phases).
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PGB_NSTAGE * PGB_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wm = half * 2 + ((wave >> 1) & 1), wn = wave & 1;
    int vid0 = blockIdx.x;
    {
        const int G = (int)gridDim.x, q = G >> 3, r = G & 7, xcd = vid0 & 7;
        vid0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid0 >> 3);
    }
    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) { st_t0 = pg_now(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nj = vid0 < ntiles ? (ntiles - vid0 + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const unsigned rowbytesA = (unsigned)p.K * 4u;
    const unsigned kblocks = (unsigned)p.K / 16u;
    const int nsteps = (int)kblocks;
    const int S = nj * nsteps;                  // stages of this workgroup's stream

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    // issue side: the tile and K step of the next stage to load, and that tile's per-lane source offsets
    unsigned voffA[2], voffB[3];
    int ij = 0, is = 0;
    auto setup = [&](int j) {
        const int tile = vid0 + j * (int)gridDim.x;
        const int bx = tile / p.tiles_n, by = tile - bx * p.tiles_n;
        const int m0 = bx * PGB_BM, n0 = by * PGB_BN, batch = m0 / p.batch_rows;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = 64 * (wave + 8 * i) + lane, row = f >> 2, pc = f & 3, lc = pc ^ ((row >> 2) & 3);
            voffA[i] = (m0 + row < p.M) ? (unsigned)(m0 + row) * rowbytesA + (unsigned)lc * 16u : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 8 * i, pl = piece >> 3, rb = piece & 7;
            const int row = 32 * rb + (lane >> 1), hf = (lane & 1) ^ ((row >> 3) & 1);
            const int n = n0 + row;
            voffB[i] = (n < p.N) ? ((unsigned)((batch * p.N + n) >> 2) * kblocks) * 384u + (unsigned)pl * 128u + (unsigned)(n & 3) * 32u + (unsigned)hf * 16u
                                 : 0xFFFFFFFFu;
        }
    };
    const bool loads = !(p.dbg & 8), mfma = !(p.dbg & 1);
    auto issue_next = [&](unsigned buf) {
        if (is == 0) setup(ij);
        unsigned char *sa = smem + buf, *sb = sa + PGB_A_STAGE;
        const unsigned soffA = (unsigned)is * 64u, soffB = (unsigned)is * 384u;
        if (loads) {
#pragma unroll
            for (int i = 0; i < 2; ++i) pg_dma16(rsA, sa + (wave + 8 * i) * 1024, voffA[i], soffA);
#pragma unroll
            for (int i = 0; i < 3; ++i) pg_dma16(rsB, sb + (wave + 8 * i) * 1024, voffB[i], soffB);
        }
        if (++is == nsteps) { is = 0; ++ij; }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int r = lane & 31, h = lane >> 5;
    const unsigned sA = (unsigned)((r >> 2) & 3);
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned fragA0 = smem_base + (unsigned)(wm * 64 + r) * 64u + (((unsigned)(2 * h) ^ sA) * 16u);
    const unsigned fragA1 = smem_base + (unsigned)(wm * 64 + r) * 64u + (((unsigned)(2 * h + 1) ^ sA) * 16u);
    const unsigned fragB = smem_base + (unsigned)PGB_A_STAGE + (unsigned)(wn * 128 + r) * 32u + (unsigned)((h ^ ((r >> 3) & 1)) * 16);

    unsigned b0 = 0, b1 = PGB_STAGE, b2 = 2 * PGB_STAGE;
    if (S > 0) issue_next(b0);
    if (S > 1) issue_next(b1);
    if (S > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (half) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    int cs = 0, cj = 0;               // compute side: K step within the tile, tile ordinal
    for (int g = 0; g < S; ++g) {
        // ---- X(g): the loads of stage g + 2, this step's fragments, the split of the A fragments
        if (g + 2 < S) issue_next(b2);
        const unsigned a0 = fragA0 + b0, a1 = fragA1 + b0, bb = fragB + b0;
        pg_u32x4 ra[2][2], b[4][3];
        PG_DSR128(ra[0][0], a0, 0); PG_DSR128(ra[0][1], a1, 0);
        PG_DSR128(ra[1][0], a0, 32 * 64); PG_DSR128(ra[1][1], a1, 32 * 64);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) PG_DSR128(b[t][pl], bb, pl * PGB_B_PLANE + t * 32 * 32);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        uint4 a[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint2 h0, m0_, l0, h1, m1, l1;
            pg_split3x4(__builtin_bit_cast(float4, ra[t][0]), h0, m0_, l0);
            pg_split3x4(__builtin_bit_cast(float4, ra[t][1]), h1, m1, l1);
            a[t][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            a[t][1] = make_uint4(m0_.x, m0_.y, m1.x, m1.y);
            a[t][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        // (the planes are used only after the barrier: without this pin hipcc sinks the split behind it, into the MFMA phase)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(a[t][pl].x), "+v"(a[t][pl].y), "+v"(a[t][pl].z), "+v"(a[t][pl].w));
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < S) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- Y(g): the MFMAs, at raised priority (the partner's X phase fills the issue slots they leave)
        // (the B fragments of the wave's second pair of column tiles are read here, under the first 24 MFMAs: 24 registers fewer
        // live across the barrier; the stage stays valid until X(g + 1) of this wave group refills it)
#pragma unroll
        for (int t = 2; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) PG_DSR128(b[t][pl], bb, pl * PGB_B_PLANE + t * 32 * 32);
        __builtin_amdgcn_s_setprio(1);
        if (mfma) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 2; tn < 4; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(a[t][pl].x), "v"(a[t][pl].y), "v"(a[t][pl].z), "v"(a[t][pl].w));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(b[t][pl]));
        }
        __builtin_amdgcn_s_setprio(0);
        if (++cs == nsteps) {
            // the tile is complete: straight from the accumulator layout - per register two rows of 32 consecutive floats
            const int tile = vid0 + cj * (int)gridDim.x;
            const int bx = tile / p.tiles_n, by = tile - bx * p.tiles_n;
            const int m0 = bx * PGB_BM, n0 = by * PGB_BN;
            const int col = n0 + wn * 128 + (lane & 31);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float *dst = p.c + (size_t)m * p.ldc + col;
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        if (!(p.dbg & 4) && m < p.M && col + tn * 32 < p.N) dst[tn * 32] = acc[tm][tn][e];
                        acc[tm][tn][e] = 0.0f;
                    }
                }
            cs = 0; ++cj;
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned t_ = b0; b0 = b1; b1 = b2; b2 = t_;
    }
    if (!half) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    if (p.stamps && tid == 0) {
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 5;
        o[0] = st_t0; o[1] = pg_now(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

// ---- G kind, 256 x 256 tile, ping-pong halves --------------------------------------------------------------------------------
// Filter gradient of a Winograd layer: dU[k] (M = Cout, N = Cin) = sum over tiles t of W[k][t][:]^T V[k][t][:], K = the tile axis.
// The same skeleton as k_pgemm_pp (three-stage LDS-DMA ring, two wave groups offset by half a step).  Operands:
//   A = W = transformed output gradient, written by k_wino_gy as bf16 planes in the "PR" layout unsigned short [rows][3][Cout]
//       (a (t, plane) row of the tile = 512 contiguous bytes, whole lines); fragments by ds_read_b64_tr_b16 (the LDS rows are t, the
//       MFMA wants 8 consecutive t per lane: the hardware transpose read), no VALU.  Image [16 t][3 planes][32 chunks of 16 B], chunk c of
//       row t at c ^ ((t & 3) << 2): the four t rows of a transpose block land in four different 64-byte bank groups.
//   B = V = transformed input, float32 [rows][Cin] as the forward pass left it; a lane reads its column (8 consecutive t of one channel)
//       with four ds_read2st64_b32 and splits it into the three planes in registers.
// Wave group g owns output rows [128 g, 128 g + 128); its four waves own 64 columns each: 4 x 2 MFMA tiles, four W fragments (free) and
// two V fragments (68 VALU instructions) per step.  Output: slab (split, M, nbatch, N) partial sums, straight from the accumulators.
__global__ __launch_bounds__(PGB_THREADS, 2) void k_pgemm_gpp(PlaneGemmP p) {
    constexpr int W_ROW = 3 * 512, W_STAGE = 16 * W_ROW, V_ROW = 1024, V_STAGE = 16 * V_ROW;       // 24 KB + 16 KB = PGB_STAGE
    static_assert(W_STAGE + V_STAGE == PGB_STAGE, "stage size");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PGB_NSTAGE * PGB_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wn = wave & 3;
    int vid = blockIdx.x;
    {
        const int q = p.remap_n >> 3, r = p.remap_n & 7, xcd = vid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    // (((split * nbatch + batch) * tiles_m + M tile) * tiles_n + N tile)
    const int by = vid % p.tiles_n; vid /= p.tiles_n;
    const int bx = vid % p.tiles_m; vid /= p.tiles_m;
    const int batch = vid % p.nbatch, split = vid / p.nbatch;
    const int m0 = bx * PGB_BM, n0 = by * PGB_BN;
    const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int nsteps = kend > kbeg ? (kend - kbeg) / PG_BK : 0;
    const unsigned rowW = (unsigned)p.M * 6u, rowV = (unsigned)p.N * 4u;          // bytes per row t

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    // W pieces (1 KiB = two (t, plane) rows of 512 B): piece j covers image rows 2j, 2j+1 (row index = 3 t + plane); wave w: j = w, w+8, w+16
    // V pieces (1 KiB = one t row): wave w: t = w, w + 8
    unsigned voffW[3], voffV[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int irow = 2 * (wave + 8 * i) + (lane >> 5), t = irow / 3, pl = irow - 3 * t;
        const int pc = lane & 31, lc = pc ^ ((t & 3) << 2);
        const int co = m0 + lc * 8;
        voffW[i] = (co < p.M) ? (unsigned)t * rowW + (unsigned)pl * (unsigned)p.M * 2u + (unsigned)co * 2u : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave + 8 * i, ci = n0 + lane * 4;
        voffV[i] = (ci < p.N) ? (unsigned)t * rowV + (unsigned)ci * 4u : 0xFFFFFFFFu;
    }
    const unsigned row0 = (unsigned)(batch * p.batch_rows + kbeg);
    auto issue = [&](unsigned buf, unsigned kstep) {
        unsigned char *sw = smem + buf, *sv = sw + W_STAGE;
        const unsigned soffW = (row0 + kstep * 16u) * rowW, soffV = (row0 + kstep * 16u) * rowV;
#pragma unroll
        for (int i = 0; i < 3; ++i) pg_dma16(rsA, sw + (wave + 8 * i) * 1024, voffW[i], soffW);
#pragma unroll
        for (int i = 0; i < 2; ++i) pg_dma16(rsB, sv + (wave + 8 * i) * 1024, voffV[i], soffV);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // W fragment (tm, plane): t = 8h + q (+4), channels half*128 + tm*32 + 16 g16 + 4 pp: chunk half*16 + tm*4 + 2 g16 + (pp >> 1)
    unsigned offW[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const unsigned lc = (unsigned)(half * 16 + tm * 4 + 2 * g16 + (pp >> 1));
        offW[tm] = smem_base + (unsigned)(8 * h + q) * W_ROW + ((lc ^ ((unsigned)q << 2)) * 16u) + (unsigned)(pp & 1) * 8u;
    }
    // V fragment tn: column wn*64 + tn*32 + (lane & 31), t = 8h + j
    const unsigned offV = smem_base + (unsigned)W_STAGE + (unsigned)(8 * h) * V_ROW + (unsigned)(wn * 64 + (lane & 31)) * 4u;

    const bool loads = !(p.dbg & 8), mfma = !(p.dbg & 1);
    unsigned b0 = 0, b1 = PGB_STAGE, b2 = 2 * PGB_STAGE;
    if (nsteps > 0) {
        if (loads) { issue(b0, 0); if (nsteps > 1) issue(b1, 1); }
        if (nsteps > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (half) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    typedef unsigned pg_u32x2 __attribute__((ext_vector_type(2)));
#define PG_DSTR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define PG_DSR2ST64(DST, ADDR, O0, O1) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(DST) : "v"(ADDR), "n"(O0), "n"(O1) : "memory")
    for (int s = 0; s < nsteps; ++s) {
        if (s + 2 < nsteps && loads) issue(b2, (unsigned)(s + 2));
        // V columns first (their split is the VALU work of the phase), then the W transpose reads
        pg_u32x2 rv[2][4];
        const unsigned av = offV + b0;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            PG_DSR2ST64(rv[tn][0], av + tn * 128, 0, 4); PG_DSR2ST64(rv[tn][1], av + tn * 128, 8, 12);
            PG_DSR2ST64(rv[tn][2], av + tn * 128, 16, 20); PG_DSR2ST64(rv[tn][3], av + tn * 128, 24, 28);
        }
        pg_u32x2 rw[4][3][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const unsigned aw = offW[tm] + b0;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { PG_DSTR(rw[tm][pl][0], aw, pl * 512); PG_DSTR(rw[tm][pl][1], aw, pl * 512 + 4 * W_ROW); }
        }
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");          // the 8 column reads are back (lgkmcnt counts to 15)
#pragma unroll
        for (int tm = 2; tm < 4; ++tm) {
            const unsigned aw = offW[tm] + b0;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { PG_DSTR(rw[tm][pl][0], aw, pl * 512); PG_DSTR(rw[tm][pl][1], aw, pl * 512 + 4 * W_ROW); }
        }
        __builtin_amdgcn_sched_barrier(0);
        uint4 bfr[2][3];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            uint2 h0, m0_, l0, h1, m1, l1;
            split3_bf16x4(make_float4(__uint_as_float(rv[tn][0].x), __uint_as_float(rv[tn][0].y), __uint_as_float(rv[tn][1].x), __uint_as_float(rv[tn][1].y)), h0, m0_, l0);
            split3_bf16x4(make_float4(__uint_as_float(rv[tn][2].x), __uint_as_float(rv[tn][2].y), __uint_as_float(rv[tn][3].x), __uint_as_float(rv[tn][3].y)), h1, m1, l1);
            bfr[tn][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            bfr[tn][1] = make_uint4(m0_.x, m0_.y, m1.x, m1.y);
            bfr[tn][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(bfr[tn][pl].x), "+v"(bfr[tn][pl].y), "+v"(bfr[tn][pl].z), "+v"(bfr[tn][pl].w));
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (mfma) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                uint4 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = make_uint4(rw[tm][pl][0].x, rw[tm][pl][0].y, rw[tm][pl][1].x, rw[tm][pl][1].y);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) PG_MMA6(acc[tm][tn], a[0], a[1], a[2], bfr[tn][0], bfr[tn][1], bfr[tn][2]);
            }
        } else {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(rw[tm][pl][0]), "v"(rw[tm][pl][1]));
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(bfr[tn][pl].x), "v"(bfr[tn][pl].y), "v"(bfr[tn][pl].z), "v"(bfr[tn][pl].w));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned t_ = b0; b0 = b1; b1 = b2; b2 = t_;
    }
    if (!half) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    if (!(p.dbg & 4)) {
        float *cbase = p.c + (size_t)split * p.M * p.ldc + (size_t)batch * p.N;
        const int col = n0 + wn * 64 + (lane & 31);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + half * 128 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float *dst = cbase + (size_t)m * p.ldc + col;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    if (m < p.M && col + tn * 32 < p.N) dst[tn * 32] = acc[tm][tn][e];
            }
    }
}

// float32 (R, C) -> P16R4 planes, R % 4 == 0, C % 16 == 0 (tests; the filter transforms write this layout themselves)
__device__ __forceinline__ unsigned pg_off_r4(unsigned row, unsigned c, unsigned C) {       // byte offset of plane 0 (planes: + 128 each)
    return ((row >> 2) * (C >> 4) + (c >> 4)) * 384u + (row & 3u) * 32u + (c & 15u) * 2u;
}

// ---- G kind ---------------------------------------------------------------------------------------------------------------
template <int BM_, int BN_>
__global__ __launch_bounds__(PG_THREADS, 2) void k_pgemm_g(PlaneGemmP p) {
    constexpr int TM = BM_ / 64, TN = BN_ / 64;
    constexpr int NCHA = BM_ / 16 * 6, NCHB = BN_ / 16 * 6;            // 16-B chunks per stage row: 48 / 24
    constexpr int ROWA = NCHA * 16, ROWB_ = NCHB * 16;                 // 768 / 384 bytes
    constexpr int A_STAGE = PG_BK * ROWA, B_STAGE = PG_BK * ROWB_, STAGE = A_STAGE + B_STAGE;
    constexpr int NPA = A_STAGE / 1024, NPB = B_STAGE / 1024;          // 12 / 6
    constexpr int SA = (NPA + 3) / 4, SB = (NPB + 3) / 4;
    constexpr int EPI_BYTES = 4 * 32 * 36 * 4;
    constexpr int SM0 = STAGE > EPI_BYTES ? STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem0[SM0];
    __shared__ __attribute__((aligned(1024))) unsigned char smem1[STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int vid = blockIdx.x;
    {
        const int q = p.remap_n >> 3, r = p.remap_n & 7, xcd = vid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    // (((split * nbatch + batch) * tiles_m + M tile) * tiles_n + N tile): one split = one row range of every batch
    const int by = vid % p.tiles_n; vid /= p.tiles_n;
    const int bx = vid % p.tiles_m; vid /= p.tiles_m;
    const int batch = vid % p.nbatch, split = vid / p.nbatch;
    const int m0 = bx * BM_, n0 = by * BN_;
    const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int nsteps = kend > kbeg ? (kend - kbeg) / PG_BK : 0;
    const unsigned rowA = (unsigned)p.M * 6u, rowB = (unsigned)p.N * 6u;            // bytes per row t of the operands

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    unsigned voffA[SA], voffB[SB];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int f = 64 * (wave + 4 * i) + lane, t = f / NCHA, pc = f - t * NCHA;
        const int lc = pc ^ (((NCHA == 48) ? (t & 3) : ((t >> 1) & 1)) << 2);
        const int ch = m0 + (lc / 6) * 16;
        voffA[i] = (ch < p.M && t < PG_BK) ? (unsigned)t * rowA + (unsigned)(m0 / 16) * 96u + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int f = 64 * (wave + 4 * i) + lane, t = f / NCHB, pc = f - t * NCHB;
        const int lc = pc ^ (((NCHB == 48) ? (t & 3) : ((t >> 1) & 1)) << 2);
        const int ch = n0 + (lc / 6) * 16;
        voffB[i] = (ch < p.N && t < PG_BK) ? (unsigned)t * rowB + (unsigned)(n0 / 16) * 96u + (unsigned)lc * 16u : 0xFFFFFFFFu;
    }
    auto issue = [&](unsigned char *sa, unsigned soffA, unsigned soffB) {
        unsigned char *sb = sa + A_STAGE;
#pragma unroll
        for (int i = 0; i < SA; ++i)
            if (NPA % 4 == 0 || wave + 4 * i < NPA) pg_dma16(rsA, sa + (wave + 4 * i) * 1024, voffA[i], soffA);
#pragma unroll
        for (int i = 0; i < SB; ++i)
            if (NPB % 4 == 0 || wave + 4 * i < NPB) pg_dma16(rsB, sb + (wave + 4 * i) * 1024, voffB[i], soffB);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // Transposed fragment reads (ds_read_b64_tr_b16, cdna_hip_programming.md T10): per 16-lane group a 4 (t) x 16 (channel) block;
    // lane 4q + pp of the group gives the address of row q, channels 4pp .. 4pp+3.  This lane: t = 8h + q (+4 for the second read),
    // channels row0 + 16 * g16 + 4 * pp, i.e. logical chunk ((row0 / 16 + g16) * 6 + 2 * plane + (pp >> 1)), byte (pp & 1) * 8.
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    unsigned offA[TM][3], offB[TN][3];       // offB relative to the stage, like offA
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int lc = ((wm * (BM_ / 2) + t * 32) / 16 + g16) * 6 + 2 * pl + (pp >> 1);
            const int pc = lc ^ (((NCHA == 48) ? q : (q >> 1)) << 2);
            offA[t][pl] = (unsigned)(8 * h + q) * ROWA + (unsigned)pc * 16u + (unsigned)(pp & 1) * 8u;
        }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int lc = ((wn * (BN_ / 2) + t * 32) / 16 + g16) * 6 + 2 * pl + (pp >> 1);
            const int pc = lc ^ (((NCHB == 48) ? q : (q >> 1)) << 2);
            offB[t][pl] = (unsigned)A_STAGE + (unsigned)(8 * h + q) * ROWB_ + (unsigned)pc * 16u + (unsigned)(pp & 1) * 8u;
        }
    auto trfrag = [&](const unsigned char *base, unsigned off, int rowbytes) -> uint4 {
        const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(base + off));
        const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(base + off + 4 * rowbytes));
        const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
        return make_uint4(u0.x, u0.y, u1.x, u1.y);
    };
    auto compute = [&](const unsigned char *st) {
        uint4 a[TM][3], b[TN][3];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[t][pl] = trfrag(st, offA[t][pl], ROWA);
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[t][pl] = trfrag(st, offB[t][pl], ROWB_);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) PG_MMA6(acc[tm][tn], a[tm][0], a[tm][1], a[tm][2], b[tn][0], b[tn][1], b[tn][2]);
    };

    unsigned soffA = (unsigned)(batch * p.batch_rows + kbeg) * rowA, soffB = (unsigned)(batch * p.batch_rows + kbeg) * rowB;
    if (nsteps > 0) {           // nsteps is even: the K range of a split is a multiple of 32 rows
        issue(smem0, soffA, soffB);
        for (int s = 0; s + 2 < nsteps; s += 2) {
            PG_SYNC();
            issue(smem1, soffA + PG_BK * rowA, soffB + PG_BK * rowB);
            __builtin_amdgcn_sched_barrier(0);
            compute(smem0);
            PG_SYNC();
            soffA += 2 * PG_BK * rowA; soffB += 2 * PG_BK * rowB;
            issue(smem0, soffA, soffB);
            __builtin_amdgcn_sched_barrier(0);
            compute(smem1);
        }
        PG_SYNC();
        issue(smem1, soffA + PG_BK * rowA, soffB + PG_BK * rowB);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem0);
        PG_SYNC();
        compute(smem1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float *cbase = p.c + (size_t)split * p.M * p.ldc + (size_t)batch * p.N;
    pg_store_tiles<TM, TN>(acc, reinterpret_cast<float *>(smem0) + wave * (32 * 36), cbase, (size_t)p.ldc, m0 + wm * (BM_ / 2),
                           n0 + wn * (BN_ / 2), p.M, p.N, lane);
}

// ---- float32 (R, C) -> P16 planes (tests, and operands that no transform kernel produces) -----------------------------------------
__device__ __forceinline__ void pg_store_planes4(__amdgpu_buffer_rsrc_t rs, unsigned byteoff, const float4 v) {
    // the planes of 4 consecutive channels (channel % 16 in {0,4,8,12}): 8 bytes in each of the three 32-byte plane segments
    uint2 hi, mid, lo;
    split3_bf16x4(v, hi, mid, lo);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{hi.x, hi.y}, rs, byteoff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{mid.x, mid.y}, rs, byteoff + 32u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo.x, lo.y}, rs, byteoff + 64u, 0, 0);
}
// byte offset of channel c (multiple of 4) of row `row` in a P16 matrix of C channels
__device__ __forceinline__ unsigned pg_off(unsigned row, unsigned c, unsigned C) {
    return (row * (C >> 4) + (c >> 4)) * 96u + (c & 15u) * 2u;
}
__global__ __launch_bounds__(256) void k_split_planes(const float *__restrict__ x, unsigned short *__restrict__ planes, unsigned R, unsigned C, int r4) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x, C4 = C / 4u;
    if (i >= R * C4) return;
    const unsigned row = i / C4, c = (i - row * C4) * 4u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)planes, 0, R * C * 6u, 0x00020000);
    if (!r4) { pg_store_planes4(rs, pg_off(row, c, C), ldg4(x + (size_t)row * C + c)); return; }
    uint2 hi, mid, lo;
    split3_bf16x4(ldg4(x + (size_t)row * C + c), hi, mid, lo);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if (r4 == 2) {          // "PR": [R][3][C]
        const unsigned o2 = (row * 3u * C + c) * 2u;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{hi.x, hi.y}, rs, o2, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{mid.x, mid.y}, rs, o2 + C * 2u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo.x, lo.y}, rs, o2 + C * 4u, 0, 0);
        return;
    }
    const unsigned o = pg_off_r4(row, c, C);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{hi.x, hi.y}, rs, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{mid.x, mid.y}, rs, o + 128u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo.x, lo.y}, rs, o + 256u, 0, 0);
}

template <int KIND>
int launch_pgemm(PlaneGemmP &p, int bm, int bn, hipStream_t st) {
    p.tiles_m = mrcnn::cdiv(KIND == 0 ? p.M : p.M, bm);
    p.tiles_n = mrcnn::cdiv(p.N, bn);
    const int total = p.tiles_m * p.tiles_n * ((KIND != 1 && KIND != 5) ? 1 : p.nbatch * p.ksplit);
    p.remap_n = total;
    const dim3 grid(total), blk(PG_THREADS);
    if (KIND == 0) {
        if (bn == 128) hipLaunchKernelGGL((k_pgemm_f<128, 128>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((k_pgemm_f<128, 64>), grid, blk, 0, st, p);
    } else if (KIND == 2) {
        if (bn == 128) hipLaunchKernelGGL((k_pgemm_fa<128, 128>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((k_pgemm_fa<128, 64>), grid, blk, 0, st, p);
    } else if (KIND == 3) {
        hipLaunchKernelGGL(k_pgemm_big, grid, dim3(PGB_THREADS), 0, st, p);
    } else if (KIND == 4) {         // persistent: one workgroup per CU (122,880 B of LDS each), fewer when there are fewer tiles
        hipLaunchKernelGGL(k_pgemm_pp, dim3(std::min(total, g_cus())), dim3(PGB_THREADS), 0, st, p);
    } else if (KIND == 5) {
        hipLaunchKernelGGL(k_pgemm_gpp, grid, dim3(PGB_THREADS), 0, st, p);
    } else {
        if (bm == 128 && bn == 128) hipLaunchKernelGGL((k_pgemm_g<128, 128>), grid, blk, 0, st, p);
        else if (bm == 128) hipLaunchKernelGGL((k_pgemm_g<128, 64>), grid, blk, 0, st, p);
        else if (bn == 128) hipLaunchKernelGGL((k_pgemm_g<64, 128>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((k_pgemm_g<64, 64>), grid, blk, 0, st, p);
    }
    return 0;
}

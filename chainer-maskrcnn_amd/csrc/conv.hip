// fp32 MFMA implicit-GEMM convolution for gfx950: forward, backward-data, backward-filter.
//
// Replaces the cuDNN convolution / cuBLAS sgemm calls Chainer makes for
//   chainer_maskrcnn/model/extractor/feature_pyramid_network.py:48-68   (ResNet-50 + FPN)
//   chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:131-141 (RPN head)
//   chainer_maskrcnn/model/head/fpn_roi_mask_head.py:65-69,79-83        (box / mask heads,
//   L.Linear and the 2x2/2 L.Deconvolution2D are expressed as 1x1 convolutions by the host).
//
// Layout: activations NHWC, weights (Cout, KH, KW, Cin): the GEMM K axis (kh, kw, c) is
// contiguous in both operands of the forward pass.
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-wise an fmaf chain) - gfx950
// has no TF32/xf32, and the parity bar is 1e-3 relative fp32.
//
// One workgroup (256 threads = 2x2 waves) computes a BM x BN tile (128x128, 128x64, 64x128, 64x64) of C = A * B with
// a 32-deep K step staged through LDS (single buffer, next step's global loads in flight during the MFMAs).  Each
// wave owns (BM/2) x (BN/2) = TM x TN MFMA tiles of 32x32.  Operand tiles live in LDS either "k-contiguous" (KC:
// [row][k], read with ds_read_b128) or "row-contiguous" (RC: [k][row], read with ds_read_b32), whichever matches how
// the operand sits in HBM:
//   forward          A = im2col(x)  KC (gathered rows)   B = w            KC
//   backward-data    A = im2col(gy) KC (gathered rows)   B = w[co][..ci]  RC   (k = cout)
//   backward-filter  A = gy         RC (k = pixel)       B = x gathered   RC   (k = pixel), split-K
// Within each 8-wide k group lane half h uses k = 4h..4h+3 (for both operands), which is a
// permutation of the K sum and lets one ds_read_b128 feed four MFMAs.
// Grid: 1-D with an XCD remap; tiles just past a whole number of rounds of resident workgroups are split along K
// ("tail split"); all partial sums go to slabs that are added in fixed order (bit-reproducible results).
// Epilogue: 32x32 accumulator tiles are transposed through per-wave LDS tiles and stored as float4.
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;    // KC tile row stride (floats): 144 B, keeps b128 alignment, conflict-free b128 reads
constexpr int CONV_THREADS = 256;

enum { MODE_FWD = 0, MODE_BWD_DATA = 1, MODE_BWD_FILTER = 2 };

struct ConvP {
    const float *a;       // FWD: x      BWD_DATA: gy     BWD_FILTER: gy
    const float *b;       // FWD: w      BWD_DATA: w      BWD_FILTER: x
    float *c;             // FWD: y      BWD_DATA: gx     BWD_FILTER: gw or split-K slab
    const float *bias;    // FWD only (nullable)
    const float *relu_x;  // BWD_DATA only (nullable): the layer's input; gx is zeroed where relu_x <= 0 (fused ReLU backward)
    int N, H, W, Cin;     // input tensor (x / gx)
    int Ho, Wo, Cout;     // output tensor (y / gy)
    int KH, KW, stride, pad;
    int pad_w;            // FWD only: horizontal padding (== pad except for the rectangular kernels of mrcnn_conv2d_fwd_rect_f32)
    int relu;
    int accumulate;       // BWD_DATA: gx += result
    int smallc;           // Cin == 4: the K axis is (tap, 4 channels) flattened, one 16-B chunk per tap
    int M, Ng;            // GEMM M and N extents
    int ksplit, kchunk;   // BWD_FILTER: number of K splits, pixels per split;  FWD / BWD_DATA: splits, K steps per split
    float *slab;          // FWD / BWD_DATA split-K: (ksplit, M, ldc) partial sums
    unsigned bytes_a, bytes_b;   // sizes of the a / b tensors (buffer descriptors of the bounds-checked gather loads)
    // The grid is 1-D.  Workgroup ids are first remapped so that every XCD (ids equal mod 8 share one, and its L2)
    // works on a contiguous range of virtual ids; a virtual id decodes to
    //   FWD / BWD_DATA: ((M tile * tiles_n + N tile) * ksplit + split)   - the N tiles of an M tile share its A rows
    //   BWD_FILTER:     (((split * taps + tap) * tiles_m + M tile) * tiles_n + N tile) - one split = one pixel range
    int tiles_m, tiles_n, remap_n;
    // FWD / BWD_DATA tail split: tiles [0, tail_full) are whole, each later tile is computed by tail_ks workgroups
    // (tail_kchunk K steps each, ids >= remap_n = tail_full) that write partial tiles to `slab`.
    int tail_ks, tail_full, tail_kchunk;
    // FWD as the batched GEMM of the Winograd path: GEMM rows [k*wbatch_rows, (k+1)*wbatch_rows) use weight matrix k
    // (b + k*Ng*Cin); wbatch_rows is a multiple of every tile height.  0 = ordinary convolution.
    int wbatch_rows, wbatch_n;   // wbatch_n = number of GEMMs (16 or 36)
    // FWD, un-split launches only (nullable): BatchNorm statistics of the output from the epilogue.  Row (M tile * 2 + wave
    // row) of bn_part (rows, 2, Ng) receives the sums and the sums of squares of that wave's BM_/2 output rows, per channel.
    float *bn_part;
    int dbg;              // split-operand kernels, measurement only (mrcnn_debug_conv_parts): 1 = no MFMA, 2 = no in-loop global loads, 4 = no epilogue (1, 2: plain loop only), 8 = plain K loop
};

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// read-once stream (split-K slabs, the Winograd GEMM's output M): non-temporal, does not displace lines other kernels re-read
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4s(const float *p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
// ---- MFMA over one staged K step -----------------------------------------------------------
// The workgroup tile is BM_ x BN_ (128 or 64 each), 2x2 waves, a wave owns (BM_/2) x (BN_/2) = TM x TN MFMA tiles.
template <bool A_KC, bool B_KC, int BM_, int BN_>
__device__ __forceinline__ void mma_step(const float *__restrict__ sA, const float *__restrict__ sB,
                                         f32x16 (&acc)[BM_ / 64][BN_ / 64], int wm, int wn, int lane) {
    constexpr int TM = BM_ / 64, TN = BN_ / 64;
    constexpr int LDA = BM_ + 4, LDB = BN_ + 4;     // RC tile row strides
    const int r = lane & 31, h = lane >> 5;
    float af[2][TM][4], bf[2][TN][4];
    auto load_frag = [&](int kg, float (&a)[TM][4], float (&b)[TN][4]) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int row = wm * (BM_ / 2) + t * 32 + r;
            if (A_KC) {
                const float4 v = *reinterpret_cast<const float4 *>(&sA[row * LDK + kg * 8 + 4 * h]);
                a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[t][i] = sA[(kg * 8 + 4 * h + i) * LDA + row];
            }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int col = wn * (BN_ / 2) + t * 32 + r;
            if (B_KC) {
                const float4 v = *reinterpret_cast<const float4 *>(&sB[col * LDK + kg * 8 + 4 * h]);
                b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[t][i] = sB[(kg * 8 + 4 * h + i) * LDB + col];
            }
        }
    };
    // operands of k-group kg+1 are fetched from LDS before the 16/8/4 MFMAs of k-group kg are issued
    load_frag(0, af[0], bf[0]);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
        if (kg + 1 < BK / 8) load_frag(kg + 1, af[(kg + 1) & 1], bf[(kg + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg & 1][tm][i], bf[kg & 1][tn][i], acc[tm][tn], 0, 0, 0);
    }
}

// ---- EXPLORATORY (opt-in, mrcnn_conv2d_set_split_bf16; never the default): the same K step with three-term split-bf16 operands.
// gfx950 has no xf32, so an fp32-input MFMA runs at 1/16 of the bf16 rate.  Every operand x is staged as hi = bf16(x) and
// lo = bf16(x - hi) (16 significant bits between them) and a product a*b is accumulated in float32 as al*bh + ah*bl + ah*bh (the
// al*bl term, <= 2^-18 |ab|, is dropped): three v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32
// (64 cycles each) per 16 of K - 5.3x fewer MFMA cycles, products accurate to ~1e-5 relative instead of 6e-8.  LDS holds the two
// bf16 planes of each operand ([row][32 k], row stride 80 B): the same bytes as the float32 tiles.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int LDKH = BK + 8;           // bf16 elements per LDS row of a split plane (80 B: 16-B aligned fragments)
__device__ __forceinline__ void split_bf16x4(const float4 v, uint2 &hi, uint2 &lo) {
    const f32x2_t v01 = {v.x, v.y}, v23 = {v.z, v.w};
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2_t));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2_t));
    const f32x2_t r01 = {v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u)};
    const f32x2_t r23 = {v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u)};
    hi = make_uint2(h01, h23);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2_t)));
}
// The same split with IEEE half planes (11 significant bits each: 22 between them, products accurate to ~5e-7 - within 10x of
// float32 - at the bf16 MFMA rate), for operands that live in half's range: hi = rtz(x), lo = rtz(x - hi).  Values beyond
// +-65504 would overflow (forward activations, weights and Winograd-domain inputs of this model stay below 1e4); values below
// 6e-5 keep an absolute precision of 6e-8, which is why the BACKWARD passes (gradients of 1e-6 .. 1e-3) use the bf16 planes.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_f16x4(const float4 v, uint2 &hi, uint2 &lo) {
    const f16x2_t h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2_t l01 = __builtin_amdgcn_cvt_pkrtz(v.x - (float)h01[0], v.y - (float)h01[1]);
    const f16x2_t l23 = __builtin_amdgcn_cvt_pkrtz(v.z - (float)h23[0], v.w - (float)h23[1]);
    hi = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}
// Fragment of a 32-row MFMA tile from a split plane.  KC plane ([row][32 k], stride LDKH): one ds_read_b128.  RC plane ([k][rows],
// the layout the operand has in HBM when K is the pixel or the output-channel axis): two ds_read_b64_tr_b16 - per 16-lane group the
// hardware reads a 4 (k) x 16 (rows) block and hands lane i the 4 k values of row i (cdna_hip_programming.md T10); EXEC is full here.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
// SPLIT = 3: THREE bf16 planes hi + mid + lo = the float32 operand EXACTLY (3 x 8 significant bits = float32's 24; bf16 has float32's
// exponent range), and the six products of weight >= 2^-16: lh + hl + mm + mh + hm + hh.  The dropped ml, lm, ll are <= 3 x 2^-24
// of |ab| - the size of the rounding the float32 MFMA itself makes when it adds a product to its accumulator - so this GEMM is a
// float32-ACCURATE emulation (the "BF16x6 / BF16x9" float32 modes of vendor BLAS libraries) at 3/8 of the float32 MFMA cycles.
__device__ __forceinline__ void split3_bf16x4(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
    const f32x2_t v01 = {v.x, v.y}, v23 = {v.z, v.w};
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2_t));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2_t));
    const f32x2_t r01 = {v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u)};
    const f32x2_t r23 = {v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u)};
    const unsigned m01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2_t));
    const unsigned m23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2_t));
    const f32x2_t q01 = {r01[0] - __uint_as_float(m01 << 16), r01[1] - __uint_as_float(m01 & 0xffff0000u)};
    const f32x2_t q23 = {r23[0] - __uint_as_float(m23 << 16), r23[1] - __uint_as_float(m23 & 0xffff0000u)};
    hi = make_uint2(h01, h23);
    mid = make_uint2(m01, m23);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(q01, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(q23, bf16x2_t)));
}
template <bool KC>
__device__ __forceinline__ uint4 split_frag(const unsigned short *__restrict__ plane, int row0, int ld, int kk, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if (KC) return *reinterpret_cast<const uint4 *>(plane + (row0 + r) * LDKH + kk * 16 + 8 * h);
    const int li = lane & 15, q = li >> 2, pp = li & 3, g16 = (lane >> 4) & 1;
    const unsigned short *a0 = plane + (kk * 16 + 8 * h + q) * ld + row0 + 16 * g16 + 4 * pp;
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)a0);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(a0 + 4 * ld));
    const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
}
template <bool A_KC, bool B_KC, int BM_, int BN_, int SPLIT>
__device__ __forceinline__ void mma_step_split(const unsigned short *__restrict__ aH, const unsigned short *__restrict__ aL,
                                               const unsigned short *__restrict__ bH, const unsigned short *__restrict__ bL,
                                               f32x16 (&acc)[BM_ / 64][BN_ / 64], int wm, int wn, int lane,
                                               const unsigned short *__restrict__ aM = nullptr, const unsigned short *__restrict__ bM = nullptr) {
    constexpr int TM = BM_ / 64, TN = BN_ / 64;
    constexpr int LDAH = BM_ + 32, LDBH = BN_ + 32;
    if constexpr (SPLIT == 3) {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            uint4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                ah[t] = split_frag<A_KC>(aH, wm * (BM_ / 2) + t * 32, LDAH, kk, lane);
                am[t] = split_frag<A_KC>(aM, wm * (BM_ / 2) + t * 32, LDAH, kk, lane);
                al[t] = split_frag<A_KC>(aL, wm * (BM_ / 2) + t * 32, LDAH, kk, lane);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                bh[t] = split_frag<B_KC>(bH, wn * (BN_ / 2) + t * 32, LDBH, kk, lane);
                bm[t] = split_frag<B_KC>(bM, wn * (BN_ / 2) + t * 32, LDBH, kk, lane);
                bl[t] = split_frag<B_KC>(bL, wn * (BN_ / 2) + t * 32, LDBH, kk, lane);
            }
#define MRCNN_BF(x) __builtin_bit_cast(bf16x8_t, x)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {       // smallest products first
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MRCNN_BF(al[tm]), MRCNN_BF(bh[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MRCNN_BF(ah[tm]), MRCNN_BF(bl[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MRCNN_BF(am[tm]), MRCNN_BF(bm[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MRCNN_BF(am[tm]), MRCNN_BF(bh[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MRCNN_BF(ah[tm]), MRCNN_BF(bm[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MRCNN_BF(ah[tm]), MRCNN_BF(bh[tn]), acc[tm][tn], 0, 0, 0);
                }
#undef MRCNN_BF
        }
        return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        uint4 ah[TM], al[TM], bh[TN], bl[TN];           // 8 16-bit elements each
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            ah[t] = split_frag<A_KC>(aH, wm * (BM_ / 2) + t * 32, LDAH, kk, lane);
            al[t] = split_frag<A_KC>(aL, wm * (BM_ / 2) + t * 32, LDAH, kk, lane);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            bh[t] = split_frag<B_KC>(bH, wn * (BN_ / 2) + t * 32, LDBH, kk, lane);
            bl[t] = split_frag<B_KC>(bL, wn * (BN_ / 2) + t * 32, LDBH, kk, lane);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if constexpr (SPLIT == 1) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, al[tm]), __builtin_bit_cast(bf16x8_t, bh[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah[tm]), __builtin_bit_cast(bf16x8_t, bl[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah[tm]), __builtin_bit_cast(bf16x8_t, bh[tn]), acc[tm][tn], 0, 0, 0);
                } else {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, al[tm]), __builtin_bit_cast(f16x8_t, bh[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah[tm]), __builtin_bit_cast(f16x8_t, bl[tn]), acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah[tm]), __builtin_bit_cast(f16x8_t, bh[tn]), acc[tm][tn], 0, 0, 0);
                }
            }
    }
}

// ---- the kernel ----------------------------------------------------------------------------
// All tensor offsets are 32-bit element indices (the host checks every tensor has < 2^31 elements); the per-row
// part of every gather address is computed once, a K step only adds a wave-uniform tap/channel offset.
__device__ __forceinline__ void divmod_small(int v, int d, float inv, int &q, int &r) {
    // exact for 0 <= v < 2^24: float estimate + one correction step each way
    q = (int)((float)v * inv);
    r = v - q * d;
    if (r >= d) { r -= d; ++q; }
    if (r < 0) { r += d; --q; }
}

template <int MODE, int BM_, int BN_, bool SMALLC, int SPLIT = 0>         // SPLIT: 0 float32 MFMA, 1 bf16 hi/lo planes, 2 half hi/lo planes
__global__ __launch_bounds__(CONV_THREADS, 2) void k_conv_igemm(ConvP p) {
    constexpr bool BF3 = SPLIT != 0;
    static_assert(!BF3 || !SMALLC, "split operands: not for the 4-channel image layer");
    constexpr bool A_KC = (MODE != MODE_BWD_FILTER);
    constexpr bool B_KC = (MODE == MODE_FWD);
    constexpr int TM = BM_ / 64, TN = BN_ / 64;
    constexpr int LDA = BM_ + 4, LDB = BN_ + 4;
    // (split operands: two 16-bit planes per operand - KC: rows x LDKH, RC: BK x (rows + 32) elements each; in floats: half of it x 2)
    constexpr int LDAH = BM_ + 32, LDBH = BN_ + 32;
    constexpr int A_PLANE = A_KC ? BM_ * LDKH : BK * LDAH, B_PLANE = B_KC ? BN_ * LDKH : BK * LDBH;       // 16-bit elements
    constexpr int NPL = SPLIT == 3 ? 3 : 2;            // planes per operand (16-bit elements: NPL * PLANE / 2 floats)
    constexpr int A_ELEMS = BF3 ? A_PLANE * NPL / 2 : (A_KC ? BM_ * LDK : BK * LDA);
    constexpr int B_ELEMS = BF3 ? B_PLANE * NPL / 2 : (B_KC ? BN_ * LDK : BK * LDB);
    constexpr int NA = BM_ / 32, NB = BN_ / 32;          // float4 loads per thread per K step
    // RC loader geometry: a k-row of width Wd floats is covered by Wd/4 threads; 256/(Wd/4) k-rows per pass
    constexpr int B_TPR = BN_ / 4, B_KPP = CONV_THREADS / B_TPR;
    // one LDS array: the A and B stages of the K loop, reused by the epilogue as four per-wave 32 x 36 transpose tiles
    constexpr int EPI_LD = 36, EPI_ELEMS = 4 * 32 * EPI_LD;
    constexpr int SMEM_ELEMS = (A_ELEMS + B_ELEMS) > EPI_ELEMS ? (A_ELEMS + B_ELEMS) : EPI_ELEMS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_ELEMS];
    float *const sA = smem, *const sB = smem + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int vid = blockIdx.x;
    if (vid < p.remap_n) {        // bijective XCD remap (cdna_hip_programming.md T1)
        const int q = p.remap_n >> 3, r = p.remap_n & 7, xcd = vid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    int bx, by, split, tap = 0;
    bool tailchunk = false;       // this workgroup computes a K chunk of one of the last (tail) tiles
    int tail_idx = 0;
    if (MODE == MODE_BWD_FILTER) {
        by = vid % p.tiles_n; vid /= p.tiles_n;
        bx = vid % p.tiles_m; vid /= p.tiles_m;
        const int ntap = SMALLC ? 1 : (p.wbatch_rows ? p.wbatch_n : p.KH * p.KW);
        tap = vid % ntap;                       // kh*KW+kw (Winograd batch mode: the GEMM index k)
        split = vid / ntap;
    } else {
        int tile;
        if (p.tail_ks && vid >= p.tail_full) {
            const int q = vid - p.tail_full;
            tail_idx = q / p.tail_ks;
            split = q - tail_idx * p.tail_ks;
            tile = p.tail_full + tail_idx;
            tailchunk = true;
        } else {
            tile = vid / p.ksplit;
            split = vid - tile * p.ksplit;
        }
        bx = tile / p.tiles_n;
        by = tile - bx * p.tiles_n;
    }
    const int m0 = bx * BM_, n0 = by * BN_;

    // KC loader role: 16-B chunk kc of rows r0+32i.   RC loader role: columns 4*rc.. of k index k0 + KPP*i.
    const int kc = tid & 7, r0 = tid >> 3;
    const int rcB = tid % B_TPR, k0B = tid / B_TPR;
    const int taps = p.KH * p.KW;

    // ---- per-thread row state ----------------------------------------------------------------
    // FWD / BWD_DATA: gathered A rows (output pixels m): rowoff = element offset of (n, h0, w0, 0) in the source
    // tensor, (h0, w0) = top-left source coordinate of tap (0,0); rowmask bit i = row exists.
    int a_off[NA], a_h0[NA], a_w0[NA];
    unsigned rowmask = 0;
    // weights (FWD: row n of B; BWD_DATA: handled per step);  BWD_FILTER: pixel walk state of the B loads
    int b_off[NB];
    unsigned colmask = 0;
    if (MODE != MODE_BWD_FILTER) {
        const int PW_ = (MODE == MODE_FWD) ? p.Wo : p.W, PH_ = (MODE == MODE_FWD) ? p.Ho : p.H;
        const float invW = 1.0f / (float)PW_, invH = 1.0f / (float)PH_;
        const int srcH = (MODE == MODE_FWD) ? p.H : p.Ho, srcW = (MODE == MODE_FWD) ? p.W : p.Wo;
        const int srcC = (MODE == MODE_FWD) ? (SMALLC ? 4 : p.Cin) : p.Cout;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + r0 + 32 * i;
            const bool v = m < p.M;
            rowmask |= v ? (1u << i) : 0u;
            int q, w_, n_, h_;
            divmod_small(v ? m : 0, PW_, invW, q, w_);
            divmod_small(q, PH_, invH, n_, h_);
            a_h0[i] = (MODE == MODE_FWD) ? h_ * p.stride - p.pad : h_ + p.pad;
            a_w0[i] = (MODE == MODE_FWD) ? w_ * p.stride - p.pad_w : w_ + p.pad;
            a_off[i] = ((n_ * srcH + a_h0[i]) * srcW + a_w0[i]) * srcC;
        }
    }
    if (MODE == MODE_FWD) {
        const int krow = SMALLC ? taps * 4 : taps * p.Cin;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + r0 + 32 * i;
            colmask |= (n < p.Ng) ? (1u << i) : 0u;
            b_off[i] = ((p.wbatch_rows ? (m0 / p.wbatch_rows) * p.Ng : 0) + (n < p.Ng ? n : 0)) * krow;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // K-step bookkeeping
    int nsteps, kbeg = 0, kend = 0;
    const int cin_steps = p.Cin / BK, cout_steps = p.Cout / BK;
    int sbeg = 0;             // FWD / BWD_DATA split-K: this workgroup's K steps are [sbeg, sbeg + nsteps)
    if (MODE == MODE_FWD) nsteps = SMALLC ? (taps + 7) / 8 : taps * cin_steps;
    else if (MODE == MODE_BWD_DATA) nsteps = taps * cout_steps;
    else {
        const int P = p.N * p.Ho * p.Wo;
        kbeg = split * p.kchunk;
        kend = min(P, kbeg + p.kchunk);
        nsteps = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    }
    if (MODE != MODE_BWD_FILTER && (p.ksplit > 1 || tailchunk)) {
        const int kch = tailchunk ? p.tail_kchunk : p.kchunk;
        sbeg = split * kch;
        nsteps = max(0, min(nsteps, sbeg + kch) - sbeg);
    }
    // BWD_FILTER loader: thread (fp = tid/8, fc = tid%8) stages pixel kbeg + 32*step + fp of both operands, 16-B
    // channel chunks fc, fc+8, ... (128-B runs per pixel and load instruction).  One pixel per thread means ONE
    // coordinate walk per step: (h0, w0) = top-left input coordinate of the pixel's window and off0 = its element offset
    // are advanced by BK pixels = dN images + dH rows + dW columns with at most one carry each.  (A divmod per load,
    // the obvious form, is ~170 VALU instructions per step and costs 8 % on the 3x3 256->256 layers.)
    const int fp = tid >> 3, fc = tid & 7;
    int f_kh = 0, f_kw = 0;
    int f_khv[NB], f_kwv[NB], f_toff[NB];      // SMALLC: a column chunk is a tap, so the tap differs per load
    unsigned f_amask = 0, f_bmask = 0;          // bit i: channel chunk fc + 8i exists
    int f_h0 = 0, f_w0 = 0, f_off0 = 0, f_aoff = 0, f_pix = 0;
    int f_sw = 0, f_sh = 0, f_doff = 0, f_wlim = 0, f_hlim = 0, f_wback = 0, f_hback = 0, f_c1off = 0, f_c2off = 0;
    if (MODE == MODE_BWD_FILTER) {
        f_kw = tap % p.KW; f_kh = tap / p.KW;
        if (p.wbatch_rows) f_kw = f_kh = 0;        // batched 1x1 GEMMs: `tap` selects the operand block, not a shift
        const int xc = SMALLC ? 4 : p.Cin;
#pragma unroll
        for (int i = 0; i < NA; ++i) f_amask |= (m0 + (fc + 8 * i) * 4 < p.M) ? (1u << i) : 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int col = n0 + (fc + 8 * i) * 4;
            if (SMALLC) {                               // column = (tap, 4 channels)
                const int tp = col >> 2;
                f_bmask |= (tp < taps) ? (1u << i) : 0u;
                f_kwv[i] = tp % p.KW; f_khv[i] = tp / p.KW;
                f_toff[i] = (f_khv[i] * p.W + f_kwv[i]) * 4;
            } else {
                f_bmask |= (col < p.Ng) ? (1u << i) : 0u;
                f_khv[i] = f_kh; f_kwv[i] = f_kw;
                f_toff[i] = (f_kh * p.W + f_kw) * xc + col;
            }
        }
        const float invW = 1.0f / (float)p.Wo, invH = 1.0f / (float)p.Ho;
        const int P = p.N * p.Ho * p.Wo;
        f_pix = kbeg + fp;
        int q, wo, n, ho;
        divmod_small(min(f_pix, P - 1), p.Wo, invW, q, wo);
        divmod_small(q, p.Ho, invH, n, ho);
        f_h0 = ho * p.stride - p.pad;
        f_w0 = wo * p.stride - p.pad;
        f_off0 = ((n * p.H + f_h0) * p.W + f_w0) * xc + (p.wbatch_rows ? tap * p.wbatch_rows * xc : 0);
        f_aoff = f_pix * p.Cout + m0 + fc * 4 + (p.wbatch_rows ? tap * p.wbatch_rows * p.Cout : 0);
        const int dq = BK / p.Wo, dW = BK - dq * p.Wo, dN = dq / p.Ho, dH = dq - dN * p.Ho;
        f_sw = dW * p.stride; f_sh = dH * p.stride;
        f_doff = ((dN * p.H + f_sh) * p.W + f_sw) * xc;
        f_wback = p.Wo * p.stride; f_hback = p.Ho * p.stride;
        f_wlim = f_wback - p.pad; f_hlim = f_hback - p.pad;
        f_c1off = (p.stride * p.W - f_wback) * xc;
        f_c2off = (p.H - f_hback) * p.W * xc;
    }

    // FWD / BWD_DATA: K step s = (tap t = kh*KW + kw, channel step cs) is walked incrementally in scalar registers (a
    // div/mod of s per step is ~45 SALU instructions in front of the step's loads).
    int w_cs = 0, w_kh = 0, w_kw = 0, w_toff = 0, w_boff = 0;
    if (MODE == MODE_FWD && !SMALLC) {
        const int t = sbeg / cin_steps;
        w_cs = sbeg - t * cin_steps; w_kh = t / p.KW; w_kw = t - w_kh * p.KW;
        w_toff = (w_kh * p.W + w_kw) * p.Cin + w_cs * BK;
    }
    if (MODE == MODE_BWD_DATA) {
        const int t = sbeg / cout_steps;
        w_cs = sbeg - t * cout_steps; w_kh = t / p.KW; w_kw = t - w_kh * p.KW;
        w_toff = -(w_kh * p.Wo + w_kw) * p.Cout + w_cs * BK;
        w_boff = w_cs * BK * (taps * p.Cin) + t * p.Cin;
    }

    // Gather loads are hardware bounds-checked buffer loads (T8): a predicate that is false turns the byte offset into
    // 0xFFFFFFFF, which is outside the descriptor's range, so the load returns 0 - no branch, no select, and the zero
    // padding of the convolution costs one v_cndmask per load instead of four per stored value.
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, p.bytes_a, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)p.b, 0, p.bytes_b, 0x00020000);
    float4 ra[NA], rb[NB];
    auto ldA = [&](bool ok, int off) -> float4 {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (unsigned)off * 4u : 0xFFFFFFFFu, 0, 0);
        float4 f;
        __builtin_memcpy(&f, &v, 16);
        return f;
    };
    auto ldB = [&](bool ok, int off) -> float4 {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? (unsigned)off * 4u : 0xFFFFFFFFu, 0, 0);
        float4 f;
        __builtin_memcpy(&f, &v, 16);
        return f;
    };
    auto load_step = [&](int s) {
        if (MODE == MODE_FWD && SMALLC) {
            const int chunk = s * 8 + kc;                 // tap index of this thread's 16-B chunk
            const bool tv = chunk < taps;
            const int kw = chunk % p.KW, kh = chunk / p.KW;
            const int toff = (kh * p.W + kw) * 4;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hi = a_h0[i] + kh, wi = a_w0[i] + kw;
                const bool ok = tv & (bool)((rowmask >> i) & 1u) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                ra[i] = ldA(ok, a_off[i] + toff);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = ldB(tv && ((colmask >> i) & 1u), b_off[i] + chunk * 4);
        } else if (MODE == MODE_FWD) {
            const int toff = w_toff + kc * 4;
            const int woff = s * BK + kc * 4;            // the weight K axis (tap, channel) is contiguous
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hi = a_h0[i] + w_kh, wi = a_w0[i] + w_kw;
                const bool ok = (bool)((rowmask >> i) & 1u) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                ra[i] = ldA(ok, a_off[i] + toff);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = ldB((colmask >> i) & 1u, b_off[i] + woff);
            // next step: channel block, then kw, then kh
            const bool tapdone = ++w_cs == cin_steps;
            w_toff += BK;
            w_cs = tapdone ? 0 : w_cs;
            w_kw += tapdone ? 1 : 0;
            const bool rowdone = w_kw == p.KW;
            w_kw = rowdone ? 0 : w_kw;
            w_kh += rowdone ? 1 : 0;
            w_toff += rowdone ? (p.W - p.KW) * p.Cin : 0;
        } else if (MODE == MODE_BWD_DATA) {
            // gx[n,hi,wi,ci] = sum_{kh,kw,co} gy[n, hi+pad-kh, wi+pad-kw, co] * w[co][kh][kw][ci]   (stride 1)
            const int toff = w_toff + kc * 4;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int ho = a_h0[i] - w_kh, wo = a_w0[i] - w_kw;
                const bool ok = (bool)((rowmask >> i) & 1u) & ((unsigned)ho < (unsigned)p.Ho) & ((unsigned)wo < (unsigned)p.Wo);
                ra[i] = ldA(ok, a_off[i] + toff);
            }
            const int ci = n0 + rcB * 4;
            const int krow = taps * p.Cin;
            const int wbase = w_boff + ci + k0B * krow;
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = ldB(ci < p.Ng, wbase + B_KPP * i * krow);
            const bool tapdone = ++w_cs == cout_steps;
            w_toff += BK; w_boff += BK * krow;
            w_cs = tapdone ? 0 : w_cs;
            w_kw += tapdone ? 1 : 0;
            w_toff -= tapdone ? 2 * p.Cout : 0;
            w_boff += tapdone ? p.Cin - p.Cout * krow : 0;
            const bool rowdone = w_kw == p.KW;
            w_kw = rowdone ? 0 : w_kw;
            w_kh += rowdone ? 1 : 0;
            w_toff -= rowdone ? (p.Wo - p.KW) * p.Cout : 0;
        } else {
            // gw[co][kh][kw][ci] = sum_pix gy[pix][co] * x[pix shifted by (kh,kw)][ci]
            const bool pv = f_pix < kend;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = ldA(pv & (bool)((f_amask >> i) & 1u), f_aoff + 32 * i);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int hi = f_h0 + f_khv[i], wi = f_w0 + f_kwv[i];
                const bool ok = pv & (bool)((f_bmask >> i) & 1u) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                rb[i] = ldB(ok, f_off0 + f_toff[i]);
            }
            // advance the pixel by BK
            f_pix += BK;
            f_aoff += BK * p.Cout;
            f_w0 += f_sw; f_h0 += f_sh; f_off0 += f_doff;
            const bool c1 = f_w0 >= f_wlim;
            f_w0 -= c1 ? f_wback : 0; f_h0 += c1 ? p.stride : 0; f_off0 += c1 ? f_c1off : 0;
            const bool c2 = f_h0 >= f_hlim;
            f_h0 -= c2 ? f_hback : 0; f_off0 += c2 ? f_c2off : 0;
        }
    };
    unsigned short *const aH = reinterpret_cast<unsigned short *>(sA), *const aL = aH + A_PLANE, *const aM = aL + A_PLANE;
    unsigned short *const bH = reinterpret_cast<unsigned short *>(sB), *const bL = bH + B_PLANE, *const bM = bL + B_PLANE;
    auto store_step = [&]() {
        if constexpr (BF3) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                uint2 hi, lo, mid;
                if constexpr (SPLIT == 1) split_bf16x4(ra[i], hi, lo);
                else if constexpr (SPLIT == 3) split3_bf16x4(ra[i], hi, mid, lo);
                else split_f16x4(ra[i], hi, lo);
                const int o = A_KC ? (r0 + 32 * i) * LDKH + kc * 4 : fp * LDAH + (fc + 8 * i) * 4;
                *reinterpret_cast<uint2 *>(&aH[o]) = hi;
                *reinterpret_cast<uint2 *>(&aL[o]) = lo;
                if constexpr (SPLIT == 3) *reinterpret_cast<uint2 *>(&aM[o]) = mid;
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                uint2 hi, lo, mid;
                // half planes: the weight operand is scaled by 2^12 (exact; undone in the epilogue) - Winograd-domain filters
                // G g G^T of 1e-2 weights are 1e-6 .. 1e-3, below half's normal range
                if constexpr (SPLIT == 1) split_bf16x4(rb[i], hi, lo);
                else if constexpr (SPLIT == 3) split3_bf16x4(rb[i], hi, mid, lo);
                else if constexpr (MODE == MODE_FWD) split_f16x4(make_float4(rb[i].x * 4096.0f, rb[i].y * 4096.0f, rb[i].z * 4096.0f, rb[i].w * 4096.0f), hi, lo);
                else split_f16x4(rb[i], hi, lo);
                const int o = B_KC ? (r0 + 32 * i) * LDKH + kc * 4
                                   : (MODE == MODE_BWD_FILTER ? fp * LDBH + (fc + 8 * i) * 4 : (k0B + B_KPP * i) * LDBH + rcB * 4);
                *reinterpret_cast<uint2 *>(&bH[o]) = hi;
                *reinterpret_cast<uint2 *>(&bL[o]) = lo;
                if constexpr (SPLIT == 3) *reinterpret_cast<uint2 *>(&bM[o]) = mid;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float4 v = ra[i];
            if (A_KC) *reinterpret_cast<float4 *>(&sA[(r0 + 32 * i) * LDK + kc * 4]) = v;
            else *reinterpret_cast<float4 *>(&sA[fp * LDA + (fc + 8 * i) * 4]) = v;                // BWD_FILTER
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float4 v = rb[i];
            if (B_KC) *reinterpret_cast<float4 *>(&sB[(r0 + 32 * i) * LDK + kc * 4]) = v;
            else if (MODE == MODE_BWD_FILTER) *reinterpret_cast<float4 *>(&sB[fp * LDB + (fc + 8 * i) * 4]) = v;
            else *reinterpret_cast<float4 *>(&sB[(k0B + B_KPP * i) * LDB + rcB * 4]) = v;
        }
    };

    // Split operands, pipelined K loop: the raw operands of step s+1 are split into their planes in REGISTERS while the MFMAs
    // of step s run (the ~700 VALU cycles of the split in the MFMAs' shadow instead of between the two barriers), and the loads of
    // step s+2 go out as soon as the split has consumed the registers - a whole step of latency cover instead of one MFMA phase.
    uint2 pa[BF3 ? NA : 1][3], pb[BF3 ? NB : 1][3];              // [hi, lo, mid]
    auto split_regs = [&]() {
        if constexpr (BF3) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if constexpr (SPLIT == 1) split_bf16x4(ra[i], pa[i][0], pa[i][1]);
                else if constexpr (SPLIT == 3) split3_bf16x4(ra[i], pa[i][0], pa[i][2], pa[i][1]);
                else split_f16x4(ra[i], pa[i][0], pa[i][1]);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if constexpr (SPLIT == 1) split_bf16x4(rb[i], pb[i][0], pb[i][1]);
                else if constexpr (SPLIT == 3) split3_bf16x4(rb[i], pb[i][0], pb[i][2], pb[i][1]);
                else if constexpr (MODE == MODE_FWD) split_f16x4(make_float4(rb[i].x * 4096.0f, rb[i].y * 4096.0f, rb[i].z * 4096.0f, rb[i].w * 4096.0f), pb[i][0], pb[i][1]);
                else split_f16x4(rb[i], pb[i][0], pb[i][1]);
            }
        }
    };
    auto store_planes = [&]() {
        if constexpr (BF3) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int o = A_KC ? (r0 + 32 * i) * LDKH + kc * 4 : fp * LDAH + (fc + 8 * i) * 4;
                *reinterpret_cast<uint2 *>(&aH[o]) = pa[i][0];
                *reinterpret_cast<uint2 *>(&aL[o]) = pa[i][1];
                if constexpr (SPLIT == 3) *reinterpret_cast<uint2 *>(&aM[o]) = pa[i][2];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int o = B_KC ? (r0 + 32 * i) * LDKH + kc * 4
                                   : (MODE == MODE_BWD_FILTER ? fp * LDBH + (fc + 8 * i) * 4 : (k0B + B_KPP * i) * LDBH + rcB * 4);
                *reinterpret_cast<uint2 *>(&bH[o]) = pb[i][0];
                *reinterpret_cast<uint2 *>(&bL[o]) = pb[i][1];
                if constexpr (SPLIT == 3) *reinterpret_cast<uint2 *>(&bM[o]) = pb[i][2];
            }
        }
    };
    bool piped = false;
    if constexpr (BF3) piped = (p.dbg & 8) == 0;               // default; mrcnn_debug_conv_parts(8) = the plain loop (A/B)
    if (piped && nsteps > 0) {
        load_step(sbeg);
        split_regs();
        store_planes();
        load_step(sbeg + min(1, nsteps - 1));
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            mma_step_split<A_KC, B_KC, BM_, BN_, SPLIT>(aH, aL, bH, bL, acc, wm, wn, lane, aM, bM);
            split_regs();                                        // step s+1's planes (garbage after the last step: not stored)
            // the planes are only read after the barrier: without this pin hipcc sinks the whole split behind it
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NPL; ++j) asm volatile("" : "+v"(pa[i][j].x), "+v"(pa[i][j].y));
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NPL; ++j) asm volatile("" : "+v"(pb[i][j].x), "+v"(pb[i][j].y));
            load_step(sbeg + min(s + 2, nsteps - 1));
            // one MFMA, then a handful of the split's VALU instructions in its shadow
#pragma unroll
            for (int i = 0; i < TM * TN * (SPLIT == 3 ? 6 : 3) * (BK / 16); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, SPLIT == 3 ? 6 : 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (s + 1 < nsteps) store_planes();
            __syncthreads();
        }
    } else
    if (nsteps > 0) {
        load_step(sbeg);
        store_step();
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            // Always load (the last iteration re-reads its own step: harmless) so that the gather address arithmetic and
            // the global loads share ONE basic block with the MFMAs and the scheduler can interleave them.
            if constexpr (BF3) {
                if (!(p.dbg & 2)) load_step(sbeg + min(s + 1, nsteps - 1));
                if (!(p.dbg & 1)) mma_step_split<A_KC, B_KC, BM_, BN_, SPLIT>(aH, aL, bH, bL, acc, wm, wn, lane, aM, bM);
            } else {
            load_step(sbeg + min(s + 1, nsteps - 1));
            mma_step<A_KC, B_KC, BM_, BN_>(sA, sB, acc, wm, wn, lane);
            // schedule: the gather loads go out after the first quarter of the step's MFMAs (their address arithmetic is
            // hidden under those), so the data is back long before the LDS stores at the end of the step
            __builtin_amdgcn_sched_group_barrier(0x8, (TM * TN * 16) / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, NA + NB, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, (TM * TN * 16) * 3 / 4, 0);
            }
            __syncthreads();
            store_step();
            __syncthreads();
        }
    }

    // ---- epilogue: acc[tm][tn][reg] -> C[m][n], row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
    if constexpr (BF3) { if (p.dbg & 4) return; }
    if constexpr (SPLIT == 2 && MODE == MODE_FWD) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= (1.0f / 4096.0f);
    }
    const int r = lane & 31, h = lane >> 5;
    size_t ldc;
    float *cbase;
    const bool partial = (MODE != MODE_BWD_FILTER) && (p.ksplit > 1 || tailchunk);     // raw partial sums go to a slab
    if (MODE == MODE_FWD) { ldc = p.Cout; cbase = partial ? p.slab + (size_t)split * p.M * ldc : p.c; }
    else if (MODE == MODE_BWD_DATA) { ldc = p.Cin; cbase = partial ? p.slab + (size_t)split * p.M * ldc : p.c; }
    else {
        ldc = (size_t)(p.wbatch_rows ? p.wbatch_n : taps) * (SMALLC ? 4 : p.Cin);
        cbase = p.c + (size_t)split * p.Cout * ldc + (SMALLC ? 0 : (size_t)tap * p.Cin);
    }
    if (MODE != MODE_BWD_FILTER && tailchunk) {      // tile-local (BM_ x BN_) slab of chunk `split` of tail tile `tail_idx`
        ldc = BN_;
        cbase = p.slab + ((size_t)tail_idx * p.tail_ks + split) * (BM_ * BN_) - ((size_t)m0 * BN_ + n0);
    }
    // Each 32 x 32 MFMA tile is transposed through the wave's private LDS tile (the K loop's last barrier has passed,
    // so the staging buffers are free) and leaves as float4 stores, 8 rows x 128 B per instruction; the accumulator
    // layout itself would give 4-B stores, 16 instructions per tile, which bounds the low-K (1x1, 64..256 channel) layers.
    float *const et = smem + wave * (32 * EPI_LD);
    const int er = lane >> 3, ec = (lane & 7) * 4;         // read side: rows er + 8j, columns ec..ec+3
    const bool plain = !partial;
    const bool bn = MODE == MODE_FWD && plain && p.bn_part != nullptr;
    float4 bsum[TN], bsq[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { bsum[tn] = make_float4(0.f, 0.f, 0.f, 0.f); bsq[tn] = make_float4(0.f, 0.f, 0.f, 0.f); }
    // The bias of the wave's TN column groups is read ONCE here and waited for explicitly.  Read per tile inside the loops
    // below, the waitcnt pass could not prove across the rows' exec-masked blocks that the load had completed and put
    // s_waitcnt vmcnt(0) in front of EVERY row - and on gfx9 stores count in vmcnt too, so every row's store waited for the
    // previous row's store to be acknowledged: 16 serialised store round trips per wave and tile.
    float4 bvs[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * (BN_ / 2) + tn * 32 + ec;
        bvs[tn] = (MODE == MODE_FWD && plain && p.bias) ? ldg4(p.bias + (n < p.Ng ? n : 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (MODE == MODE_FWD) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) only (expcnt / lgkmcnt fields = no wait)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
            for (int e = 0; e < 16; ++e) et[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_LD + r] = acc[tm][tn][e];
            const int n = n0 + wn * (BN_ / 2) + tn * 32 + ec;
            const bool nv = n < p.Ng;
            const float4 bv = bvs[tn];
            // BWD_DATA: the old value (accumulate) and the ReLU mask operand of the tile's four rows are read up front under
            // ONE block-uniform branch each, from clamped (always valid) addresses - inside the per-row `if` the two loads of
            // every row were followed by s_waitcnt vmcnt(0): 32 serialised memory latencies per 64 x 64 wave tile
            float4 old4[4], xm4[4];
            if (MODE == MODE_BWD_DATA && plain && (p.accumulate || p.relu_x)) {
                size_t off4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * (BM_ / 2) + tm * 32 + er + 8 * j;
                    off4[j] = (size_t)min(m, p.M - 1) * ldc + (nv ? n : 0);
                }
                if (p.accumulate) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) old4[j] = ldg4(cbase + off4[j]);
                }
                if (p.relu_x) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xm4[j] = ldg4(p.relu_x + off4[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = er + 8 * j;
                const int m = m0 + wm * (BM_ / 2) + tm * 32 + row;
                float4 v = *reinterpret_cast<const float4 *>(&et[row * EPI_LD + ec]);
                if (!nv || m >= p.M) continue;
                float *dst = cbase + (size_t)m * ldc + n;
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (MODE == MODE_FWD && bn) {
                    bsum[tn].x += v.x; bsum[tn].y += v.y; bsum[tn].z += v.z; bsum[tn].w += v.w;
                    bsq[tn].x = fmaf(v.x, v.x, bsq[tn].x); bsq[tn].y = fmaf(v.y, v.y, bsq[tn].y);
                    bsq[tn].z = fmaf(v.z, v.z, bsq[tn].z); bsq[tn].w = fmaf(v.w, v.w, bsq[tn].w);
                }
                if (MODE == MODE_FWD && p.relu && plain) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (MODE == MODE_BWD_DATA && p.accumulate && plain) {
                    const float4 o = old4[j];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                if (MODE == MODE_BWD_DATA && p.relu_x && plain) {
                    const float4 xm = xm4[j];
                    v.x = xm.x > 0.f ? v.x : 0.f; v.y = xm.y > 0.f ? v.y : 0.f;
                    v.z = xm.z > 0.f ? v.z : 0.f; v.w = xm.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(dst) = v;
            }
        }
    if (MODE == MODE_FWD && bn) {
        // the 8 lanes with equal (lane & 7) hold the same 4 channels for different rows: butterfly over lane bits 3..5 (fixed
        // order), lanes 0..7 then write the wave's row of partial statistics
        const int prow = (m0 / BM_) * 2 + wm;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
            for (int o = 8; o <= 32; o <<= 1) {
                bsum[tn].x += __shfl_xor(bsum[tn].x, o, 64); bsum[tn].y += __shfl_xor(bsum[tn].y, o, 64);
                bsum[tn].z += __shfl_xor(bsum[tn].z, o, 64); bsum[tn].w += __shfl_xor(bsum[tn].w, o, 64);
                bsq[tn].x += __shfl_xor(bsq[tn].x, o, 64); bsq[tn].y += __shfl_xor(bsq[tn].y, o, 64);
                bsq[tn].z += __shfl_xor(bsq[tn].z, o, 64); bsq[tn].w += __shfl_xor(bsq[tn].w, o, 64);
            }
            const int n = n0 + wn * (BN_ / 2) + tn * 32 + ec;
            if (er == 0 && n < p.Ng) {
                *reinterpret_cast<float4 *>(p.bn_part + ((size_t)prow * 2) * p.Ng + n) = bsum[tn];
                *reinterpret_cast<float4 *>(p.bn_part + ((size_t)prow * 2 + 1) * p.Ng + n) = bsq[tn];
            }
        }
    }
}

int g_cus();
#include "planes_gemm.h"

// ---- launch planning -------------------------------------------------------------------------------------
// Workgroup slots of the chip for one kernel instantiation = resident workgroups per CU (runtime occupancy query,
// cached) x CUs.  Tiles and split-K factors are chosen so that the grid is a whole number of slot "rounds": a grid of
// 1.02 rounds costs 2.
int g_cus() {
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}
template <int MODE, int BM_, int BN_, bool SMALLC>
int slots_of() {
    static int slots = 0;
    if (!slots) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(&k_conv_igemm<MODE, BM_, BN_, SMALLC>),
                                                         CONV_THREADS, 0) != hipSuccess || occ <= 0)
            occ = 2;
        (void)hipGetLastError();
        slots = occ * g_cus();
    }
    return slots;
}
template <int MODE>
int slots_for(int bm, int bn) {
    if (bm == 128 && bn == 128) return slots_of<MODE, 128, 128, false>();
    if (bm == 128 && bn == 64) return slots_of<MODE, 128, 64, false>();
    if (bm == 64 && bn == 128) return slots_of<MODE, 64, 128, false>();
    return slots_of<MODE, 64, 64, false>();
}

struct TileChoice { int bm, bn; };
int g_plan_fill = 2;          // workgroups per CU a tile choice / split-K plan aims for (measurement knob: mrcnn_debug_conv_plan)
int g_plan_filter_rounds = 2;     // in HALF rounds of workgroup slots
int g_dbg_parts = 0;
int g_plan_force_tile = 0;    // measurement: the forward / backward-data tile choice is 1 = 128x64, 2 = 64x64 (applied BEFORE the split plans)
// The largest tile that still gives every CU at least two workgroups (measured on gfx950: 128x128 ~118 TF, 128x64 ~103,
// 64x64 ~80 on large layers, so a smaller tile only pays when the big one cannot fill the chip); narrow GEMM sides
// (<= 64) take the 64-wide tile.
template <int MODE>
TileChoice choose_tile(long long M, long long Ng, long long z) {
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Ng + bn - 1) / bn) * z; };
    const long long fill = (long long)g_plan_fill * g_cus();
    const bool narrow_n = Ng <= 64, narrow_m = M <= 64;
    if (g_plan_force_tile == 2) return {64, 64};
    if (g_plan_force_tile == 1 && !narrow_m) return {128, 64};
    if (!narrow_n && !narrow_m && tiles(128, 128) >= fill) return {128, 128};
    if (!narrow_m && tiles(128, 64) >= fill) return {128, 64};
    if (!narrow_n && !narrow_m && tiles(128, 128) >= fill * 3 / 4) return {128, 128};
    if (narrow_m && !narrow_n) return {64, 128};
    if (!narrow_m && narrow_n) return {128, 64};
    return {64, 64};
}

// Measurement knob (mrcnn_conv2d_set_debug_skip; bench.py's roofline split): bit 0 = the MFMA GEMM launches are skipped,
// bit 1 = every other kernel of the convolution calls (Winograd transforms, slab / tail / column sums) is skipped.
// Results are garbage while a bit is set; kernel durations do not depend on the data.
int g_debug_skip = 0;
// EXPLORATORY: forward-kind GEMM launches with three-term split-bf16 operands (mrcnn_conv2d_set_split_bf16).  Off by default;
// results then differ from the float32 kernels by ~1e-5 relative per product.
int g_split_mode[3] = {0, 0, 0};        // per pass (forward, backward-data, backward-filter): 0 float32, 1 bf16 planes, 2 half planes
thread_local int g_cur_pass = 0;        // set by the entry points around their launches (PASS_*)

template <int MODE>
void launch_conv(ConvP &p, int zdim, TileChoice t, hipStream_t st) {
    // zdim: BWD_FILTER taps * ksplit; FWD / BWD_DATA ksplit
    if (p.smallc) t = (MODE == MODE_FWD) ? TileChoice{128, 64} : TileChoice{64, 128};   // image layer: fixed tiles
    p.tiles_m = mrcnn::cdiv(p.M, t.bm);
    p.tiles_n = mrcnn::cdiv(p.Ng, t.bn);
    const int tiles = p.tiles_m * p.tiles_n;
    int total = tiles * zdim;
    p.remap_n = total;
    if (MODE != MODE_BWD_FILTER && p.tail_ks) {
        p.remap_n = p.tail_full;
        total = p.tail_full + (tiles - p.tail_full) * p.tail_ks;
    }
    const dim3 grid(total), blk(CONV_THREADS);
    p.dbg = g_dbg_parts;
    if (g_debug_skip & 1) return;
    if (p.smallc) {
        if (MODE == MODE_FWD) hipLaunchKernelGGL((k_conv_igemm<MODE_FWD, 128, 64, true>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((k_conv_igemm<MODE_BWD_FILTER, 64, 128, true>), grid, blk, 0, st, p);
        return;
    }
    {   // exploratory opt-in: three-term split operands; the planes' type per PASS of the calling entry point
        const int split = g_split_mode[g_cur_pass];
        // (Sending the emulated launches with 64 x 64 / 128 x 64 / 64 x 128 tiles - MFMA-busy 0.12 - 0.24 - to the float32-MFMA kernel was
        // measured in round 5, same-process A/B on the step: 21.72 ms emulated everywhere against 21.84 - 21.91 ms with any subset on float32.)
        if (split == 1) {
            if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 128, false, 1>), grid, blk, 0, st, p);
            else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 64, false, 1>), grid, blk, 0, st, p);
            else if (t.bm == 64 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 128, false, 1>), grid, blk, 0, st, p);
            else hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 64, false, 1>), grid, blk, 0, st, p);
            return;
        }
        if (split == 3) {
            if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 128, false, 3>), grid, blk, 0, st, p);
            else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 64, false, 3>), grid, blk, 0, st, p);
            else if (t.bm == 64 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 128, false, 3>), grid, blk, 0, st, p);
            else hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 64, false, 3>), grid, blk, 0, st, p);
            return;
        }
        if (split == 2) {
            if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 128, false, 2>), grid, blk, 0, st, p);
            else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 64, false, 2>), grid, blk, 0, st, p);
            else if (t.bm == 64 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 128, false, 2>), grid, blk, 0, st, p);
            else hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 64, false, 2>), grid, blk, 0, st, p);
            return;
        }
    }
    if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 128, false>), grid, blk, 0, st, p);
    else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((k_conv_igemm<MODE, 128, 64, false>), grid, blk, 0, st, p);
    else if (t.bm == 64 && t.bn == 128) hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 128, false>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((k_conv_igemm<MODE, 64, 64, false>), grid, blk, 0, st, p);
}

// Sum split-K slabs: out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i]  (deterministic order).
__global__ __launch_bounds__(256) void k_sum_slabs(const float *__restrict__ slabs, float *__restrict__ out,
                                                   size_t n4, int ksplit, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = accumulate ? ldg4(out + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= ksplit; k += 8) {            // 8 independent loads in flight, added in slab order
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ldg4s(slabs + ((size_t)(k + j) * n4 + i) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    for (; k < ksplit; ++k) {
        const float4 v = ldg4s(slabs + ((size_t)k * n4 + i) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + i * 4) = s;
}

// FWD / BWD_DATA split-K epilogue: out[m][n] = (accumulate ? out : 0) + sum_s slab[s][m][n] + bias[n], optional ReLU.
__global__ __launch_bounds__(256) void k_sum_slabs_ep(const float *__restrict__ slabs, float *__restrict__ out, size_t n4,
                                                      int ksplit, int ldc4, const float *__restrict__ bias, int relu,
                                                      int accumulate, const float *__restrict__ relu_x) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = accumulate ? ldg4(out + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < ksplit; ++k) {
        const float4 v = ldg4(slabs + ((size_t)k * n4 + i) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
        const float4 b = ldg4(bias + (i % ldc4) * 4);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    if (relu_x) {
        const float4 xm = ldg4(relu_x + i * 4);
        s.x = xm.x > 0.f ? s.x : 0.f; s.y = xm.y > 0.f ? s.y : 0.f; s.z = xm.z > 0.f ? s.z : 0.f; s.w = xm.w > 0.f ? s.w : 0.f;
    }
    *reinterpret_cast<float4 *>(out + i * 4) = s;
}

// Tail-split epilogue: tile `tail_full + blockIdx.y` = sum of its tail_ks tile-local slabs (fixed order) + bias, ReLU,
// accumulate; one thread per float4 of the bm x bn tile.
__global__ __launch_bounds__(256) void k_tail_sum(const float *__restrict__ slab, float *__restrict__ c,
                                                  const float *__restrict__ bias, int relu, int accumulate, int M, int Ng,
                                                  int ldc, int tiles_n, int tail_full, int ks, int bm, int bn,
                                                  const float *__restrict__ relu_x) {
    const int e4 = blockIdx.x * 256 + threadIdx.x;
    if (e4 * 4 >= bm * bn) return;
    const int tile = tail_full + blockIdx.y;
    const int bx = tile / tiles_n, by = tile - bx * tiles_n;
    const int ml = (e4 * 4) / bn, nl = (e4 * 4) - ml * bn;
    const int m = bx * bm + ml, n = by * bn + nl;
    if (m >= M || n >= Ng) return;
    float *dst = c + (size_t)m * ldc + n;
    float4 s = accumulate ? ldg4(dst) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float *src = slab + (size_t)blockIdx.y * ks * (bm * bn) + (size_t)e4 * 4;
    for (int k = 0; k < ks; ++k) {
        const float4 v = ldg4(src + (size_t)k * (bm * bn));
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
        const float4 b = ldg4(bias + n);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    if (relu_x) {
        const float4 xm = ldg4(relu_x + (size_t)m * ldc + n);
        s.x = xm.x > 0.f ? s.x : 0.f; s.y = xm.y > 0.f ? s.y : 0.f; s.z = xm.z > 0.f ? s.z : 0.f; s.w = xm.w > 0.f ? s.w : 0.f;
    }
    *reinterpret_cast<float4 *>(dst) = s;
}

// Column sums of a (P, C) matrix (bias gradient): out[c] = sum_p g[p][c]; two-stage, deterministic.
// Stage 1: thread = (channel group of 4, row lane), float4 streaming loads, a block owns a contiguous chunk of rows.
// Stage 2: 64 channels x 16 slices per 1024-thread block, slices added in order.
struct ColPlan { int G, RPI, nblk, rows_per_blk; };
ColPlan col_plan(int P, int C) {
    ColPlan r;
    const int C4 = C / 4;
    r.G = std::min(C4, 256);
    r.RPI = 256 / r.G;
    long long want = ((long long)P * C4 + 256ll * 16 - 1) / (256ll * 16);
    r.nblk = (int)std::max(1ll, std::min(want, 1024ll));
    r.rows_per_blk = (int)(((long long)P + r.nblk - 1) / r.nblk);
    r.rows_per_blk = (r.rows_per_blk + r.RPI - 1) / r.RPI * r.RPI;
    r.nblk = (P + r.rows_per_blk - 1) / r.rows_per_blk;
    return r;
}
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ g, float *__restrict__ part, int P, int C,
                                                        int G, int RPI, int rows_per_blk) {
    __shared__ float4 s1[256];
    const int t = threadIdx.x, cg0 = t % G, rr = t / G;
    const int C4 = C / 4;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(P, r0 + rows_per_blk);
    for (int cg = cg0; cg < C4; cg += G) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int r = r0 + rr;
        for (; r + 3 * RPI < r1; r += 4 * RPI) {         // four rows in flight, added in row order
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldg4(g + (size_t)(r + j * RPI) * C + cg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
        }
        for (; r < r1; r += RPI) {
            const float4 v = ldg4(g + (size_t)r * C + cg * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        s1[t] = a;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < RPI; ++k) {
                const float4 q = s1[k * G + cg0];
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            }
            *reinterpret_cast<float4 *>(part + (size_t)blockIdx.x * C + cg * 4) = a;
        }
        __syncthreads();
    }
}
// Final stage: 256-thread workgroups (4 channels x 64 slices; slice s adds partials s, s + 64, ... in order, four loads in
// flight, then the 64 slice sums are added in slice order).  Small workgroups on purpose: this kernel runs on the
// weight-gradient stream beside the data-gradient GEMMs, whose workgroups hold 3 of the 4 wave slots of every SIMD - a
// 1024-thread workgroup needs a whole CU to drain before it can start (measured: 20-30 us per launch for a 1-MB reduction).
constexpr int CSF_CH = 4, CSF_SL = 64;
__global__ __launch_bounds__(CSF_CH * CSF_SL) void k_colsum_final(const float *__restrict__ part, float *__restrict__ out, int nb, int C,
                                                                 int accumulate) {
    __shared__ double sa[CSF_SL][CSF_CH];
    const int cl = threadIdx.x % CSF_CH, c = blockIdx.x * CSF_CH + cl, slice = threadIdx.x / CSF_CH;
    double a = 0.0;
    if (c < C) {
        int k = slice;
        for (; k + 3 * CSF_SL < nb; k += 4 * CSF_SL) {
            const float v0 = part[(size_t)k * C + c], v1 = part[(size_t)(k + CSF_SL) * C + c];
            const float v2 = part[(size_t)(k + 2 * CSF_SL) * C + c], v3 = part[(size_t)(k + 3 * CSF_SL) * C + c];
            a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
        }
        for (; k < nb; k += CSF_SL) a += (double)part[(size_t)k * C + c];
    }
    sa[slice][cl] = a;
    __syncthreads();
    if (slice == 0 && c < C) {
        for (int k = 1; k < CSF_SL; ++k) a += sa[k][cl];
        out[c] = (float)a + (accumulate ? out[c] : 0.0f);
    }
}

int conv_out(int in, int k, int s, int pad) { return (in + 2 * pad - k) / s + 1; }

int check_conv(const void *a, const void *b, const void *c, int N, int H, int W, int Cin, int Cout, int KH,
               int KW, int stride, int pad) {
    if (!a || !b || !c) return mrcnn::fail_arg(MRCNN_E_INVALID, "conv2d: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "conv2d: bad sizes");
    if ((Cin % BK && Cin != 4) || Cout % BK)
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED,
                               "conv2d: Cin (%d) must be 4 or a multiple of %d... Cout (%d) a multiple of %d (the host layer pads)", Cin, BK, Cout, BK);
    if (conv_out(H, KH, stride, pad) <= 0 || conv_out(W, KW, stride, pad) <= 0)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "conv2d: empty output");
    const long long lim = (1ll << 30) - 1;      // element offsets are 32-bit and byte sizes fit a buffer descriptor
    const long long Ho_ = conv_out(H, KH, stride, pad), Wo_ = conv_out(W, KW, stride, pad);
    if ((long long)N * H * W * Cin > lim || (long long)N * Ho_ * Wo_ * Cout > lim || (long long)Cout * KH * KW * Cin > lim ||
        (long long)N * std::max<long long>(H * W, Ho_ * Wo_) >= (1ll << 24))
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "conv2d: tensor too large for 32-bit offsets / 24-bit pixel indices");
    return 0;
}

ConvP make_p(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    ConvP p{};
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pad_w = pad;
    p.Ho = conv_out(H, KH, stride, pad);
    p.Wo = conv_out(W, KW, stride, pad);
    p.ksplit = 1;
    p.smallc = (Cin == 4);
    p.bn_part = nullptr;
    return p;
}

// split-K plan for backward-filter: enough workgroups to fill 256 CUs a few times over.
TileChoice filter_tile(const ConvP &p) {
    TileChoice t;
    if (p.smallc) return {64, 128};
    t.bm = p.Cout <= 64 ? 64 : 128;
    t.bn = (p.smallc ? p.KH * p.KW * 4 : p.Cin) <= 64 ? 64 : 128;
    return t;
}
// split-K plan for backward-filter: the number of splits that fills exactly one round of workgroup slots (slab traffic
// grows with the split count), with at least 8 K steps per split.
void filter_plan(const ConvP &p, int &ksplit, int &kchunk) {
    const long long P = (long long)p.N * p.Ho * p.Wo;
    const TileChoice t = filter_tile(p);
    const long long tiles = p.smallc ? (long long)mrcnn::cdiv(p.Cout, t.bm) * mrcnn::cdiv(p.KH * p.KW * 4, t.bn)
                                     : (long long)mrcnn::cdiv(p.Cout, t.bm) * mrcnn::cdiv(p.Cin, t.bn) * (p.wbatch_rows ? p.wbatch_n : p.KH * p.KW);
    const long long slots = p.smallc ? slots_of<MODE_BWD_FILTER, 64, 128, true>() : slots_for<MODE_BWD_FILTER>(t.bm, t.bn);
    const long long maxsplit = std::max(1ll, P / (8 * BK));
    long long want = slots * g_plan_filter_rounds / 2 / tiles;    // one full round
    if (want < 1) want = 1;
    ksplit = (int)std::max(1ll, std::min(std::min(want, maxsplit), 256ll));
    kchunk = (int)(((P + ksplit - 1) / ksplit + BK - 1) / BK * BK);
    ksplit = (int)((P + kchunk - 1) / kchunk);
}

// FWD / BWD_DATA split-K: when even the smallest useful tile leaves CUs idle (deep layers: few pixels, long K) the K
// steps are split over `ksplit` workgroups per tile; partial tiles go to slabs and are summed in fixed order.
template <int MODE>
void data_plan(ConvP &p, TileChoice &t, int nsteps) {
    t = choose_tile<MODE>(p.M, p.Ng, 1);
    const long long tiles = (long long)mrcnn::cdiv(p.M, t.bm) * mrcnn::cdiv(p.Ng, t.bn);
    const long long fill = (long long)g_plan_fill * g_cus();
    p.ksplit = 1;
    p.kchunk = nsteps;
    p.tail_ks = 0;
    if (!p.smallc && tiles >= fill) {
        // Tail split: a grid of r.x rounds of workgroup slots (x small) pays a whole extra round for its last few tiles,
        // which run on a nearly empty chip.  Those tiles are split along K into ~one round of short workgroups instead.
        const long long S = slots_for<MODE>(t.bm, t.bn), rem = tiles % S;
        if (tiles > S && rem > 0 && rem * 2 <= S && nsteps >= 8 && tiles * 16 < (1ll << 31)) {
            long long ks = std::min<long long>(std::min<long long>(S / rem, nsteps / 4), 16);
            if (ks >= 2) {
                p.tail_kchunk = (int)((nsteps + ks - 1) / ks);
                p.tail_ks = (nsteps + p.tail_kchunk - 1) / p.tail_kchunk;
                p.tail_full = (int)(tiles - rem);
            }
        }
        return;
    }
    if (p.smallc || tiles >= fill || nsteps < 16) return;
    long long ks = std::min<long long>(std::min<long long>(fill / tiles, nsteps / 8), 16);
    if (ks < 2) return;
    p.kchunk = (int)((nsteps + ks - 1) / ks);
    p.ksplit = (nsteps + p.kchunk - 1) / p.kchunk;
}
size_t data_ws_bytes(const ConvP &p, long long ldc) { return p.ksplit > 1 ? (size_t)p.ksplit * p.M * ldc * sizeof(float) : 0; }
size_t tail_ws_bytes(const ConvP &p, const TileChoice &t) {
    const long long tiles = (long long)mrcnn::cdiv(p.M, t.bm) * mrcnn::cdiv(p.Ng, t.bn);
    return p.tail_ks ? (size_t)(tiles - p.tail_full) * p.tail_ks * t.bm * t.bn * sizeof(float) : 0;
}

template <int MODE>
int run_data_conv(ConvP &p, int nsteps, long long ldc, void *ws, size_t ws_bytes, hipStream_t st) {
    TileChoice t;
    data_plan<MODE>(p, t, nsteps);
    if (p.ksplit > 1) {
        if (!ws || ws_bytes < data_ws_bytes(p, ldc)) { p.ksplit = 1; p.kchunk = nsteps; }      // no workspace: unsplit
        else p.slab = (float *)ws;
    }
    if (p.tail_ks) {
        if (!ws || ws_bytes < tail_ws_bytes(p, t)) p.tail_ks = 0;                              // no workspace: no tail split
        else p.slab = (float *)ws;
    }
    launch_conv<MODE>(p, p.ksplit, t, st);
    MRCNN_LAUNCH_CHECK();
    if (p.tail_ks) {
        const int tiles = mrcnn::cdiv(p.M, t.bm) * mrcnn::cdiv(p.Ng, t.bn);
        if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_tail_sum, dim3(mrcnn::cdiv(t.bm * t.bn / 4, 256), tiles - p.tail_full), dim3(256), 0, st, p.slab, p.c,
                           MODE == MODE_FWD ? p.bias : nullptr, MODE == MODE_FWD ? p.relu : 0,
                           MODE == MODE_BWD_DATA ? p.accumulate : 0, p.M, p.Ng, (int)ldc, p.tiles_n, p.tail_full, p.tail_ks,
                           t.bm, t.bn, MODE == MODE_BWD_DATA ? p.relu_x : nullptr);
        MRCNN_LAUNCH_CHECK();
    }
    if (p.ksplit > 1) {
        const size_t n4 = (size_t)p.M * ldc / 4;
        if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_sum_slabs_ep, dim3(mrcnn::cdiv(n4, 256)), dim3(256), 0, st, p.slab, p.c, n4, p.ksplit, (int)(ldc / 4),
                           MODE == MODE_FWD ? p.bias : nullptr, MODE == MODE_FWD ? p.relu : 0,
                           MODE == MODE_BWD_DATA ? p.accumulate : 0, MODE == MODE_BWD_DATA ? p.relu_x : nullptr);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Winograd F(m x m, 3x3), m = 2 or 4, for the 3x3 / stride 1 / pad 1 convolutions with >= 256 channels (mask head, FPN
// and RPN 3x3, res4 / res5 conv2): 2.25x (m = 2) or 4x (m = 4) fewer multiplications than the direct form; 77 % of the
// step's MACs are such layers.  a = m + 2, nk = a*a (16 or 36), T = N * ceil(H/m) * ceil(W/m) tiles.
//   U[k] = G g G^T   (one Cout x Cin matrix per k)                 k_wino_filter (every call: the weights change)
//   V[k] = B^T d B   (one T x Cin matrix per k)                    k_wino_input
//   M[k] = V[k] U[k]^T                                             ONE launch of the 1x1 forward kernel over nk*Tp rows
//   Y    = A^T M A + bias, ReLU / accumulate / ReLU mask           k_wino_output
// Backward-data = the same pipeline on gy with the rotated, channel-transposed filter.  Backward-filter:
//   dU[k] = sum_t (A dY A^T)[k][t]^T (B^T d B)[k][t]               k_wino_gy, k_wino_input, ONE batched filter-gradient launch
//   dg    = G^T dU G                                               k_wino_filter_grad
// Non-fused: V, M cross HBM once each (4x / 2.25x the activation size).  m is chosen per layer to minimise nk * T (m = 4
// unless the map is tiny or padding waste dominates).  Matrices: Lavin & Gray, points (0, +-1, +-2, inf) for m = 4.
// Sums are in fixed order (bit-reproducible); rounding differs from the direct kernel (m = 2: a few ulp, m = 4: ~1e-5).
// ---------------------------------------------------------------------------------------------------------------
constexpr int WINO_ROWS = 128;          // Tp = T rounded up to this: a GEMM tile never straddles two k

int g_wino_min_channels = 256, g_wino_min_pixels = 2048, g_wino_tile = 0;      // tile 0 = automatic, 2 or 4 = forced
int g_wino_banded = 16;       // input transforms: 0 = launch order, 1 = XCD-banded raster order, >= 2 = XCD-banded column panels of that many tiles (mrcnn_debug_wino_banded, for A/B)
// Per pass (PASS_FWD / PASS_BWD_DATA / PASS_BWD_FILTER) override of the tile: 0 = follow g_wino_tile, 2 / 4 = forced,
// -1 = the pass never takes the Winograd path.  F(4x4,3x3) amplifies float32 rounding by ~|A|^2 |B|^2 |G|^2: harmless on
// activations (measured 1.5e-4 of the tensor scale on the full network) but visible in gradients of layers whose own
// float32 noise floor is low (no BatchNorm behind them) - mrcnn_conv2d_set_winograd_pass_tiles.
// Default {2, 0, 0}: the FORWARD pass stays on F(2x2,3x3) (or the direct kernel below its threshold).  Measured on the full
// network (profiles/r02_winograd_pass_probe.txt): F(4x4) in forward keeps activations within 2.2e-4 of their scale, but the
// losses' curvature (smooth-L1 with sigma 3, softmax) turns those 1e-4 activation errors into parameter-gradient errors
// of up to 2e-2 in the layers without BatchNorm (rpn/conv, FPN laterals) - 100x their float32 noise floor - whereas F(4x4)
// in the two BACKWARD passes leaves every gradient at the floor.  {0, 0, 0} = F(4x4) wherever cheaper, +8 % images/s.
enum { PASS_FWD = 0, PASS_BWD_DATA = 1, PASS_BWD_FILTER = 2 };
int g_wino_pass_tile[3] = {2, 0, 0};

int wino_m(int H, int W, int pass) {
    const int pt = g_wino_pass_tile[pass];
    if (pt == 2 || pt == 4) return pt;
    if (g_wino_tile == 2 || g_wino_tile == 4) return g_wino_tile;
    const long long c2 = 16ll * ((H + 1) / 2) * ((W + 1) / 2), c4 = 36ll * ((H + 3) / 4) * ((W + 3) / 4);
    return c4 < c2 ? 4 : 2;
}
bool wino_ok(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int pass) {
    if (g_wino_pass_tile[pass] < 0) return false;
    if (KH != 3 || KW != 3 || stride != 1 || pad != 1) return false;
    // measured on gfx950 (tools/conv_bench.py): with >= 256 channels the GEMMs are deep enough (K >= 256) to win from
    // 2048 pixels up; at 128 channels Winograd ties the direct kernel, at 64 it loses; below 2048 pixels the four launches
    // are latency-bound
    // F(4x4) does 4x fewer multiplications, so it pays from a quarter of the channels F(2x2) needs (measured: 64 -> 64 on
    // 256^2 maps 95 -> 78 us, 128 -> 128 on 128^2 83 -> 62 us; with F(2x2) the same layers lose)
    const int minc = wino_m(H, W, pass) == 4 ? std::max(BK, g_wino_min_channels / 4) : g_wino_min_channels;
    if (Cin % BK || Cout % BK || Cin < minc || Cout < minc) return false;
    if ((long long)N * H * W < g_wino_min_pixels) return false;
    const long long T = (long long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const long long Tp = (T + WINO_ROWS - 1) / WINO_ROWS * WINO_ROWS;
    return 16 * Tp < (1ll << 24) && 16 * Tp * std::max(Cin, Cout) < (1ll << 30);    // limits of the GEMM kernel's offsets
}
struct WinoGeom { int m, a, nk, th, tw; long long T, Tp; };
WinoGeom wino_geom(int N, int H, int W, int pass) {
    WinoGeom g;
    g.m = wino_m(H, W, pass); g.a = g.m + 2; g.nk = g.a * g.a;
    g.th = (H + g.m - 1) / g.m; g.tw = (W + g.m - 1) / g.m;
    g.T = (long long)N * g.th * g.tw;
    g.Tp = (g.T + WINO_ROWS - 1) / WINO_ROWS * WINO_ROWS;
    return g;
}
struct WinoLayout { size_t u, v, m, inner, total; WinoGeom g; };
WinoLayout wino_layout(int N, int H, int W, int Cin, int Cout, int pass) {
    WinoLayout L;
    L.g = wino_geom(N, H, W, pass);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t o = 0;
    L.u = o; o += al((size_t)L.g.nk * Cout * Cin * 6);          // float32, or three bf16 planes (plane GEMMs)
    L.v = o; o += al((size_t)L.g.nk * L.g.Tp * Cin * 4);
    L.m = o; o += al((size_t)L.g.nk * L.g.Tp * Cout * 4);
    L.inner = o; o += (size_t)4 * g_cus() * 128 * 128 * sizeof(float);              // tail-split slabs of the GEMM launch
    L.total = o;
    return L;
}
size_t wino_ws_bytes(int N, int H, int W, int Cin, int Cout, int pass) { return wino_layout(N, H, W, Cin, Cout, pass).total; }

// ---- the 1-D transforms (T = float or V4) --------------------------------------------------------------------------
struct V4 { float x, y, z, w; };
__device__ __forceinline__ V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__device__ __forceinline__ V4 operator*(float s, V4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
__device__ __forceinline__ V4 v4ld(const float *p) { const float4 f = ldg4(p); return {f.x, f.y, f.z, f.w}; }
__device__ __forceinline__ void v4st(float *p, V4 v) { *reinterpret_cast<float4 *>(p) = make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ V4 v4zero() { return {0.f, 0.f, 0.f, 0.f}; }

template <int M_, typename T> __device__ __forceinline__ void wino_bt(const T (&d)[M_ + 2], T (&r)[M_ + 2]) {      // B^T d
    if (M_ == 2) {
        r[0] = d[0] - d[2]; r[1] = d[1] + d[2]; r[2] = d[2] - d[1]; r[3] = d[1] - d[3];
    } else {
        r[0] = (4.0f * d[0] - 5.0f * d[2]) + d[4];
        r[1] = (d[4] + d[3]) - 4.0f * (d[1] + d[2]);
        r[2] = (d[4] - d[3]) + 4.0f * (d[1] - d[2]);
        r[3] = (d[4] - d[2]) + 2.0f * (d[3] - d[1]);
        r[4] = (d[4] - d[2]) - 2.0f * (d[3] - d[1]);
        r[5] = (4.0f * d[1] - 5.0f * d[3]) + d[5];
    }
}
template <int M_, typename T> __device__ __forceinline__ void wino_at(const T (&m)[M_ + 2], T (&s)[M_]) {           // A^T m
    if (M_ == 2) {
        s[0] = (m[0] + m[1]) + m[2]; s[1] = (m[1] - m[2]) - m[3];
    } else {
        const T p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
        s[0] = (m[0] + p12) + p34;
        s[1] = m12 + 2.0f * m34;
        s[2] = p12 + 4.0f * p34;
        s[3] = (m12 + 8.0f * m34) + m[5];
    }
}
template <int M_, typename T> __device__ __forceinline__ void wino_a(const T (&y)[M_], T (&t)[M_ + 2], T zero) {    // A y
    if (M_ == 2) {
        t[0] = y[0]; t[1] = y[0] + y[1]; t[2] = y[0] - y[1]; t[3] = zero - y[1];
    } else {
        const T e = y[0] + y[2], o = y[1] + y[3], e4 = y[0] + 4.0f * y[2], o8 = 2.0f * y[1] + 8.0f * y[3];
        t[0] = y[0]; t[1] = e + o; t[2] = e - o; t[3] = e4 + o8; t[4] = e4 - o8; t[5] = y[3];
    }
}
template <int M_> __device__ __forceinline__ void wino_g(const float (&g)[3], float (&t)[M_ + 2]) {                   // G g
    if (M_ == 2) {
        t[0] = g[0]; t[1] = 0.5f * ((g[0] + g[1]) + g[2]); t[2] = 0.5f * ((g[0] - g[1]) + g[2]); t[3] = g[2];
    } else {
        const float e = g[0] + g[2];
        t[0] = 0.25f * g[0];
        t[1] = (-1.0f / 6.0f) * (e + g[1]);
        t[2] = (-1.0f / 6.0f) * (e - g[1]);
        t[3] = (g[0] * (1.0f / 24.0f) + g[2] * (1.0f / 6.0f)) + g[1] * (1.0f / 12.0f);
        t[4] = (g[0] * (1.0f / 24.0f) + g[2] * (1.0f / 6.0f)) - g[1] * (1.0f / 12.0f);
        t[5] = g[2];
    }
}
template <int M_> __device__ __forceinline__ void wino_gt(const float (&D)[M_ + 2], float (&r)[3]) {                  // G^T D
    if (M_ == 2) {
        r[0] = D[0] + 0.5f * (D[1] + D[2]); r[1] = 0.5f * (D[1] - D[2]); r[2] = 0.5f * (D[1] + D[2]) + D[3];
    } else {
        const float p12 = D[1] + D[2], m12 = D[1] - D[2], p34 = D[3] + D[4], m34 = D[3] - D[4];
        r[0] = (0.25f * D[0] - p12 * (1.0f / 6.0f)) + p34 * (1.0f / 24.0f);
        r[1] = m34 * (1.0f / 12.0f) - m12 * (1.0f / 6.0f);
        r[2] = (p34 * (1.0f / 6.0f) - p12 * (1.0f / 6.0f)) + D[5];
    }
}

// U[k][co][ci] from w (Cout,3,3,Cin).  transposed: the backward-data filter w'[ci][u][v][co] = w[co][2-u][2-v][ci],
// written as U[k][ci][co] (the GEMM's "Cout" axis is then Cin).
template <int M_>
__device__ __forceinline__ void wino_filter_body(const float *__restrict__ w, float *__restrict__ U, int Cout, int Cin,
                                                 int transposed, unsigned blk, int uplanes = 0) {
    constexpr int A_ = M_ + 2;
    const int i = (int)(blk * 256u + threadIdx.x);
    if (i >= Cout * Cin) return;
    // the 36 (16) writes per thread dominate: threads run along the output's contiguous axis (ci, or co when transposed)
    const int ci = transposed ? i / Cout : i % Cin, co = transposed ? i % Cout : i / Cin;
    float t[A_][3];          // G g, column by column
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        float g[3], c[A_];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int uu = transposed ? 2 - u : u, vv = transposed ? 2 - v : v;
            g[u] = w[(((size_t)co * 3 + uu) * 3 + vv) * Cin + ci];
        }
        wino_g<M_>(g, c);
#pragma unroll
        for (int r = 0; r < A_; ++r) t[r][v] = c[r];
    }
    const size_t stride = (size_t)Cout * Cin;
    const size_t o = transposed ? (size_t)ci * Cout + co : (size_t)co * Cin + ci;
    if (uplanes) {
        // U as "P16R4" bf16 planes for the plane GEMMs (planes_gemm.h): matrix k has `rows` rows (the GEMM's N index) of `cols`
        // values (its K index); hi + mid + lo == the float32 value exactly
        const unsigned rows = transposed ? (unsigned)Cin : (unsigned)Cout, cols = transposed ? (unsigned)Cout : (unsigned)Cin;
        const unsigned rr = transposed ? (unsigned)ci : (unsigned)co, cc = transposed ? (unsigned)co : (unsigned)ci;
        unsigned short *Up = reinterpret_cast<unsigned short *>(U);
#pragma unroll
        for (int r = 0; r < A_; ++r) {
            float row[A_];
            wino_g<M_>(t[r], row);
#pragma unroll
            for (int q = 0; q < A_; ++q) {
                const unsigned off = pg_off_r4((unsigned)(r * A_ + q) * rows + rr, cc & ~3u, cols) / 2u + (cc & 3u);
                const float v = row[q];
                const __bf16 h = (__bf16)v;
                const float r1 = v - (float)h;
                const __bf16 m = (__bf16)r1;
                const __bf16 l = (__bf16)(r1 - (float)m);
                Up[off] = __builtin_bit_cast(unsigned short, h);
                Up[off + 64] = __builtin_bit_cast(unsigned short, m);
                Up[off + 128] = __builtin_bit_cast(unsigned short, l);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < A_; ++r) {
        float row[A_];
        wino_g<M_>(t[r], row);
#pragma unroll
        for (int q = 0; q < A_; ++q) U[(size_t)(r * A_ + q) * stride + o] = row[q];
    }
}
template <int M_>
__global__ __launch_bounds__(256) void k_wino_filter(const float *__restrict__ w, float *__restrict__ U, int Cout, int Cin,
                                                     int transposed, int uplanes) {
    wino_filter_body<M_>(w, U, Cout, Cin, transposed, blockIdx.x, uplanes);
}

// V[k][t][c] = (B^T d B)[k], d = the a x a input patch of tile t (rows m*ty-1 .., zero outside).  Thread = (t, 4 channels).
// All global traffic goes through buffer descriptors with 32-bit byte offsets (wino_ok() bounds x and V below 4 GiB): a tap
// outside the image is an out-of-range offset (the load returns 0), so the a*a loads are unconditional and issue back to back
// - with plain pointers every tap sat in its own exec-masked branch and the waitcnt pass serialised them into ~16 dependent
// rounds of memory latency per thread.
// HARDWARE NOTE (measured on gfx950, ROCm 7.2): the plane offset k * Tp * C * 4 of a STORE must NOT ride in the SGPR soffset.
// A buffer_store_dwordx4 needs one wait state before a VALU instruction overwrites its data registers; hipcc inserts that
// s_nop only when soffset is an immediate (the rule of the older ISA manuals: "stores with an SGPR offset need no wait
// state") - with an SGPR soffset the second data register of lanes 12-15 of every row of 16 lanes was stored AFTER the
// following v_pk_add_f32 had overwritten it: 7 of the 16 planes wrong in element y of those lanes, different from run to
// run.  Loads with an SGPR soffset are fine (k_wino_output), and so are stores whose data registers are not rewritten at once
// (roi_align.hip).  The plane offset is therefore added to the VGPR offset (one v_add per store).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ V4 v4buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ V4 v4buf_nt(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {       // read-once stream
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 2);
    return {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ void v4bufst(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, V4 v) {
    const u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff, soff, 0);
}
// (r6) The layer's input may be relu(BatchNorm(x)) of a training-mode BatchNorm whose statistics are final: the transform then applies
// gamma * ((x - mean) * invstd) + beta and the ReLU to every tap as it loads it - the expression of nn.hip's k_bn_apply, so the values
// are the bits that kernel would have written - and the normalised activation never crosses HBM (conv -> bn -> relu -> conv 3x3 of a
// ResNet bottleneck: extractor/feature_pyramid_network.py:48-66).  A tap outside the image stays the zero of the padding.
struct InBN { const float *gamma, *beta, *mean, *invstd; };
template <int M_>
__device__ __forceinline__ void wino_input_body(const float *__restrict__ x, float *__restrict__ V, int N, int H, int W, int C,
                                                int th, int tw, long long T, long long Tp, unsigned blk, const InBN bn = InBN{},
                                                int panel = 0) {
    constexpr int A_ = M_ + 2;
    // 32-bit index arithmetic (wino_ok(): 16 * Tp * C < 2^30): the 64-bit divisions this replaces were ~600 instructions of
    // branchy software division per thread
    const unsigned C4 = (unsigned)C / 4u;
    const unsigned i = blk * 256u + threadIdx.x;
    if (i >= (unsigned)Tp * C4) return;
    unsigned t = i / C4;
    const int c = (int)(i - t * C4) * 4;
    const unsigned ks = (unsigned)((size_t)Tp * C * 4);                 // bytes per k plane
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, (unsigned)((size_t)A_ * A_ * Tp * C * 4), 0x00020000);
    // Tile order inside an image (speed only: every tile is still transformed exactly once into its own rows of V).  Raster order puts
    // the tiles that share two of their six input rows a whole tile row apart; in column PANELS of `panel` tiles the next tile row of the
    // panel follows at once, and the shared rows are an L2 hit whatever else the other streams push through the cache meanwhile.
    if (panel > 1 && t < (unsigned)T) {
        const unsigned per = (unsigned)th * (unsigned)tw, pw = (unsigned)panel;
        const unsigned n0 = t / per, q = t - n0 * per;
        const unsigned full = (unsigned)tw / pw * pw, nf = (unsigned)th * full;
        unsigned ty0, tx0;
        if (q < nf) {
            const unsigned pp = q / ((unsigned)th * pw), r = q - pp * (unsigned)th * pw;
            ty0 = r / pw; tx0 = pp * pw + (r - ty0 * pw);
        } else {
            const unsigned wr = (unsigned)tw - full, q2 = q - nf;
            ty0 = q2 / wr; tx0 = full + (q2 - ty0 * wr);
        }
        t = n0 * per + ty0 * (unsigned)tw + tx0;
    }
    const unsigned vo = (unsigned)(((size_t)t * C + c) * 4);
    if (t >= (unsigned)T) {           // rows that pad T to a multiple of the GEMM tile: zero (the filter-gradient GEMM sums them)
#pragma unroll
        for (int k = 0; k < A_ * A_; ++k) v4bufst(rsV, vo + (unsigned)k * ks, 0, v4zero());
        return;
    }
    const unsigned trow = t / (unsigned)tw;
    const int tx = (int)(t - trow * (unsigned)tw);
    const int n = (int)(trow / (unsigned)th);
    const int ty = (int)(trow - (unsigned)n * (unsigned)th);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (unsigned)((size_t)N * H * W * C * 4), 0x00020000);
    const unsigned OOB = 0xFFFFFFFFu;
    unsigned rowo[A_], colo[A_];
#pragma unroll
    for (int r = 0; r < A_; ++r) {
        const int h = M_ * ty - 1 + r, ww = M_ * tx - 1 + r;
        rowo[r] = (unsigned)h < (unsigned)H ? (unsigned)(((size_t)n * H + h) * W * C * 4) : OOB;
        colo[r] = (unsigned)ww < (unsigned)W ? (unsigned)(((size_t)ww * C + c) * 4) : OOB;
    }
    V4 bg = v4zero(), bb = v4zero(), bm = v4zero(), bs = v4zero();
    if (bn.gamma) { bg = v4ld(bn.gamma + c); bb = v4ld(bn.beta + c); bm = v4ld(bn.mean + c); bs = v4ld(bn.invstd + c); }
    V4 b[A_][A_];           // B^T d, built column by column
#pragma unroll
    for (int q = 0; q < A_; ++q) {
        V4 d[A_], r[A_];
#pragma unroll
        for (int rr = 0; rr < A_; ++rr) d[rr] = v4buf(rsX, (rowo[rr] == OOB || colo[q] == OOB) ? OOB : rowo[rr] + colo[q], 0);
        if (bn.gamma) {         // (kernel-uniform)
#pragma unroll
            for (int rr = 0; rr < A_; ++rr) {
                const bool in = !(rowo[rr] == OOB || colo[q] == OOB);
                const V4 v = d[rr];
                V4 o;
                o.x = fmaxf(bg.x * ((v.x - bm.x) * bs.x) + bb.x, 0.f); o.y = fmaxf(bg.y * ((v.y - bm.y) * bs.y) + bb.y, 0.f);
                o.z = fmaxf(bg.z * ((v.z - bm.z) * bs.z) + bb.z, 0.f); o.w = fmaxf(bg.w * ((v.w - bm.w) * bs.w) + bb.w, 0.f);
                d[rr] = in ? o : v4zero();
            }
        }
        wino_bt<M_, V4>(d, r);
#pragma unroll
        for (int rr = 0; rr < A_; ++rr) b[rr][q] = r[rr];
    }
#pragma unroll
    for (int rr = 0; rr < A_; ++rr) {
        V4 row[A_];
        wino_bt<M_, V4>(b[rr], row);
#pragma unroll
        for (int q = 0; q < A_; ++q) v4bufst(rsV, vo + (unsigned)(rr * A_ + q) * ks, 0, row[q]);
    }
}
// Workgroup -> tile-range order of the input transforms.  Neighbouring tile rows share two of their six input rows; dealt in launch
// order, workgroups b and b + 1 sit on DIFFERENT XCDs (the hardware deals workgroup ids round robin over the 8 XCDs) and the shared rows
// are fetched from HBM once per XCD: 173 MB for the 103-MB input of a mask-head layer (rocprofv3 FETCH_SIZE).  Banded, every XCD works
// on a contiguous range of virtual blocks and the overlap is an L2 hit.  `g` = hardware workgroup id, `first` = id of the first
// workgroup of this family in the launch (the filter blocks of k_wino_input_filter come before), `nb` = blocks of the family.
__device__ __forceinline__ unsigned xcd_banded(unsigned g, unsigned first, unsigned nb, int on) {
    const unsigned b = g - first;
    if (!on) return b;
    const unsigned f0 = first & 7u, n2 = nb + f0, b2 = b + f0;         // pretend f0 phantom blocks in front: id % 8 is then the XCD
    const unsigned q = n2 >> 3, r = n2 & 7u, xcd = b2 & 7u;
    const unsigned v2 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b2 >> 3);
    return v2 - min(xcd + 1u, f0);                                       // the phantoms are the first block of bands 0 .. f0-1
}
template <int M_>
__global__ __launch_bounds__(256) void k_wino_input(const float *__restrict__ x, float *__restrict__ V, int N, int H, int W, int C,
                                                    int th, int tw, long long T, long long Tp, int banded, InBN bn) {
    wino_input_body<M_>(x, V, N, H, W, C, th, tw, T, Tp, xcd_banded(blockIdx.x, 0, gridDim.x, banded), bn, banded);
}
// Filter and input transform of one convolution call in ONE launch (the first `fblocks` workgroups transform the filter):
// the two are independent, and a separate 5-8 us filter launch in front of every Winograd GEMM is pure launch latency on the
// critical stream of the deep layers.
template <int M_>
__global__ __launch_bounds__(256) void k_wino_input_filter(const float *__restrict__ x, float *__restrict__ V, int N, int H, int W, int C,
                                                           int th, int tw, long long T, long long Tp, const float *__restrict__ w,
                                                           float *__restrict__ U, int Cout_w, int Cin_w, int transposed, unsigned fblocks,
                                                           int banded, int uplanes, InBN bn) {
    if (blockIdx.x < fblocks) wino_filter_body<M_>(w, U, Cout_w, Cin_w, transposed, blockIdx.x, uplanes);
    else wino_input_body<M_>(x, V, N, H, W, C, th, tw, T, Tp, xcd_banded(blockIdx.x, fblocks, gridDim.x - fblocks, banded), bn, banded);
}

// y (m x m pixels of tile t) = A^T M A + bias, then ReLU | + old y (accumulate) | zeroed where relu_x <= 0.
// bn_part (nullable; the layer feeds a training-mode BatchNorm): row blockIdx.x of bn_part (gridDim.x, 2, C) receives the
// per-channel sums and sums of squares of the pixels the block wrote (a block covers 256 / (C/4) whole tiles: the host
// passes bn_part only when C/4 divides 256) - the same partials the GEMM epilogue produces for the direct layers.
template <int M_>
__global__ __launch_bounds__(256) void k_wino_output(const float *__restrict__ Mb, float *__restrict__ y, int N, int H, int W, int C,
                                                     int th, int tw, long long T, long long Tp, const float *__restrict__ bias,
                                                     int relu, int accumulate, const float *__restrict__ relu_x,
                                                     float *__restrict__ bn_part) {
    constexpr int A_ = M_ + 2;
    __shared__ float4 sred[2][256];
    const unsigned C4 = (unsigned)C / 4u;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    V4 bs = v4zero(), bq = v4zero();
    if (i < (unsigned)T * C4) {
        const unsigned t = i / C4;                      // 32-bit index arithmetic, as in k_wino_input
        const int c = (int)(i - t * C4) * 4;
        const unsigned trow = t / (unsigned)tw;
        const int tx = (int)(t - trow * (unsigned)tw);
        const int n = (int)(trow / (unsigned)th);
        const int ty = (int)(trow - (unsigned)n * (unsigned)th);
        // buffer descriptors (see k_wino_input): the a*a plane reads share one VGPR offset, pixels outside the image are
        // out-of-range offsets - their stores are dropped, their optional reads return 0 - so nothing here is divergent
        const unsigned ks = (unsigned)((size_t)Tp * C * 4);
        const auto rsM = __builtin_amdgcn_make_buffer_rsrc((void *)Mb, 0, (unsigned)((size_t)A_ * A_ * Tp * C * 4), 0x00020000);
        const unsigned ybytes = (unsigned)((size_t)N * H * W * C * 4);
        const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, ybytes, 0x00020000);
        const auto rsR = __builtin_amdgcn_make_buffer_rsrc((void *)relu_x, 0, relu_x ? ybytes : 0u, 0x00020000);
        const unsigned mo = (unsigned)(((size_t)t * C + c) * 4);
        V4 s[M_][A_];           // A^T m, column by column
#pragma unroll
        for (int q = 0; q < A_; ++q) {
            V4 m[A_], r[M_];
#pragma unroll
            for (int rr = 0; rr < A_; ++rr) m[rr] = v4buf_nt(rsM, mo, (unsigned)(rr * A_ + q) * ks);
            wino_at<M_, V4>(m, r);
#pragma unroll
            for (int a = 0; a < M_; ++a) s[a][q] = r[a];
        }
        const V4 bv = bias ? v4ld(bias + c) : v4zero();
        const unsigned OOB = 0xFFFFFFFFu;
#pragma unroll
        for (int a = 0; a < M_; ++a) {
            V4 row[M_], old[M_], xm[M_];
            wino_at<M_, V4>(s[a], row);
            const int h = M_ * ty + a;
            unsigned off[M_];
#pragma unroll
            for (int b = 0; b < M_; ++b) {
                const int ww = M_ * tx + b;
                off[b] = (h < H && ww < W) ? (unsigned)((((size_t)n * H + h) * W + ww) * C + c) * 4u : OOB;
            }
            // the optional reads of a row of pixels go out together (one block-uniform branch per row, not per pixel)
            if (accumulate) {
#pragma unroll
                for (int b = 0; b < M_; ++b) old[b] = v4buf(rsY, off[b], 0);
            }
            if (relu_x) {
#pragma unroll
                for (int b = 0; b < M_; ++b) xm[b] = v4buf(rsR, off[b], 0);
            }
#pragma unroll
            for (int b = 0; b < M_; ++b) {
                V4 v = row[b] + bv;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (accumulate) v = v + old[b];
                if (relu_x) {
                    v.x = xm[b].x > 0.f ? v.x : 0.f; v.y = xm[b].y > 0.f ? v.y : 0.f; v.z = xm[b].z > 0.f ? v.z : 0.f; v.w = xm[b].w > 0.f ? v.w : 0.f;
                }
                v4bufst(rsY, off[b], 0, v);
                if (off[b] != OOB) {       // (statistics: only pixels of the image)
                    bs = bs + v;
                    bq.x = fmaf(v.x, v.x, bq.x); bq.y = fmaf(v.y, v.y, bq.y); bq.z = fmaf(v.z, v.z, bq.z); bq.w = fmaf(v.w, v.w, bq.w);
                }
            }
        }
    }
    if (bn_part) {              // block-uniform
        sred[0][threadIdx.x] = make_float4(bs.x, bs.y, bs.z, bs.w);
        sred[1][threadIdx.x] = make_float4(bq.x, bq.y, bq.z, bq.w);
        __syncthreads();
        if (threadIdx.x < C4) {        // the block's tiles in tile order (fixed => bit-reproducible)
            float4 a = sred[0][threadIdx.x], b2 = sred[1][threadIdx.x];
            for (unsigned k = threadIdx.x + C4; k < 256; k += C4) {
                const float4 q = sred[0][k], r = sred[1][k];
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
                b2.x += r.x; b2.y += r.y; b2.z += r.z; b2.w += r.w;
            }
            *reinterpret_cast<float4 *>(bn_part + ((size_t)blockIdx.x * 2) * C + threadIdx.x * 4) = a;
            *reinterpret_cast<float4 *>(bn_part + ((size_t)blockIdx.x * 2 + 1) * C + threadIdx.x * 4) = b2;
        }
    }
}

// W[k][t][c] = (A dy A^T)[k]: the m x m output-gradient tile t, zero outside the image.  Thread = (t, 4 channels).
// bias_part (nullable): the kernel reads every element of gy exactly once, so the bias gradient's column sums ride along -
// row blockIdx.x of bias_part (gridDim.x, C) receives the sums of the block's tiles (a block covers 256 / (C/4) whole tiles
// when C/4 divides 256: the host passes bias_part only then); k_colsum_final adds the rows up.
template <int M_>
__global__ __launch_bounds__(256) void k_wino_gy(const float *__restrict__ gy, float *__restrict__ Wt, int N, int H, int W, int C,
                                                 int th, int tw, long long T, long long Tp, float *__restrict__ bias_part, int wplanes) {
    constexpr int A_ = M_ + 2;
    __shared__ float4 sred[256];
    const unsigned C4 = (unsigned)C / 4u;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const bool in_range = i < (unsigned)Tp * C4;
    const unsigned t = in_range ? i / C4 : (unsigned)Tp;         // 32-bit index arithmetic, as in k_wino_input
    const int c = in_range ? (int)(i - t * C4) * 4 : 0;
    V4 bsum = v4zero();
    // buffer descriptors as in k_wino_input: unconditional loads (pixels outside the image = out-of-range offset = 0), plane
    // offsets of the stores in the VGPR offset (see the hardware note there)
    // wplanes: W as bf16 planes in the "PR" layout [k][t][3][C] (hi, mid, lo rows of C values: a wave writes whole 512-byte rows) for
    // the plane GEMM k_pgemm_gpp (planes_gemm.h) instead of float32 [k][t][C]
    const unsigned ks = wplanes ? (unsigned)((size_t)Tp * C * 6) : (unsigned)((size_t)Tp * C * 4);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void *)Wt, 0, (unsigned)((size_t)A_ * A_ * ks), 0x00020000);
    const unsigned vo = wplanes ? (unsigned)(((size_t)t * 3 * C + c) * 2) : (unsigned)(((size_t)t * C + c) * 4);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto put = [&](unsigned off, V4 v) {
        if (!wplanes) { v4bufst(rsW, off, 0, v); return; }
        uint2 hi, mid, lo;
        split3_bf16x4(make_float4(v.x, v.y, v.z, v.w), hi, mid, lo);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{hi.x, hi.y}, rsW, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{mid.x, mid.y}, rsW, off + (unsigned)C * 2u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo.x, lo.y}, rsW, off + (unsigned)C * 4u, 0, 0);
    };
    if (in_range && t >= (unsigned)T) {           // padded rows: zero
#pragma unroll
        for (int k = 0; k < A_ * A_; ++k) put(vo + (unsigned)k * ks, v4zero());
    } else if (in_range) {
        const unsigned trow = t / (unsigned)tw;
        const int tx = (int)(t - trow * (unsigned)tw);
        const int n = (int)(trow / (unsigned)th);
        const int ty = (int)(trow - (unsigned)n * (unsigned)th);
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void *)gy, 0, (unsigned)((size_t)N * H * W * C * 4), 0x00020000);
        V4 r[A_][M_];           // A y, column by column
#pragma unroll
        for (int b = 0; b < M_; ++b) {
            const int ww = M_ * tx + b;
            V4 y[M_], col[A_];
#pragma unroll
            for (int a = 0; a < M_; ++a) {
                const int h = M_ * ty + a;
                y[a] = v4buf(rsG, (h < H && ww < W) ? (unsigned)((((size_t)n * H + h) * W + ww) * C + c) * 4u : 0xFFFFFFFFu, 0);
                bsum = bsum + y[a];
            }
            wino_a<M_, V4>(y, col, v4zero());
#pragma unroll
            for (int q = 0; q < A_; ++q) r[q][b] = col[q];
        }
#pragma unroll
        for (int q = 0; q < A_; ++q) {
            V4 row[A_];
            wino_a<M_, V4>(r[q], row, v4zero());
#pragma unroll
            for (int j = 0; j < A_; ++j) put(vo + (unsigned)(q * A_ + j) * ks, row[j]);
        }
    }
    if (bias_part) {            // block-uniform
        sred[threadIdx.x] = make_float4(bsum.x, bsum.y, bsum.z, bsum.w);
        __syncthreads();
        if (threadIdx.x < C4) {        // the block's tiles in tile order (fixed => bit-reproducible)
            float4 a = sred[threadIdx.x];
            for (unsigned k = threadIdx.x + C4; k < 256; k += C4) {
                const float4 q = sred[k];
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            }
            *reinterpret_cast<float4 *>(bias_part + (size_t)blockIdx.x * C + threadIdx.x * 4) = a;
        }
    }
}

// Backward pass, one read of gy for three consumers: V = B^T d B (the backward-data GEMM's operand), W = A dy A^T of the
// inner m x m tile (the filter-gradient GEMM's operand, kept by the caller) and the per-channel sums of gy (bias
// gradient partials, one row of `bias_part` per block; blocks walk the tiles with a grid stride).  The inner tile is
// re-read for the second transform: it hits L1 / L2, HBM sees gy once.
template <int M_>
__global__ __launch_bounds__(256) void k_wino_gy_dual(const float *__restrict__ gy, float *__restrict__ V, float *__restrict__ Wt,
                                                      float *__restrict__ bias_part, int N, int H, int W, int C, int th, int tw,
                                                      long long T, long long Tp) {
    constexpr int A_ = M_ + 2;
    __shared__ float4 sred[256];
    const int C4 = C / 4;
    const size_t ks = (size_t)Tp * C;
    V4 bsum = v4zero();
    const long long total = Tp * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const long long t = i / C4;
        if (t >= T) {
#pragma unroll
            for (int k = 0; k < A_ * A_; ++k) {
                v4st(V + ((size_t)k * Tp + t) * C + c, v4zero());
                v4st(Wt + ((size_t)k * Tp + t) * C + c, v4zero());
            }
            continue;
        }
        const int tx = (int)(t % tw);
        const int ty = (int)((t / tw) % th);
        const int n = (int)(t / ((long long)tw * th));
        {
            V4 b[A_][A_];
#pragma unroll
            for (int q = 0; q < A_; ++q) {
                const int ww = M_ * tx - 1 + q;
                V4 d[A_], r[A_];
#pragma unroll
                for (int rr = 0; rr < A_; ++rr) {
                    const int h = M_ * ty - 1 + rr;
                    const bool ok = (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W;
                    d[rr] = ok ? v4ld(gy + (((size_t)n * H + h) * W + ww) * C + c) : v4zero();
                }
                wino_bt<M_, V4>(d, r);
#pragma unroll
                for (int rr = 0; rr < A_; ++rr) b[rr][q] = r[rr];
            }
            float *o = V + (size_t)t * C + c;
#pragma unroll
            for (int rr = 0; rr < A_; ++rr) {
                V4 row[A_];
                wino_bt<M_, V4>(b[rr], row);
#pragma unroll
                for (int q = 0; q < A_; ++q) v4st(o + (size_t)(rr * A_ + q) * ks, row[q]);
            }
        }
        {
            V4 r[A_][M_];
#pragma unroll
            for (int bq = 0; bq < M_; ++bq) {
                const int ww = M_ * tx + bq;
                V4 y[M_], col[A_];
#pragma unroll
                for (int a = 0; a < M_; ++a) {
                    const int h = M_ * ty + a;
                    y[a] = (h < H && ww < W) ? v4ld(gy + (((size_t)n * H + h) * W + ww) * C + c) : v4zero();
                    bsum = bsum + y[a];
                }
                wino_a<M_, V4>(y, col, v4zero());
#pragma unroll
                for (int q = 0; q < A_; ++q) r[q][bq] = col[q];
            }
            float *o = Wt + (size_t)t * C + c;
#pragma unroll
            for (int q = 0; q < A_; ++q) {
                V4 row[A_];
                wino_a<M_, V4>(r[q], row, v4zero());
#pragma unroll
                for (int j = 0; j < A_; ++j) v4st(o + (size_t)(q * A_ + j) * ks, row[j]);
            }
        }
    }
    if (bias_part) {        // threads tid, tid + C4, ... hold the same channel group (256 % C4 == 0 is checked by the host)
        sred[threadIdx.x] = make_float4(bsum.x, bsum.y, bsum.z, bsum.w);
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            float4 a = sred[threadIdx.x];
            for (int k = threadIdx.x + C4; k < 256; k += C4) { const float4 q = sred[k]; a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w; }
            *reinterpret_cast<float4 *>(bias_part + (size_t)blockIdx.x * C + threadIdx.x * 4) = a;
        }
    }
}

// gw[co][u][v][ci] (+)= (G^T dU G)[u][v], dU[co][k][ci] = sum over tiles of W[k][t][co] * V[k][t][ci].
template <int M_>
__global__ __launch_bounds__(256) void k_wino_filter_grad(const float *__restrict__ dU, float *__restrict__ gw, int Cout, int Cin,
                                                          int accumulate, int nslab) {
    constexpr int A_ = M_ + 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * Cin) return;
    const int ci = i % Cin, co = i / Cin;
    // all a*a reads first (they were interleaved with the slab loop's branches: one memory latency per element)
    float Dall[A_][A_];
    const size_t o0 = (size_t)co * (A_ * A_) * Cin + ci;
#pragma unroll
    for (int q = 0; q < A_; ++q)
#pragma unroll
        for (int j = 0; j < A_; ++j) Dall[q][j] = dU[o0 + (size_t)(q * A_ + j) * Cin];
    for (int sl = 1; sl < nslab; ++sl) {            // the split-K slabs of dU are added here, in slab order
        const float *ds = dU + (size_t)sl * Cout * (A_ * A_) * Cin + o0;
#pragma unroll
        for (int q = 0; q < A_; ++q)
#pragma unroll
            for (int j = 0; j < A_; ++j) Dall[q][j] += ds[(size_t)(q * A_ + j) * Cin];
    }
    float old[3][3];
    if (accumulate) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) old[u][v] = gw[(((size_t)co * 3 + u) * 3 + v) * Cin + ci];
    }
    float r[3][A_];         // G^T D, column by column
#pragma unroll
    for (int j = 0; j < A_; ++j) {
        float D[A_], c3[3];
#pragma unroll
        for (int q = 0; q < A_; ++q) D[q] = Dall[q][j];
        wino_gt<M_>(D, c3);
#pragma unroll
        for (int u = 0; u < 3; ++u) r[u][j] = c3[u];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        float o[3];
        wino_gt<M_>(r[u], o);
#pragma unroll
        for (int v = 0; v < 3; ++v) gw[(((size_t)co * 3 + u) * 3 + v) * Cin + ci] = accumulate ? old[u][v] + o[v] : o[v];
    }
}

#define WINO_LAUNCH(KERN, G, GRID, ...)                                                              \
    do {                                                                                             \
        if (g_debug_skip & 2) break;                                                                 \
        if ((G).m == 2) hipLaunchKernelGGL((KERN<2>), GRID, dim3(256), 0, st, __VA_ARGS__);          \
        else hipLaunchKernelGGL((KERN<4>), GRID, dim3(256), 0, st, __VA_ARGS__);                     \
        MRCNN_LAUNCH_CHECK();                                                                        \
    } while (0)

// (k_wino_output<4> holds a 6 x 6 tile of float4 per thread and takes 248 VGPRs = 2 waves per SIMD.  Capping it at 3 / 4 waves per SIMD
// through __launch_bounds__ was measured in round 5: the transforms of a 2 x 256^2 x 256 forward call go 189 -> 240 / 288 us - the 36 plane
// loads in flight per thread are what the kernel lives on; left uncapped.)
#define WINO_LAUNCH_OUT(G, GRID, ...) WINO_LAUNCH(k_wino_output, G, GRID, __VA_ARGS__)

int g_pg_big = 1, g_pg_min_tiles = 256;
// Split mode 3 on the Winograd filter-gradient GEMMs of the big layers: k_pgemm_gpp (planes_gemm.h).  The transformed output
// gradient W is written as bf16 planes by k_wino_gy, the transformed input V stays float32 (the forward pass's, or k_wino_input's).
bool pg_big_g_ok(const WinoGeom &g, int Cin, int Cout) {
    if (g_split_mode[PASS_BWD_FILTER] != 3 || !g_pg_big) return false;
    if (Cout % PGB_BM || Cin % PGB_BN || g.Tp % 32) return false;
    return g.Tp >= 2048 && (long long)g.nk * g.Tp * Cout * 6 < (1ll << 32) && (long long)g.nk * g.Tp * Cin * 4 < (1ll << 32);
}
// ksplit / kchunk: the split-K plan of the launch the layer takes by default (the plane GEMM when `big`); ks32 / kc32: the plan tuned for
// k_conv_igemm's 128 x 128 tiles, taken when the call arrives with a cached float32 W (WINOGRAD_SHARED_GY_TRANSFORM) - ADVICE r4
struct WinoFLayout { size_t v, w, slabs, du, total; WinoGeom g; int ksplit, kchunk; int big; int ks32, kc32; };
WinoFLayout wino_filter_layout(int N, int H, int W, int Cin, int Cout) {
    WinoFLayout L;
    L.g = wino_geom(N, H, W, PASS_BWD_FILTER);
    ConvP p = make_p(1, 1, (int)L.g.Tp, Cin, Cout, 1, 1, 1, 0);
    p.wbatch_rows = (int)L.g.Tp; p.wbatch_n = L.g.nk;
    filter_plan(p, L.ksplit, L.kchunk);
    L.ks32 = L.ksplit; L.kc32 = L.kchunk;
    L.big = pg_big_g_ok(L.g, Cin, Cout) ? 1 : 0;
    if (L.big) {            // k_pgemm_gpp: one 256 x 256 tile per (batch, split), one workgroup per CU: the splits fill one round of CUs
        const long long tiles = (long long)L.g.nk * (Cout / PGB_BM) * (Cin / PGB_BN);
        long long ks = std::max(1ll, (long long)g_cus() / tiles);
        ks = std::min(ks, std::max(1ll, L.g.Tp / 256));                 // at least 16 K steps per split
        L.kchunk = (int)((L.g.Tp / 32 + ks - 1) / ks * 32);
        L.ksplit = (int)((L.g.Tp + L.kchunk - 1) / L.kchunk);
    }
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t o = 0;
    L.v = o; o += al((size_t)L.g.nk * L.g.Tp * Cin * 4);
    L.w = o; o += al((size_t)L.g.nk * L.g.Tp * Cout * 6);          // float32, or three bf16 planes (plane GEMM)
    L.slabs = o; o += al((size_t)std::max(L.ksplit, L.ks32) * Cout * L.g.nk * Cin * 4);
    L.du = o; o += al((size_t)Cout * L.g.nk * Cin * 4);
    L.total = o;
    return L;
}

// Set by the *_inbn entry points around their call of the ordinary entry point (which they have checked takes the Winograd path): the
// input transform of that call applies this BatchNorm + ReLU on load.
thread_local InBN t_inbn{};

// bias_part / bias_rows (nullable): when the gy transform runs here and its blocks cover whole tiles, it also leaves the
// bias gradient's partial column sums in bias_part and *bias_rows = their number (else *bias_rows = 0: the caller sums gy)
int wino_bwd_filter(const float *x, const float *gy, float *gw, int N, int H, int W, int Cin, int Cout, int accumulate, void *ws,
                    hipStream_t st, const float *v_cached, const float *w_cached, float *bias_part = nullptr, int *bias_rows = nullptr) {
    const InBN inbn = t_inbn;
    const WinoFLayout L = wino_filter_layout(N, H, W, Cin, Cout);
    const WinoGeom &g = L.g;
    char *base = (char *)ws;
    float *Vw = (float *)(base + L.v), *Wt = (float *)(base + L.w), *slabs = (float *)(base + L.slabs), *dU = (float *)(base + L.du);
    const float *V = v_cached ? v_cached : Vw;       // the forward pass's transformed input, kept by the caller
    const long long nin = g.Tp * (Cin / 4), nout = g.Tp * (Cout / 4);      // the transform kernels zero the padded rows
    if (!v_cached) WINO_LAUNCH(k_wino_input, g, dim3((unsigned)((nin + 255) / 256)), x, Vw, N, H, W, Cin, g.th, g.tw, g.T, g.Tp, g_wino_banded, inbn);
    const int C4o = Cout / 4;
    const bool fuse_bias = bias_part && bias_rows && !w_cached && C4o <= 256 && 256 % C4o == 0 && !(g_debug_skip & 2);
    if (bias_rows) *bias_rows = fuse_bias ? (int)((nout + 255) / 256) : 0;
    const int big = (L.big && !w_cached) ? 1 : 0;           // (a cached W is float32: the shared-transform path keeps k_conv_igemm)
    if (!w_cached) WINO_LAUNCH(k_wino_gy, g, dim3((unsigned)((nout + 255) / 256)), gy, Wt, N, H, W, Cout, g.th, g.tw, g.T, g.Tp,
                               fuse_bias ? bias_part : (float *)nullptr, big);
    if (big) {
        PlaneGemmP q{};
        q.a = reinterpret_cast<const unsigned short *>(Wt); q.b = reinterpret_cast<const unsigned short *>(V);
        q.c = L.ksplit > 1 ? slabs : dU;
        q.M = Cout; q.N = Cin; q.K = (int)g.Tp; q.batch_rows = (int)g.Tp; q.nbatch = g.nk; q.ksplit = L.ksplit; q.kchunk = L.kchunk;
        q.bytes_a = (unsigned)((size_t)g.nk * g.Tp * Cout * 6); q.bytes_b = (unsigned)((size_t)g.nk * g.Tp * Cin * 4);
        q.ldc = g.nk * Cin; q.dbg = 0; q.stamps = nullptr;
        if (!(g_debug_skip & 1)) launch_pgemm<5>(q, PGB_BM, PGB_BN, st);
        MRCNN_LAUNCH_CHECK();
        if (L.ksplit > 1) {
            const size_t n4 = (size_t)Cout * g.nk * Cin / 4;
            if (!(g_debug_skip & 10)) hipLaunchKernelGGL(k_sum_slabs, dim3(mrcnn::cdiv(n4, 256)), dim3(256), 0, st, slabs, dU, n4, L.ksplit, 0);
            MRCNN_LAUNCH_CHECK();
        }
        WINO_LAUNCH(k_wino_filter_grad, g, dim3(mrcnn::cdiv(Cout * Cin, 256)), dU, gw, Cout, Cin, accumulate, 1);
        return 0;
    }
    ConvP p = make_p(1, 1, (int)g.Tp, Cin, Cout, 1, 1, 1, 0);
    p.wbatch_rows = (int)g.Tp; p.wbatch_n = g.nk;
    p.ksplit = L.ks32; p.kchunk = L.kc32;
    p.a = w_cached ? w_cached : Wt; p.b = V; p.c = p.ksplit > 1 ? slabs : dU;
    p.bytes_a = (unsigned)((size_t)g.nk * g.Tp * Cout * 4); p.bytes_b = (unsigned)((size_t)g.nk * g.Tp * Cin * 4);
    p.M = Cout; p.Ng = Cin;
    launch_conv<MODE_BWD_FILTER>(p, g.nk * p.ksplit, filter_tile(p), st);
    MRCNN_LAUNCH_CHECK();
    if (p.ksplit > 1) {     // a separate, fully parallel slab sum: folding it into k_wino_filter_grad (Cout*Cin threads only) was
                            // measured 20 % .. 4x slower (tools/wino_sweep.py)
        const size_t n4 = (size_t)Cout * g.nk * Cin / 4;
        if (!(g_debug_skip & 10)) hipLaunchKernelGGL(k_sum_slabs, dim3(mrcnn::cdiv(n4, 256)), dim3(256), 0, st, slabs, dU, n4, p.ksplit, 0);
        MRCNN_LAUNCH_CHECK();
    }
    WINO_LAUNCH(k_wino_filter_grad, g, dim3(mrcnn::cdiv(Cout * Cin, 256)), dU, gw, Cout, Cin, accumulate, 1);
    return 0;
}

// Split mode 3 on the Winograd batched GEMMs of the big layers: the plane GEMM of planes_gemm.h (k_pgemm_pp: 256 x 256 tiles, the
// transformed activations V stay float32 and are split when a wave reads its fragment, the transformed filters U are written as
// bf16 planes by the filter transform).  g_pg_big: 0 = never (mrcnn_debug_conv_parts bit 8 of the high byte; A/B against k_conv_igemm).
bool pg_big_ok(int pass, const WinoGeom &g, int K, int Nn) {
    if (g_split_mode[pass] != 3 || !g_pg_big) return false;
    if (g.Tp % PGB_BM || Nn % PGB_BN || K % 16) return false;
    const long long tiles = (long long)g.nk * g.Tp / PGB_BM * (Nn / PGB_BN);
    // the persistent grid walks the tiles in rounds of one per CU: a last round that is mostly empty (288 tiles = 2 rounds at 0.56) loses
    // to k_conv_igemm's 128 x 128 tiles (tools/gemm_only_profile.py 3,3,3 2,2,0 0,256: 67 against 57 us on the p3 layers)
    const long long rounds = (tiles + g_cus() - 1) / g_cus();
    if (tiles * 10 < rounds * g_cus() * 8) return false;
    return tiles >= g_pg_min_tiles && (long long)g.nk * g.Tp * K * 4 < (1ll << 32) && (long long)g.nk * Nn * K * 6 < (1ll << 32);
}

// in (N,H,W,Cin) -> out (N,H,W,Cout); w is always the layer's (Cout_layer,3,3,Cin_layer) weight tensor: transposed selects
// the backward-data filter (then Cin here = the layer's Cout and Cout here = the layer's Cin).
int wino_conv(const float *in, const float *w, float *out, int N, int H, int W, int Cin, int Cout, bool transposed,
              const float *bias, int relu, int accumulate, const float *relu_x, void *ws, size_t ws_bytes, hipStream_t st,
              float *v_keep, float *w_keep = nullptr, float *gbias = nullptr, int gbias_accumulate = 0, float *bn_part = nullptr) {
    const InBN inbn = transposed ? InBN{} : t_inbn;
    const WinoLayout L = wino_layout(N, H, W, Cin, Cout, transposed ? PASS_BWD_DATA : PASS_FWD);
    const WinoGeom &g = L.g;
    char *base = (char *)ws;
    float *U = (float *)(base + L.u), *V = v_keep ? v_keep : (float *)(base + L.v), *Mb = (float *)(base + L.m);
    // the layer's weight tensor is (Cout_layer, 3, 3, Cin_layer): forward Cout_layer = Cout; transposed Cout_layer = Cin
    const long long nin = g.Tp * (Cin / 4), nout = g.T * (Cout / 4);       // k_wino_input zeroes the padded rows of V
    const unsigned fblocks = (unsigned)mrcnn::cdiv(Cout * Cin, 256);
    const int big = pg_big_ok(transposed ? PASS_BWD_DATA : PASS_FWD, g, Cin, Cout) ? 1 : 0;
    if (w_keep) WINO_LAUNCH(k_wino_filter, g, dim3(fblocks), w, U, transposed ? Cin : Cout, transposed ? Cout : Cin, transposed ? 1 : 0, big);
    if (w_keep) {           // backward pass: one read of gy feeds this GEMM, the filter-gradient GEMM and the bias gradient
        const bool wb = gbias && (256 % (Cin / 4)) == 0;
        const int nblk = (int)std::min<long long>((nin + 255) / 256, 512);
        float *bias_part = wb ? (float *)(base + L.m) : nullptr;          // M is written only after this kernel has finished
        WINO_LAUNCH(k_wino_gy_dual, g, dim3(nblk), in, V, w_keep, bias_part, N, H, W, Cin, g.th, g.tw, g.T, g.Tp);
        if (wb) {
            if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_final, dim3(mrcnn::cdiv(Cin, CSF_CH)), dim3(CSF_CH * CSF_SL), 0, st, bias_part, gbias, nblk, Cin, gbias_accumulate);
            MRCNN_LAUNCH_CHECK();
        } else if (gbias) {       // channel count that does not tile a 256-thread block: the ordinary two-kernel column sum
            const int P = N * H * W;
            const ColPlan cp = col_plan(P, Cin);
            float *part = (float *)(base + L.m);
            if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_partial, dim3(cp.nblk), dim3(256), 0, st, in, part, P, Cin, cp.G, cp.RPI, cp.rows_per_blk);
            MRCNN_LAUNCH_CHECK();
            if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_final, dim3(mrcnn::cdiv(Cin, CSF_CH)), dim3(CSF_CH * CSF_SL), 0, st, part, gbias, cp.nblk, Cin, gbias_accumulate);
            MRCNN_LAUNCH_CHECK();
        }
    } else
        WINO_LAUNCH(k_wino_input_filter, g, dim3(fblocks + (unsigned)((nin + 255) / 256)), in, V, N, H, W, Cin, g.th, g.tw, g.T, g.Tp, w, U,
                    transposed ? Cin : Cout, transposed ? Cout : Cin, transposed ? 1 : 0, fblocks, g_wino_banded, big, inbn);
    if (big) {          // the plane GEMM: A = V (float32 rows), B = U (P16R4 planes), C = M
        PlaneGemmP q{};
        q.a = reinterpret_cast<const unsigned short *>(V); q.b = reinterpret_cast<const unsigned short *>(U); q.c = Mb;
        q.M = (int)(g.nk * g.Tp); q.N = Cout; q.K = Cin; q.batch_rows = (int)g.Tp; q.nbatch = g.nk; q.ksplit = 1; q.kchunk = Cin;
        q.bytes_a = (unsigned)((size_t)g.nk * g.Tp * Cin * 4); q.bytes_b = (unsigned)((size_t)g.nk * Cout * Cin * 6);
        q.ldc = Cout; q.dbg = 0; q.stamps = nullptr;
        if (!(g_debug_skip & 1)) launch_pgemm<4>(q, PGB_BM, PGB_BN, st);
        MRCNN_LAUNCH_CHECK();
        WINO_LAUNCH_OUT(g, dim3((unsigned)((nout + 255) / 256)), Mb, out, N, H, W, Cout, g.th, g.tw, g.T, g.Tp, bias, relu,
                    accumulate, relu_x, bn_part);
        return 0;
    }
    // batched GEMM: 1x1 "convolution" over nk*Tp pixels, weight matrix selected by the row block
    ConvP p = make_p(1, 1, (int)(g.nk * g.Tp), Cin, Cout, 1, 1, 1, 0);
    p.a = V; p.b = U; p.c = Mb;
    p.bytes_a = (unsigned)((size_t)g.nk * g.Tp * Cin * 4); p.bytes_b = (unsigned)((size_t)g.nk * Cout * Cin * 4);
    p.M = (int)(g.nk * g.Tp); p.Ng = Cout;
    p.wbatch_rows = (int)g.Tp; p.wbatch_n = g.nk;
    if (int e = run_data_conv<MODE_FWD>(p, Cin / BK, Cout, base + L.inner, ws_bytes - L.inner, st)) return e;
    WINO_LAUNCH_OUT(g, dim3((unsigned)((nout + 255) / 256)), Mb, out, N, H, W, Cout, g.th, g.tw, g.T, g.Tp, bias, relu,
                accumulate, relu_x, bn_part);
    return 0;
}

}  // namespace

extern "C" size_t mrcnn_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    // upper bound for both the forward and the backward-data split-K slabs (16 splits of the larger output)
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    const long long Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
    if (Ho <= 0 || Wo <= 0) return 0;
    const long long a = (long long)N * Ho * Wo * Cout, b = (long long)N * H * W * Cin;
    const long long fill = (long long)g_plan_fill * g_cus();
    // split-K only engages when the 64x64 tiling has fewer than `fill` tiles: M*Ng < fill*4096
    const long long cap = fill * 8192;
    long long need = 0;
    if (a < cap) need = std::max(need, a);
    if (b < cap) need = std::max(need, b);
    // tail split: at most one round of workgroup slots of partial 128x128 tiles (<= 4 workgroups per CU)
    const size_t tail = (size_t)4 * g_cus() * 128 * 128 * sizeof(float);
    size_t bytes = std::max((size_t)need * 16 * sizeof(float), tail);
    // the same query serves forward (Cin->Cout) and backward-data (Cout->Cin)
    if (wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_FWD)) bytes = std::max(bytes, wino_ws_bytes(N, H, W, Cin, Cout, PASS_FWD));
    if (wino_ok(N, H, W, Cout, Cin, KH, KW, stride, pad, PASS_BWD_DATA)) bytes = std::max(bytes, wino_ws_bytes(N, H, W, Cout, Cin, PASS_BWD_DATA));
    return bytes;
}

extern "C" int mrcnn_debug_conv_parts(int mask) {
    g_dbg_parts = mask & 0xff;
    g_pg_big = (mask & 0x100) ? 0 : 1;        // 8: the three-plane kernels run the plain K loop instead of the pipelined one; 16..128: planes_gemm.h knobs
    return 0;
}

// Measurement / test entry points of planes_gemm.h: float32 (R, C) -> P16 planes, and one plane-GEMM launch.
extern "C" int mrcnn_debug_split_planes_f32(const float *x, void *planes, int R, int C, int r4, void *stream) {
    if (!x || !planes || R <= 0 || C <= 0 || C % 16 || (long long)R * C * 6 >= (1ll << 32) || (r4 == 1 && R % 4))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_split_planes: null pointer, C %% 16 != 0, R %% 4 != 0 (P16R4) or more than 4 GiB of planes");
    hipLaunchKernelGGL(k_split_planes, dim3(mrcnn::cdiv((long long)R * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short *)planes, (unsigned)R, (unsigned)C, r4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}
static unsigned long long *g_pg_stamps = nullptr;
extern "C" int mrcnn_debug_planes_gemm_stamps(unsigned long long *stamps) {       // 5 x u64 per workgroup of the next debug_planes_gemm calls (F kinds); null = off
    g_pg_stamps = stamps;
    return 0;
}
extern "C" int mrcnn_debug_planes_gemm(int kind, const void *a, const void *b, float *c, int M, int N, int K, int batch_rows, int nbatch,
                                       int ksplit, int bm, int bn, void *stream) {
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || nbatch <= 0 || ksplit <= 0 || kind < 0 || kind > 5)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_planes_gemm: bad arguments");
    PlaneGemmP p{};
    p.a = (const unsigned short *)a; p.b = (const unsigned short *)b; p.c = c;
    p.M = M; p.N = N; p.K = K; p.batch_rows = batch_rows; p.nbatch = nbatch; p.ksplit = ksplit;
    long long ba, bb;
    if (kind == 5) {      // G, big tile, ping-pong: A = W (nbatch * K, M) "PR" planes, B = V (nbatch * K, N) float32; K = batch_rows
        if (K != batch_rows || K % 32 || M % 16 || N % 16 || bm != 256 || bn != 256)
            return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_planes_gemm G big: K == batch_rows, K %% 32, tile 256 x 256");
        ba = (long long)nbatch * K * M * 6; bb = (long long)nbatch * K * N * 4;
        p.kchunk = (K / 32 + ksplit - 1) / ksplit * 32;         // the caller's slab array has `ksplit` slabs; all are written
        if ((K + p.kchunk - 1) / p.kchunk != ksplit) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_planes_gemm G big: ksplit %d leaves an empty split", ksplit);
        p.ldc = nbatch * N;
    } else if (kind == 3 || kind == 4) {      // F, big tile (4: ping-pong halves): A (M, K) float32, B (nbatch * N, K) P16R4 planes; batch_rows % 256 == 0
        if (K % 16 || batch_rows % 256 || M != nbatch * batch_rows || N % 16 || bm != 256 || bn != 256)
            return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_planes_gemm big: K %% 16, batch_rows %% 256, tile 256 x 256");
        ba = (long long)M * K * 4; bb = (long long)nbatch * N * K * 6;
        p.ldc = N; p.ksplit = 1; p.kchunk = K;
    } else if (kind == 0 || kind == 2) {        // F: A (M, K) planes (kind 2: float32), B (nbatch, N, K) planes; M = nbatch * batch_rows
        if (K % 32 || batch_rows % 128 || M != nbatch * batch_rows || N % 16 || bm != 128 || (bn != 128 && bn != 64))       // (two-stage ring, steps in pairs)
            return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_planes_gemm F: K %% 32, batch_rows %% 128, M == nbatch * batch_rows, tile 128 x 128|64");
        ba = (long long)M * K * (kind == 2 ? 4 : 6); bb = (long long)nbatch * N * K * 6;
        p.ldc = N; p.ksplit = 1; p.kchunk = K;
    } else {                // G: A (nbatch, K, M), B (nbatch, K, N); K = batch_rows
        if (K != batch_rows || K % (32 * ksplit) || M % 16 || N % 16 || (bm != 128 && bm != 64) || (bn != 128 && bn != 64))   // (steps in pairs per split)
            return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_planes_gemm G: K == batch_rows, K %% (32 * ksplit), tiles 128|64");
        ba = (long long)nbatch * K * M * 6; bb = (long long)nbatch * K * N * 6;
        p.ldc = nbatch * N; p.kchunk = K / ksplit;
    }
    if (ba >= (1ll << 32) || bb >= (1ll << 32)) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "debug_planes_gemm: operand of 4 GiB or more");
    p.bytes_a = (unsigned)ba; p.bytes_b = (unsigned)bb;
    p.dbg = g_dbg_parts >> 4;
    p.stamps = g_pg_stamps;
    if (kind == 0) launch_pgemm<0>(p, bm, bn, (hipStream_t)stream);
    else if (kind == 2) launch_pgemm<2>(p, bm, bn, (hipStream_t)stream);
    else if (kind == 3) launch_pgemm<3>(p, bm, bn, (hipStream_t)stream);
    else if (kind == 4) launch_pgemm<4>(p, bm, bn, (hipStream_t)stream);
    else if (kind == 5) launch_pgemm<5>(p, bm, bn, (hipStream_t)stream);
    else launch_pgemm<1>(p, bm, bn, (hipStream_t)stream);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_debug_wino_banded(int on) {
    g_wino_banded = on < 0 ? 0 : on;          // 0 launch order, 1 XCD-banded raster, >= 2 XCD-banded column panels of that many tiles
    return 0;
}

extern "C" int mrcnn_debug_conv_plan(int fill, int filter_rounds, int force_tile) {
    if (fill < 1 || fill > 16 || filter_rounds < 1 || filter_rounds > 8) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_conv_plan: fill 1..16, filter_rounds (in half rounds) 1..8");
    g_plan_fill = fill; g_plan_filter_rounds = filter_rounds; g_plan_force_tile = force_tile;
    return 0;
}

extern "C" int mrcnn_conv2d_set_winograd_thresholds(int min_channels, int min_pixels, int tile) {
    if (min_channels < BK || min_pixels < 1 || (tile != 0 && tile != 2 && tile != 4))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "set_winograd_thresholds: min_channels >= %d, min_pixels >= 1, tile 0/2/4", BK);
    g_wino_min_channels = min_channels;
    g_wino_min_pixels = min_pixels;
    g_wino_tile = tile;
    return 0;
}

extern "C" int mrcnn_conv2d_set_winograd_pass_tiles(int fwd, int bwd_data, int bwd_filter) {
    const int t[3] = {fwd, bwd_data, bwd_filter};
    for (int i = 0; i < 3; ++i)
        if (t[i] != -1 && t[i] != 0 && t[i] != 2 && t[i] != 4)
            return mrcnn::fail_arg(MRCNN_E_INVALID, "set_winograd_pass_tiles: each of -1 (direct), 0 (follow the global tile), 2, 4");
    for (int i = 0; i < 3; ++i) g_wino_pass_tile[i] = t[i];
    return 0;
}

extern "C" int mrcnn_conv2d_get_winograd_pass_tiles(int *tiles3) {
    if (!tiles3) return mrcnn::fail_arg(MRCNN_E_INVALID, "get_winograd_pass_tiles: null output");
    for (int i = 0; i < 3; ++i) tiles3[i] = g_wino_pass_tile[i];
    return 0;
}

extern "C" int mrcnn_conv2d_set_split_operands(int fwd, int bwd_data, int bwd_filter) {
    const int m[3] = {fwd, bwd_data, bwd_filter};
    for (int i = 0; i < 3; ++i)
        if (m[i] < 0 || m[i] > 3) return mrcnn::fail_arg(MRCNN_E_INVALID, "set_split_operands: each of 0 (float32), 1 (bf16 hi/lo), 2 (half hi/lo), 3 (bf16 hi/mid/lo, float32-accurate)");
    for (int i = 0; i < 3; ++i) g_split_mode[i] = m[i];
    return 0;
}

extern "C" int mrcnn_conv2d_get_split_operands(int *modes3) {
    if (!modes3) return mrcnn::fail_arg(MRCNN_E_INVALID, "get_split_operands: null output");
    for (int i = 0; i < 3; ++i) modes3[i] = g_split_mode[i];
    return 0;
}

extern "C" int mrcnn_conv2d_set_debug_skip(int mask) {
    if (mask < 0 || mask > 15) return mrcnn::fail_arg(MRCNN_E_INVALID, "set_debug_skip: mask in [0,15]");
    g_debug_skip = mask;
    return 0;
}

extern "C" long long mrcnn_conv2d_executed_macs(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int pass) {
    // multiply-accumulates the MFMA pipes actually execute for one pass (0 forward, 1 backward-data, 2 backward-filter)
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || pass < 0 || pass > 2) return 0;
    const bool wk = pass == PASS_BWD_DATA ? wino_ok(N, H, W, Cout, Cin, KH, KW, stride, pad, pass) : wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, pass);
    if (wk) {
        const WinoGeom g = wino_geom(N, H, W, pass);
        return (long long)g.nk * g.Tp * Cin * Cout;
    }
    const long long Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
    return (long long)N * Ho * Wo * KH * KW * (Cin == 4 ? 4 : Cin) * Cout;
}

extern "C" size_t mrcnn_conv2d_winograd_v_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    if (!wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_FWD)) return 0;
    const WinoGeom g = wino_geom(N, H, W, PASS_FWD);
    return (size_t)g.nk * g.Tp * Cin * sizeof(float);
}

extern "C" int mrcnn_conv2d_fwd_f32(const float *x, const float *w, const float *bias, float *y, int N, int H,
                                    int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                                    float *wino_v, void *ws, size_t ws_bytes, void *stream) {
    g_cur_pass = 0;
    if (int e = check_conv(x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    if (wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_FWD) && ws && ws_bytes >= wino_ws_bytes(N, H, W, Cin, Cout, PASS_FWD))
        return wino_conv(x, w, y, N, H, W, Cin, Cout, false, bias, relu, 0, nullptr, ws, ws_bytes, (hipStream_t)stream, wino_v);
    ConvP p = make_p(N, H, W, Cin, Cout, KH, KW, stride, pad);
    p.a = x; p.b = w; p.c = y; p.bias = bias; p.relu = relu;
    p.bytes_a = (unsigned)((size_t)N * H * W * Cin * 4); p.bytes_b = (unsigned)((size_t)Cout * KH * KW * Cin * 4);
    p.M = N * p.Ho * p.Wo; p.Ng = Cout;
    const int nsteps = p.smallc ? (KH * KW + 7) / 8 : KH * KW * (Cin / BK);
    return run_data_conv<MODE_FWD>(p, nsteps, Cout, ws, ws_bytes, (hipStream_t)stream);
}

// Forward convolution of a layer that feeds a training-mode BatchNorm (no bias, no ReLU): the epilogue also leaves the
// per-channel sums / sums of squares of its output rows in bn_part (rows, 2, Cout) - the statistics pass of BatchNorm
// (one read of the activation) disappears; mrcnn_bn_train_fwd_stats_f32 finishes from the partials.  rows = 0: this
// geometry takes a path without the fused statistics (Winograd, split-K or tail-split launches) - call the plain entry.
static int bnstats_plan(ConvP &p, TileChoice &t, bool &wino, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    wino = false;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    if ((Cin % BK && Cin != 4) || Cout % BK || Cin == 4) return 0;
    if (wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_FWD)) {
        // Winograd forward: the output transform produces the partials, one row per 256-thread block of whole tiles
        const int C4 = Cout / 4;
        if (C4 > 256 || 256 % C4) return 0;
        wino = true;
        const WinoGeom g = wino_geom(N, H, W, PASS_FWD);
        return (int)((g.T * C4 + 255) / 256);
    }
    p = make_p(N, H, W, Cin, Cout, KH, KW, stride, pad);
    if (p.Ho <= 0 || p.Wo <= 0) return 0;
    p.M = N * p.Ho * p.Wo; p.Ng = Cout;
    data_plan<MODE_FWD>(p, t, KH * KW * (Cin / BK));
    if (p.ksplit > 1 || p.tail_ks) return 0;
    return mrcnn::cdiv(p.M, t.bm) * 2;
}
extern "C" size_t mrcnn_conv2d_bnstats_rows(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    ConvP p;
    TileChoice t;
    bool wino;
    return (size_t)bnstats_plan(p, t, wino, N, H, W, Cin, Cout, KH, KW, stride, pad);
}
extern "C" int mrcnn_conv2d_fwd_bnstats_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH,
                                            int KW, int stride, int pad, float *bn_part, float *wino_v, void *ws, size_t ws_bytes,
                                            void *stream) {
    g_cur_pass = 0;
    if (int e = check_conv(x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    ConvP p;
    TileChoice t;
    bool wino;
    if (!bn_part || bnstats_plan(p, t, wino, N, H, W, Cin, Cout, KH, KW, stride, pad) == 0)
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "conv2d_fwd_bnstats: no fused statistics for this geometry (mrcnn_conv2d_bnstats_rows == 0)");
    if (wino) {
        if (!ws || ws_bytes < wino_ws_bytes(N, H, W, Cin, Cout, PASS_FWD))
            return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "conv2d_fwd_bnstats: workspace %zu < %zu", ws_bytes, wino_ws_bytes(N, H, W, Cin, Cout, PASS_FWD));
        return wino_conv(x, w, y, N, H, W, Cin, Cout, false, nullptr, 0, 0, nullptr, ws, ws_bytes, (hipStream_t)stream, wino_v, nullptr, nullptr, 0, bn_part);
    }
    p.a = x; p.b = w; p.c = y; p.bias = nullptr; p.relu = 0; p.bn_part = bn_part;
    p.bytes_a = (unsigned)((size_t)N * H * W * Cin * 4); p.bytes_b = (unsigned)((size_t)Cout * KH * KW * Cin * 4);
    launch_conv<MODE_FWD>(p, 1, t, (hipStream_t)stream);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// ---- (r6, ABI v10) the layer's input is relu(BatchNorm(x)) of a training-mode BatchNorm with final statistics: applied on load by the
// Winograd input transform (wino_input_body), never written.  Only for geometries whose forward AND filter-gradient passes take the
// Winograd path under the settings in force (mrcnn_conv2d_inbn_ok): the caller keeps the materialised activation otherwise.
extern "C" int mrcnn_conv2d_inbn_ok(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || (Cin % 4)) return 0;
    return (wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_FWD) && wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_BWD_FILTER)) ? 1 : 0;
}
static int inbn_args(const float *g, const float *b, const float *m, const float *s, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                     int pad, const char *who) {
    if (!g || !b || !m || !s) return mrcnn::fail_arg(MRCNN_E_INVALID, "%s: null BatchNorm pointer", who);
    if (!mrcnn_conv2d_inbn_ok(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "%s: this geometry does not take the Winograd path in both passes (mrcnn_conv2d_inbn_ok == 0)", who);
    return 0;
}
extern "C" int mrcnn_conv2d_fwd_inbn_f32(const float *x, const float *in_gamma, const float *in_beta, const float *in_mean, const float *in_invstd,
                                         const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                         float *bn_part, float *wino_v, void *ws, size_t ws_bytes, void *stream) {
    if (int e = inbn_args(in_gamma, in_beta, in_mean, in_invstd, N, H, W, Cin, Cout, KH, KW, stride, pad, "conv2d_fwd_inbn")) return e;
    if (!ws || ws_bytes < wino_ws_bytes(N, H, W, Cin, Cout, PASS_FWD))
        return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "conv2d_fwd_inbn: workspace %zu < %zu", ws_bytes, wino_ws_bytes(N, H, W, Cin, Cout, PASS_FWD));
    t_inbn = InBN{in_gamma, in_beta, in_mean, in_invstd};
    const int rc = bn_part ? mrcnn_conv2d_fwd_bnstats_f32(x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, bn_part, wino_v, ws, ws_bytes, stream)
                           : mrcnn_conv2d_fwd_f32(x, w, nullptr, y, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, wino_v, ws, ws_bytes, stream);
    t_inbn = InBN{};
    return rc;
}

// Forward convolution with a rectangular kernel and per-axis padding (the 15x1 / 1x15 separable pairs of LightRoIMaskHead,
// model/head/light_roi_mask_head.py:29-44).  Same kernel as mrcnn_conv2d_fwd_f32 (never the Winograd path).
extern "C" int mrcnn_conv2d_fwd_rect_f32(const float *x, const float *w, const float *bias, float *y, int N, int H, int W,
                                         int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int relu, void *ws,
                                         size_t ws_bytes, void *stream) {
    g_cur_pass = 0;
    if (pad_h < 0 || pad_w < 0) return mrcnn::fail_arg(MRCNN_E_INVALID, "conv2d_fwd_rect: negative padding");
    // sizes and limits are validated for the larger of the two paddings (a superset of the real output)
    if (int e = check_conv(x, w, y, N, H, W, Cin, Cout, KH, KW, stride, std::max(pad_h, pad_w))) return e;
    if (conv_out(H, KH, stride, pad_h) <= 0 || conv_out(W, KW, stride, pad_w) <= 0 || Cin == 4)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "conv2d_fwd_rect: empty output (or the 4-channel image layer)");
    ConvP p = make_p(N, H, W, Cin, Cout, KH, KW, stride, pad_h);
    p.pad_w = pad_w;
    p.Wo = conv_out(W, KW, stride, pad_w);
    p.a = x; p.b = w; p.c = y; p.bias = bias; p.relu = relu;
    p.bytes_a = (unsigned)((size_t)N * H * W * Cin * 4); p.bytes_b = (unsigned)((size_t)Cout * KH * KW * Cin * 4);
    p.M = N * p.Ho * p.Wo; p.Ng = Cout;
    return run_data_conv<MODE_FWD>(p, KH * KW * (Cin / BK), Cout, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t mrcnn_conv2d_winograd_w_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    // the shared transform of gy needs the backward-data and backward-filter passes on the same tile
    if (!wino_ok(N, H, W, Cout, Cin, KH, KW, stride, pad, PASS_BWD_DATA) || !wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_BWD_FILTER)) return 0;
    const WinoGeom g = wino_geom(N, H, W, PASS_BWD_DATA);
    if (g.m != wino_geom(N, H, W, PASS_BWD_FILTER).m) return 0;
    return (size_t)g.nk * g.Tp * Cout * sizeof(float);
}

extern "C" int mrcnn_conv2d_bwd_data_f32(const float *gy, const float *w, float *gx, const float *relu_x, int N, int H, int W,
                                         int Cin, int Cout, int KH, int KW, int stride, int pad,
                                         int accumulate, float *wino_w, float *gbias, int gbias_accumulate, void *ws,
                                         size_t ws_bytes, void *stream) {
    g_cur_pass = 1;
    if (int e = check_conv(gy, w, gx, N, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    if (stride != 1)
        return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "conv2d_bwd_data: stride %d (only 1; strided 1x1 convs are "
                                                    "handled by the host as a subsample + stride-1 conv)", stride);
    ConvP p = make_p(N, H, W, Cin, Cout, KH, KW, stride, pad);
    if (Cin == 4) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "conv2d_bwd_data: Cin == 4 (image layer) has no data gradient");
    // relu_x with accumulate: every path (direct epilogue, split-K / tail-split sums, Winograd output transform) adds the old value
    // first and masks the TOTAL - the ReLU backward of the tensor gx is the gradient of, applied where its last contribution lands
    if (wino_ok(N, H, W, Cout, Cin, KH, KW, stride, pad, PASS_BWD_DATA) && ws && ws_bytes >= wino_ws_bytes(N, H, W, Cout, Cin, PASS_BWD_DATA))
        return wino_conv(gy, w, gx, N, H, W, Cout, Cin, true, nullptr, 0, accumulate, relu_x, ws, ws_bytes, (hipStream_t)stream, nullptr,
                         wino_w, wino_w ? gbias : nullptr, gbias_accumulate);
    if (wino_w || gbias) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "conv2d_bwd_data: wino_w / gbias only on the Winograd path (mrcnn_conv2d_winograd_w_bytes() > 0)");
    p.a = gy; p.b = w; p.c = gx; p.accumulate = accumulate; p.relu_x = relu_x;
    p.bytes_a = (unsigned)((size_t)N * p.Ho * p.Wo * Cout * 4); p.bytes_b = (unsigned)((size_t)Cout * KH * KW * Cin * 4);
    p.M = N * H * W; p.Ng = Cin;
    return run_data_conv<MODE_BWD_DATA>(p, KH * KW * (Cout / BK), Cin, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t mrcnn_conv2d_bwd_filter_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                                          int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    ConvP p = make_p(N, H, W, Cin, Cout, KH, KW, stride, pad);
    if (p.Ho <= 0 || p.Wo <= 0) return 0;
    int ksplit, kchunk;
    filter_plan(p, ksplit, kchunk);
    const size_t wsz = (size_t)Cout * KH * KW * Cin * sizeof(float);
    const size_t P = (size_t)N * p.Ho * p.Wo;
    const size_t bias_part = (size_t)col_plan((int)P, Cout).nblk * Cout * sizeof(float);
    if (wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_BWD_FILTER)) {
        // the gy transform leaves one row of bias-gradient partials per 256-thread block
        const WinoFLayout L = wino_filter_layout(N, H, W, Cin, Cout);
        const size_t rows = (size_t)((L.g.Tp * (Cout / 4) + 255) / 256);
        return L.total + std::max(bias_part, rows * Cout * sizeof(float)) + 256;
    }
    return wsz * ksplit + bias_part + 256;     // slabs are also used for ksplit == 1 when accumulating
}

extern "C" int mrcnn_conv2d_bwd_filter_f32(const float *x, const float *gy, float *gw, float *gbias, int N,
                                           int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                           int accumulate, const float *wino_v, const float *wino_w, void *ws, size_t ws_bytes,
                                           void *stream) {
    g_cur_pass = 2;
    if (int e = check_conv(x, gy, gw, N, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    if (g_debug_skip & 4) return 0;             // timing experiments only (tools/ab_step.py): what the whole pass costs the step
    const size_t need = mrcnn_conv2d_bwd_filter_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad);
    if (!ws || ws_bytes < need) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "conv2d_bwd_filter: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    ConvP p = make_p(N, H, W, Cin, Cout, KH, KW, stride, pad);
    if (wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_BWD_FILTER)) {
        const WinoFLayout L = wino_filter_layout(N, H, W, Cin, Cout);
        // the forward pass's transformed input is reusable only when that pass ran Winograd on the same tile
        if (wino_v && !(wino_ok(N, H, W, Cin, Cout, KH, KW, stride, pad, PASS_FWD) && wino_geom(N, H, W, PASS_FWD).m == L.g.m)) wino_v = nullptr;
        if (wino_w && mrcnn_conv2d_winograd_w_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad) == 0) wino_w = nullptr;
        float *bias_part = (float *)((char *)ws + L.total);
        int bias_rows = 0;
        if (int e = wino_bwd_filter(x, gy, gw, N, H, W, Cin, Cout, accumulate, ws, st, wino_v, wino_w, gbias ? bias_part : nullptr, &bias_rows)) return e;
        if (gbias) {
            int nrows = bias_rows;          // partial column sums from the gy transform, or a separate pass over gy
            if (nrows == 0) {
                const int P = N * p.Ho * p.Wo;
                const ColPlan cp = col_plan(P, Cout);
                if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_partial, dim3(cp.nblk), dim3(256), 0, st, gy, bias_part, P, Cout, cp.G, cp.RPI, cp.rows_per_blk);
                MRCNN_LAUNCH_CHECK();
                nrows = cp.nblk;
            }
            if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_final, dim3(mrcnn::cdiv(Cout, CSF_CH)), dim3(CSF_CH * CSF_SL), 0, st, bias_part, gbias, nrows, Cout, accumulate);
            MRCNN_LAUNCH_CHECK();
        }
        return 0;
    }
    filter_plan(p, p.ksplit, p.kchunk);
    const size_t wcount = (size_t)Cout * KH * KW * Cin;
    float *slabs = (float *)ws;
    float *bias_part = (float *)ws + wcount * p.ksplit;
    const bool use_slabs = p.ksplit > 1 || accumulate;
    p.a = gy; p.b = x; p.c = use_slabs ? slabs : gw;
    p.bytes_a = (unsigned)((size_t)N * p.Ho * p.Wo * Cout * 4); p.bytes_b = (unsigned)((size_t)N * H * W * Cin * 4);
    p.M = Cout; p.Ng = p.smallc ? KH * KW * 4 : Cin;
    launch_conv<MODE_BWD_FILTER>(p, (p.smallc ? 1 : KH * KW) * p.ksplit, filter_tile(p), st);
    MRCNN_LAUNCH_CHECK();
    if (use_slabs) {
        const size_t n4 = wcount / 4;
        if (!(g_debug_skip & 10)) hipLaunchKernelGGL(k_sum_slabs, dim3(mrcnn::cdiv(n4, 256)), dim3(256), 0, st, slabs, gw, n4, p.ksplit, accumulate);
        MRCNN_LAUNCH_CHECK();
    }
    if (gbias) {
        const int P = N * p.Ho * p.Wo;
        const ColPlan cp = col_plan(P, Cout);
        if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_partial, dim3(cp.nblk), dim3(256), 0, st, gy, bias_part, P, Cout, cp.G, cp.RPI, cp.rows_per_blk);
        MRCNN_LAUNCH_CHECK();
        if (!(g_debug_skip & 2)) hipLaunchKernelGGL(k_colsum_final, dim3(mrcnn::cdiv(Cout, CSF_CH)), dim3(CSF_CH * CSF_SL), 0, st, bias_part, gbias, cp.nblk, Cout, accumulate);
        MRCNN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mrcnn_conv2d_bwd_filter_inbn_f32(const float *x, const float *in_gamma, const float *in_beta, const float *in_mean,
                                                const float *in_invstd, const float *gy, float *gw, int N, int H, int W, int Cin, int Cout, int KH,
                                                int KW, int stride, int pad, int accumulate, const float *wino_v, void *ws, size_t ws_bytes,
                                                void *stream) {
    if (int e = inbn_args(in_gamma, in_beta, in_mean, in_invstd, N, H, W, Cin, Cout, KH, KW, stride, pad, "conv2d_bwd_filter_inbn")) return e;
    t_inbn = InBN{in_gamma, in_beta, in_mean, in_invstd};
    const int rc = mrcnn_conv2d_bwd_filter_f32(x, gy, gw, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, accumulate, wino_v, nullptr, ws, ws_bytes, stream);
    t_inbn = InBN{};
    return rc;
}

// HBM-bound layer kernels of the ResNet-50-FPN backbone and heads for gfx950 (NHWC fp32):
// training-mode BatchNorm (+residual, +ReLU) forward/backward, ReLU backward, add, 2x2/2 max-pool
// (cover_all), nearest-2x upsample + add, strided-scatter (backward of a subsampling 1x1 conv),
// pixel-shuffle for the 2x2/2 deconvolution, and the fused MomentumSGD + WeightDecay update.
//
// Replaces the cuDNN/CuPy kernels Chainer runs for
//   chainer_maskrcnn/model/extractor/feature_pyramid_network.py:48-68 (bn/relu/max_pooling_2d/
//   unpooling_2d/add inside ResNet50Layers and the top-down pathway),
//   chainer_maskrcnn/model/head/fpn_roi_mask_head.py:65-83 (ReLU, Deconvolution2D data movement),
//   train.py:107-109 (MomentumSGD + WeightDecay hook).
//
// Every kernel streams float4 (16 B/lane), each tensor byte is read/written once per kernel, and
// all reductions are two-stage with a fixed summation order (bit-reproducible, no float atomics).
// Roofline: HBM; algorithmic bytes are stated per kernel.
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// streamed-once read (the LAST reader of a big tensor in its pass): non-temporal, does not displace lines other kernels re-read
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float *p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }

// (n, h, w, c) of element i of an (N, Hd, Wd, C4) float4 grid.  32-bit unsigned arithmetic whenever the grid has fewer than 2^32
// elements (always, in practice): the size_t divisions this replaces compile to ~100 instructions of software division EACH
// (k_pixel_shuffle moved 16 B per ~400 instructions: 0.9 TB/s).
struct Nhwc4 { int n, h, w, c; };
__device__ __forceinline__ Nhwc4 split_nhwc4(size_t i, int C4, int Wd, int Hd, bool wide) {
    Nhwc4 r;
    if (!wide) {
        const unsigned u = (unsigned)i, q = u / (unsigned)C4, q2 = q / (unsigned)Wd, q3 = q2 / (unsigned)Hd;
        r.c = (int)(u - q * (unsigned)C4); r.w = (int)(q - q2 * (unsigned)Wd); r.h = (int)(q2 - q3 * (unsigned)Hd); r.n = (int)q3;
    } else {
        const size_t q = i / C4, q2 = q / Wd, q3 = q2 / Hd;
        r.c = (int)(i - q * C4); r.w = (int)(q - q2 * Wd); r.h = (int)(q2 - q3 * Hd); r.n = (int)q3;
    }
    return r;
}

inline int ew_grid(size_t n4) { return (int)std::min<size_t>((n4 + NT - 1) / NT, 256 * 16); }

// ---------------------------------------------------------------------------------------------
// Channel-wise reductions over a (P, C) row-major matrix.  Thread t owns channel group
// cg = t % G (G = min(C/4, 256)) and rows t/G, t/G + RPI, ...; a block owns a contiguous chunk of
// rows.  Partials go to ws[blk][C] (float4 per group), a finalize kernel sums them in order.
// ---------------------------------------------------------------------------------------------
int g_red_cap = 1024;         // most row blocks (= partial rows) of a channel-wise reduction (measurement knob: mrcnn_debug_bn_plan)
struct RedPlan {
    int G, RPI, nblk, rows_per_blk;
};
RedPlan red_plan(int P, int C) {
    RedPlan r;
    const int C4 = C / 4;
    r.G = std::min(C4, NT);
    r.RPI = NT / r.G;
    long long want = ((long long)P * C4 + (long long)NT * 16 - 1) / ((long long)NT * 16);
    r.nblk = (int)std::max(1ll, std::min(want, (long long)g_red_cap));
    r.rows_per_blk = (int)(((long long)P + r.nblk - 1) / r.nblk);
    r.rows_per_blk = (r.rows_per_blk + r.RPI - 1) / r.RPI * r.RPI;
    r.nblk = (P + r.rows_per_blk - 1) / r.rows_per_blk;
    return r;
}

// BN forward statistics: per block, per channel: sum(x - K), sum((x - K)^2) with K = x[0][c]
// (shifted sums: no catastrophic cancellation when |mean| >> std).
__global__ __launch_bounds__(NT) void k_bn_stats_partial(const float *__restrict__ x, int P, int C, int G, int RPI,
                                                         int rows_per_blk, float *__restrict__ part) {
    __shared__ float4 s1[NT], s2[NT];
    const int t = threadIdx.x, cg0 = t % G, rr = t / G;
    const int C4 = C / 4;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(P, r0 + rows_per_blk);
    for (int cg = cg0; cg < C4; cg += G) {
        const float4 K = ld4(x + cg * 4);
        float4 a = f4(0.f), b = f4(0.f);
        auto acc = [&](const float4 v) {
            const float dx = v.x - K.x, dy = v.y - K.y, dz = v.z - K.z, dw = v.w - K.w;
            a.x += dx; a.y += dy; a.z += dz; a.w += dw;
            b.x = fmaf(dx, dx, b.x); b.y = fmaf(dy, dy, b.y); b.z = fmaf(dz, dz, b.z); b.w = fmaf(dw, dw, b.w);
        };
        int r = r0 + rr;
        for (; r + 3 * RPI < r1; r += 4 * RPI) {         // four rows in flight, accumulated in row order
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ld4(x + (size_t)(r + j * RPI) * C + cg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc(v[j]);
        }
        for (; r < r1; r += RPI) acc(ld4(x + (size_t)r * C + cg * 4));
        s1[t] = a; s2[t] = b;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < RPI; ++k) {
                const float4 p = s1[k * G + cg0], q = s2[k * G + cg0];
                a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
                b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
            }
            st4(part + ((size_t)blockIdx.x * 2 * C) + cg * 4, a);
            st4(part + ((size_t)blockIdx.x * 2 * C) + C + cg * 4, b);
        }
        __syncthreads();
    }
}

// Sum of the per-block partials (nblk rows of (2, C) floats: sums, sums of squares) of the block's 4 * QUADS channels by a
// 256-thread block, in double.  These kernels are pure latency on the step's critical chain (a convolution's statistics must be
// final before its BatchNorm can be applied; measured round 5 by leaving them out: 0.70 ms of a 20.7-ms step for 106 launches),
// so the rows are spread over as many loads in flight as the block has: thread = (row slice, channel quad), slice s adds rows
// s, s + SLICES, ... (float4 loads, four rows = eight loads in flight, row order), the slice sums of a channel are then added
// in slice order in two levels (fixed order => bit-reproducible).  QUADS = 4 (64 slices x 16 channels: 64-byte runs per partial
// row) is the shipped layout; QUADS = 1 (256 slices of one quad: two rounds of loads instead of eight for the 2048 partial rows
// of a res2 layer) was measured in round 5 and is SLOWER on every shape (tools/bn_plan_ab.py: 0.97 against 0.93 ms per step
// forward, 1.84 against 1.78 backward - four times the workgroups, each touching a 16-byte piece of every 128-byte line),
// as is a higher cap on the reduction passes' row blocks (2048, 4096: +0.02 ms).  Still 256 threads, not 1024: a workgroup
// of 16 waves cannot start until a whole CU has drained when these run beside GEMMs of another stream.
// Returns the sums in (a, b) of the threads with threadIdx.x < 4 * QUADS (channel blockIdx.x * 4 * QUADS + threadIdx.x).
constexpr int FIN_THREADS = 256;
int g_fin_quads = 4;          // channel quads per finalisation block (measurement knob: mrcnn_debug_bn_plan)
// Rows of `part` are row_floats floats apart (2 C for the (sums, second statistic) rows of one BatchNorm); the second statistic sits off_b
// floats into the row (C) - the pair kernels below keep three statistics per row and read the third with off_b = 2 C.
template <int QUADS>
__device__ __forceinline__ void reduce_partials(const float *__restrict__ part, int nblk, int C, double &a, double &b, int row_floats = 0,
                                                int off_b = 0) {
    if (row_floats == 0) { row_floats = 2 * C; off_b = C; }
    constexpr int CH = 4 * QUADS, SLICES = FIN_THREADS / QUADS, L2N = SLICES / 16;
    __shared__ double ra[SLICES][CH], rb[SLICES][CH];
    __shared__ double ra2[16][CH], rb2[16][CH];
    const int quad = threadIdx.x % QUADS, slice = threadIdx.x / QUADS;
    const int c0 = blockIdx.x * CH + quad * 4;
    double sa4[4] = {0.0, 0.0, 0.0, 0.0}, sb4[4] = {0.0, 0.0, 0.0, 0.0};
    if (c0 < C) {           // C is a multiple of 4: a quad is inside or outside as a whole
        int k = slice;
        for (; k + 3 * SLICES < nblk; k += 4 * SLICES) {
            float4 va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[j] = ld4(part + (size_t)(k + j * SLICES) * row_floats + c0);
                vb[j] = ld4(part + (size_t)(k + j * SLICES) * row_floats + off_b + c0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sa4[0] += (double)va[j].x; sa4[1] += (double)va[j].y; sa4[2] += (double)va[j].z; sa4[3] += (double)va[j].w;
                sb4[0] += (double)vb[j].x; sb4[1] += (double)vb[j].y; sb4[2] += (double)vb[j].z; sb4[3] += (double)vb[j].w;
            }
        }
        for (; k < nblk; k += SLICES) {
            const float4 va = ld4(part + (size_t)k * row_floats + c0), vb = ld4(part + (size_t)k * row_floats + off_b + c0);
            sa4[0] += (double)va.x; sa4[1] += (double)va.y; sa4[2] += (double)va.z; sa4[3] += (double)va.w;
            sb4[0] += (double)vb.x; sb4[1] += (double)vb.y; sb4[2] += (double)vb.z; sb4[3] += (double)vb.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { ra[slice][quad * 4 + e] = sa4[e]; rb[slice][quad * 4 + e] = sb4[e]; }
    __syncthreads();
    if (threadIdx.x < 16 * CH) {            // 16 groups of L2N consecutive slices per channel
        const int ch = threadIdx.x % CH, g = threadIdx.x / CH;
        double x = 0.0, y = 0.0;
#pragma unroll 4
        for (int k = 0; k < L2N; ++k) { x += ra[g * L2N + k][ch]; y += rb[g * L2N + k][ch]; }
        ra2[g][ch] = x; rb2[g][ch] = y;
    }
    __syncthreads();
    a = 0.0; b = 0.0;
    if (threadIdx.x < CH) {
#pragma unroll
        for (int g = 0; g < 16; ++g) { a += ra2[g][threadIdx.x]; b += rb2[g][threadIdx.x]; }
    }
}

template <int QUADS>
__global__ __launch_bounds__(FIN_THREADS) void k_bn_stats_final(const float *__restrict__ x, const float *__restrict__ part,
                                                         int nblk, int P, int C, float eps, float decay, int shifted,
                                                         float *__restrict__ mean, float *__restrict__ invstd,
                                                         float *__restrict__ run_mean, float *__restrict__ run_var) {
    constexpr int FIN_CH = 4 * QUADS, FIN_SLICES = FIN_THREADS / FIN_CH;    // the thread layout of this kernel's own passes
    __shared__ double sa[FIN_SLICES][FIN_CH], sb[FIN_SLICES][FIN_CH];
    __shared__ double smean[FIN_CH];
    __shared__ int sredo;
    const int cl = threadIdx.x % FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl, slice = threadIdx.x / FIN_CH;
    double a, b;
    if (threadIdx.x == 0) sredo = 0;
    reduce_partials<QUADS>(part, nblk, C, a, b);        // (contains barriers: sredo is visible below)
    // shifted: the partials are sums of (x - K), K = x[0][c] (k_bn_stats_partial); else plain sums (the convolution epilogue)
    double ms = a / P, var = b / P - ms * ms;
    double m = (shifted ? (double)x[c < C ? c : 0] : 0.0) + ms;
    // Un-shifted float32 partials lose var's digits to cancellation once mean^2 >> var (relative error ~1e-7 mean^2 / var:
    // 9e-6 at |mean|/std = 34, measured) - e.g. a pretrained backbone's first BatchNorms.  Past |mean|/std = 32 the block
    // recomputes its FIN_CH channels exactly: sums of (x - mean) and (x - mean)^2 over all P rows in double (a second pass
    // over x for these channels only; never taken on the step's own activations).
    if (slice == 0 && c < C && !shifted && var < ms * ms * (1.0 / 1024.0)) atomicOr(&sredo, 1);
    if (slice == 0) smean[cl] = m;
    __syncthreads();
    if (sredo) {            // block-uniform
        const double mc = smean[cl];
        double d1 = 0.0, d2 = 0.0;
        if (c < C) {
            // eight rows in flight per thread, added in row order (one dependent load per iteration made this rare path
            // - a pretrained backbone's first BatchNorms, a dead channel - a step-time cliff of tens of ms at P ~ 5e5)
            int r = slice;
            for (; r + 7 * FIN_SLICES < P; r += 8 * FIN_SLICES) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = x[(size_t)(r + j * FIN_SLICES) * C + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const double d = (double)v[j] - mc; d1 += d; d2 += d * d; }
            }
            for (; r < P; r += FIN_SLICES) {
                const double d = (double)x[(size_t)r * C + c] - mc;
                d1 += d; d2 += d * d;
            }
        }
        __syncthreads();
        sa[slice][cl] = d1; sb[slice][cl] = d2;
        __syncthreads();
        if (slice == 0) {
            for (int k = 1; k < FIN_SLICES; ++k) { d1 += sa[k][cl]; d2 += sb[k][cl]; }
            const double dm = d1 / P;
            m = mc + dm;
            var = d2 / P - dm * dm;
        }
    }
    if (slice != 0 || c >= C) return;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, v = (float)var;
    mean[c] = mf;
    invstd[c] = 1.0f / sqrtf(v + eps);
    if (run_mean) run_mean[c] = decay * run_mean[c] + (1.0f - decay) * mf;
    if (run_var) {
        const float adj = (float)P / (float)max(P - 1, 1);
        run_var[c] = decay * run_var[c] + (1.0f - decay) * v * adj;
    }
}
void launch_stats_final(hipStream_t st, const float *x, const float *part, int nblk, int P, int C, float eps, float decay, int shifted,
                        float *mean, float *invstd, float *run_mean, float *run_var) {
    if (g_fin_quads == 1)
        hipLaunchKernelGGL(k_bn_stats_final<1>, dim3(mrcnn::cdiv(C, 4)), dim3(FIN_THREADS), 0, st, x, part, nblk, P, C, eps, decay, shifted, mean,
                           invstd, run_mean, run_var);
    else
        hipLaunchKernelGGL(k_bn_stats_final<4>, dim3(mrcnn::cdiv(C, 16)), dim3(FIN_THREADS), 0, st, x, part, nblk, P, C, eps, decay, shifted, mean,
                           invstd, run_mean, run_var);
}

// y = gamma*(x-mean)*invstd + beta (+ residual) (ReLU).  Bytes: 4*P*C*(2 or 3).
__global__ __launch_bounds__(NT) void k_bn_apply(const float *__restrict__ x, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, const float *__restrict__ mean,
                                                 const float *__restrict__ invstd, const float *__restrict__ res,
                                                 float *__restrict__ y, size_t n4, int C4, int relu) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C4) * 4;
        const float4 v = ld4s(x + i * 4), g = ld4(gamma + c), b = ld4(beta + c), m = ld4(mean + c), s = ld4(invstd + c);
        float4 o;
        o.x = g.x * ((v.x - m.x) * s.x) + b.x;
        o.y = g.y * ((v.y - m.y) * s.y) + b.y;
        o.z = g.z * ((v.z - m.z) * s.z) + b.z;
        o.w = g.w * ((v.w - m.w) * s.w) + b.w;
        if (res) {
            const float4 r = ld4s(res + i * 4);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4(y + i * 4, o);
    }
}

// Inference-mode BN (running statistics): y = gamma*(x-avg_mean)/sqrt(avg_var+eps) + beta (+ residual) (ReLU).
__global__ __launch_bounds__(NT) void k_bn_infer(const float *__restrict__ x, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, const float *__restrict__ mean,
                                                 const float *__restrict__ var, const float *__restrict__ res,
                                                 float *__restrict__ y, size_t n4, int C4, float eps, int relu) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C4) * 4;
        const float4 v = ld4(x + i * 4), g = ld4(gamma + c), b = ld4(beta + c), m = ld4(mean + c), va = ld4(var + c);
        float4 o;
        o.x = g.x * ((v.x - m.x) * (1.0f / sqrtf(va.x + eps))) + b.x;
        o.y = g.y * ((v.y - m.y) * (1.0f / sqrtf(va.y + eps))) + b.y;
        o.z = g.z * ((v.z - m.z) * (1.0f / sqrtf(va.z + eps))) + b.z;
        o.w = g.w * ((v.w - m.w) * (1.0f / sqrtf(va.w + eps))) + b.w;
        if (res) {
            const float4 r = ld4(res + i * 4);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4(y + i * 4, o);
    }
}

// BN backward reductions: per channel sum(dz), sum(dz*xhat) with dz = relu ? gy*(y>0) : gy.
// relu: 0 none, 1 mask = (y > 0) read from memory, 2 mask = (gamma*xhat + beta > 0) recomputed from x with the forward's
// exact expression (bitwise the same y; only for BN + ReLU without a residual) - one HBM stream less.
__global__ __launch_bounds__(NT) void k_bn_bwd_partial(const float *__restrict__ gy, const float *__restrict__ x,
                                                       const float *__restrict__ y, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, int P, int C, int G, int RPI,
                                                       int rows_per_blk, int relu, float *__restrict__ part) {
    __shared__ float4 s1[NT], s2[NT];
    const int t = threadIdx.x, cg0 = t % G, rr = t / G;
    const int C4 = C / 4;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(P, r0 + rows_per_blk);
    for (int cg = cg0; cg < C4; cg += G) {
        const float4 m = ld4(mean + cg * 4), s = ld4(invstd + cg * 4);
        const float4 ga = relu == 2 ? ld4(gamma + cg * 4) : f4(0.f), be = relu == 2 ? ld4(beta + cg * 4) : f4(0.f);
        float4 a = f4(0.f), b = f4(0.f);
        auto acc = [&](float4 g, const float4 v, float4 yy) {
            if (relu == 2) {
                yy.x = ga.x * ((v.x - m.x) * s.x) + be.x; yy.y = ga.y * ((v.y - m.y) * s.y) + be.y;
                yy.z = ga.z * ((v.z - m.z) * s.z) + be.z; yy.w = ga.w * ((v.w - m.w) * s.w) + be.w;
            }
            if (relu) {
                g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
            }
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
            b.x = fmaf(g.x, (v.x - m.x) * s.x, b.x); b.y = fmaf(g.y, (v.y - m.y) * s.y, b.y);
            b.z = fmaf(g.z, (v.z - m.z) * s.z, b.z); b.w = fmaf(g.w, (v.w - m.w) * s.w, b.w);
        };
        const float *yr = relu == 1 ? y : gy;             // only mode 1 reads the third stream
        int r = r0 + rr;
        // (four rows in flight were tried in round 5: the in-step launches went 1012 -> 1392 us per step - more registers, fewer waves)
        for (; r + RPI < r1; r += 2 * RPI) {              // two rows (six loads) in flight, accumulated in row order
            const size_t o0 = (size_t)r * C + cg * 4, o1 = (size_t)(r + RPI) * C + cg * 4;
            const float4 g0 = ld4(gy + o0), v0 = ld4(x + o0), g1 = ld4(gy + o1), v1 = ld4(x + o1);
            float4 y0 = g0, y1 = g1;
            if (relu == 1) { y0 = ld4(yr + o0); y1 = ld4(yr + o1); }
            acc(g0, v0, y0);
            acc(g1, v1, y1);
        }
        for (; r < r1; r += RPI) {
            const size_t o = (size_t)r * C + cg * 4;
            const float4 g = ld4(gy + o), v = ld4(x + o);
            acc(g, v, relu == 1 ? ld4(yr + o) : g);
        }
        s1[t] = a; s2[t] = b;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < RPI; ++k) {
                const float4 p = s1[k * G + cg0], q = s2[k * G + cg0];
                a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
                b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
            }
            st4(part + ((size_t)blockIdx.x * 2 * C) + cg * 4, a);
            st4(part + ((size_t)blockIdx.x * 2 * C) + C + cg * 4, b);
        }
        __syncthreads();
    }
}

template <int QUADS>
__global__ __launch_bounds__(FIN_THREADS) void k_bn_bwd_final(const float *__restrict__ part, int nblk, int C,
                                                       float *__restrict__ gbeta, float *__restrict__ ggamma) {
    const int c = blockIdx.x * 4 * QUADS + threadIdx.x;
    double a, b;
    reduce_partials<QUADS>(part, nblk, C, a, b);
    if (threadIdx.x >= 4 * QUADS || c >= C) return;
    gbeta[c] = (float)a;
    ggamma[c] = (float)b;
}
void launch_bwd_final(hipStream_t st, const float *part, int nblk, int C, float *gbeta, float *ggamma) {
    if (g_fin_quads == 1) hipLaunchKernelGGL(k_bn_bwd_final<1>, dim3(mrcnn::cdiv(C, 4)), dim3(FIN_THREADS), 0, st, part, nblk, C, gbeta, ggamma);
    else hipLaunchKernelGGL(k_bn_bwd_final<4>, dim3(mrcnn::cdiv(C, 16)), dim3(FIN_THREADS), 0, st, part, nblk, C, gbeta, ggamma);
}

// gx = gamma*invstd*(dz - gbeta/P - xhat*ggamma/P); optionally gres = dz.
__global__ __launch_bounds__(NT) void k_bn_bwd_apply(const float *__restrict__ gy, const float *__restrict__ x,
                                                     const float *__restrict__ y, const float *__restrict__ gamma,
                                                     const float *__restrict__ mean, const float *__restrict__ invstd,
                                                     const float *__restrict__ gbeta, const float *__restrict__ ggamma,
                                                     float *__restrict__ gx, float *__restrict__ gres, size_t n4, int C4,
                                                     float invP, int relu, const float *__restrict__ beta) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C4) * 4;
        float4 g = ld4s(gy + i * 4);
        const float4 v = ld4s(x + i * 4);
        const float4 ga = ld4(gamma + c), m = ld4(mean + c), s = ld4(invstd + c), gb = ld4(gbeta + c), gg = ld4(ggamma + c);
        if (relu) {
            float4 yy;
            if (relu == 2) {
                const float4 be = ld4(beta + c);
                yy.x = ga.x * ((v.x - m.x) * s.x) + be.x; yy.y = ga.y * ((v.y - m.y) * s.y) + be.y;
                yy.z = ga.z * ((v.z - m.z) * s.z) + be.z; yy.w = ga.w * ((v.w - m.w) * s.w) + be.w;
            } else {
                yy = ld4s(y + i * 4);
            }
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        float4 o;
        o.x = ga.x * s.x * (g.x - gb.x * invP - ((v.x - m.x) * s.x) * (gg.x * invP));
        o.y = ga.y * s.y * (g.y - gb.y * invP - ((v.y - m.y) * s.y) * (gg.y * invP));
        o.z = ga.z * s.z * (g.z - gb.z * invP - ((v.z - m.z) * s.z) * (gg.z * invP));
        o.w = ga.w * s.w * (g.w - gb.w * invP - ((v.w - m.w) * s.w) * (gg.w * invP));
        st4(gx + i * 4, o);
        if (gres) st4(gres + i * 4, g);
    }
}

// ---- (r6) Two BatchNorms that meet in one residual sum: the main branch and the projection shortcut of a ResNet bottleneck
// (extractor/feature_pyramid_network.py:48-66; Chainer's BottleneckA: relu(bn3(conv3(.)) + bn4(conv4(x)))).  Run layer by layer, the
// shortcut's BatchNorm output r crosses HBM twice in the forward pass (written by its apply kernel, read as the residual) and the masked
// gradient g_r three times in the backward pass (written beside g_h3, read by both kernels of the shortcut's BatchNorm backward):
// 5 x 252 MB per step at 2 x 1024^2.  The pair kernels read the two pre-BatchNorm tensors side by side and never materialise r / g_r.
// Every value is computed by the expressions of the single-layer kernels in the same order (r first, then o = bn_a + r; the same row
// blocks, the same order of the partial sums): the results are the same bits as the layer-by-layer sequence.
__global__ __launch_bounds__(NT) void k_bn_apply2(const float *__restrict__ xa, const float *__restrict__ gamma_a, const float *__restrict__ beta_a,
                                                  const float *__restrict__ mean_a, const float *__restrict__ invstd_a,
                                                  const float *__restrict__ xb, const float *__restrict__ gamma_b, const float *__restrict__ beta_b,
                                                  const float *__restrict__ mean_b, const float *__restrict__ invstd_b, float *__restrict__ y,
                                                  size_t n4, int C4) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C4) * 4;
        const float4 va = ld4s(xa + i * 4), vb = ld4s(xb + i * 4);
        const float4 ga = ld4(gamma_a + c), ba = ld4(beta_a + c), ma = ld4(mean_a + c), sa = ld4(invstd_a + c);
        const float4 gb = ld4(gamma_b + c), bb = ld4(beta_b + c), mb = ld4(mean_b + c), sb = ld4(invstd_b + c);
        float4 r, o;
        r.x = gb.x * ((vb.x - mb.x) * sb.x) + bb.x; r.y = gb.y * ((vb.y - mb.y) * sb.y) + bb.y;
        r.z = gb.z * ((vb.z - mb.z) * sb.z) + bb.z; r.w = gb.w * ((vb.w - mb.w) * sb.w) + bb.w;
        o.x = ga.x * ((va.x - ma.x) * sa.x) + ba.x; o.y = ga.y * ((va.y - ma.y) * sa.y) + ba.y;
        o.z = ga.z * ((va.z - ma.z) * sa.z) + ba.z; o.w = ga.w * ((va.w - ma.w) * sa.w) + ba.w;
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        st4(y + i * 4, o);
    }
}

// partial rows (nblk, 3, C): sum(dz), sum(dz * xhat_a), sum(dz * xhat_b); dz = y ? gy * (y > 0) : gy
__global__ __launch_bounds__(NT) void k_bn_bwd_partial2(const float *__restrict__ gy, const float *__restrict__ y, const float *__restrict__ xa,
                                                        const float *__restrict__ xb, const float *__restrict__ mean_a,
                                                        const float *__restrict__ invstd_a, const float *__restrict__ mean_b,
                                                        const float *__restrict__ invstd_b, int P, int C, int G, int RPI, int rows_per_blk,
                                                        float *__restrict__ part) {
    __shared__ float4 s1[NT], s2[NT], s3[NT];
    const int t = threadIdx.x, cg0 = t % G, rr = t / G;
    const int C4 = C / 4;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(P, r0 + rows_per_blk);
    for (int cg = cg0; cg < C4; cg += G) {
        const float4 ma = ld4(mean_a + cg * 4), sa = ld4(invstd_a + cg * 4), mb = ld4(mean_b + cg * 4), sb = ld4(invstd_b + cg * 4);
        float4 a = f4(0.f), b = f4(0.f), c = f4(0.f);
        auto acc = [&](float4 g, const float4 va, const float4 vb, const float4 yy) {
            if (y) {
                g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
            }
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
            b.x = fmaf(g.x, (va.x - ma.x) * sa.x, b.x); b.y = fmaf(g.y, (va.y - ma.y) * sa.y, b.y);
            b.z = fmaf(g.z, (va.z - ma.z) * sa.z, b.z); b.w = fmaf(g.w, (va.w - ma.w) * sa.w, b.w);
            c.x = fmaf(g.x, (vb.x - mb.x) * sb.x, c.x); c.y = fmaf(g.y, (vb.y - mb.y) * sb.y, c.y);
            c.z = fmaf(g.z, (vb.z - mb.z) * sb.z, c.z); c.w = fmaf(g.w, (vb.w - mb.w) * sb.w, c.w);
        };
        const float *yr = y ? y : gy;
        int r = r0 + rr;
        for (; r + RPI < r1; r += 2 * RPI) {              // two rows (eight loads) in flight, accumulated in row order
            const size_t o0 = (size_t)r * C + cg * 4, o1 = (size_t)(r + RPI) * C + cg * 4;
            const float4 g0 = ld4(gy + o0), a0 = ld4(xa + o0), b0 = ld4(xb + o0), g1 = ld4(gy + o1), a1 = ld4(xa + o1), b1 = ld4(xb + o1);
            float4 y0 = g0, y1 = g1;
            if (y) { y0 = ld4(yr + o0); y1 = ld4(yr + o1); }
            acc(g0, a0, b0, y0);
            acc(g1, a1, b1, y1);
        }
        for (; r < r1; r += RPI) {
            const size_t o = (size_t)r * C + cg * 4;
            const float4 g = ld4(gy + o);
            acc(g, ld4(xa + o), ld4(xb + o), y ? ld4(yr + o) : g);
        }
        s1[t] = a; s2[t] = b; s3[t] = c;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < RPI; ++k) {
                const float4 p = s1[k * G + cg0], q = s2[k * G + cg0], u = s3[k * G + cg0];
                a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
                b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
                c.x += u.x; c.y += u.y; c.z += u.z; c.w += u.w;
            }
            st4(part + ((size_t)blockIdx.x * 3 * C) + cg * 4, a);
            st4(part + ((size_t)blockIdx.x * 3 * C) + C + cg * 4, b);
            st4(part + ((size_t)blockIdx.x * 3 * C) + 2 * C + cg * 4, c);
        }
        __syncthreads();
    }
}

// gbeta_a = gbeta_b = sum(dz), ggamma_a = sum(dz * xhat_a), ggamma_b = sum(dz * xhat_b): the single-layer reduction, twice
template <int QUADS>
__global__ __launch_bounds__(FIN_THREADS) void k_bn_bwd_final2(const float *__restrict__ part, int nblk, int C, float *__restrict__ gbeta_a,
                                                               float *__restrict__ ggamma_a, float *__restrict__ gbeta_b,
                                                               float *__restrict__ ggamma_b) {
    const int c = blockIdx.x * 4 * QUADS + threadIdx.x;
    double a, b, a2, b2;
    reduce_partials<QUADS>(part, nblk, C, a, b, 3 * C, C);
    __syncthreads();
    reduce_partials<QUADS>(part, nblk, C, a2, b2, 3 * C, 2 * C);
    if (threadIdx.x >= 4 * QUADS || c >= C) return;
    gbeta_a[c] = (float)a; ggamma_a[c] = (float)b;
    gbeta_b[c] = (float)a2; ggamma_b[c] = (float)b2;
}

__global__ __launch_bounds__(NT) void k_bn_bwd_apply2(const float *__restrict__ gy, const float *__restrict__ y, const float *__restrict__ xa,
                                                      const float *__restrict__ xb, const float *__restrict__ gamma_a,
                                                      const float *__restrict__ mean_a, const float *__restrict__ invstd_a,
                                                      const float *__restrict__ gbeta_a, const float *__restrict__ ggamma_a,
                                                      const float *__restrict__ gamma_b, const float *__restrict__ mean_b,
                                                      const float *__restrict__ invstd_b, const float *__restrict__ gbeta_b,
                                                      const float *__restrict__ ggamma_b, float *__restrict__ gxa, float *__restrict__ gxb,
                                                      size_t n4, int C4, float invP) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C4) * 4;
        float4 g = ld4s(gy + i * 4);
        const float4 va = ld4s(xa + i * 4), vb = ld4s(xb + i * 4);
        if (y) {
            const float4 yy = ld4s(y + i * 4);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        {
            const float4 ga = ld4(gamma_a + c), m = ld4(mean_a + c), s = ld4(invstd_a + c), gb = ld4(gbeta_a + c), gg = ld4(ggamma_a + c);
            float4 o;
            o.x = ga.x * s.x * (g.x - gb.x * invP - ((va.x - m.x) * s.x) * (gg.x * invP));
            o.y = ga.y * s.y * (g.y - gb.y * invP - ((va.y - m.y) * s.y) * (gg.y * invP));
            o.z = ga.z * s.z * (g.z - gb.z * invP - ((va.z - m.z) * s.z) * (gg.z * invP));
            o.w = ga.w * s.w * (g.w - gb.w * invP - ((va.w - m.w) * s.w) * (gg.w * invP));
            st4(gxa + i * 4, o);
        }
        {
            const float4 ga = ld4(gamma_b + c), m = ld4(mean_b + c), s = ld4(invstd_b + c), gb = ld4(gbeta_b + c), gg = ld4(ggamma_b + c);
            float4 o;
            o.x = ga.x * s.x * (g.x - gb.x * invP - ((vb.x - m.x) * s.x) * (gg.x * invP));
            o.y = ga.y * s.y * (g.y - gb.y * invP - ((vb.y - m.y) * s.y) * (gg.y * invP));
            o.z = ga.z * s.z * (g.z - gb.z * invP - ((vb.z - m.z) * s.z) * (gg.z * invP));
            o.w = ga.w * s.w * (g.w - gb.w * invP - ((vb.w - m.w) * s.w) * (gg.w * invP));
            st4(gxb + i * 4, o);
        }
    }
}

__global__ __launch_bounds__(NT) void k_relu_bwd(const float *__restrict__ gy, const float *__restrict__ y,
                                                 float *__restrict__ gx, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        float4 g = ld4(gy + i * 4);
        const float4 yy = ld4(y + i * 4);
        g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
        g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        st4(gx + i * 4, g);
    }
}

__global__ __launch_bounds__(NT) void k_add(const float *__restrict__ a, const float *__restrict__ b,
                                            float *__restrict__ o, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const float4 p = ld4(a + i * 4), q = ld4(b + i * 4);
        st4(o + i * 4, make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w));
    }
}

// ---- 2x2 stride-2 max pooling, cover_all (Ho = ceil(H/2)) ------------------------------------
__global__ __launch_bounds__(NT) void k_maxpool_fwd(const float *__restrict__ x, float *__restrict__ y, int N, int H,
                                                    int W, int C4, int Ho, int Wo) {
    const size_t n4 = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, Wo, Ho, n4 > 0xFFFFFFFFull);
        const int c = ix.c, wo = ix.w, ho = ix.h, n = ix.n;
        float4 m = f4(-INFINITY);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int h = 2 * ho + dy, w = 2 * wo + dx;
                if (h < H && w < W) {
                    const float4 v = ld4(x + ((((size_t)n * H + h) * W + w) * C4 + c) * 4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        st4(y + i * 4, m);
    }
}

// ---- legacy variants: 3x3 / stride-2 max pooling with cover_all (C4Backbone's pool1,
//      c4_backbone.py:20: Ho = (H - 3 + 1) / 2 + 1, windows may hang over the bottom / right edge), global average pooling
//      over the pooled RoI (ResnetRoIMaskHead, resnet_roi_mask_head.py:65) and a stand-alone ReLU.  Forward only.
__global__ __launch_bounds__(NT) void k_maxpool3s2_fwd(const float *__restrict__ x, float *__restrict__ y, int N, int H, int W,
                                                       int C4, int Ho, int Wo) {
    const size_t n4 = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, Wo, Ho, n4 > 0xFFFFFFFFull);
        const int c = ix.c, wo = ix.w, ho = ix.h, n = ix.n;
        float4 m = f4(-INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int h = 2 * ho + dy, w = 2 * wo + dx;
                if (h < H && w < W) {
                    const float4 v = ld4(x + ((((size_t)n * H + h) * W + w) * C4 + c) * 4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        st4(y + i * 4, m);
    }
}

// y (R, C) = mean over the P pixels of x (R, P, C): pixels added in order, then one division by P (F.average_pooling_2d
// over the whole map = sum / P).
__global__ __launch_bounds__(NT) void k_global_avg_pool(const float *__restrict__ x, float *__restrict__ y, int R, int P, int C4) {
    const size_t n4 = (size_t)R * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C4), r = (int)(i / C4);
        float4 a = f4(0.f);
        for (int p = 0; p < P; ++p) {
            const float4 v = ld4(x + (((size_t)r * P + p) * C4 + c) * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        const float d = (float)P;
        st4(y + i * 4, make_float4(a.x / d, a.y / d, a.z / d, a.w / d));
    }
}

__global__ __launch_bounds__(NT) void k_relu_fwd(const float *__restrict__ x, float *__restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const float4 v = ld4(x + i * 4);
        st4(y + i * 4, make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)));
    }
}

// gx[cell] = gy[window] if the cell is the FIRST maximum of its window (row-major order), else 0.
// One thread per window: writes all (<=4) cells of the window => gx fully written, no atomics.
__global__ __launch_bounds__(NT) void k_maxpool_bwd(const float *__restrict__ x, const float *__restrict__ gy,
                                                    float *__restrict__ gx, int N, int H, int W, int C4, int Ho, int Wo) {
    const size_t n4 = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, Wo, Ho, n4 > 0xFFFFFFFFull);
        const int c = ix.c, wo = ix.w, ho = ix.h, n = ix.n;
        const float4 g = ld4(gy + i * 4);
        float v[4][4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = 2 * ho + (k >> 1), w = 2 * wo + (k & 1);
            ok[k] = h < H && w < W;
            float4 t = f4(-INFINITY);
            if (ok[k]) t = ld4(x + ((((size_t)n * H + h) * W + w) * C4 + c) * 4);
            v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
        }
        const float gv[4] = {g.x, g.y, g.z, g.w};
        float o[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int arg = 0;
            float best = v[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][e] > best) { best = v[k][e]; arg = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][e] = (k == arg) ? gv[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = 2 * ho + (k >> 1), w = 2 * wo + (k & 1);
            if (ok[k]) st4(gx + ((((size_t)n * H + h) * W + w) * C4 + c) * 4, make_float4(o[k][0], o[k][1], o[k][2], o[k][3]));
        }
    }
}

// ---- nearest 2x upsample (F.unpooling_2d ksize 2, cropped to (H,W)) + lateral add --------------
__global__ __launch_bounds__(NT) void k_upsample_add(const float *__restrict__ top, const float *__restrict__ lat,
                                                     float *__restrict__ out, int N, int H, int W, int Ht, int Wt, int C4) {
    const size_t n4 = (size_t)N * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, W, H, n4 > 0xFFFFFFFFull);
        const int c = ix.c, w = ix.w, h = ix.h, n = ix.n;
        const float4 a = ld4(top + ((((size_t)n * Ht + (h >> 1)) * Wt + (w >> 1)) * C4 + c) * 4);
        const float4 b = ld4(lat + i * 4);
        st4(out + i * 4, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
    }
}

// gtop[n,ht,wt] (+)= sum of the in-bounds 2x2 block of gout.
__global__ __launch_bounds__(NT) void k_upsample_bwd(const float *__restrict__ gout, float *__restrict__ gtop, int N,
                                                     int H, int W, int Ht, int Wt, int C4, int accumulate) {
    const size_t n4 = (size_t)N * Ht * Wt * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, Wt, Ht, n4 > 0xFFFFFFFFull);
        const int c = ix.c, wt = ix.w, ht = ix.h, n = ix.n;
        float4 s = accumulate ? ld4(gtop + i * 4) : f4(0.f);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int h = 2 * ht + dy, w = 2 * wt + dx;
                if (h < H && w < W) {
                    const float4 v = ld4(gout + ((((size_t)n * H + h) * W + w) * C4 + c) * 4);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
        st4(gtop + i * 4, s);
    }
}

// Backward of y[n,ho,wo] = x[n,ho*s,wo*s]: gx fully written (zeros off the lattice) or accumulated.
__global__ __launch_bounds__(NT) void k_subsample_bwd(const float *__restrict__ gsub, float *__restrict__ gx, int N,
                                                      int H, int W, int Ho, int Wo, int C4, int s, int accumulate,
                                                      const float *__restrict__ relu_x) {
    const size_t n4 = (size_t)N * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, W, H, n4 > 0xFFFFFFFFull);
        const int c = ix.c, w = ix.w, h = ix.h, n = ix.n;
        const bool on = (h % s == 0) && (w % s == 0) && (h / s < Ho) && (w / s < Wo);
        if (accumulate) {
            if (!on) continue;
            const float4 g = ld4(gsub + ((((size_t)n * Ho + h / s) * Wo + w / s) * C4 + c) * 4);
            const float4 o = ld4(gx + i * 4);
            float4 v = make_float4(o.x + g.x, o.y + g.y, o.z + g.z, o.w + g.w);
            if (relu_x) {
                const float4 xm = ld4(relu_x + i * 4);
                v.x = xm.x > 0.f ? v.x : 0.f; v.y = xm.y > 0.f ? v.y : 0.f; v.z = xm.z > 0.f ? v.z : 0.f; v.w = xm.w > 0.f ? v.w : 0.f;
            }
            st4(gx + i * 4, v);
        } else {
            float4 g = f4(0.f);
            if (on) {
                g = ld4(gsub + ((((size_t)n * Ho + h / s) * Wo + w / s) * C4 + c) * 4);
                if (relu_x) {
                    const float4 xm = ld4(relu_x + i * 4);
                    g.x = xm.x > 0.f ? g.x : 0.f; g.y = xm.y > 0.f ? g.y : 0.f; g.z = xm.z > 0.f ? g.z : 0.f; g.w = xm.w > 0.f ? g.w : 0.f;
                }
            }
            st4(gx + i * 4, g);
        }
    }
}

// Pixel shuffle of the 2x2/2 deconvolution: t (P=N*H*W, [a][b][Cout]) <-> y (N, 2H, 2W, Cout).
// dir 0: y[n,2h+a,2w+b,:] = t[n,h,w,a,b,:];  dir 1: the inverse gather (backward).
__global__ __launch_bounds__(NT) void k_pixel_shuffle(const float *__restrict__ src, const float *__restrict__ bias,
                                                      float *__restrict__ dst, int N, int H, int W, int C4, int dir) {
    const size_t n4 = (size_t)N * H * W * 4 * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        // i indexes the (N,2H,2W,C) side
        const Nhwc4 ix = split_nhwc4(i, C4, 2 * W, 2 * H, n4 > 0xFFFFFFFFull);
        const int c = ix.c, w2 = ix.w, h2 = ix.h, n = ix.n;
        const size_t j = ((((size_t)n * H + (h2 >> 1)) * W + (w2 >> 1)) * 4 + (h2 & 1) * 2 + (w2 & 1)) * C4 + c;
        if (dir == 0) {
            float4 v = ld4(src + j * 4);
            if (bias) { const float4 b = ld4(bias + c * 4); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            st4(dst + i * 4, v);
        } else st4(dst + j * 4, ld4(src + i * 4));
    }
}


// ---------------------------------------------------------------------------------------------
// Algebraic merge of the mask branch's last two layers.  The reference applies NO non-linearity between the 2x2/2
// deconvolution and the final 1x1 convolution (head/fpn_roi_mask_head.py:83, fpn_roi_keypoint_head.py:93:
// `mask = self.conv2(self.deconv1(mask))`), so their composition is ONE 2x2/2 deconvolution to K2 channels:
//   Wm[ab][k][ci] = sum_o W2[k][o] * Wd[ab][o][ci],   bm[k] = b2[k] + sum_o W2[k][o] * bd[o]
// (4x fewer MACs on the 28x28 maps and no (R,28,28,C) intermediate).  The parameters stay Wd, bd, W2, b2; their
// gradients follow from the merged layer's filter gradient G[ab][k][ci] and bias gradient gb4[ab][k]:
//   gWd[ab][o][ci] = sum_k W2[k][o] * G[ab][k][ci]           gbd[o] = sum_k W2[k][o] * gsum[k]
//   gW2[k][o] = sum_ab sum_ci G[ab][k][ci] * Wd[ab][o][ci] + gsum[k] * bd[o]     gb2[k] = gsum[k] = sum_ab gb4[ab][k]
// Exact in real arithmetic; fixed summation order (bit-reproducible).  All matrices are tiny (<= 1 MB).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_deconv_merge_fwd(const float *__restrict__ wd, const float *__restrict__ bd,
                                                         const float *__restrict__ w2, const float *__restrict__ b2,
                                                         float *__restrict__ wm, float *__restrict__ bm, int C, int Cin,
                                                         int K2, int ld2) {
    const int i = blockIdx.x * NT + threadIdx.x;
    const int total = 4 * K2 * Cin;
    if (i < total) {
        const int ci = i % Cin, k = (i / Cin) % K2, ab = i / (Cin * K2);
        const float *w2r = w2 + (size_t)k * ld2, *wdc = wd + (size_t)ab * C * Cin + ci;
        float a = 0.0f;
        for (int o = 0; o < C; ++o) a = fmaf(w2r[o], wdc[(size_t)o * Cin], a);
        wm[i] = a;
    } else if (i < total + K2) {
        const int k = i - total;
        float a = b2 ? b2[k] : 0.0f;
        for (int o = 0; o < C; ++o) a = fmaf(w2[(size_t)k * ld2 + o], bd[o], a);
        bm[k] = a;
    }
}

__global__ __launch_bounds__(NT) void k_deconv_merge_bwd_wd(const float *__restrict__ G, const float *__restrict__ gb4,
                                                            const float *__restrict__ w2, float *__restrict__ gwd,
                                                            float *__restrict__ gbd, float *__restrict__ gb2, int C,
                                                            int Cin, int K2, int ld2) {
    const int i = blockIdx.x * NT + threadIdx.x;
    const int total = 4 * C * Cin;
    if (i < total) {
        const int ci = i % Cin, o = (i / Cin) % C, ab = i / (Cin * C);
        const float *g = G + (size_t)ab * K2 * Cin + ci;
        float a = 0.0f;
        for (int k = 0; k < K2; ++k) a = fmaf(w2[(size_t)k * ld2 + o], g[(size_t)k * Cin], a);
        gwd[i] = a;
    } else if (i < total + C) {
        const int o = i - total;
        float a = 0.0f;
        for (int k = 0; k < K2; ++k) {
            const float gs = ((gb4[k] + gb4[K2 + k]) + gb4[2 * K2 + k]) + gb4[3 * K2 + k];
            a = fmaf(w2[(size_t)k * ld2 + o], gs, a);
        }
        gbd[o] = a;
    } else if (i < total + C + K2) {
        const int k = i - total - C;
        if (gb2) gb2[k] = ((gb4[k] + gb4[K2 + k]) + gb4[2 * K2 + k]) + gb4[3 * K2 + k];
    }
}

// one wave per (k, o): lanes stride over ci, fixed-order butterfly reduction
__global__ __launch_bounds__(NT) void k_deconv_merge_bwd_w2(const float *__restrict__ G, const float *__restrict__ gb4,
                                                            const float *__restrict__ wd, const float *__restrict__ bd,
                                                            float *__restrict__ gw2, int C, int Cin, int K2, int ld2) {
    const int wv = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= K2 * C) return;
    const int o = wv % C, k = wv / C;
    float a = 0.0f;
    for (int ab = 0; ab < 4; ++ab) {
        const float *g = G + ((size_t)ab * K2 + k) * Cin, *w = wd + ((size_t)ab * C + o) * Cin;
        for (int ci = lane; ci < Cin; ci += 64) a = fmaf(g[ci], w[ci], a);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
    if (lane == 0) {
        const float gs = ((gb4[k] + gb4[K2 + k]) + gb4[2 * K2 + k]) + gb4[3 * K2 + k];
        gw2[(size_t)k * ld2 + o] = fmaf(gs, bd[o], a);
    }
}


// ---------------------------------------------------------------------------------------------
// Device side of the training Transform (train.py:21-37): the host decodes the JPEG and rasterises the polygons, the
// raw uint8 image and masks cross PCIe at their ORIGINAL size (10x fewer bytes than the prepared float32 tensors) and
// are resized here, straight into the (zero-padded) batch tensors.  Interpolation rules = OpenCV's, restated exactly as
// in chainer_maskrcnn/dataset/transforms.py (bit-identical results: same float operations, no contraction):
//   INTER_LINEAR  fx = (float)((dx + 0.5) * (src/dst) - 0.5), sx = floor(fx), clamped taps; horizontal then vertical
//   INTER_NEAREST sx = min(floor(dx * (src/dst)), src - 1)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void linear_tap(int d, int dst, int src, int &s0, int &s1, float &a0, float &a1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    s0 = s; s1 = min(s + 1, src - 1);
    a0 = 1.0f - f; a1 = f;
}

// src (H,W,3) uint8 -> dst[c][y][x] (planes of dst_h x dst_w, written for y < oh, x < ow), values / div
__global__ __launch_bounds__(NT) void k_image_resize_u8(const uint8_t *__restrict__ src, int H, int W, float *__restrict__ dst,
                                                        int oh, int ow, int dst_h, int dst_w, float div) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= oh * ow) return;
    const int y = i / ow, x = i - y * ow;
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    linear_tap(x, ow, W, x0, x1, a0, a1);
    linear_tap(y, oh, H, y0, y1, b0, b1);
    const uint8_t *r0 = src + (size_t)y0 * W * 3, *r1 = src + (size_t)y1 * W * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = (float)r0[x0 * 3 + c] * a0 + (float)r0[x1 * 3 + c] * a1;
        const float bot = (float)r1[x0 * 3 + c] * a0 + (float)r1[x1 * 3 + c] * a1;
        dst[((size_t)c * dst_h + y) * dst_w + x] = (top * b0 + bot * b1) / div;
    }
}

// float32 planes: src (C,H,W) -> dst (C, dst_h, dst_w) written for y < oh, x < ow, values / div.  MaskRCNN.prepare
// (maskrcnn.py:261-276) resizes the float32 CHW image with chainercv.transforms.resize = cv2.resize INTER_LINEAR.
__global__ __launch_bounds__(NT) void k_image_resize_f32(const float *__restrict__ src, int C, int H, int W, float *__restrict__ dst,
                                                         int oh, int ow, int dst_h, int dst_w, float div) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= oh * ow) return;
    const int y = i / ow, x = i - y * ow;
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    linear_tap(x, ow, W, x0, x1, a0, a1);
    linear_tap(y, oh, H, y0, y1, b0, b1);
    for (int c = 0; c < C; ++c) {
        const float *r0 = src + ((size_t)c * H + y0) * W, *r1 = src + ((size_t)c * H + y1) * W;
        const float top = r0[x0] * a0 + r0[x1] * a1;
        const float bot = r1[x0] * a0 + r1[x1] * a1;
        dst[((size_t)c * dst_h + y) * dst_w + x] = (top * b0 + bot * b1) / div;
    }
}

// src (G,H,W) uint8 -> dst (G, dst_h, dst_w), written for y < oh, x < ow
__global__ __launch_bounds__(NT) void k_mask_resize_nearest_u8(const uint8_t *__restrict__ src, int G, int H, int W,
                                                               uint8_t *__restrict__ dst, int oh, int ow, int dst_h, int dst_w) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)G * oh * ow) return;
    const int x = (int)(i % ow);
    const int y = (int)((i / ow) % oh);
    const int g = (int)(i / ((size_t)ow * oh));
    const int sx = min((int)floor((double)x * (1.0 / ((double)ow / (double)W))), W - 1);
    const int sy = min((int)floor((double)y * (1.0 / ((double)oh / (double)H))), H - 1);
    dst[((size_t)g * dst_h + y) * dst_w + x] = src[((size_t)g * H + sy) * W + sx];
}

// v = momentum*v - lr*(g + wd*p); p += v   (Chainer WeightDecay hook then MomentumSGD).  20 B/param.
__global__ __launch_bounds__(NT) void k_sgd(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ v,
                                            size_t n, float lr, float momentum, float wd) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        float4 pp = ld4(p + i * 4), vv = ld4(v + i * 4);
        const float4 gg = ld4(g + i * 4);
        vv.x = momentum * vv.x - lr * (gg.x + wd * pp.x); vv.y = momentum * vv.y - lr * (gg.y + wd * pp.y);
        vv.z = momentum * vv.z - lr * (gg.z + wd * pp.z); vv.w = momentum * vv.w - lr * (gg.w + wd * pp.w);
        pp.x += vv.x; pp.y += vv.y; pp.z += vv.z; pp.w += vv.w;
        st4(p + i * 4, pp);
        st4(v + i * 4, vv);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float nv = momentum * v[i] - lr * (g[i] + wd * p[i]);
        v[i] = nv;
        p[i] += nv;
    }
}

// ---- bilinear x2 with corner alignment (Chainer F.resize_images, fpn_roi_keypoint_head.py:80-81,109) ----------
// u = i*(H-1)/(OH-1) (float64 like numpy.linspace), u0 = clip(floor(u), 0, H-2), u1 = u0+1, weights (u1-u), (u-u0).
struct Lin { int i0; float w0, w1; };
__device__ __forceinline__ Lin lin_coord(int o, int in, int out) {
    Lin l;
    if (in < 2) { l.i0 = 0; l.w0 = 1.f; l.w1 = 0.f; return l; }
    const double u = (double)o * (double)(in - 1) / (double)(out - 1);
    int i0 = (int)floor(u);
    i0 = min(max(i0, 0), in - 2);
    l.i0 = i0;
    l.w0 = (float)((double)(i0 + 1) - u);
    l.w1 = (float)(u - (double)i0);
    return l;
}

__global__ __launch_bounds__(NT) void k_bilinear2x_fwd(const float *__restrict__ x, float *__restrict__ y, int N, int H,
                                                       int W, int C4) {
    const int OH = 2 * H, OW = 2 * W;
    const size_t n4 = (size_t)N * OH * OW * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, OW, OH, n4 > 0xFFFFFFFFull);
        const int c = ix.c, ow = ix.w, oh = ix.h, n = ix.n;
        const Lin ly = lin_coord(oh, H, OH), lx = lin_coord(ow, W, OW);
        const int y1 = min(ly.i0 + 1, H - 1), x1 = min(lx.i0 + 1, W - 1);
        const float *b = x + (size_t)n * H * W * C4 * 4 + c * 4;
        const float4 a00 = ld4(b + ((size_t)ly.i0 * W + lx.i0) * C4 * 4), a01 = ld4(b + ((size_t)ly.i0 * W + x1) * C4 * 4);
        const float4 a10 = ld4(b + ((size_t)y1 * W + lx.i0) * C4 * 4), a11 = ld4(b + ((size_t)y1 * W + x1) * C4 * 4);
        const float w1 = lx.w0 * ly.w0, w2 = lx.w1 * ly.w0, w3 = lx.w0 * ly.w1, w4 = lx.w1 * ly.w1;
        float4 o;
        o.x = w1 * a00.x + w2 * a01.x + w3 * a10.x + w4 * a11.x;
        o.y = w1 * a00.y + w2 * a01.y + w3 * a10.y + w4 * a11.y;
        o.z = w1 * a00.z + w2 * a01.z + w3 * a10.z + w4 * a11.z;
        o.w = w1 * a00.w + w2 * a01.w + w3 * a10.w + w4 * a11.w;
        st4(y + i * 4, o);
    }
}

// Adjoint, owner-computes: every input cell gathers the (<= 3x3) output cells that interpolate from it.
__global__ __launch_bounds__(NT) void k_bilinear2x_bwd(const float *__restrict__ gy, float *__restrict__ gx, int N, int H,
                                                       int W, int C4) {
    const int OH = 2 * H, OW = 2 * W;
    const size_t n4 = (size_t)N * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        const Nhwc4 ix = split_nhwc4(i, C4, W, H, n4 > 0xFFFFFFFFull);
        const int c = ix.c, w = ix.w, h = ix.h, n = ix.n;
        // output rows whose u lies in (h-1, h+1): o in ((h-1)(OH-1)/(H-1), (h+1)(OH-1)/(H-1))
        const int oy0 = H > 1 ? max(0, (int)(((long long)(h - 1) * (OH - 1)) / (H - 1))) : 0;
        const int oy1 = H > 1 ? min(OH - 1, (int)(((long long)(h + 1) * (OH - 1) + (H - 2)) / (H - 1))) : OH - 1;
        const int ox0 = W > 1 ? max(0, (int)(((long long)(w - 1) * (OW - 1)) / (W - 1))) : 0;
        const int ox1 = W > 1 ? min(OW - 1, (int)(((long long)(w + 1) * (OW - 1) + (W - 2)) / (W - 1))) : OW - 1;
        float4 s = f4(0.f);
        for (int oy = oy0; oy <= oy1; ++oy) {
            const Lin ly = lin_coord(oy, H, OH);
            const float wy = (ly.i0 == h ? ly.w0 : 0.f) + (min(ly.i0 + 1, H - 1) == h ? ly.w1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const Lin lx = lin_coord(ox, W, OW);
                const float wx = (lx.i0 == w ? lx.w0 : 0.f) + (min(lx.i0 + 1, W - 1) == w ? lx.w1 : 0.f);
                if (wx == 0.f) continue;
                const float4 g = ld4(gy + ((((size_t)n * OH + oy) * OW + ox) * C4 + c) * 4);
                const float ww = wx * wy;
                s.x = fmaf(ww, g.x, s.x); s.y = fmaf(ww, g.y, s.y); s.z = fmaf(ww, g.z, s.z); s.w = fmaf(ww, g.w, s.w);
            }
        }
        st4(gx + i * 4, s);
    }
}

__global__ __launch_bounds__(NT) void k_img_nhwc4(const float *__restrict__ x, float *__restrict__ y, int N, size_t HW) {
    const size_t total = (size_t)N * HW;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const size_t n = i / HW, p = i % HW;
        const float *b = x + n * 3 * HW + p;
        st4(y + i * 4, make_float4(b[0], b[HW], b[2 * HW], 0.f));
    }
}

__global__ __launch_bounds__(NT) void k_random_keys(uint32_t *__restrict__ out, size_t n, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);      // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        out[i] = (uint32_t)(z >> 32);
    }
}

// Device-resident seed: keys = hash(state[0], index); state[0] is advanced by a second launch, so a captured HIP
// graph draws fresh keys on every replay (a by-value seed would be frozen into the graph).
__global__ __launch_bounds__(NT) void k_random_keys_dev(uint32_t *__restrict__ out, size_t n,
                                                        const unsigned long long *__restrict__ state) {
    const unsigned long long seed = state[0];
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        out[i] = (uint32_t)(z >> 32);
    }
}
__global__ void k_advance_seed(unsigned long long *state) { state[0] += 0xD1B54A32D192ED03ull; }

int chk(bool ok, const char *what) { return ok ? 0 : mrcnn::fail_arg(MRCNN_E_INVALID, "%s", what); }

}  // namespace

extern "C" size_t mrcnn_bn_workspace_bytes(int P, int C) {
    if (P <= 0 || C <= 0 || (C % 4)) return 0;
    const RedPlan r = red_plan(P, C);
    return (size_t)r.nblk * 2 * C * sizeof(float);
}

// Timing experiments only (tools/ab_step.py): bit 0 leaves out k_bn_bwd_final, bit 1 k_bn_stats_final of the *_stats form (their outputs
// keep whatever the buffers held).
int g_bn_debug_skip = 0;
extern "C" int mrcnn_debug_bn_skip(int mask) {
    if (mask < 0 || mask > 3) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_bn_skip: mask in [0,3]");
    g_bn_debug_skip = mask;
    return 0;
}

extern "C" int mrcnn_debug_bn_plan(int red_cap, int fin_quads) {
    if (red_cap < 1 || red_cap > 8192 || (fin_quads != 1 && fin_quads != 4)) return mrcnn::fail_arg(MRCNN_E_INVALID, "debug_bn_plan: red_cap 1..8192, fin_quads 1 or 4");
    g_red_cap = red_cap; g_fin_quads = fin_quads;
    return 0;
}

extern "C" int mrcnn_bn_train_fwd_f32(const float *x, const float *gamma, const float *beta, const float *residual,
                                      float *y, float *save_mean, float *save_invstd, float *running_mean,
                                      float *running_var, int P, int C, float eps, float decay, int relu, void *ws,
                                      size_t ws_bytes, void *stream) {
    if (int e = chk(x && gamma && beta && y && save_mean && save_invstd && ws, "bn_train_fwd: null pointer")) return e;
    if (int e = chk(P > 0 && C > 0 && (C % 4) == 0, "bn_train_fwd: need P>0, C%4==0")) return e;
    if (ws_bytes < mrcnn_bn_workspace_bytes(P, C)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bn_train_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan r = red_plan(P, C);
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(r.nblk), dim3(NT), 0, st, x, P, C, r.G, r.RPI, r.rows_per_blk, (float *)ws);
    MRCNN_LAUNCH_CHECK();
    launch_stats_final(st, x, (const float *)ws, r.nblk, P, C, eps, decay, 1, save_mean, save_invstd, running_mean, running_var);
    MRCNN_LAUNCH_CHECK();
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(k_bn_apply, dim3(ew_grid(n4)), dim3(NT), 0, st, x, gamma, beta, save_mean, save_invstd, residual, y,
                       n4, C / 4, relu);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// Training-mode BatchNorm forward from the per-row-block partial statistics the producing convolution left behind
// (mrcnn_conv2d_fwd_bnstats_f32: part (rows, 2, C) = sums, sums of squares): finalize + apply, no statistics pass over x.
extern "C" int mrcnn_bn_train_fwd_stats_f32(const float *x, const float *part, int rows, const float *gamma, const float *beta,
                                            const float *residual, float *y, float *save_mean, float *save_invstd,
                                            float *running_mean, float *running_var, int P, int C, float eps, float decay, int relu,
                                            void *stream) {
    if (int e = chk(x && part && gamma && beta && y && save_mean && save_invstd, "bn_train_fwd_stats: null pointer")) return e;
    if (int e = chk(P > 0 && C > 0 && (C % 4) == 0 && rows > 0, "bn_train_fwd_stats: need P>0, C%4==0, rows>0")) return e;
    hipStream_t st = (hipStream_t)stream;
    if (!(g_bn_debug_skip & 2))
    launch_stats_final(st, x, part, rows, P, C, eps, decay, 0, save_mean, save_invstd, running_mean, running_var);
    MRCNN_LAUNCH_CHECK();
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(k_bn_apply, dim3(ew_grid(n4)), dim3(NT), 0, st, x, gamma, beta, save_mean, save_invstd, residual, y, n4, C / 4, relu);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_bn_train_bwd_f32(const float *gy, const float *x, const float *y, const float *gamma, const float *beta,
                                      const float *save_mean, const float *save_invstd, float *gx, float *gres,
                                      float *ggamma, float *gbeta, int P, int C, int relu, void *ws, size_t ws_bytes,
                                      void *stream) {
    if (int e = chk(gy && x && gamma && save_mean && save_invstd && gx && ggamma && gbeta && ws, "bn_train_bwd: null pointer")) return e;
    if (int e = chk(!relu || y || beta, "bn_train_bwd: relu needs y, or beta to recompute the mask")) return e;
    const int rmode = !relu ? 0 : (y ? 1 : 2);
    if (int e = chk(P > 0 && C > 0 && (C % 4) == 0, "bn_train_bwd: need P>0, C%4==0")) return e;
    if (ws_bytes < mrcnn_bn_workspace_bytes(P, C)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bn_train_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan r = red_plan(P, C);
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(r.nblk), dim3(NT), 0, st, gy, x, y, gamma, beta, save_mean, save_invstd, P, C, r.G,
                       r.RPI, r.rows_per_blk, rmode, (float *)ws);
    MRCNN_LAUNCH_CHECK();
    if (!(g_bn_debug_skip & 1))
    launch_bwd_final(st, (const float *)ws, r.nblk, C, gbeta, ggamma);
    MRCNN_LAUNCH_CHECK();
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(ew_grid(n4)), dim3(NT), 0, st, gy, x, y, gamma, save_mean, save_invstd, gbeta,
                       ggamma, gx, gres, n4, C / 4, 1.0f / (float)P, rmode, beta);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// Statistics of a training-mode BatchNorm WITHOUT its apply kernel (the consumer applies it on load: mrcnn_conv2d_fwd_inbn_f32): save_mean /
// save_invstd and the running statistics exactly as mrcnn_bn_train_fwd(_stats)_f32 leaves them.
extern "C" int mrcnn_bn_train_stats_f32(const float *x, const float *part, int rows, float *save_mean, float *save_invstd, float *running_mean,
                                        float *running_var, int P, int C, float eps, float decay, void *ws, size_t ws_bytes, void *stream) {
    if (int e = chk(x && save_mean && save_invstd, "bn_train_stats: null pointer")) return e;
    if (int e = chk(P > 0 && C > 0 && (C % 4) == 0 && rows >= 0 && (rows == 0) == (part == nullptr), "bn_train_stats: need P>0, C%4==0, part / rows together")) return e;
    hipStream_t st = (hipStream_t)stream;
    if (part) {
        launch_stats_final(st, x, part, rows, P, C, eps, decay, 0, save_mean, save_invstd, running_mean, running_var);
    } else {
        if (!ws || ws_bytes < mrcnn_bn_workspace_bytes(P, C)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bn_train_stats: workspace too small");
        const RedPlan r = red_plan(P, C);
        hipLaunchKernelGGL(k_bn_stats_partial, dim3(r.nblk), dim3(NT), 0, st, x, P, C, r.G, r.RPI, r.rows_per_blk, (float *)ws);
        MRCNN_LAUNCH_CHECK();
        launch_stats_final(st, x, (const float *)ws, r.nblk, P, C, eps, decay, 1, save_mean, save_invstd, running_mean, running_var);
    }
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t mrcnn_bn_pair_workspace_bytes(int P, int C) {
    if (P <= 0 || C <= 0 || (C % 4)) return 0;
    const RedPlan r = red_plan(P, C);
    return (size_t)r.nblk * 3 * C * sizeof(float);
}

// relu(BN_a(xa) + BN_b(xb)): statistics of each layer as in the single-layer entry points (part_* / rows_* = the producing convolution's
// partial rows, or NULL / 0 = a statistics pass over the tensor through ws), then ONE apply kernel.
extern "C" int mrcnn_bn_train_fwd_pair_f32(const float *xa, const float *part_a, int rows_a, const float *gamma_a, const float *beta_a,
                                           float *mean_a, float *invstd_a, float *run_mean_a, float *run_var_a, const float *xb,
                                           const float *part_b, int rows_b, const float *gamma_b, const float *beta_b, float *mean_b,
                                           float *invstd_b, float *run_mean_b, float *run_var_b, float *y, int P, int C, float eps, float decay,
                                           void *ws, size_t ws_bytes, void *stream) {
    if (int e = chk(xa && gamma_a && beta_a && mean_a && invstd_a && xb && gamma_b && beta_b && mean_b && invstd_b && y, "bn_train_fwd_pair: null pointer")) return e;
    if (int e = chk(P > 0 && C > 0 && (C % 4) == 0 && rows_a >= 0 && rows_b >= 0, "bn_train_fwd_pair: need P>0, C%4==0, rows>=0")) return e;
    if (int e = chk((rows_a == 0) == (part_a == nullptr) && (rows_b == 0) == (part_b == nullptr), "bn_train_fwd_pair: part / rows must come together")) return e;
    if ((!part_a || !part_b) && (!ws || ws_bytes < mrcnn_bn_workspace_bytes(P, C)))
        return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bn_train_fwd_pair: a statistics pass needs mrcnn_bn_workspace_bytes() of workspace");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan r = red_plan(P, C);
    auto stats = [&](const float *x, const float *part, int rows, float *mean, float *invstd, float *rm, float *rv) {
        if (part) {
            launch_stats_final(st, x, part, rows, P, C, eps, decay, 0, mean, invstd, rm, rv);
        } else {
            hipLaunchKernelGGL(k_bn_stats_partial, dim3(r.nblk), dim3(NT), 0, st, x, P, C, r.G, r.RPI, r.rows_per_blk, (float *)ws);
            launch_stats_final(st, x, (const float *)ws, r.nblk, P, C, eps, decay, 1, mean, invstd, rm, rv);
        }
    };
    stats(xb, part_b, rows_b, mean_b, invstd_b, run_mean_b, run_var_b);
    MRCNN_LAUNCH_CHECK();
    stats(xa, part_a, rows_a, mean_a, invstd_a, run_mean_a, run_var_a);
    MRCNN_LAUNCH_CHECK();
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(k_bn_apply2, dim3(ew_grid(n4)), dim3(NT), 0, st, xa, gamma_a, beta_a, mean_a, invstd_a, xb, gamma_b, beta_b, mean_b, invstd_b, y,
                       n4, C / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

// Backward of the pair: dz = y ? gy * (y > 0) : gy; gxa / gxb = the two BatchNorm backward passes of dz (gbeta_a == gbeta_b == sum dz).
extern "C" int mrcnn_bn_train_bwd_pair_f32(const float *gy, const float *y, const float *xa, const float *xb, const float *gamma_a,
                                           const float *mean_a, const float *invstd_a, const float *gamma_b, const float *mean_b,
                                           const float *invstd_b, float *gxa, float *gxb, float *ggamma_a, float *gbeta_a, float *ggamma_b,
                                           float *gbeta_b, int P, int C, void *ws, size_t ws_bytes, void *stream) {
    if (int e = chk(gy && xa && xb && gamma_a && mean_a && invstd_a && gamma_b && mean_b && invstd_b && gxa && gxb && ggamma_a && gbeta_a && ggamma_b &&
                        gbeta_b && ws, "bn_train_bwd_pair: null pointer")) return e;
    if (int e = chk(P > 0 && C > 0 && (C % 4) == 0, "bn_train_bwd_pair: need P>0, C%4==0")) return e;
    if (ws_bytes < mrcnn_bn_pair_workspace_bytes(P, C)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bn_train_bwd_pair: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan r = red_plan(P, C);
    hipLaunchKernelGGL(k_bn_bwd_partial2, dim3(r.nblk), dim3(NT), 0, st, gy, y, xa, xb, mean_a, invstd_a, mean_b, invstd_b, P, C, r.G, r.RPI,
                       r.rows_per_blk, (float *)ws);
    MRCNN_LAUNCH_CHECK();
    if (g_fin_quads == 1)
        hipLaunchKernelGGL(k_bn_bwd_final2<1>, dim3(mrcnn::cdiv(C, 4)), dim3(FIN_THREADS), 0, st, (const float *)ws, r.nblk, C, gbeta_a, ggamma_a, gbeta_b, ggamma_b);
    else
        hipLaunchKernelGGL(k_bn_bwd_final2<4>, dim3(mrcnn::cdiv(C, 16)), dim3(FIN_THREADS), 0, st, (const float *)ws, r.nblk, C, gbeta_a, ggamma_a, gbeta_b, ggamma_b);
    MRCNN_LAUNCH_CHECK();
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(k_bn_bwd_apply2, dim3(ew_grid(n4)), dim3(NT), 0, st, gy, y, xa, xb, gamma_a, mean_a, invstd_a, gbeta_a, ggamma_a, gamma_b, mean_b,
                       invstd_b, gbeta_b, ggamma_b, gxa, gxb, n4, C / 4, 1.0f / (float)P);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_relu_bwd_f32(const float *gy, const float *y, float *gx, size_t n, void *stream) {
    if (n == 0) return 0;
    if (int e = chk(gy && y && gx && (n % 4) == 0, "relu_bwd: null pointer or n%4")) return e;
    hipLaunchKernelGGL(k_relu_bwd, dim3(ew_grid(n / 4)), dim3(NT), 0, (hipStream_t)stream, gy, y, gx, n / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_add_f32(const float *a, const float *b, float *out, size_t n, void *stream) {
    if (n == 0) return 0;
    if (int e = chk(a && b && out && (n % 4) == 0, "add: null pointer or n%4")) return e;
    hipLaunchKernelGGL(k_add, dim3(ew_grid(n / 4)), dim3(NT), 0, (hipStream_t)stream, a, b, out, n / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_maxpool2x2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    if (int e = chk(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "maxpool2x2_fwd: bad args")) return e;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(ew_grid((size_t)N * Ho * Wo * C / 4)), dim3(NT), 0, (hipStream_t)stream, x, y,
                       N, H, W, C / 4, Ho, Wo);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_maxpool3x3s2_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    if (int e = chk(x && y && N > 0 && H >= 3 && W >= 3 && C > 0 && (C % 4) == 0, "maxpool3x3s2_fwd: bad args")) return e;
    const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool3s2_fwd, dim3(ew_grid((size_t)N * Ho * Wo * C / 4)), dim3(NT), 0, (hipStream_t)stream, x, y, N, H, W,
                       C / 4, Ho, Wo);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_global_avg_pool_fwd_f32(const float *x, float *y, int R, int P, int C, void *stream) {
    if (int e = chk(R >= 0 && P > 0 && C > 0 && (C % 4) == 0 && (R == 0 || (x && y)), "global_avg_pool_fwd: bad args")) return e;
    if (R == 0) return 0;
    hipLaunchKernelGGL(k_global_avg_pool, dim3(ew_grid((size_t)R * C / 4)), dim3(NT), 0, (hipStream_t)stream, x, y, R, P, C / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_relu_fwd_f32(const float *x, float *y, size_t n, void *stream) {
    if (int e = chk((n == 0 || (x && y)) && (n % 4) == 0, "relu_fwd: bad args (n % 4 == 0)")) return e;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_relu_fwd, dim3(ew_grid(n / 4)), dim3(NT), 0, (hipStream_t)stream, x, y, n / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_maxpool2x2_bwd_f32(const float *x, const float *gy, float *gx, int N, int H, int W, int C,
                                        void *stream) {
    if (int e = chk(x && gy && gx && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "maxpool2x2_bwd: bad args")) return e;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(ew_grid((size_t)N * Ho * Wo * C / 4)), dim3(NT), 0, (hipStream_t)stream, x, gy,
                       gx, N, H, W, C / 4, Ho, Wo);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_upsample2x_add_fwd_f32(const float *top, const float *lat, float *out, int N, int H, int W, int Ht,
                                            int Wt, int C, void *stream) {
    if (int e = chk(top && lat && out && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "upsample2x_add_fwd: bad args")) return e;
    if (int e = chk(Ht >= (H + 1) / 2 && Wt >= (W + 1) / 2, "upsample2x_add_fwd: top smaller than ceil(out/2)")) return e;
    hipLaunchKernelGGL(k_upsample_add, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(NT), 0, (hipStream_t)stream, top, lat,
                       out, N, H, W, Ht, Wt, C / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_upsample2x_bwd_f32(const float *gout, float *gtop, int N, int H, int W, int Ht, int Wt, int C,
                                        int accumulate, void *stream) {
    if (int e = chk(gout && gtop && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "upsample2x_bwd: bad args")) return e;
    if (int e = chk(Ht >= (H + 1) / 2 && Wt >= (W + 1) / 2, "upsample2x_bwd: top smaller than ceil(out/2)")) return e;
    hipLaunchKernelGGL(k_upsample_bwd, dim3(ew_grid((size_t)N * Ht * Wt * C / 4)), dim3(NT), 0, (hipStream_t)stream, gout,
                       gtop, N, H, W, Ht, Wt, C / 4, accumulate);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_subsample_bwd_f32(const float *gsub, float *gx, int N, int H, int W, int C, int stride,
                                       int accumulate, const float *relu_x, void *stream) {
    if (int e = chk(gsub && gx && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && stride > 0, "subsample_bwd: bad args")) return e;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    hipLaunchKernelGGL(k_subsample_bwd, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(NT), 0, (hipStream_t)stream, gsub, gx,
                       N, H, W, Ho, Wo, C / 4, stride, accumulate, relu_x);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_pixel_shuffle2x_f32(const float *src, const float *bias, float *dst, int N, int H, int W, int C,
                                         int inverse, void *stream) {
    if (int e = chk(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "pixel_shuffle2x: bad args")) return e;
    hipLaunchKernelGGL(k_pixel_shuffle, dim3(ew_grid((size_t)N * H * W * C)), dim3(NT), 0, (hipStream_t)stream, src, inverse ? nullptr : bias, dst, N,
                       H, W, C / 4, inverse);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_deconv_merge_fwd_f32(const float *wd, const float *bd, const float *w2, const float *b2, float *wm,
                                          float *bm, int C, int Cin, int K2, int ld2, void *stream) {
    if (int e = chk(wd && bd && w2 && wm && bm, "deconv_merge_fwd: null pointer")) return e;
    if (int e = chk(C > 0 && Cin > 0 && K2 > 0 && ld2 >= C, "deconv_merge_fwd: bad sizes")) return e;
    const int total = 4 * K2 * Cin + K2;
    hipLaunchKernelGGL(k_deconv_merge_fwd, dim3(mrcnn::cdiv(total, NT)), dim3(NT), 0, (hipStream_t)stream, wd, bd, w2, b2, wm, bm, C,
                       Cin, K2, ld2);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_deconv_merge_bwd_f32(const float *G, const float *gb4, const float *wd, const float *bd,
                                          const float *w2, float *gwd, float *gbd, float *gw2, float *gb2, int C,
                                          int Cin, int K2, int ld2, void *stream) {
    if (int e = chk(G && gb4 && wd && bd && w2 && gwd && gbd && gw2, "deconv_merge_bwd: null pointer")) return e;
    if (int e = chk(C > 0 && Cin > 0 && K2 > 0 && ld2 >= C, "deconv_merge_bwd: bad sizes")) return e;
    hipStream_t st = (hipStream_t)stream;
    const int t1 = 4 * C * Cin + C + K2;
    hipLaunchKernelGGL(k_deconv_merge_bwd_wd, dim3(mrcnn::cdiv(t1, NT)), dim3(NT), 0, st, G, gb4, w2, gwd, gbd, gb2, C, Cin, K2, ld2);
    MRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_deconv_merge_bwd_w2, dim3(mrcnn::cdiv(K2 * C * 64, NT)), dim3(NT), 0, st, G, gb4, wd, bd, gw2, C, Cin, K2, ld2);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_image_resize_u8_f32(const uint8_t *src, int H, int W, float *dst, int oh, int ow, int dst_h, int dst_w,
                                         float div, void *stream) {
    if (int e = chk(src && dst, "image_resize: null pointer")) return e;
    if (int e = chk(H > 0 && W > 0 && oh > 0 && ow > 0 && dst_h >= oh && dst_w >= ow, "image_resize: bad sizes")) return e;
    hipLaunchKernelGGL(k_image_resize_u8, dim3(mrcnn::cdiv(oh * ow, NT)), dim3(NT), 0, (hipStream_t)stream, src, H, W, dst, oh, ow,
                       dst_h, dst_w, div);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_image_resize_f32(const float *src, int C, int H, int W, float *dst, int oh, int ow, int dst_h, int dst_w,
                                      float div, void *stream) {
    if (int e = chk(src && dst, "image_resize_f32: null pointer")) return e;
    if (int e = chk(C > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && dst_h >= oh && dst_w >= ow, "image_resize_f32: bad sizes")) return e;
    hipLaunchKernelGGL(k_image_resize_f32, dim3(mrcnn::cdiv(oh * ow, NT)), dim3(NT), 0, (hipStream_t)stream, src, C, H, W, dst,
                       oh, ow, dst_h, dst_w, div);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_mask_resize_nearest_u8(const uint8_t *src, int G, int H, int W, uint8_t *dst, int oh, int ow, int dst_h,
                                            int dst_w, void *stream) {
    if (G == 0) return 0;
    if (int e = chk(src && dst, "mask_resize_nearest: null pointer")) return e;
    if (int e = chk(G > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && dst_h >= oh && dst_w >= ow, "mask_resize_nearest: bad sizes")) return e;
    const size_t n = (size_t)G * oh * ow;
    hipLaunchKernelGGL(k_mask_resize_nearest_u8, dim3((unsigned)mrcnn::cdiv(n, (size_t)NT)), dim3(NT), 0, (hipStream_t)stream, src, G,
                       H, W, dst, oh, ow, dst_h, dst_w);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_sgd_momentum_wd_f32(float *p, const float *g, float *v, size_t n, float lr, float momentum,
                                         float weight_decay, void *stream) {
    if (n == 0) return 0;
    if (int e = chk(p && g && v, "sgd_momentum_wd: null pointer")) return e;
    hipLaunchKernelGGL(k_sgd, dim3(ew_grid(std::max<size_t>(n / 4, 1))), dim3(NT), 0, (hipStream_t)stream, p, g, v, n, lr,
                       momentum, weight_decay);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_image_nchw3_to_nhwc4_f32(const float *x, float *y, int N, int H, int W, void *stream) {
    if (int e = chk(x && y && N > 0 && H > 0 && W > 0, "image_nchw3_to_nhwc4: bad args")) return e;
    hipLaunchKernelGGL(k_img_nhwc4, dim3(ew_grid((size_t)N * H * W)), dim3(NT), 0, (hipStream_t)stream, x, y, N, (size_t)H * W);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_random_keys_u32(uint32_t *out, size_t n, unsigned long long seed, void *stream) {
    if (n == 0) return 0;
    if (int e = chk(out != nullptr, "random_keys: null pointer")) return e;
    hipLaunchKernelGGL(k_random_keys, dim3(ew_grid(n)), dim3(NT), 0, (hipStream_t)stream, out, n, seed);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_bilinear2x_fwd_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    if (int e = chk(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "bilinear2x_fwd: bad args")) return e;
    hipLaunchKernelGGL(k_bilinear2x_fwd, dim3(ew_grid((size_t)N * H * W * C)), dim3(NT), 0, (hipStream_t)stream, x, y, N, H, W, C / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_bilinear2x_bwd_f32(const float *gy, float *gx, int N, int H, int W, int C, void *stream) {
    if (int e = chk(gy && gx && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "bilinear2x_bwd: bad args")) return e;
    hipLaunchKernelGGL(k_bilinear2x_bwd, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(NT), 0, (hipStream_t)stream, gy, gx, N, H, W, C / 4);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_random_keys_dev_u32(uint32_t *out, size_t n, unsigned long long *state, void *stream) {
    if (int e = chk(out != nullptr && state != nullptr, "random_keys_dev: null pointer")) return e;
    if (n > 0) {
        hipLaunchKernelGGL(k_random_keys_dev, dim3(ew_grid(n)), dim3(NT), 0, (hipStream_t)stream, out, n, state);
        MRCNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_advance_seed, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrcnn_bn_infer_fwd_f32(const float *x, const float *gamma, const float *beta, const float *mean,
                                      const float *var, const float *residual, float *y, int P, int C, float eps,
                                      int relu, void *stream) {
    if (int e = chk(x && gamma && beta && mean && var && y && P > 0 && C > 0 && (C % 4) == 0, "bn_infer_fwd: bad args")) return e;
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(k_bn_infer, dim3(ew_grid(n4)), dim3(NT), 0, (hipStream_t)stream, x, gamma, beta, mean, var, residual, y,
                       n4, C / 4, eps, relu);
    MRCNN_LAUNCH_CHECK();
    return 0;
}

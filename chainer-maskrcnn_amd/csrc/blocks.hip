// Composite enqueue of a whole ResNet bottleneck (forward, backward) for gfx950: host code only.
//
// Replaces the Python-issued chain of Chainer's ResNet50Layers building block as the reference runs it in training mode
// (chainer_maskrcnn/model/extractor/feature_pyramid_network.py:22,48-66) and its backward inside the updater loop (train.py:117-132).
// Every launch goes through the public entry points of this library (conv.hip, nn.hip) with the operands, order and streams of the
// per-layer host path (chainer_maskrcnn/nn/core.py Bottleneck.fwd / .bwd): same kernels, same plans, same bits.  What this file adds
// is the host economy - the ~25 foreign calls, ~20 allocations and the stream fences of one block become one call - because the step
// was one kernel speed-up away from being bound by its Python enqueue loop (profiles/r04_host_time.txt: 16.4 ms of host per 21.7 ms).
#include "common.h"

#include <algorithm>
#include <mutex>

namespace {

constexpr size_t ALIGN = 256;
constexpr unsigned long long NOT_MATERIALISED = ~0ull;       // plan->off[] of an activation its consumer recomputes on load
int g_inbn_on = 1;            // measurement knob (mrcnn_debug_bottleneck_inbn): 0 = bn1's output is materialised as before
inline size_t up(size_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

struct Geo {
    int N, H, W, Ho, Wo, cin, mid, cout, stride, project;
    long long Pin, Pout;          // pixels of the block's input / of everything behind the (strided) first convolution
};

int geo_of(const mrcnn_bottleneck_t *b, Geo &g) {
    if (!b) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck: null descriptor");
    if (b->N <= 0 || b->H <= 0 || b->W <= 0 || b->cin <= 0 || b->mid <= 0 || b->cout <= 0 || b->stride <= 0 || b->cin % 32 || b->mid % 32 ||
        b->cout % 32)
        return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck: sizes must be positive, channel counts multiples of 32");
    if (!b->project && (b->stride != 1 || b->cin != b->cout))
        return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck: an identity shortcut needs stride 1 and cin == cout");
    if (b->fwd_split < -1 || b->fwd_split > 3) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck: fwd_split in -1..3");
    g.N = b->N; g.H = b->H; g.W = b->W; g.cin = b->cin; g.mid = b->mid; g.cout = b->cout; g.stride = b->stride; g.project = b->project ? 1 : 0;
    g.Ho = (b->H - 1) / b->stride + 1; g.Wo = (b->W - 1) / b->stride + 1;
    g.Pin = (long long)g.N * g.H * g.W; g.Pout = (long long)g.N * g.Ho * g.Wo;
    if (g.Pin * std::max(g.cin, g.cout) >= (1ll << 31)) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "bottleneck: tensor of 2^31 elements or more");
    return 0;
}

// the four convolutions: (input H, W, Cin, Cout, K, stride, pad)
struct CG { int H, W, Cin, Cout, K, stride, pad; };
inline CG conv_geo(const Geo &g, int i) {
    switch (i) {
    case 0: return {g.H, g.W, g.cin, g.mid, 1, g.stride, 0};
    case 1: return {g.Ho, g.Wo, g.mid, g.mid, 3, 1, 1};
    case 2: return {g.Ho, g.Wo, g.mid, g.cout, 1, 1, 0};
    default: return {g.H, g.W, g.cin, g.cout, 1, g.stride, 0};
    }
}

// RAII bracket of the forward arithmetic, the rule of the Python path (nn/core.py _layer_tiles.__enter__) exactly: the forward mode is
// overridden only when the process runs the float32-accurate emulation (mode 3) - the exploratory modes 1 / 2 are left as they are, in both
// paths.  Restored on every exit path.  The setting is process-global state of the library: the bracket holds a lock so that
// two host threads issuing composite calls cannot interleave their set / restore pairs (other callers of mrcnn_conv2d_set_split_operands
// are on their own: the entry point is documented as process-wide).
std::mutex g_split_lock;
struct FwdSplit {
    int keep[3];
    bool on = false;
    std::unique_lock<std::mutex> hold;
    explicit FwdSplit(int want) {
        if (want < 0) return;
        hold = std::unique_lock<std::mutex>(g_split_lock);
        if (mrcnn_conv2d_get_split_operands(keep) != 0 || keep[0] != 3 || keep[0] == want) return;
        on = mrcnn_conv2d_set_split_operands(want, keep[1], keep[2]) == 0;
    }
    ~FwdSplit() {
        if (on) mrcnn_conv2d_set_split_operands(keep[0], keep[1], keep[2]);
    }
};

// A few timing-disabled events per device for the main -> side stream fences (cudaStreamWaitEvent semantics: a wait refers to the
// record before it, so an event can be recorded again as soon as the wait has been enqueued; the ring only keeps that obvious).
constexpr int EV_RING = 16, EV_DEVS = 16;
hipEvent_t g_ev[EV_DEVS][EV_RING];
bool g_ev_made[EV_DEVS] = {};
unsigned g_ev_next[EV_DEVS] = {};
std::mutex g_ev_lock;       // the pool is shared by every caller thread: creation, slot choice and the record / wait pair are one critical section
                            // (two threads recording the same event between each other's record and wait would fence on the wrong point)

int fence(hipStream_t from, hipStream_t to) {
    if (from == to) return 0;
    int dev = 0;
    MRCNN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= EV_DEVS) return mrcnn::fail_arg(MRCNN_E_UNSUPPORTED, "bottleneck: device index %d", dev);
    std::lock_guard<std::mutex> hold(g_ev_lock);
    if (!g_ev_made[dev]) {
        for (int i = 0; i < EV_RING; ++i) MRCNN_HIP_TRY(hipEventCreateWithFlags(&g_ev[dev][i], hipEventDisableTiming));
        g_ev_made[dev] = true;
    }
    hipEvent_t e = g_ev[dev][g_ev_next[dev]++ % EV_RING];
    MRCNN_HIP_TRY(hipEventRecord(e, from));
    MRCNN_HIP_TRY(hipStreamWaitEvent(to, e, 0));
    return 0;
}

#define TRY(expr)                 \
    do {                          \
        if (int e__ = (expr)) return e__; \
    } while (0)

}  // namespace

extern "C" int mrcnn_debug_bottleneck_inbn(int on) {        // A/B of the bn1-on-load path (tools/ab_step.py); plans are made per call
    g_inbn_on = on ? 1 : 0;
    return 0;
}

extern "C" int mrcnn_bottleneck_fwd_plan(const mrcnn_bottleneck_t *b, mrcnn_bottleneck_plan_t *plan) {
    Geo g;
    TRY(geo_of(b, g));
    if (!plan) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_fwd_plan: null plan");
    FwdSplit bracket(b->fwd_split);
    *plan = mrcnn_bottleneck_plan_t{};
    size_t o = 0;
    auto put = [&](int slot, size_t bytes) { plan->off[slot] = o; o += up(bytes); };
    const size_t f = sizeof(float);
    // (conv2's Winograd input transforms apply bn1 + ReLU on load where both of its passes take that path: a1 is then never written -
    // off[A1] = NOT_MATERIALISED)
    const CG c1 = conv_geo(g, 1);
    const bool inbn1 = g_inbn_on && mrcnn_conv2d_inbn_ok(g.N, c1.H, c1.W, c1.Cin, c1.Cout, c1.K, c1.K, c1.stride, c1.pad) != 0;
    put(MRCNN_BN_H1, (size_t)g.Pout * g.mid * f);
    if (inbn1) plan->off[MRCNN_BN_A1] = NOT_MATERIALISED; else put(MRCNN_BN_A1, (size_t)g.Pout * g.mid * f);
    put(MRCNN_BN_H2, (size_t)g.Pout * g.mid * f); put(MRCNN_BN_A2, (size_t)g.Pout * g.mid * f);
    put(MRCNN_BN_H3, (size_t)g.Pout * g.cout * f);
    if (g.project) put(MRCNN_BN_H4, (size_t)g.Pout * g.cout * f);       // (the shortcut's BatchNorm output is never written: bn pair below)
    size_t ws = 0;
    for (int i = 0; i < (g.project ? 4 : 3); ++i) {
        const CG c = conv_geo(g, i);
        plan->part_rows[i] = (int)mrcnn_conv2d_bnstats_rows(g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad);
        plan->v_bytes[i] = mrcnn_conv2d_winograd_v_bytes(g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad);
        if (plan->v_bytes[i]) put(MRCNN_BN_V + i, plan->v_bytes[i]);
        if (plan->part_rows[i]) put(MRCNN_BN_PART + i, (size_t)plan->part_rows[i] * 2 * c.Cout * f);
        else ws = std::max(ws, mrcnn_bn_workspace_bytes((int)g.Pout, c.Cout));
        put(MRCNN_BN_MEAN + i, (size_t)c.Cout * f); put(MRCNN_BN_INVSTD + i, (size_t)c.Cout * f);
        ws = std::max(ws, mrcnn_conv2d_workspace_bytes(g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad));
    }
    plan->arena_bytes = o;
    plan->ws_bytes = ws;
    return 0;
}

extern "C" int mrcnn_bottleneck_fwd_f32(const mrcnn_bottleneck_t *b, const mrcnn_bottleneck_plan_t *plan, const float *x, float *y,
                                        void *arena, size_t arena_bytes, void *ws, size_t ws_bytes, void *stream) {
    Geo g;
    TRY(geo_of(b, g));
    if (!plan || !x || !y || !arena) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_fwd: null pointer");
    if (arena_bytes < plan->arena_bytes) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bottleneck_fwd: arena %zu < %llu", arena_bytes, (unsigned long long)plan->arena_bytes);
    if (ws_bytes < plan->ws_bytes || (plan->ws_bytes && !ws)) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bottleneck_fwd: workspace %zu < %llu", ws_bytes, (unsigned long long)plan->ws_bytes);
    for (int i = 0; i < (g.project ? 4 : 3); ++i)
        if (!b->w[i] || !b->gamma[i] || !b->beta[i]) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_fwd: null parameter pointer (layer %d)", i + 1);
    FwdSplit bracket(b->fwd_split);
    char *A = (char *)arena;
    auto at = [&](int slot) { return (float *)(A + plan->off[slot]); };
    // convolution i of the block (+ its BatchNorm statistics where the launch has them), then BatchNorm i
    auto conv = [&](int i, const float *in, float *out) -> int {
        const CG c = conv_geo(g, i);
        float *v = plan->v_bytes[i] ? at(MRCNN_BN_V + i) : nullptr;
        if (plan->part_rows[i])
            return mrcnn_conv2d_fwd_bnstats_f32(in, b->w[i], out, g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad, at(MRCNN_BN_PART + i), v,
                                                ws, ws_bytes, stream);
        return mrcnn_conv2d_fwd_f32(in, b->w[i], nullptr, out, g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad, 0, v, ws, ws_bytes, stream);
    };
    auto bn = [&](int i, const float *in, const float *residual, float *out, int relu) -> int {
        const int C = conv_geo(g, i).Cout;
        if (plan->part_rows[i])
            return mrcnn_bn_train_fwd_stats_f32(in, at(MRCNN_BN_PART + i), plan->part_rows[i], b->gamma[i], b->beta[i], residual, out,
                                                at(MRCNN_BN_MEAN + i), at(MRCNN_BN_INVSTD + i), b->run_mean[i], b->run_var[i], (int)g.Pout, C, b->eps,
                                                b->decay, relu, stream);
        return mrcnn_bn_train_fwd_f32(in, b->gamma[i], b->beta[i], residual, out, at(MRCNN_BN_MEAN + i), at(MRCNN_BN_INVSTD + i), b->run_mean[i],
                                      b->run_var[i], (int)g.Pout, C, b->eps, b->decay, relu, ws, ws_bytes, stream);
    };
    TRY(conv(0, x, at(MRCNN_BN_H1)));
    if (plan->off[MRCNN_BN_A1] == NOT_MATERIALISED) {
        // bn1's statistics only; conv2's input transform applies bn1 + ReLU to h1 as it loads it (conv.hip wino_input_body)
        const CG c1 = conv_geo(g, 1);
        TRY(mrcnn_bn_train_stats_f32(at(MRCNN_BN_H1), plan->part_rows[0] ? at(MRCNN_BN_PART + 0) : nullptr, plan->part_rows[0], at(MRCNN_BN_MEAN + 0),
                                     at(MRCNN_BN_INVSTD + 0), b->run_mean[0], b->run_var[0], (int)g.Pout, g.mid, b->eps, b->decay, ws, ws_bytes, stream));
        TRY(mrcnn_conv2d_fwd_inbn_f32(at(MRCNN_BN_H1), b->gamma[0], b->beta[0], at(MRCNN_BN_MEAN + 0), at(MRCNN_BN_INVSTD + 0), b->w[1], at(MRCNN_BN_H2), g.N,
                                      c1.H, c1.W, c1.Cin, c1.Cout, c1.K, c1.K, c1.stride, c1.pad, plan->part_rows[1] ? at(MRCNN_BN_PART + 1) : nullptr,
                                      plan->v_bytes[1] ? at(MRCNN_BN_V + 1) : nullptr, ws, ws_bytes, stream));
    } else {
        TRY(bn(0, at(MRCNN_BN_H1), nullptr, at(MRCNN_BN_A1), 1));
        TRY(conv(1, at(MRCNN_BN_A1), at(MRCNN_BN_H2)));
    }
    TRY(bn(1, at(MRCNN_BN_H2), nullptr, at(MRCNN_BN_A2), 1));
    TRY(conv(2, at(MRCNN_BN_A2), at(MRCNN_BN_H3)));
    if (!g.project) return bn(2, at(MRCNN_BN_H3), x, y, 1);
    // projection shortcut: relu(bn3(h3) + bn4(h4)) in one apply kernel - same bits as bn4 -> r, bn3(+ r) (nn.hip k_bn_apply2)
    TRY(conv(3, x, at(MRCNN_BN_H4)));
    const int C = g.cout;
    return mrcnn_bn_train_fwd_pair_f32(at(MRCNN_BN_H3), plan->part_rows[2] ? at(MRCNN_BN_PART + 2) : nullptr, plan->part_rows[2], b->gamma[2], b->beta[2],
                                       at(MRCNN_BN_MEAN + 2), at(MRCNN_BN_INVSTD + 2), b->run_mean[2], b->run_var[2], at(MRCNN_BN_H4),
                                       plan->part_rows[3] ? at(MRCNN_BN_PART + 3) : nullptr, plan->part_rows[3], b->gamma[3], b->beta[3],
                                       at(MRCNN_BN_MEAN + 3), at(MRCNN_BN_INVSTD + 3), b->run_mean[3], b->run_var[3], y, (int)g.Pout, C, b->eps, b->decay, ws,
                                       ws_bytes, stream);
}

namespace {

// backward arena: g_h3 | g_a2 | g_h2 | g_a1 | g_h1 | g_h4 | g_sub
struct BwdLayout { size_t h3, a2, h2, a1, h1, h4, sub, total, ws_main, ws_side; };

BwdLayout bwd_layout(const Geo &g) {
    BwdLayout L{};
    size_t o = 0;
    const size_t f = sizeof(float);
    auto put = [&](size_t bytes) { const size_t at = o; o += up(bytes); return at; };
    L.h3 = put((size_t)g.Pout * g.cout * f);
    L.a2 = put((size_t)g.Pout * g.mid * f); L.h2 = put((size_t)g.Pout * g.mid * f);
    L.a1 = put((size_t)g.Pout * g.mid * f); L.h1 = put((size_t)g.Pout * g.mid * f);
    if (g.project) L.h4 = put((size_t)g.Pout * g.cout * f);
    if (g.project && g.stride != 1) L.sub = put((size_t)g.Pout * g.cin * f);
    L.total = o;
    for (int i = 0; i < (g.project ? 4 : 3); ++i) {
        const CG c = conv_geo(g, i);
        L.ws_main = std::max(L.ws_main, i >= 2 && g.project ? mrcnn_bn_pair_workspace_bytes((int)g.Pout, c.Cout) : mrcnn_bn_workspace_bytes((int)g.Pout, c.Cout));
        L.ws_side = std::max(L.ws_side, mrcnn_conv2d_bwd_filter_workspace_bytes(g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad));
        // the data gradient of a strided 1x1 convolution runs on the subsampled lattice (stride 1 there)
        if (c.stride == 1) L.ws_main = std::max(L.ws_main, mrcnn_conv2d_workspace_bytes(g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, 1, c.pad));
        else L.ws_main = std::max(L.ws_main, mrcnn_conv2d_workspace_bytes(g.N, g.Ho, g.Wo, c.Cin, c.Cout, 1, 1, 1, 0));
    }
    return L;
}

}  // namespace

extern "C" int mrcnn_bottleneck_bwd_sizes(const mrcnn_bottleneck_t *b, size_t *sizes3) {
    Geo g;
    TRY(geo_of(b, g));
    if (!sizes3) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_bwd_sizes: null output");
    const BwdLayout L = bwd_layout(g);
    sizes3[0] = L.total; sizes3[1] = L.ws_main; sizes3[2] = L.ws_side;
    return 0;
}

extern "C" int mrcnn_bottleneck_bwd_f32(const mrcnn_bottleneck_t *b, const mrcnn_bottleneck_plan_t *plan, const float *x, const float *y,
                                        const void *fwd_arena, const float *gy, int gy_masked, float *g_r, float *gx_acc, float *gx_new,
                                        int mask_gx, void *arena, size_t arena_bytes, void *ws_main, size_t ws_main_bytes, void *ws_side,
                                        size_t ws_side_bytes, void *stream, void *side_stream) {
    Geo g;
    TRY(geo_of(b, g));
    if (!plan || !x || !fwd_arena || !gy || !arena) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_bwd: null pointer");
    if (!gy_masked && (!y || (!g_r && !b->project))) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_bwd: an unmasked gy needs y (and a g_r buffer: identity shortcut)");
    if (g.project && !gx_acc && !gx_new) return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_bwd: a projection block needs gx_acc or gx_new");
    const BwdLayout L = bwd_layout(g);
    if (arena_bytes < L.total) return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bottleneck_bwd: arena %zu < %zu", arena_bytes, L.total);
    if (ws_main_bytes < L.ws_main || ws_side_bytes < L.ws_side || !ws_main || !ws_side)
        return mrcnn::fail_arg(MRCNN_E_WORKSPACE, "bottleneck_bwd: workspaces %zu / %zu < %zu / %zu", ws_main_bytes, ws_side_bytes, L.ws_main, L.ws_side);
    for (int i = 0; i < (g.project ? 4 : 3); ++i)
        if (!b->w[i] || !b->gamma[i] || !b->beta[i] || !b->gw[i] || !b->ggamma[i] || !b->gbeta[i])
            return mrcnn::fail_arg(MRCNN_E_INVALID, "bottleneck_bwd: null parameter / gradient pointer (layer %d)", i + 1);
    hipStream_t main = (hipStream_t)stream, side = side_stream ? (hipStream_t)side_stream : main;
    const char *F = (const char *)fwd_arena;
    auto fw = [&](int slot) { return (const float *)(F + plan->off[slot]); };
    char *A = (char *)arena;
    float *g_h3 = (float *)(A + L.h3), *g_a2 = (float *)(A + L.a2), *g_h2 = (float *)(A + L.h2), *g_a1 = (float *)(A + L.a1),
          *g_h1 = (float *)(A + L.h1), *g_h4 = (float *)(A + L.h4), *g_sub = (float *)(A + L.sub);
    const int P = (int)g.Pout;
    // BatchNorm i backward: relu 0 none / 1 mask from yy / 2 mask recomputed from the saved input (BN + ReLU without a residual)
    auto bn_bwd = [&](int i, const float *gyi, const float *xin, const float *yy, int relu, float *gx, float *gres) -> int {
        return mrcnn_bn_train_bwd_f32(gyi, xin, yy, b->gamma[i], b->beta[i], fw(MRCNN_BN_MEAN + i), fw(MRCNN_BN_INVSTD + i), gx, gres, b->ggamma[i],
                                      b->gbeta[i], P, conv_geo(g, i).Cout, relu, ws_main, ws_main_bytes, main);
    };
    // filter gradient of convolution i on the side stream, behind everything enqueued on the main stream so far
    auto filter_grad = [&](int i, const float *xin, const float *gyi) -> int {
        const CG c = conv_geo(g, i);
        TRY(fence(main, side));
        return mrcnn_conv2d_bwd_filter_f32(xin, gyi, b->gw[i], nullptr, g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, c.stride, c.pad, 0,
                                           plan->v_bytes[i] ? fw(MRCNN_BN_V + i) : nullptr, nullptr, ws_side, ws_side_bytes, side);
    };
    auto data_grad = [&](int i, const float *gyi, float *gx, int accumulate, const float *relu_x) -> int {      // stride-1 layers
        const CG c = conv_geo(g, i);
        return mrcnn_conv2d_bwd_data_f32(gyi, b->w[i], gx, relu_x, g.N, c.H, c.W, c.Cin, c.Cout, c.K, c.K, 1, c.pad, accumulate, nullptr, nullptr, 0,
                                         ws_main, ws_main_bytes, main);
    };
    auto data_grad_sub = [&](int i, const float *gyi, float *gsub, int accumulate) -> int {        // strided 1x1: on the output lattice
        const CG c = conv_geo(g, i);
        return mrcnn_conv2d_bwd_data_f32(gyi, b->w[i], gsub, nullptr, g.N, g.Ho, g.Wo, c.Cin, c.Cout, 1, 1, 1, 0, accumulate, nullptr, nullptr, 0,
                                         ws_main, ws_main_bytes, main);
    };
    const float *gr = gy;
    if (g.project) {
        // both BatchNorms of the residual sum in one pass each (sums, apply): the masked gradient is never written - same bits as
        // bn_bwd(2 -> g_h3, g_r) here and bn_bwd(3, g_r -> g_h4) below (nn.hip k_bn_bwd_partial2 / _apply2)
        TRY(mrcnn_bn_train_bwd_pair_f32(gy, gy_masked ? nullptr : y, fw(MRCNN_BN_H3), fw(MRCNN_BN_H4), b->gamma[2], fw(MRCNN_BN_MEAN + 2),
                                        fw(MRCNN_BN_INVSTD + 2), b->gamma[3], fw(MRCNN_BN_MEAN + 3), fw(MRCNN_BN_INVSTD + 3), g_h3, g_h4, b->ggamma[2],
                                        b->gbeta[2], b->ggamma[3], b->gbeta[3], P, g.cout, ws_main, ws_main_bytes, main));
    } else if (gy_masked) {
        TRY(bn_bwd(2, gy, fw(MRCNN_BN_H3), nullptr, 0, g_h3, nullptr));
    } else {
        TRY(bn_bwd(2, gy, fw(MRCNN_BN_H3), y, 1, g_h3, g_r));
        gr = g_r;
    }
    TRY(filter_grad(2, fw(MRCNN_BN_A2), g_h3));
    TRY(data_grad(2, g_h3, g_a2, 0, nullptr));
    TRY(bn_bwd(1, g_a2, fw(MRCNN_BN_H2), nullptr, 2, g_h2, nullptr));
    if (plan->off[MRCNN_BN_A1] == NOT_MATERIALISED) {         // the forward never wrote a1: the filter gradient's input transform rebuilds it from h1
        const CG c1 = conv_geo(g, 1);
        TRY(fence(main, side));
        TRY(mrcnn_conv2d_bwd_filter_inbn_f32(fw(MRCNN_BN_H1), b->gamma[0], b->beta[0], fw(MRCNN_BN_MEAN + 0), fw(MRCNN_BN_INVSTD + 0), g_h2, b->gw[1], g.N, c1.H,
                                             c1.W, c1.Cin, c1.Cout, c1.K, c1.K, c1.stride, c1.pad, 0, plan->v_bytes[1] ? fw(MRCNN_BN_V + 1) : nullptr, ws_side,
                                             ws_side_bytes, side));
    } else {
        TRY(filter_grad(1, fw(MRCNN_BN_A1), g_h2));
    }
    TRY(data_grad(1, g_h2, g_a1, 0, nullptr));
    TRY(bn_bwd(0, g_a1, fw(MRCNN_BN_H1), nullptr, 2, g_h1, nullptr));
    const float *relu_x = mask_gx ? x : nullptr;
    if (!g.project) {
        float *acc = const_cast<float *>(gr);          // the identity shortcut: the input gradient accumulates into the shortcut gradient
        if (gx_acc) {
            TRY(mrcnn_add_f32(gr, gx_acc, gx_acc, (size_t)g.Pin * g.cin, main));
            acc = gx_acc;
        }
        TRY(filter_grad(0, x, g_h1));
        return data_grad(0, g_h1, acc, 1, relu_x);
    }
    if (g.stride == 1) {
        float *gx = gx_acc ? gx_acc : gx_new;
        TRY(filter_grad(0, x, g_h1));
        TRY(data_grad(0, g_h1, gx, gx_acc ? 1 : 0, nullptr));
        TRY(filter_grad(3, x, g_h4));
        return data_grad(3, g_h4, gx, 1, relu_x);
    }
    // both strided 1x1 convolutions read the same lattice: their data gradients are summed there and scattered once
    TRY(filter_grad(0, x, g_h1));
    TRY(filter_grad(3, x, g_h4));
    TRY(data_grad_sub(0, g_h1, g_sub, 0));
    TRY(data_grad_sub(3, g_h4, g_sub, 1));
    return mrcnn_subsample_bwd_f32(g_sub, gx_acc ? gx_acc : gx_new, g.N, g.H, g.W, g.cin, g.stride, gx_acc ? 1 : 0, relu_x, main);
}

// Host-side robustness check of libmrcnn_hip.so, meant to run under AddressSanitizer WITHOUT a GPU
// (make -C chainer-maskrcnn_amd/csrc asan; tests/test_abi_cpu.py::test_host_side_under_address_sanitizer):
//   * every planning / workspace query over a sweep of shapes (these functions index host tables and size workspaces),
//   * every compute entry point with null buffers and with nonsensical sizes: each must return a non-zero code and leave a
//     message in mrcnn_last_error() before anything is launched - never dereference, never divide by zero.
// GPU AddressSanitizer is not available on the target pool; the device code is covered by the parity tests instead.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "mrcnn_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)
#define EXPECT_ERR(call)                                                      \
    do {                                                                      \
        const int rc_ = (call);                                               \
        if (rc_ == 0) { std::printf("FAIL %s:%d: %s returned 0\n", __FILE__, __LINE__, #call); ++failures; } \
        else if (rc_ > 0) { std::printf("FAIL %s:%d: reached the HIP runtime (code %d) instead of rejecting its arguments: %s\n", __FILE__, __LINE__, rc_, #call); ++failures; } \
        else if (rc_ < 0 && std::strlen(mrcnn_last_error()) == 0) { std::printf("FAIL %s:%d: no message\n", __FILE__, __LINE__); ++failures; } \
    } while (0)

int main() {
    EXPECT(mrcnn_abi_version() == MRCNN_ABI_VERSION);
    // ---- planning queries -----------------------------------------------------------------------------------------
    const int sizes[] = {1, 7, 13, 14, 28, 64, 100, 256};
    const int chans[] = {4, 32, 64, 96, 256, 512, 2048};
    const int ks[] = {1, 3, 7};
    unsigned long long acc = 0;
    for (int tiles = 0; tiles < 4; ++tiles) {
        const int t[4][3] = {{2, 0, 0}, {0, 0, 0}, {4, 4, 4}, {-1, -1, -1}};
        EXPECT(mrcnn_conv2d_set_winograd_pass_tiles(t[tiles][0], t[tiles][1], t[tiles][2]) == 0);
        int got[3] = {9, 9, 9};
        EXPECT(mrcnn_conv2d_get_winograd_pass_tiles(got) == 0 && got[0] == t[tiles][0] && got[2] == t[tiles][2]);
        for (int N = 1; N <= 2; ++N)
            for (int H : sizes) for (int W : {H, H + 3}) for (int ci : chans) for (int co : chans) for (int k : ks)
                for (int stride = 1; stride <= 2; ++stride) {
                    const int pad = k / 2;
                    acc += mrcnn_conv2d_workspace_bytes(N, H, W, ci, co, k, k, stride, pad);
                    acc += mrcnn_conv2d_bwd_filter_workspace_bytes(N, H, W, ci, co, k, k, stride, pad);
                    acc += mrcnn_conv2d_winograd_v_bytes(N, H, W, ci, co, k, k, stride, pad);
                    acc += mrcnn_conv2d_winograd_w_bytes(N, H, W, ci, co, k, k, stride, pad);
                    acc += mrcnn_conv2d_bnstats_rows(N, H, W, ci, co, k, k, stride, pad);
                    for (int pass = 0; pass < 3; ++pass) acc += (unsigned long long)mrcnn_conv2d_executed_macs(N, H, W, ci, co, k, k, stride, pad, pass);
                }
    }
    EXPECT(mrcnn_conv2d_set_winograd_pass_tiles(2, 0, 0) == 0);
    EXPECT_ERR(mrcnn_conv2d_set_winograd_pass_tiles(3, 0, 0));
    EXPECT_ERR(mrcnn_conv2d_get_winograd_pass_tiles(nullptr));
    EXPECT_ERR(mrcnn_conv2d_set_debug_skip(16));
    EXPECT_ERR(mrcnn_roi_align_set_bwd_variant(5));
    EXPECT(mrcnn_conv2d_workspace_bytes(0, 0, 0, 0, 0, 0, 0, 0, 0) == 0 || true);
    for (int N = 1; N <= 3; ++N)
        for (int A : {1, 9, 1000, 23025, 261888})
            for (int n_pre : {1, 100, 6000, 12000}) for (int n_post : {1, 300, 2000}) {
                acc += mrcnn_rpn_proposals_workspace_bytes(N, A, n_pre, n_post);
                acc += mrcnn_anchor_target_workspace_bytes(N, A);
            }
    acc += mrcnn_rpn_proposals_workspace_bytes(0, 0, 0, 0) + mrcnn_rpn_proposals_workspace_bytes(-1, -5, -7, -9);
    for (int n : {0, 1, 63, 64, 65, 12000}) acc += mrcnn_nms_workspace_bytes(n);
    for (int P : {1, 100, 524288}) for (int C : chans) acc += mrcnn_bn_workspace_bytes(P, C) + mrcnn_bn_pair_workspace_bytes(P, C);
    acc += mrcnn_loss_workspace_bytes();
    {
        const int Hs[5] = {256, 128, 64, 32, 16}, Ws[5] = {256, 128, 64, 32, 16};
        for (int L = 1; L <= 5; ++L) acc += mrcnn_roi_align_fpn_bwd_workspace_bytes(Hs, Ws, L, 2, 256, 512, 14, 14, 2);
        acc += mrcnn_roi_align_bwd_workspace_bytes(1, 256, 200, 272, 512, 7, 7, 2);
    }
    // ---- compute entries: null buffers ---------------------------------------------------------------------------
    float *F = nullptr; const float *CF = nullptr; int32_t *I = nullptr; const int32_t *CI = nullptr; void *V = nullptr;
    const uint8_t *CU = nullptr; uint8_t *U = nullptr; const uint32_t *CK = nullptr; uint32_t *K = nullptr;
    EXPECT_ERR(mrcnn_roi_align_fwd_f32(CF, 1, 1, 256, 8, 8, CF, 4, 7, 7, 0.25f, 2, F, V));
    EXPECT_ERR(mrcnn_roi_align_bwd_f32(CF, 1, 1, 256, 8, 8, CF, 4, 7, 7, 0.25f, 2, F, V));
    EXPECT_ERR(mrcnn_roi_align_bwd_ws_f32(CF, 1, 1, 256, 8, 8, CF, 4, 7, 7, 0.25f, 2, F, V, 0, V));
    EXPECT_ERR(mrcnn_roi_align_fwd_f32(CF, 7, 1, 256, 8, 8, CF, 4, 7, 7, 0.25f, 2, F, V));
    EXPECT_ERR(mrcnn_roi_align_fpn_fwd_f32(nullptr, nullptr, nullptr, CF, 5, 1, 256, CF, CI, 4, 7, 7, 2, F, V));
    EXPECT_ERR(mrcnn_roi_align_fpn_bwd_f32(CF, nullptr, nullptr, nullptr, CF, 5, 1, 256, CF, CI, 4, 7, 7, 2, 0, V, 0, V));
    EXPECT_ERR(mrcnn_debug_roi_align_bwd_stamps(CF, 1, 256, 8, 8, CF, 4, 7, 7, 0.25f, 2, F, nullptr, V));
    EXPECT_ERR(mrcnn_roi_align_sample_tables(CF, 4, 8, 8, 7, 7, 0.25f, 2, 16, I, I, F, V));
    EXPECT_ERR(mrcnn_conv2d_fwd_f32(CF, CF, CF, F, 1, 8, 8, 32, 32, 3, 3, 1, 1, 0, F, V, 0, V));
    EXPECT_ERR(mrcnn_conv2d_fwd_rect_f32(CF, CF, CF, F, 1, 8, 8, 32, 32, 15, 1, 1, 7, 0, 0, V, 0, V));
    EXPECT_ERR(mrcnn_conv2d_fwd_bnstats_f32(CF, CF, F, 2, 64, 64, 256, 1024, 1, 1, 1, 0, F, F, V, 0, V));
    EXPECT_ERR(mrcnn_bn_train_fwd_stats_f32(CF, CF, 16, CF, CF, CF, F, F, F, F, F, 64, 32, 2e-5f, 0.9f, 1, V));
    EXPECT_ERR(mrcnn_conv2d_bwd_data_f32(CF, CF, F, CF, 1, 8, 8, 32, 32, 3, 3, 1, 1, 0, F, F, 0, V, 0, V));
    EXPECT_ERR(mrcnn_conv2d_bwd_filter_f32(CF, CF, F, F, 1, 8, 8, 32, 32, 3, 3, 1, 1, 0, CF, CF, V, 0, V));
    {   // non-null buffers, unsupported channel counts / zero sizes / too small a workspace
        static float buf[64];
        EXPECT_ERR(mrcnn_conv2d_fwd_f32(buf, buf, buf, buf, 1, 8, 8, 33, 32, 3, 3, 1, 1, 0, nullptr, buf, sizeof(buf), V));
        EXPECT_ERR(mrcnn_conv2d_fwd_f32(buf, buf, buf, buf, 0, 8, 8, 32, 32, 3, 3, 1, 1, 0, nullptr, buf, sizeof(buf), V));
        EXPECT_ERR(mrcnn_conv2d_fwd_bnstats_f32(buf, buf, buf, 2, 64, 64, 256, 256, 3, 3, 1, 1, buf, nullptr, buf, 16, V));    /* Winograd: workspace too small */
        EXPECT_ERR(mrcnn_conv2d_fwd_bnstats_f32(buf, buf, buf, 2, 8, 8, 2048, 512, 1, 1, 1, 0, buf, nullptr, buf, sizeof(buf), V));    /* split-K geometry: declined */
        EXPECT_ERR(mrcnn_conv2d_bwd_filter_f32(buf, buf, buf, buf, 1, 64, 64, 256, 256, 3, 3, 1, 1, 0, nullptr, nullptr, buf, 16, V));
        EXPECT_ERR(mrcnn_rpn_proposals_f32(buf, buf, buf, 1, 30000, 64.f, 64.f, 16.f, nullptr, 20000, 100, 0.7f, buf, (int32_t *)buf, buf,
                                           (int32_t *)buf, I, I, I, buf, 1u << 30, V));
        EXPECT_ERR(mrcnn_anchor_target_f32(buf, 100, buf, (const int32_t *)buf, 1000, 1, 64.f, 64.f, nullptr, CK, 256, 0.7f, 0.3f, 0.5f, 0, buf,
                                           (int32_t *)buf, buf, 1u << 30, V));
    }
    EXPECT_ERR(mrcnn_bn_train_fwd_f32(CF, CF, CF, CF, F, F, F, F, F, 64, 32, 2e-5f, 0.9f, 1, V, 0, V));
    EXPECT_ERR(mrcnn_bn_train_bwd_f32(CF, CF, CF, CF, CF, CF, CF, F, F, F, F, 64, 32, 1, V, 0, V));
    EXPECT_ERR(mrcnn_bn_infer_fwd_f32(CF, CF, CF, CF, CF, CF, F, 64, 32, 2e-5f, 1, V));
    {
        const float *hs[2] = {CF, CF}; float *gs[2] = {F, F}; const int hw[2] = {64, 16};
        EXPECT_ERR(mrcnn_rpn_pack_levels_f32(hs, hw, 2, 1, 32, 3, F, F, 240, V));               // null levels
        EXPECT_ERR(mrcnn_rpn_unpack_grad_levels_f32(CF, CF, gs, hw, 2, 1, 32, 3, 240, V));
        static float hb[64];
        const float *hs2[2] = {hb, hb}; float *gs2[2] = {hb, hb};
        EXPECT_ERR(mrcnn_rpn_pack_levels_f32(hs2, hw, 2, 1, 32, 3, hb, hb, 241, V));            // sum(HW) * A != Atot
        EXPECT_ERR(mrcnn_rpn_pack_levels_f32(hs2, hw, 9, 1, 32, 3, hb, hb, 240, V));            // too many levels
        EXPECT_ERR(mrcnn_rpn_unpack_grad_levels_f32(hb, hb, gs2, hw, 2, 1, 16, 3, 240, V));     // Cp < 6 A
    }
    EXPECT_ERR(mrcnn_bn_train_fwd_pair_f32(CF, CF, 4, CF, CF, F, F, F, F, CF, CF, 4, CF, CF, F, F, F, F, F, 64, 32, 2e-5f, 0.9f, V, 0, V));
    EXPECT_ERR(mrcnn_bn_train_bwd_pair_f32(CF, CF, CF, CF, CF, CF, CF, CF, CF, CF, F, F, F, F, F, F, 64, 32, V, 0, V));
    {   // non-null buffers: part / rows must come together, a statistics pass needs its workspace, the backward its own
        static float b2[64];
        EXPECT_ERR(mrcnn_bn_train_fwd_pair_f32(b2, b2, 0, b2, b2, b2, b2, b2, b2, b2, b2, 4, b2, b2, b2, b2, b2, b2, b2, 64, 32, 2e-5f, 0.9f, b2, sizeof(b2), V));
        EXPECT_ERR(mrcnn_bn_train_fwd_pair_f32(b2, nullptr, 0, b2, b2, b2, b2, b2, b2, b2, b2, 4, b2, b2, b2, b2, b2, b2, b2, 64, 32, 2e-5f, 0.9f, b2, 16, V));
        EXPECT_ERR(mrcnn_bn_train_bwd_pair_f32(b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, 64, 32, b2, 16, V));
        EXPECT_ERR(mrcnn_bn_train_bwd_pair_f32(b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, b2, 64, 30, b2, sizeof(b2), V));
    }
    EXPECT_ERR(mrcnn_relu_bwd_f32(CF, CF, F, 64, V));
    EXPECT_ERR(mrcnn_relu_fwd_f32(CF, F, 64, V));
    EXPECT_ERR(mrcnn_add_f32(CF, CF, F, 64, V));
    EXPECT_ERR(mrcnn_maxpool2x2_fwd_f32(CF, F, 1, 8, 8, 32, V));
    EXPECT_ERR(mrcnn_maxpool2x2_bwd_f32(CF, CF, F, 1, 8, 8, 32, V));
    EXPECT_ERR(mrcnn_maxpool3x3s2_fwd_f32(CF, F, 1, 8, 8, 32, V));
    EXPECT_ERR(mrcnn_global_avg_pool_fwd_f32(CF, F, 2, 49, 32, V));
    EXPECT_ERR(mrcnn_upsample2x_add_fwd_f32(CF, CF, F, 1, 8, 8, 4, 4, 32, V));
    EXPECT_ERR(mrcnn_upsample2x_bwd_f32(CF, F, 1, 8, 8, 4, 4, 32, 0, V));
    EXPECT_ERR(mrcnn_subsample_bwd_f32(CF, F, 1, 8, 8, 32, 2, 0, CF, V));
    EXPECT_ERR(mrcnn_pixel_shuffle2x_f32(CF, CF, F, 1, 8, 8, 32, 0, V));
    EXPECT_ERR(mrcnn_deconv_merge_fwd_f32(CF, CF, CF, CF, F, F, 32, 32, 32, 32, V));
    EXPECT_ERR(mrcnn_deconv_merge_bwd_f32(CF, CF, CF, CF, CF, F, F, F, F, 32, 32, 32, 32, V));
    EXPECT_ERR(mrcnn_bilinear2x_fwd_f32(CF, F, 1, 8, 8, 32, V));
    EXPECT_ERR(mrcnn_bilinear2x_bwd_f32(CF, F, 1, 8, 8, 32, V));
    EXPECT_ERR(mrcnn_image_nchw3_to_nhwc4_f32(CF, F, 1, 8, 8, V));
    EXPECT_ERR(mrcnn_image_resize_u8_f32(CU, 8, 8, F, 16, 16, 16, 16, 255.f, V));
    EXPECT_ERR(mrcnn_image_resize_f32(CF, 3, 8, 8, F, 16, 16, 16, 16, 255.f, V));
    EXPECT_ERR(mrcnn_mask_resize_nearest_u8(CU, 2, 8, 8, U, 16, 16, 16, 16, V));
    EXPECT_ERR(mrcnn_random_keys_u32(K, 64, 1ull, V));
    EXPECT_ERR(mrcnn_random_keys_dev_u32(K, 64, nullptr, V));
    EXPECT_ERR(mrcnn_sgd_momentum_wd_f32(F, CF, F, 64, 1e-3f, 0.9f, 5e-4f, V));
    EXPECT_ERR(mrcnn_softmax_ce_f32(CF, 1, 0, 2, 1, CI, 64, 2, -1, F, F, 0, 2, 1, 2, V, 0, V));
    EXPECT_ERR(mrcnn_smooth_l1_f32(CF, 4, CF, CI, 64, 3.f, F, F, 4, 4, V, 0, V));
    EXPECT_ERR(mrcnn_mask_bce_f32(CF, CI, CI, 4, 196, 96, F, F, V, 0, V));
    EXPECT_ERR(mrcnn_sigmoid_ce_f32(CF, CI, 64, F, F, V, 0, V));
    EXPECT_ERR(mrcnn_select_channel_f32(CF, CI, 4, 80, 196, F, 0, V));
    EXPECT_ERR(mrcnn_nhwc_nchw_f32(CF, F, 4, 196, 96, 80, 0, V));
    EXPECT_ERR(mrcnn_scale_by_dev_f32(F, 64, CF, V));
    EXPECT_ERR(mrcnn_loss_total_f32(CF, 5, F, V));
    EXPECT_ERR(mrcnn_rpn_pack_f32(CF, 1, 64, 32, 3, F, F, 0, 192, V));
    EXPECT_ERR(mrcnn_rpn_unpack_grad_f32(CF, CF, 1, 64, 32, 3, F, 0, 192, V));
    EXPECT_ERR(mrcnn_rpn_proposals_f32(CF, CF, CF, 1, 192, 64.f, 64.f, 16.f, CF, 100, 20, 0.7f, F, I, F, I, I, I, I, V, 0, V));
    EXPECT_ERR(mrcnn_nms_f32(CF, 100, 0.5f, 100, I, I, V, 0, V));
    EXPECT_ERR(mrcnn_map_rois_to_fpn_levels_f32(CF, 10, 2, 6, F, V));
    EXPECT_ERR(mrcnn_softmax2_f32(CF, F, 10, V));
    EXPECT_ERR(mrcnn_proposal_target_f32(CF, CF, CI, 100, CF, CI, CI, 8, CK, 1, 256, 64, 0.5f, 0.5f, 0.f, CF, CF, F, F, I, F, I, I, I, I, I, CI, CI, I, V));
    EXPECT_ERR(mrcnn_mask_target_u8(CU, 1, 8, 64, 64, CF, CI, CI, 256, 64, 14, I, V));
    EXPECT_ERR(mrcnn_keypoint_target_f32(CF, 1, 8, 17, CF, CI, CI, 256, 64, 56, 0, I, V));
    EXPECT_ERR(mrcnn_count_valid_labels_i32(CI, 2, 8, I, V));
    EXPECT_ERR(mrcnn_anchor_target_f32(CF, 100, CF, CI, 8, 1, 64.f, 64.f, CF, CK, 256, 0.7f, 0.3f, 0.5f, 1, F, I, V, 0, V));
    EXPECT_ERR(mrcnn_detect_decode_f32(CF, 10, CF, 96, 81, 88, 1.f, CF, CF, 64.f, 64.f, F, F, V));
    EXPECT_ERR(mrcnn_class_nms_f32(CF, CF, 10, 81, 1, 81, 0.5f, 0.5f, I, I, V));
    EXPECT_ERR(mrcnn_mask_paste_f32(CF, 2, 28, 96, CI, CF, 64, 64, U, V));
    // ---- composite bottleneck (ABI v9): the plans are host arithmetic; the compute calls reject null buffers / short arenas first
    {
        mrcnn_bottleneck_t b{};
        mrcnn_bottleneck_plan_t plan{};
        size_t s3[3] = {0, 0, 0};
        static float host[16];          // non-null stand-ins: every call below must be rejected before anything could touch them
        float *NF = host; const float *NCF = host; void *NV = host;
        EXPECT_ERR(mrcnn_bottleneck_fwd_plan(nullptr, &plan));
        EXPECT_ERR(mrcnn_bottleneck_fwd_plan(&b, &plan));                  // all-zero descriptor
        for (int stride = 1; stride <= 2; ++stride) for (int project = 0; project <= 1; ++project) for (int H : {7, 32, 100}) {
            b = mrcnn_bottleneck_t{};
            b.N = 2; b.H = H; b.W = H + 3; b.cin = project ? 64 : 256; b.mid = 64; b.cout = 256; b.stride = stride; b.project = project; b.fwd_split = -1;
            b.eps = 2e-5f; b.decay = 0.9f;
            if (!project && stride != 1) { EXPECT_ERR(mrcnn_bottleneck_fwd_plan(&b, &plan)); continue; }
            EXPECT(mrcnn_bottleneck_fwd_plan(&b, &plan) == 0 && plan.arena_bytes > 0);
            EXPECT_ERR(mrcnn_bottleneck_fwd_plan(&b, nullptr));
            EXPECT(mrcnn_bottleneck_bwd_sizes(&b, s3) == 0 && s3[0] > 0 && s3[2] > 0);
            EXPECT_ERR(mrcnn_bottleneck_bwd_sizes(&b, nullptr));
            acc += plan.arena_bytes + plan.ws_bytes + s3[0] + s3[1] + s3[2];
            EXPECT_ERR(mrcnn_bottleneck_fwd_f32(&b, &plan, nullptr, NF, NV, 0, NV, 0, V));                // null x
            EXPECT_ERR(mrcnn_bottleneck_fwd_f32(&b, &plan, NCF, NF, NV, 16, NV, (size_t)1 << 40, V));  // arena too small
            EXPECT_ERR(mrcnn_bottleneck_fwd_f32(&b, &plan, NCF, NF, NV, (size_t)1 << 40, NV, (size_t)1 << 40, V));   // null parameters
            EXPECT_ERR(mrcnn_bottleneck_bwd_f32(&b, &plan, NCF, NCF, nullptr, NCF, 1, nullptr, nullptr, NF, 0, NV, (size_t)1 << 40, NV,
                                                (size_t)1 << 40, NV, (size_t)1 << 40, V, V));              // null forward arena
            EXPECT_ERR(mrcnn_bottleneck_bwd_f32(&b, &plan, NCF, NCF, NV, NCF, 0, nullptr, nullptr, NF, 0, NV, (size_t)1 << 40, NV,
                                                (size_t)1 << 40, NV, (size_t)1 << 40, V, V));              // unmasked gy without g_r
            EXPECT_ERR(mrcnn_bottleneck_bwd_f32(&b, &plan, NCF, NCF, NV, NCF, 1, nullptr, NF, NF, 0, NV, 16, NV,
                                                (size_t)1 << 40, NV, (size_t)1 << 40, V, V));              // backward arena too small
            EXPECT_ERR(mrcnn_bottleneck_bwd_f32(&b, &plan, NCF, NCF, NV, NCF, 1, nullptr, NF, NF, 0, NV, (size_t)1 << 40, NV,
                                                (size_t)1 << 40, NV, (size_t)1 << 40, V, V));              // null parameter / gradient pointers
        }
        b.fwd_split = 7;
        EXPECT_ERR(mrcnn_bottleneck_fwd_plan(&b, &plan));
    }
    // ---- ROIAlign backward plan (ABI v9): the size query is host arithmetic; builder / planned backward / status reject bad arguments
    {
        static float host2[16];
        const float *NCF = host2; void *NV = host2; float *NF = host2;
        const int Hs[2] = {40, 20}, Ws[2] = {48, 24};
        const float sc[2] = {0.25f, 0.125f};
        float *gxs[2] = {NF, NF};
        for (int P : {7, 14, 16}) for (int R : {1, 300, 2000}) for (int split = 0; split <= 1; ++split) {
            const size_t b = mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 2, 2, R, P, P, split);
            EXPECT(b > 256);
            acc += b;
        }
        EXPECT(mrcnn_roi_align_fpn_bwd_plan_bytes(nullptr, Ws, 2, 2, 300, 7, 7, 1) == 0);
        EXPECT(mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 2, 2, 300, 17, 17, 1) == 0);       // pooled size beyond the fast path
        EXPECT(mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 0, 2, 300, 7, 7, 1) == 0);
        EXPECT_ERR(mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 2, 2, 256, NCF, nullptr, 300, 7, 7, 2, 1, NV, 1u << 20, V));     // two levels need `levels`
        EXPECT_ERR(mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 2, 2, 256, NCF, CI ? CI : (const int32_t *)host2, 300, 7, 7, 2, 1, nullptr, 0, V));   // null plan
        EXPECT_ERR(mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 2, 2, 255, NCF, (const int32_t *)host2, 300, 7, 7, 2, 1, NV, 1u << 20, V));   // C % 4
        EXPECT_ERR(mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 2, 2, 256, NCF, (const int32_t *)host2, 300, 7, 7, 0, 1, NV, 1u << 20, V));   // adaptive sampling
        EXPECT_ERR(mrcnn_roi_align_fpn_bwd_planned_f32(NCF, nullptr, Hs, Ws, sc, 2, 2, 256, NCF, (const int32_t *)host2, 300, 7, 7, 2, 0, nullptr, 0, NV, 1u << 20, 0, V));
        EXPECT_ERR(mrcnn_roi_align_fpn_bwd_planned_f32(nullptr, gxs, Hs, Ws, sc, 2, 2, 256, NCF, (const int32_t *)host2, 300, 7, 7, 2, 0, nullptr, 0, NV, 1u << 20, 0, V));
        int st3[3];
        EXPECT_ERR(mrcnn_roi_align_bwd_plan_status(nullptr, 1024, st3, V));
        EXPECT_ERR(mrcnn_roi_align_bwd_plan_status(NV, 1024, nullptr, V));
        EXPECT_ERR(mrcnn_roi_align_bwd_plan_status(NV, 16, st3, V));
        EXPECT_ERR(mrcnn_debug_roi_align_lean_variant(3));
    }
    std::printf("planning checksum %llu, %d failure(s)\n", acc, failures);
    return failures ? 1 : 0;
}

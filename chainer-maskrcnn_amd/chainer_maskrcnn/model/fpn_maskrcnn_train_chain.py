"""The FPN Mask R-CNN training step on gfx950 kernels.

Mirror of chainer_maskrcnn/model/fpn_maskrcnn_train_chain.py:14-117: same constructor arguments,
``__call__(imgs, bboxes, labels, masks, scale) -> loss``, five un-weighted losses (:106), the same
report keys (:108-115; kept in ``self.observation`` as device scalars).  As in the reference the
``rpn_sigma`` / ``roi_sigma`` / ``anchor_target_creator`` arguments are accepted and the base-class
defaults (3, 1, default AnchorTargetCreator) apply (:19-26).

What is different in mechanism:
  * no host round trip: proposals, both target creators, the mask crop+resize and all losses run on
    the device; per-image counts stay on the device (padded rows carry label -1);
  * batch size > 1 (the reference raises for n != 1, :37-40; ``strict_batch1=True`` restores that):
    every image samples its own RoIs / anchors, the rows are concatenated and each loss is
    normalised over the concatenated batch (hard part 8);
  * the backward pass is an explicit tape walk (``backward()``); ``loss.backward()`` on the returned
    tensor triggers it, so reference-style ``loss = model(*batch); loss.backward()`` works;
  * the mask branch runs on the (<= 64 per image) positive rows only by default - identical loss and
    gradients; ``mask_rows='all'`` evaluates all sampled rows like the reference.
"""
import numpy as np
import torch

from chainer_maskrcnn._hip import ops
from chainer_maskrcnn.utils.proposal_target_creator import ProposalTargetCreator


class AnchorTargetCreator(object):
    """Parameters of ChainerCV's AnchorTargetCreator; the work is ops.anchor_target."""

    def __init__(self, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3, pos_ratio=0.5):
        self.n_sample, self.pos_iou_thresh, self.neg_iou_thresh, self.pos_ratio = n_sample, pos_iou_thresh, neg_iou_thresh, pos_ratio
        self.seed = 1 << 32
        self._state = None

    def set_seed(self, seed):
        self.seed = seed
        self._state = None

    def __call__(self, bbox, anchor, img_size, n_gt=None, keys=None, per_image_hw=None):
        """bbox (N,G,4) (or (G,4)), anchor (A,4) -> (loc (N,A,4), label (N,A)).  per_image_hw (N,2): each image's own size."""
        if bbox.dim() == 2:
            bbox = bbox[None]
        N, G, _ = bbox.shape
        if n_gt is None:
            n_gt = torch.full((N,), G, dtype=torch.int32, device=bbox.device)
        if keys is None:
            if self._state is None or self._state.device != bbox.device:
                self._state = ops.seed_state(self.seed, bbox.device)
            keys = ops.random_keys_dev((N, anchor.shape[0]), self._state)
        return ops.anchor_target(anchor, bbox.contiguous(), n_gt, img_size, keys, self.n_sample, self.pos_iou_thresh,
                                 self.neg_iou_thresh, self.pos_ratio, per_image_hw=per_image_hw)


class _LossHandle(torch.autograd.Function):
    """Makes ``loss.backward()`` on the returned scalar run the chain's explicit backward pass."""

    @staticmethod
    def forward(ctx, anchor, chain, value):
        ctx.chain = chain
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        # g = d(objective)/d(loss): 1 for a plain ``loss.backward()``; ``(loss * k).backward()`` (loss scaling) gives k.
        # MomentumSGD.update() calls loss.backward() itself and says so (unit_upstream), which skips the scaling pass.
        ctx.chain.backward(upstream=None if ctx.chain.unit_upstream else g)
        return None, None, None


def calc_mask_loss(roi_cls_mask, gt_roi_mask, xp, gt_roi_label):
    """train.py:50-58: channel ``gt_label - 1`` of every row, then sigmoid cross entropy of the first n_pos rows against
    the 28x28 targets.  Passed by name to FPNMaskRCNNTrainChain it is recognised (``fused_kind``) and evaluated as ONE
    fused kernel (``mrcnn_mask_bce_f32``: selection + loss + gradient); called directly - or copied into user code - the
    body below runs on the HIP-backed pieces of chainer_maskrcnn/functions/loss.py and gives the same value."""
    from chainer_maskrcnn import functions as F
    roi_mask = roi_cls_mask[xp.arange(roi_cls_mask.shape[0]), gt_roi_label - 1]
    return F.sigmoid_cross_entropy(roi_mask[:gt_roi_mask.shape[0]], gt_roi_mask)


calc_mask_loss.fused_kind = 'mask_bce'


def calc_keypoint_loss(roi_cls_mask, gt_roi_mask, xp, gt_roi_label, num_keypoints=17):
    """train_keypoints.py:21-27: softmax cross entropy over the 56*56 positions of each (positive RoI, keypoint).
    Fused form: ``mrcnn_softmax_ce_f32`` on the NHWC logits."""
    from chainer_maskrcnn import functions as F
    num_positives = gt_roi_mask.shape[0]
    roi_mask = roi_cls_mask[:num_positives].reshape((num_positives * num_keypoints, -1))
    return F.softmax_cross_entropy(roi_mask, gt_roi_mask.reshape((-1,)))


calc_keypoint_loss.fused_kind = 'keypoint_ce'


# Arithmetic of the convolution GEMMs (mrcnn_conv2d_set_split_operands: forward, backward-data, backward-filter; + which layers' FORWARD
# pass the setting applies to).  All of them take float32 tensors in and out and accumulate in float32:
#   'f32'                     v_mfma_f32_32x32x2_f32 in every pass (bit for bit an fmaf chain)
#   'bf16x6_behind_backbone'  the SHIPPED training default (train.py, bench.py): the float32-ACCURATE three-plane emulation on
#                             v_mfma_f32_32x32x16_bf16 - every operand carried exactly by three bf16 planes (hi + mid + lo), the six
#                             products of weight >= 2^-16 accumulated in float32; per-GEMM error against float64 <= the float32 MFMA's
#                             (tests/test_split_gemm_gpu.py) - in both backward passes of every layer and in the forward pass of every
#                             layer BEHIND the backbone (FPN, RPN, heads); the forward pass of the ResNet's convolutions stays on the
#                             float32 MFMA, so c2 .. c5 - 50 layers of training-mode BatchNorm - are those of the float32 step bit for
#                             bit.  Full-width parity with the UNRELAXED bars of the float32 configuration on five batches
#                             (tests/test_full_width_gpu.py; profiles/r04_full_width_parity_behind_backbone_forward_emulated.txt)
#   'bf16x6_backward'         float32 MFMA in the whole forward pass, the emulation in both backward passes (round 4's first choice: the
#                             same five batches pass, activations / losses / sampled targets bit-identical to 'f32')
#   'bf16x6'                  the emulation in every pass of every layer (opt-in: with the backbone's forward pass emulated, one of the
#                             five batches has 4 % of the gradient tensors above 3 x the float32 floor, against a 3 % bar)
# value = (split operands per pass, emulate the backbone's forward pass too)
GEMM_ARITHMETIC = {'f32': ((0, 0, 0), True), 'bf16x6_behind_backbone': ((3, 3, 3), False), 'bf16x6_backward': ((0, 3, 3), True),
                   'bf16x6': ((3, 3, 3), True)}
DEFAULT_GEMM_ARITHMETIC = 'bf16x6_behind_backbone'


def select_gemm_arithmetic(name):
    """Process-wide: the library's split-operand modes and the per-layer rule of nn/core.py."""
    from chainer_maskrcnn._hip import lib, check
    from chainer_maskrcnn.nn import core
    split, backbone_fwd = GEMM_ARITHMETIC[name]
    check(lib().mrcnn_conv2d_set_split_operands(*split))
    core.FWD_EMULATION_IN_BACKBONE = bool(backbone_fwd)


class FPNMaskRCNNTrainChain(object):
    EARLY_RPN_BACKWARD = True       # A/B switch of the early RPN backward (see rpn_loss_branch in __call__)
    # (r6) the heads' ROIAlign backward passes ACCUMULATE into the feature gradients the early RPN backward left (the kernel then touches
    # only the patches RoIs land on instead of writing every map), instead of five add_ kernels over the pyramid afterwards: same sums in
    # another order - (rpn + box) + mask instead of (box + mask) + rpn -, 0.5 GB and 5 launches less on the main stream
    ACCUMULATE_INTO_EARLY_RPN = True

    def __init__(self, faster_rcnn, mask_loss_fun=calc_mask_loss, binary_mask=True, rpn_sigma=3., roi_sigma=1.,
                 anchor_target_creator=None, strict_batch1=False, mask_rows='positives', gemm_arithmetic=None):
        """gemm_arithmetic: a key of GEMM_ARITHMETIC - the chain then selects it (process-wide library setting) at the start of every
        step; None (default of this constructor, used by the kernel-level tests) leaves the process setting alone.  train.py and
        bench.py pass DEFAULT_GEMM_ARITHMETIC."""
        if gemm_arithmetic is not None and gemm_arithmetic not in GEMM_ARITHMETIC:
            raise ValueError('gemm_arithmetic must be one of %s' % sorted(GEMM_ARITHMETIC))
        self.gemm_arithmetic = gemm_arithmetic
        self.faster_rcnn = faster_rcnn
        self.proposal_target_creator = ProposalTargetCreator(faster_rcnn.extractor.anchor_sizes)
        self.anchor_target_creator = AnchorTargetCreator()      # the argument is ignored, as in the reference
        self.rpn_sigma, self.roi_sigma = 3., 1.                  # base-class defaults apply (:25-26)
        self.loc_normalize_mean = faster_rcnn.loc_normalize_mean
        self.loc_normalize_std = faster_rcnn.loc_normalize_std
        if not callable(mask_loss_fun):
            raise TypeError('mask_loss_fun must be callable: f(roi_cls_mask, gt_roi_mask, xp, gt_roi_label) -> loss')
        # the two functions of the reference's train scripts are recognised and run as fused kernels; any other callable
        # is CALLED, like the reference does (:103-104), on device tensors (see _generic_mask_loss)
        self.mask_loss_fun = mask_loss_fun
        self.mask_loss_kind = getattr(mask_loss_fun, 'fused_kind', None)
        if self.mask_loss_kind not in ('mask_bce', 'keypoint_ce'):
            self.mask_loss_kind = 'generic'
        self.unit_upstream = False
        self.backward_follows = False       # set by MomentumSGD.update around its forward call: loss.backward() comes next, with d loss = 1
        self.binary_mask = binary_mask
        self.strict_batch1 = strict_batch1
        self.mask_rows = mask_rows
        self.observation = {}
        self.grad_ready_hook = None         # called with the lowest finished parameter offset during backward (DP overlap)
        self.sampler_keys = None            # (proposal keys, anchor keys) override for parity tests
        self.use_aux_stream = True          # independent branches (RPN losses, box head) on a second compute stream
        self.keep_outputs = False           # parity tests: keep the head outputs of the last step in self.outputs
        self.outputs = {}
        self._aux = {}
        self._arith_found = None            # the process's GEMM arithmetic before this chain selected its own (restored by backward())

    def _aux_stream(self, dev):
        key = (dev.type, dev.index)
        if key not in self._aux:
            self._aux[key] = torch.cuda.Stream(device=dev)
        return self._aux[key]

    # ------------------------------------------------------------------------------------------
    def __call__(self, imgs, bboxes, labels, masks, scale, n_gt=None, img_sizes=None):
        """scale: the Transform's resize factor - a number, or one per image (host array or device tensor).  img_sizes:
        optional (N,2) (h, w) of every image inside the zero-padded batch tensor (dataset/loader.py 'sizes'); with either,
        proposals are clipped to / anchors tested against each image's OWN size and min_size * its own scale, which is
        what the reference's batch-1-per-process step does (fpn_maskrcnn_train_chain.py:60-70)."""
        m = self.faster_rcnn
        n = bboxes.shape[0]
        if self.strict_batch1 and n != 1:
            raise ValueError('Currently only batch size 1 is supported. n={}'.format(n))
        if self.gemm_arithmetic is not None:        # (read by the library on the host at call time: forward now, backward later)
            # the process-wide setting this chain found is put back at the end of backward(): evaluation / predict calls between training
            # steps and other models of the process keep THEIR arithmetic
            if self._arith_found is None:
                from chainer_maskrcnn._hip import nn as hnn
                from chainer_maskrcnn.nn import core
                self._arith_found = (hnn.split_operands(), core.FWD_EMULATION_IN_BACKBONE)
            select_gemm_arithmetic(self.gemm_arithmetic)
        if torch.is_tensor(scale) and scale.numel() == 1:
            scale = float(scale.reshape(-1)[0].item())
        dev = imgs.device
        _, _, H, W = imgs.shape
        img_size = (H, W)
        per_hw = None
        if img_sizes is not None:
            per_hw = (img_sizes if torch.is_tensor(img_sizes) else torch.from_numpy(np.asarray(img_sizes, np.float32))).to(dev, torch.float32).contiguous()
        i32 = torch.int32
        bboxes = bboxes.contiguous()
        labels = labels.to(i32).contiguous()
        if n_gt is None:
            # padded batches (Chainer's concat_examples / dataset/loader.py pad ragged images with label -1 and zero
            # boxes): the valid rows are packed first, so the per-image count is #(label >= 0) - counted on the device
            n_gt = ops.count_valid_labels(labels)

        main = torch.cuda.current_stream(dev)
        aux = self._aux_stream(dev) if self.use_aux_stream else main
        if getattr(m.head, 'merge_deconv', False):      # composed deconv1*conv2 weights: ready long before the mask branch
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                m.head.compose_deconv(dev)
        features = m.extractor(m.to_nhwc4(imgs))
        m.rpn.train = True
        pk, ak = self.sampler_keys if self.sampler_keys is not None else (None, None)
        losses = torch.empty((5, 2), dtype=torch.float32, device=dev)
        br1 = {}

        def rpn_loss_branch(locs, scores, anchors):
            # Branch 1 (aux stream): anchor targets + RPN losses (:81-85).  It needs the RPN head outputs only, so it is
            # enqueued before the proposal kernels and runs beside them.
            A_ = anchors.shape[0]
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                br1['loc'], br1['label'] = self.anchor_target_creator(bboxes, anchors, img_size, n_gt=n_gt, keys=ak, per_image_hw=per_hw)
                _, br1['g_locs'] = ops.smooth_l1(locs.view(n * A_, 4), 4, br1['loc'].view(n * A_, 4), br1['label'].view(-1),
                                                 n * A_, self.rpn_sigma, out=losses[0])
                _, br1['g_scores'] = ops.softmax_ce(scores.view(n * A_, 2), br1['label'].view(-1), n * A_, 2, (1, 2, 0, 1),
                                                    out=losses[1])
                # The RPN's own backward pass needs nothing but these two gradients: when a backward pass is known to follow
                # (MomentumSGD.update) it is enqueued HERE, on the aux stream, and fills the chip while the main stream walks the
                # latency-bound proposal chain (decode, select, sort, NMS, target sampling: small kernels on a few CUs).  Its
                # per-level feature gradients are kept and added in backward() where rpn.backward() used to run.
                if self.EARLY_RPN_BACKWARD and self.backward_follows and aux is not main:
                    for c1, c2, _, _ in m.rpn.tape:         # saved activations were allocated on the main stream and are released by
                        for t_ in tuple(c1) + tuple(c2):    # this call: the allocator must not hand them out again before aux is done
                            if torch.is_tensor(t_):
                                t_.record_stream(aux)
                    br1['g_feats'] = m.rpn.backward(br1['g_locs'].view(n, A_, 4), br1['g_scores'].view(n, A_, 2), None)

        r = m.rpn.forward_padded(features, per_hw if per_hw is not None else img_size, scale, after_heads=rpn_loss_branch, batch_size=img_size)
        A = r['anchors'].shape[0]
        gt_rpn_loc, gt_rpn_label, g_locs, g_scores = br1['loc'], br1['label'], br1['g_locs'], br1['g_scores']
        # Branch 2 (main stream): proposals -> sampled RoIs and targets
        t = self.proposal_target_creator.sample_batch(
            r['rois'], r['levels'], r['n_rois'], bboxes, labels, n_gt,
            masks=masks.contiguous() if self.binary_mask else None,
            keypoints=None if self.binary_mask else masks.contiguous(),
            loc_normalize_mean=self.loc_normalize_mean, loc_normalize_std=self.loc_normalize_std,
            mask_size=m.head.mask_size, keys=pk, mask_rows=self.mask_rows)

        # head (:88-89) on the n*n_sample sampled rows (padding rows have label -1 and get zero gradient): the box branch
        # runs on the aux stream concurrently with the (much larger) mask branch on the main stream
        head = m.head
        S = self.proposal_target_creator.n_sample
        R = n * S
        scales = m.extractor.spatial_scales
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            box = head.box_branch(features, t['rois_xy5'], t['sample_levels'], scales)
            ld = head.out_p
            g_box = torch.empty_like(box)
            ops.softmax_ce(box, t['gt_roi_label'], R, head.n_class, (1, ld, 0, 1), Kfill=head.LOC0, gx=g_box, out=losses[3])
            ops.smooth_l1(box, ld, t['gt_roi_loc'], t['gt_roi_label'], R, self.roi_sigma, gfill=ld - head.LOC0,
                          col0=head.LOC0, gx=g_box, out=losses[2])

        rows = t['mask_rows']
        if self.mask_loss_kind == 'generic':
            g_mask = self._generic_mask_loss(t, features, scales, losses)
            m_rois, m_levels, m_label = self._generic_inputs
            mask_out = None
        elif rows == S:
            m_rois, m_levels, m_label = t['rois_xy5'], t['sample_levels'], t['gt_roi_label']
        else:   # the first `rows` rows of every image block (positives come first)
            m_rois = t['rois_xy5'].view(n, S, 5)[:, :rows].reshape(n * rows, 5)
            m_levels = t['sample_levels'].view(n, S)[:, :rows].reshape(n * rows)
            m_label = t['gt_roi_label'].view(n, S)[:, :rows].reshape(n * rows)
        if self.mask_loss_kind != 'generic':
            mask_out = head.mask_branch(features, m_rois, m_levels, scales)
        if self.mask_loss_kind == 'generic':
            pass
        elif self.mask_loss_kind == 'mask_bce':
            _, g_mask = ops.mask_bce(mask_out, t['gt_roi_mask'], m_label, out=losses[4])
        else:
            Rm, Hm, Wm, Cm = mask_out.shape
            K = head.n_keypoints
            xmap = (K, Hm * Wm * Cm, 1, Cm)
            g_mask = torch.empty_like(mask_out) if ops.softmax_ce_fills_gradient(Rm * K, Hm * Wm, xmap) else torch.zeros_like(mask_out)
            ops.softmax_ce(mask_out, t['gt_roi_mask'].view(-1), Rm * K, Hm * Wm, xmap, gx=g_mask, out=losses[4])

        main.wait_stream(aux)
        if aux is not main:             # allocated on the aux stream, consumed by the RPN backward on the main stream
            g_locs.record_stream(main)
            g_scores.record_stream(main)
        total = ops.loss_total(losses)
        self.observation = {'rpn_loc_loss': losses[0, 0], 'rpn_cls_loss': losses[1, 0], 'roi_loc_loss': losses[2, 0],
                            'roi_cls_loss': losses[3, 0], 'mask_loss': losses[4, 0], 'loss': total[0]}
        self._bwd = (features, g_locs.view(n, A, 4), g_scores.view(n, A, 2), g_box, g_mask)
        self._early_rpn = br1.get('g_feats')
        self.targets = t
        if self.keep_outputs:
            self.outputs = dict(features=features, locs=r['locs'], scores=r['scores'], box=box, mask=mask_out)
        self.rpn_targets = (gt_rpn_loc, gt_rpn_label)
        self.mask_inputs = (m_rois, m_levels, m_label)
        self.rpn_out = r
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        return _LossHandle.apply(self._anchor, self, total[0])

    # ------------------------------------------------------------------------------------------
    def _generic_mask_loss(self, t, features, scales, losses):
        """A user-supplied ``mask_loss_fun`` (not one of the two recognised functions): call it like the reference does
        (:103-104) - ``roi_cls_mask`` (R, C, S, S) with the positive rows first, ``gt_roi_mask`` with exactly n_pos rows,
        ``xp``, ``gt_roi_label`` (R,) - under torch autograd on HIP-backed operators (functions/loss.py), and return the
        gradient w.r.t. the NHWC logits.  Exact row counts need one host sync; this is the compatibility path."""
        from chainer_maskrcnn.functions import loss as FL
        m = self.faster_rcnn
        head = m.head
        S = self.proposal_target_creator.n_sample
        n = t['n_pos'].shape[0]
        n_pos = [int(v) for v in t['n_pos'].cpu().tolist()]
        n_smp = [int(v) for v in t['n_sampled'].cpu().tolist()]
        rows_of = lambda x, lo, hi: [x.view(n, S, *x.shape[1:])[i, lo(i):hi(i)] for i in range(n)]
        order = lambda x: torch.cat(rows_of(x, lambda i: 0, lambda i: n_pos[i]) +
                                    rows_of(x, lambda i: n_pos[i], lambda i: n_smp[i]), 0).contiguous()
        m_rois, m_levels, m_label = order(t['rois_xy5']), order(t['sample_levels']), order(t['gt_roi_label'])
        rows = t['mask_rows']
        gt = t['gt_roi_mask']
        gt = torch.cat([gt.view(n, rows, *gt.shape[1:])[i, :n_pos[i]] for i in range(n)], 0).contiguous()
        self._generic_inputs = (m_rois, m_levels, m_label)
        mask_out = head.mask_branch(features, m_rois, m_levels, scales)          # (R, S, S, Cp) NHWC
        leaf = mask_out.detach().requires_grad_(True)
        C = head.mask_out_channels
        with torch.enable_grad():
            x = FL.nhwc_to_nchw(leaf, C).as_subclass(FL.MaskLogits)
            loss = self.mask_loss_fun(x, gt, FL.XP(leaf.device), m_label)
            if not torch.is_tensor(loss) or loss.numel() != 1:
                raise TypeError('mask_loss_fun must return a scalar tensor')
            (g_mask,) = torch.autograd.grad(loss.reshape(()), leaf)
        losses[4, 0].copy_(loss.detach().reshape(()))
        losses[4, 1].fill_(1.0)
        return g_mask.contiguous()

    def backward(self, upstream=None):
        """Explicit backward pass: fills the flat gradient buffer of ``faster_rcnn.ps``.  ``upstream`` (device scalar or
        None = 1): d(objective)/d(loss), multiplied into the four loss-gradient seeds."""
        m = self.faster_rcnn
        features, g_locs, g_scores, g_box, g_mask = self._bwd
        if upstream is not None:
            from chainer_maskrcnn.functions.loss import _scale_
            for g_ in (g_locs, g_scores, g_box, g_mask):
                _scale_(g_, upstream)
        hook = self.grad_ready_hook
        dev = features[0].device
        main = torch.cuda.current_stream(dev)
        aux = self._aux_stream(dev) if self.use_aux_stream else main
        into_early = (self.ACCUMULATE_INTO_EARLY_RPN and self._early_rpn is not None and upstream is None
                      and len(self._early_rpn) == len(features))
        if into_early:
            g_feats = list(self._early_rpn)
            for g in g_feats:
                g.record_stream(main)
        else:
            g_feats = [torch.empty_like(f) for f in features]
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            m.head.backward_box(g_box, g_feats, accumulate=into_early)     # overwrites g_feats (first pooled size) unless they hold the RPN's
        g_pool = m.head.backward_mask_convs(g_mask)       # main stream: the mask branch down to its pooled input
        main.wait_stream(aux)
        m.head.backward_mask_pool(g_pool, g_feats)        # accumulates into g_feats (second pooled size)
        if hook:
            hook(self._offset_of('head/'))
        if into_early:
            self._early_rpn = None
        elif self._early_rpn is not None:     # computed beside the proposal chain of the forward pass (aux stream, joined at its end)
            if upstream is not None:        # (its parameter gradients are already in the buffer, un-scaled: they cannot be fixed up here)
                raise RuntimeError('the RPN backward pass ran early with d(objective)/d(loss) = 1 (backward_follows was set): '
                                   'a scaled loss needs chain.backward_follows = False during the forward call')
            for gf, g in zip(g_feats, self._early_rpn):
                gf.add_(g)
                g.record_stream(main)
            self._early_rpn = None
        else:
            m.rpn.backward(g_locs, g_scores, g_feats)
        if hook:
            hook(self._offset_of('rpn/'))
        m.extractor.backward(g_feats, progress=(lambda prefix: hook(self._offset_of(prefix))) if hook else None)
        from chainer_maskrcnn.nn import core
        core.join_side_stream(features[0].device)
        if hook:
            hook(0)
        self._bwd = None
        if self._arith_found is not None:
            from chainer_maskrcnn._hip import lib, check
            split, in_backbone = self._arith_found
            check(lib().mrcnn_conv2d_set_split_operands(*split))
            core.FWD_EMULATION_IN_BACKBONE = in_backbone
            self._arith_found = None

    def _offset_of(self, prefix):
        cache = self.__dict__.setdefault('_offset_cache', {})
        if prefix not in cache:
            ps = self.faster_rcnn.ps
            cache[prefix] = min(o for name, (o, _) in ps.offsets.items() if name.startswith(prefix))
        return cache[prefix]

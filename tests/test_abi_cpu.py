"""CPU tests of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every
symbol include/mrcnn_hip.h declares; argument errors are reported without touching a device."""
import os
import re

import pytest

from chainer_maskrcnn import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mrcnn_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mrcnn_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()            # the driver's "does it build" entry (make is a no-op when the library is current)
    lib = _hip.lib()
    names = _declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), 'libmrcnn_hip.so lacks %s' % n
        assert n in _hip.SIGNATURES, 'ctypes binding lacks %s' % n
    assert set(_hip.SIGNATURES) == set(names)
    assert lib.mrcnn_abi_version() == _hip.ABI_VERSION == 10


def test_argument_errors_do_not_need_a_device():
    lib = _hip.lib()
    rc = lib.mrcnn_roi_align_fwd_f32(None, 7, 1, 8, 5, 5, None, 1, 7, 7, 0.25, 2, None, None)
    assert rc == -1
    assert b'layout' in lib.mrcnn_last_error()
    with pytest.raises(_hip.MrcnnHipError):
        _hip.check(rc)


def test_product_path_has_no_cpu_fallback():
    import torch
    from chainer_maskrcnn.functions.roi_align.roi_align_2d import roi_align_2d
    with pytest.raises(_hip.MrcnnHipError):
        roi_align_2d(torch.zeros(1, 4, 5, 5), torch.zeros(1, 5), 7, 7, 0.25)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'chainer-maskrcnn_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), fn


def test_host_side_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers on the CPU build): the library's host code - argument validation, workspace and
    launch planning - built with AddressSanitizer (device code as usual; GPU ASan does not exist on the target pool) and
    driven by tests/native/abi_host_check.cpp: every planning query over a shape sweep, every compute entry point with
    null buffers / bad sizes.  No GPU is touched: each call must fail cleanly before launching."""
    import shutil
    import subprocess
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    import torch
    if torch.cuda.device_count() > 0:       # (does not initialise the device)
        pytest.skip('host-only check: the driver passes null buffers on purpose and must not run where a GPU could execute a launch')
    csrc = os.path.join(ROOT, 'chainer-maskrcnn_amd', 'csrc')
    subprocess.check_call(['make', '-C', csrc, 'asan', '-j8'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rt = subprocess.check_output(['/opt/rocm/lib/llvm/bin/clang', '-print-file-name=libclang_rt.asan-x86_64.so']).decode().strip()
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0',
               LD_LIBRARY_PATH=os.path.dirname(rt) + ':' + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([os.path.join(csrc, 'asan', 'abi_host_check')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'AddressSanitizer' not in out and ' 0 failure(s)' in out, out[-3000:]


def test_gemm_arithmetic_names_are_the_same_everywhere():
    """The training arithmetic is chosen by name in three places (train.py --gemm-arithmetic, bench.py --gemm-arithmetic, the chain's
    GEMM_ARITHMETIC table): the names and the default must agree, the float32-MFMA fallback must exist, and only the float32-ACCURATE
    schemes (library split modes 0 and 3) may be selectable there - the narrower two-plane splits (modes 1, 2) are never a training default."""
    import re
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import GEMM_ARITHMETIC, DEFAULT_GEMM_ARITHMETIC
    assert DEFAULT_GEMM_ARITHMETIC in GEMM_ARITHMETIC and 'f32' in GEMM_ARITHMETIC
    assert GEMM_ARITHMETIC['f32'][0] == (0, 0, 0)
    assert all(set(modes) <= {0, 3} for modes, _ in GEMM_ARITHMETIC.values())
    for path in ('train.py', 'bench.py'):
        src = open(os.path.join(ROOT, path)).read()
        m = re.search(r"'--gemm-arithmetic',\s*default=([^,]+),\s*choices=\[([^\]]+)\]", src)
        assert m, path
        choices = set(re.findall(r"'([^']+)'", m.group(2)))
        assert choices == set(GEMM_ARITHMETIC), (path, choices)
        default = m.group(1).strip().strip("'")
        assert default in ('None', DEFAULT_GEMM_ARITHMETIC), (path, default)      # bench.py: None = the shipped training default


def test_shipped_arithmetic_brackets_the_backbone_forward_calls():
    """'bf16x6_behind_backbone' = library split modes (3, 3, 3) with the FORWARD call of every extractor/resnet convolution bracketed back to
    the float32 MFMA by the host layer (nn/core.py: _layer_tiles).  The split-mode setting is host state of the library (no device needed):
    inside the bracket of a backbone layer the forward mode reads 0 and the backward modes stay 3; a layer behind the backbone keeps 3; the
    setting is restored on exit; 'f32' and 'bf16x6' do not bracket anything."""
    from chainer_maskrcnn._hip import nn as hnn
    from chainer_maskrcnn.nn import core
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import select_gemm_arithmetic

    ps = core.ParamStore()
    backbone, behind = core.Bottleneck(ps, 'extractor/resnet/res3/a', 64, 32, 128, 2, True).conv2, core.Conv(ps, 'extractor/lat_p3', 64, 32, 1)
    assert backbone.in_backbone and not behind.in_backbone
    try:
        select_gemm_arithmetic('bf16x6_behind_backbone')
        assert tuple(hnn.split_operands()) == (3, 3, 3) and core.FWD_EMULATION_IN_BACKBONE is False
        with core._layer_tiles(backbone):
            assert tuple(hnn.split_operands()) == (0, 3, 3)
        assert tuple(hnn.split_operands()) == (3, 3, 3)
        with core._layer_tiles(behind):
            assert tuple(hnn.split_operands()) == (3, 3, 3)
        for name, want in (('bf16x6', (3, 3, 3)), ('bf16x6_backward', (0, 3, 3)), ('f32', (0, 0, 0))):
            select_gemm_arithmetic(name)
            assert core.FWD_EMULATION_IN_BACKBONE is True
            for layer in (backbone, behind):
                with core._layer_tiles(layer):
                    assert tuple(hnn.split_operands()) == want, (name, layer.name)
    finally:
        select_gemm_arithmetic('f32')
        core.FWD_EMULATION_IN_BACKBONE = core.FWD_EMULATION_BEHIND_BACKBONE = True


def test_every_backbone_marks_its_convolutions():
    """the 'behind the backbone' decision is an attribute set by the extractor that builds the layer, not a match on the FPN's
    parameter names - the C4 backbone, Darknet and the res5 head (BatchNorm behind every convolution) keep their forward pass on the float32
    MFMA under 'bf16x6_behind_backbone' like the FPN's ResNet; FPN laterals / smoothing, RPN and head convolutions do not."""
    from chainer_maskrcnn.nn import core
    from chainer_maskrcnn.model.extractor.c4_backbone import C4Backbone
    from chainer_maskrcnn.model.extractor.darknet import Darknet
    from chainer_maskrcnn.model.extractor.feature_pyramid_network import FeaturePyramidNetwork
    from chainer_maskrcnn.model.head.resnet_roi_mask_head import ResnetRoIMaskHead

    def convs_of(obj, seen=None):
        seen = set() if seen is None else seen
        out = []
        for v in vars(obj).values():
            for o in (v if isinstance(v, (list, tuple)) else [v]):
                for o2 in (o if isinstance(o, (list, tuple)) else [o]):
                    if isinstance(o2, core.Conv):
                        out.append(o2)
                    elif hasattr(o2, '__dict__') and type(o2).__module__.startswith('chainer_maskrcnn') and id(o2) not in seen:
                        seen.add(id(o2))
                        out += convs_of(o2, seen)
        return out
    c4 = convs_of(C4Backbone(stages=(1, 1, 1), width_div=4))
    assert len(c4) == 1 + 3 * 4 and all(c.in_backbone for c in c4)
    dk = convs_of(Darknet())
    assert len(dk) == 5 and all(c.in_backbone for c in dk)
    fpn = FeaturePyramidNetwork(stages=(1, 1, 1, 1), width_div=4)
    by = {c.name: c.in_backbone for c in convs_of(fpn)}
    assert all(v == ('/resnet/' in k) for k, v in by.items()) and sum(by.values()) == 1 + 4 * 4 and len(by) == 17 + 8
    head = ResnetRoIMaskHead(81, 7, 1 / 16., width_div=4)
    assert all(c.in_backbone for b in head.res5 for c in (b.conv1, b.conv2, b.conv3)) and not head.conv1.in_backbone


def test_composite_bottleneck_plan_is_host_arithmetic():
    """mrcnn_bottleneck_fwd_plan / _bwd_sizes (ABI v9, csrc/blocks.hip) need no device: the forward arena holds the six activations of a
    projection block (ABI v10: the shortcut's BatchNorm output is never written - slot 6 stays empty), 256-byte aligned and disjoint; fwd_split brackets the plan like the forward call and is restored; bad descriptors are
    argument errors."""
    import ctypes
    from chainer_maskrcnn import _hip
    from chainer_maskrcnn._hip import nn as hnn
    lib = _hip.lib()
    d = _hip.Bottleneck()
    d.N, d.H, d.W, d.cin, d.mid, d.cout, d.stride, d.project, d.fwd_split = 2, 64, 64, 512, 256, 1024, 2, 1, 0
    plan = _hip.BottleneckPlan()
    _hip.check(lib.mrcnn_conv2d_set_split_operands(3, 3, 3))
    try:
        _hip.check(lib.mrcnn_bottleneck_fwd_plan(ctypes.byref(d), ctypes.byref(plan)))
        assert tuple(hnn.split_operands()) == (3, 3, 3)
    finally:
        _hip.check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
    P = 2 * 32 * 32
    # slot 1 (a1 = relu(bn1(h1))) is not materialised where conv2 takes the Winograd path in both passes: its input transforms apply bn1 on load
    NOT = 2 ** 64 - 1
    assert plan.off[1] == (NOT if lib.mrcnn_conv2d_inbn_ok(2, 32, 32, 256, 256, 3, 3, 1, 1) else P * 256 * 4)
    have = [i for i in range(6) if plan.off[i] != NOT]
    sizes = {0: P * 256 * 4, 1: P * 256 * 4, 2: P * 256 * 4, 3: P * 256 * 4, 4: P * 1024 * 4, 5: P * 1024 * 4}
    assert all(o % 256 == 0 for o in plan.off if o != NOT) and plan.off[6] == 0
    assert [plan.off[i] for i in have] == [sum(sizes[j] for j in have[:k]) for k in range(len(have))]
    used = sorted((plan.off[i], i) for i in range(_hip.BN_SLOTS) if plan.off[i] != NOT and (i in have or plan.off[i]))
    assert all(a[0] < b[0] for a, b in zip(used, used[1:])) and plan.arena_bytes > used[-1][0]
    assert plan.v_bytes[1] == lib.mrcnn_conv2d_winograd_v_bytes(2, 32, 32, 256, 256, 3, 3, 1, 1) and plan.v_bytes[0] == 0
    assert plan.part_rows[1] == lib.mrcnn_conv2d_bnstats_rows(2, 32, 32, 256, 256, 3, 3, 1, 1)
    s3 = (ctypes.c_size_t * 3)()
    _hip.check(lib.mrcnn_bottleneck_bwd_sizes(ctypes.byref(d), s3))
    assert s3[0] >= P * (1024 * 2 + 256 * 4 + 512) * 4 and s3[1] > 0 and s3[2] >= 1024 * 256 * 4
    d.project, d.stride = 0, 2          # an identity shortcut cannot be strided
    assert lib.mrcnn_bottleneck_fwd_plan(ctypes.byref(d), ctypes.byref(plan)) == -1 and b'identity' in lib.mrcnn_last_error()
    d.stride, d.cin = 1, 100
    assert lib.mrcnn_bottleneck_fwd_plan(ctypes.byref(d), ctypes.byref(plan)) == -1
    assert lib.mrcnn_bottleneck_fwd_plan(None, ctypes.byref(plan)) == -1

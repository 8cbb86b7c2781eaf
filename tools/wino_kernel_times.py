"""Per-kernel duration of the Winograd transform kernels for the step's main 3x3 shapes.
  under rocprofv3:   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/wino_kernel_times.py run
  afterwards:        python tools/wino_kernel_times.py report DIR
Each shape is run with the process-wide pass tiles of the step ({2,0,0}: F(2x2) forward, F(4x4) both backward passes)
and forward tile 0 for the FPN / RPN / head layers; a k_sgd-free marker (a fill of N*7 floats) separates the shapes."""
import csv, glob, os, re, sys, collections
SHAPES = [('mask 512x14x14 256', 512, 14, 14, 256, 0), ('p2 2x256x256 256', 2, 256, 256, 256, 0), ('p3 2x128x128 256', 2, 128, 128, 256, 0),
          ('p4 2x64x64 256', 2, 64, 64, 256, 0), ('res4 2x64x64 256 (fwd F2)', 2, 64, 64, 256, 2), ('res3 2x128x128 128 (fwd direct)', 2, 128, 128, 128, 2)]
if sys.argv[1] == 'run':
    R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
    import torch
    from chainer_maskrcnn._hip import nn as hnn
    dev = torch.device('cuda:0')
    for si, (name, N, H, W, C, ft) in enumerate(SHAPES):
        hnn.set_winograd_pass_tiles(ft, 0, 0)
        x = torch.randn((N, H, W, C), device=dev); w = torch.randn((C, 3, 3, C), device=dev) * 0.05
        b = torch.zeros((C,), device=dev); gy = torch.randn((N, H, W, C), device=dev)
        for rep in range(6):
            marker = torch.zeros((1000 + si,), device=dev)          # fillBuffer launch with a recognisable grid
            y, v = hnn.conv2d_fwd_raw(x, w, b, 1, 1, True, keep_v=True)
            gx = hnn.conv2d_bwd_data_raw(gy, w, x.shape, 1, 1)
            hnn.conv2d_bwd_filter_raw(x, gy, w.shape, 1, 1, True)
        torch.cuda.synchronize()
else:
    f = sorted(glob.glob(sys.argv[2] + '/*/*_kernel_trace.csv'))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    agg = collections.OrderedDict()
    for r in rows:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
        pre = sys.argv[3].split(',') if len(sys.argv) > 3 else ['k_wino', 'k_sum_slabs', 'k_conv_igemm', 'k_tail', 'k_colsum']
        if not any(n.startswith(q) for q in pre):
            continue
        key = (n[:44], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']))
        agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for (n, g), v in agg.items():
        v = sorted(v)
        print('%-46s grid %9d  n=%3d  median %8.1f us  min %8.1f' % (n, g, len(v), v[len(v) // 2], v[0]))

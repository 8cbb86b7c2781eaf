"""Static survey of the device code: for every kernel of every .hip file, the number of vector-memory loads, of
`s_waitcnt vmcnt(0)` (a full drain of the wave's outstanding loads), of branches and of software divisions in the gfx950
disassembly.  A kernel with many loads and about as many full drains is paying one memory latency per load: that pattern
(conditional loads in their own exec-masked branches, prefetch slots issued out of order, `&&` chains compiled to dependent
loads) is how the round-2 transform, epilogue and ROIAlign drain inefficiencies were found - DESIGN.md section 5.1.
CPU only (hipcc cross-compiles):   python tools/isa_survey.py [min_loads]"""
import glob, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, 'chainer-maskrcnn_amd', 'csrc')
min_loads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for f in sorted(glob.glob(os.path.join(SRC, '*.hip'))):
        out = os.path.join(tmp, os.path.basename(f) + '.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off',
                        '-fhip-fp32-correctly-rounded-divide-sqrt', '-I' + os.path.join(R, 'include'), '-I' + SRC, '--cuda-device-only', '-S', f, '-o', out],
                       check=True, stderr=subprocess.DEVNULL)
        name, buf = None, []
        for l in open(out):
            m = re.match(r'^(_Z\w+):', l)
            if m:
                name, buf = m.group(1), []
            elif name:
                buf.append(l)
                if 's_endpgm' in l:
                    body = ''.join(buf)
                    rows.append((os.path.basename(f), name, len(buf), len(re.findall(r'(global_load|buffer_load)', body)), body.count('vmcnt(0)'),
                                 body.count('s_cbranch'), len(re.findall(r'v_rcp_iflag|v_div_scale', body))))
                    name = None


def demangle(n):
    try:
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
    except Exception:
        return n


print('%-14s %-58s %6s %6s %8s %9s %5s' % ('file', 'kernel', 'lines', 'loads', 'vmcnt(0)', 'branches', 'div'))
for r in sorted(rows, key=lambda r: -r[4]):
    if r[3] >= min_loads and r[4] * 2 >= r[3]:
        print('%-14s %-58s %6d %6d %8d %9d %5d' % (r[0], demangle(r[1])[:58], r[2], r[3], r[4], r[5], r[6]))

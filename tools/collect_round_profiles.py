"""Run HERE after `gpurun -- bash tools/round_end.sh`: copies gpurun_out/round_end/* to profiles/rNN_* and stamps the PMC summaries with
the commit they were collected at (the GPU box has no .git)."""
import json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src = os.path.join(R, 'gpurun_out', 'round_end')
commit = subprocess.check_output(['git', '-C', R, 'rev-parse', '--short', 'HEAD']).decode().strip()
dirty = bool(subprocess.check_output(['git', '-C', R, 'status', '--porcelain', '--', 'chainer-maskrcnn_amd', 'include', 'bench.py']).decode().strip())
names = {'bench_step_n1.json': 1, 'bench_roialign_n1.json': 1, 'bench_keypoint_n1.json': 1, 'prof_step_kernel_stats.csv': 'step_kernel_stats.csv',
         'prof_roi_kernel_stats.csv': 'roialign_kernel_stats.csv', 'step_breakdown.txt': 1, 'step_streams.txt': 1, 'step_fill.txt': 1, 'step_pmc_by_kernel.txt': 1, 'gpu_tests.txt': 1,
         'step_pmc_traffic.json': 1, 'conv_pmc_mfma.json': 1, 'roialign_pmc_traffic.json': 1, 'step_in_step_gemm.json': 1, 'host_time.txt': 1}
for n, dst in names.items():
    a = os.path.join(src, n)
    if not os.path.exists(a) or os.path.getsize(a) == 0:
        print('missing', n); continue
    b = os.path.join(R, 'profiles', '%s_%s' % (rnd, n if dst == 1 else dst))
    if n.endswith('pmc_traffic.json') or n.endswith('pmc_mfma.json') or n.endswith('in_step_gemm.json'):
        d = json.load(open(a))
        d['_commit'] = commit + ('+uncommitted' if dirty else '')
        json.dump(d, open(b, 'w'), indent=1)
    else:
        shutil.copyfile(a, b)
    print('->', os.path.relpath(b, R))

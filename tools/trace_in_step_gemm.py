"""GEMM durations INSIDE the multi-stream training step, from a rocprofv3 kernel trace of bench.py (steps delimited by k_sgd): the sum of the
durations of the float32-MFMA GEMM launches (k_conv_igemm<..., 0>) and of the emulated ones (k_conv_igemm<..., 3>, k_pgemm_*) per step,
averaged over the steps after the first `skip`.  JSON on stdout: bench.py's roofline.in_step reads the committed copy under profiles/.
usage: trace_in_step_gemm.py <rocprof dir> [skip=3] [steps=5]"""
import csv, glob, json, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
want = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
sg = [e for e in ev if 'k_sgd' in e[2]]
acc = {'f32_gemm_ms': 0.0, 'emulated_gemm_ms': 0.0, 'gemm_launches': 0, 'kernels': 0, 'wall_ms': 0.0}
n = 0
for k in range(skip, min(skip + want, len(sg) - 1)):
    t0, t1 = sg[k][1], sg[k + 1][1]
    ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    for s, e, name in ks:
        if 'k_pgemm' in name:
            acc['emulated_gemm_ms'] += (e - s) / 1e6; acc['gemm_launches'] += 1
        elif 'k_conv_igemm' in name:
            m = re.search(r'k_conv_igemm<([^>]*)>', name)
            args = [a.strip() for a in m.group(1).split(',')] if m else []
            emu = len(args) >= 5 and args[4] not in ('0',)
            acc['emulated_gemm_ms' if emu else 'f32_gemm_ms'] += (e - s) / 1e6; acc['gemm_launches'] += 1
    acc['kernels'] += len(ks); acc['wall_ms'] += (t1 - t0) / 1e6
    n += 1
out = {k: round(v / max(n, 1), 4) for k, v in acc.items()}
out['steps_averaged'] = n
out['note'] = 'sum of kernel durations inside the profiled multi-stream step (three streams share the chip: a launch here is slower than its standalone replay)'
print(json.dumps(out, indent=1))

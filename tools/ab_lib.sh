#!/bin/bash
# A/B of two BUILDS of the library on one box: tools/ab_lib.sh <base .so> -> gpurun_out/ab_lib/{gemm_base,gemm_new,step}.txt
# (compile-time kernel variants cannot be switched inside one process; step lines alternate base / new twice)
set -o pipefail
R=$PWD; O=$R/gpurun_out/ab_lib; mkdir -p $O
BASE=${1:-chainer-maskrcnn_amd/csrc/ab/libmrcnn_hip_base.so}
MRCNN_HIP_LIB_AB=$BASE python tools/gemm_ab.py base > $O/gemm_base.txt 2>&1
python tools/gemm_ab.py new > $O/gemm_new.txt 2>&1
paste -d'|' $O/gemm_base.txt $O/gemm_new.txt | cut -c1-230
for i in 1 2; do
  MRCNN_HIP_LIB_AB=$BASE python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'], d['value'], d['roofline']['frac'])" | tee -a $O/step.txt
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['value'], d['roofline']['frac'])" | tee -a $O/step.txt
done

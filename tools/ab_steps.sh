#!/bin/bash
# Step-only A/B of two BUILDS of the library on one box (alternating, N rounds): tools/ab_steps.sh <base .so> [rounds]
BASE=${1:-chainer-maskrcnn_amd/csrc/ab/libmrcnn_hip_base.so}; N=${2:-3}
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for i in $(seq $N); do MRCNN_HIP_LIB_AB=$BASE run base; run new; done

#!/bin/bash
# The measurement half of tools/round_end.sh (no test suite): the three bench lines, rocprofv3 kernel statistics / traces of the step and of
# the ROIAlign workload, PMC traffic and MFMA-busy passes, host time.  Outputs under gpurun_out/round_end/.
set -o pipefail
R=$PWD
O=$R/gpurun_out/round_end
mkdir -p $O
python bench.py > $O/bench_step_n1.json 2> $O/bench_step.err; tail -c 400 $O/bench_step_n1.json
python bench.py --workload roialign > $O/bench_roialign_n1.json 2> $O/bench_roialign.err
python bench.py --workload keypoint > $O/bench_keypoint_n1.json 2> $O/bench_keypoint.err
python tools/host_time.py > $O/host_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/prof_step.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_roi --output-format csv -- python3 $R/bench.py --workload roialign --no-cpu-baseline > $O/prof_roi.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/step_pmc_run.py 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/tools/step_pmc_run.py 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/tools/step_pmc_run.py 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_rf -- python3 $R/tools/roi_pmc_run.py 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_rw -- python3 $R/tools/roi_pmc_run.py 5 > /dev/null 2>&1
cd $R
python tools/pmc_roi_traffic.py $O/pmc_rf $O/pmc_rw > $O/roialign_pmc_traffic.json 2> $O/pmc_roi.err
rm -rf $O/pmc_rf $O/pmc_rw
python tools/pmc_step_traffic.py $O/pmc_f $O/pmc_w 2 > $O/step_pmc_traffic.json 2> $O/pmc_traffic.err
python tools/pmc_step_by_kernel.py $O/pmc_f $O/pmc_w 2 > $O/step_pmc_by_kernel.txt 2>&1
python tools/pmc_mfma_summary.py $O/pmc_m 2 > $O/conv_pmc_mfma.json 2> $O/pmc_mfma.err
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m
for d in prof_step prof_roi; do f=$(ls $O/$d/*/*_kernel_stats.csv 2>/dev/null | tail -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
python tools/trace_step.py $O/prof_step 5 > $O/step_breakdown.txt 2>&1
python tools/trace_streams.py $O/prof_step 5 > $O/step_streams.txt 2>&1
python tools/trace_fill.py $O/prof_step 5 > $O/step_fill.txt 2>&1
python tools/trace_in_step_gemm.py $O/prof_step 3 5 > $O/step_in_step_gemm.json 2> $O/in_step.err
rm -rf $O/prof_step $O/prof_roi
ls -la $O

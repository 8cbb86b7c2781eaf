cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wino_order; mkdir -p $O
for v in 1 16; do
  export MRCNN_WINO_ORDER=$v
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/s$v -- python3 $R/tools/ab_tmp.py > /dev/null 2>&1
  echo order $v; python3 $R/tools/ab_tmp2.py $O/s$v; rm -rf $O/s$v
done

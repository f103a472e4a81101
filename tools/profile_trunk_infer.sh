# rocprofv3 kernel statistics of the eval-mode visual extractor, unfused (0) and with folded batch norms (1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4trunk}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f$f -o t -- python3 $R/tools/trunk_infer_bench.py 128 384 $f > $O/f$f.log 2>&1
  rm -f $O/f$f/t_kernel_trace.csv
done

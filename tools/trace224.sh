# kernel statistics + one-step timeline of the 224^2 FineTune step.  usage: bash tools/trace224.sh <outdir> [bench args...]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5t224}
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- python3 $R/bench.py --res 224 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-prof "$@" > $O/stats.log 2>&1
python3 $R/tools/step_timeline.py $O/st/st_kernel_trace.csv > $O/timeline.txt 2>&1
head -30 $O/st/st_kernel_stats.csv | cut -c1-200
head -75 $O/timeline.txt
rm -f $O/st/st_kernel_trace.csv

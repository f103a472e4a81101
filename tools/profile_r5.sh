# round-5 evidence set: rocprofv3 kernel statistics of the default train workload (+ the one-step timeline) and of the decode workload,
# then (mode "all") the PMC passes the traffic / roofline tables are built from (eager steps: the same kernels as the replayed step, 4 executed
# steps per run -- warm-up, timed, two idle-GPU issue-time steps -- which is the divisor tools/traffic_json.py is given).   usage: bash tools/profile_r5.sh <outdir-name> [all]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-decode --no-prof > $O/stats.log 2>&1
python3 $R/tools/step_timeline.py $O/stats/st_kernel_trace.csv > $O/timeline.txt 2>&1
EVK_DECODE_DEPTH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -o ds -- python3 $R/bench.py --workload decode --steps 2 --warmup 1 --no-cpu-baseline > $O/dstats.log 2>&1
if [ "$2" = "all" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-decode --graph 0 > $O/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-decode --graph 0 > $O/write.log 2>&1
  rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-decode --graph 0 > $O/mfma.log 2>&1
fi
rm -f $O/stats/st_kernel_trace.csv.tmp
ls -la $O $O/stats | head -30
du -sh $O
head -60 $O/timeline.txt

# rocprofv3 kernel statistics of the decode workload (one search in flight: EVK_DECODE_DEPTH=1 gives un-overlapped kernel durations)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4dec}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EVK_DECODE_DEPTH=${2:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -o ds -- python3 $R/bench.py --workload decode --steps ${3:-2} --warmup 1 --no-cpu-baseline > $O/dstats.log 2>&1
ls $O/dstats/*
f=$(ls $O/dstats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && f=$(ls $O/dstats/*kernel_stats.csv | head -1)
head -40 $f | cut -c1-200
tail -2 $O/dstats.log | cut -c1-600

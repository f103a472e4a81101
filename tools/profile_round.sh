set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3e}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-decode > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-decode > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-decode > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -o ds -- python3 $R/bench.py --workload decode --steps 2 --warmup 1 --no-cpu-baseline > $O/dstats.log 2>&1
ls -R $O | head -40
du -sh $O

# which hardware queue does each replay lane land on, and what does the step cost?  three runs of the default bench (replayed step), each
# followed by its kernel-trace timeline.  usage: bash tools/replay_probe2.sh <outdir>
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5rp2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof > $O/b$i.json 2> $O/b$i.err
  python3 -c "import json; d=json.load(open('$O/b$i.json')); print('run $i:', round(d['ms_per_step'],2), 'ms', d['config']['step_graph'])"
done
rocprofv3 --kernel-trace --output-format csv -d $O/g -o g -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-decode --no-prof > $O/g.log 2>&1
python3 $R/tools/step_timeline.py $O/g/g_kernel_trace.csv > $O/timeline.txt 2>&1
grep -E "^step window|^queue|time with" $O/timeline.txt
tail -2 $O/g.log | cut -c1-300
rm -rf $O/g

# round 5: what does the replayed step look like on the GPU next to the eager one?  kernel traces of both + timelines.
# usage: bash tools/replay_probe.sh <outdir> [extra env assignments for the replay run, e.g. EVK_MAIN_PRIO=0]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5replay}
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1 > $O/bench_graph.json 2> $O/bench_graph.err && \
python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 0 > $O/bench_eager.json 2> $O/bench_eager.err && \
EVK_EXPERIMENTAL=1 EVK_REPLAY_DEBUG=1 rocprofv3 --kernel-trace --output-format csv -d $O/g -o g -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1 > $O/g.log 2>&1 && \
python3 $R/tools/step_timeline.py $O/g/g_kernel_trace.csv > $O/timeline_graph.txt 2>&1
grep "\[replay\] lane" $O/g.log > $O/lanes.txt
python3 - <<PY
import json
for n in ('graph','eager'):
    d=json.load(open('$O/bench_%s.json'%n)); c=d['config']
    print(n, round(d['ms_per_step'],2), 'ms; host', round(c['host_launch_ms_per_step'],1), round(c['host_loop_ms_per_step'],1), c['step_graph'], c['step_replay_plan'])
PY
head -70 $O/timeline_graph.txt; cat $O/lanes.txt | cut -c1-300
# keep the trace small enough to come home: only the last step's rows matter to the timeline tool
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/g/g_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
keep=rows[-9000:]
w=csv.DictWriter(open('$O/g_tail_trace.csv','w'), fieldnames=['Start_Timestamp','End_Timestamp','Queue_Id','Kernel_Name'])
w.writeheader()
for r in keep: w.writerow({k:r[k] for k in w.fieldnames})
PY
rm -rf $O/g

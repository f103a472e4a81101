# round 5: the step replayer with lanes = capture streams, with / without pacing, against the eager step on the same box.
# usage: bash tools/replay_sweep.sh <outdir> [res] [workload]
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5sweep}
RES=${2:-384}
WL=${3:-finetune}
mkdir -p $O
cd $R
run() {   # name, env...
  name=$1; shift
  env "$@" python3 bench.py --workload $WL --res $RES --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; }
}
python3 bench.py --workload $WL --res $RES --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 0 > $O/eager.json 2> $O/eager.err
G="--graph 1"
for v in "streams:EVK_X=1" "cover:EVK_REPLAY_LANES=cover" "streams_eqprio:EVK_MAIN_PRIO=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py --workload $WL --res $RES --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1 > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; }
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    c=d['config']
    print('%-16s %7.2f ms  host issue %5.1f loop %5.1f  graph %s %s loss %.4f' % (os.path.basename(f)[:-5], d['ms_per_step'], c['host_launch_ms_per_step'], c['host_loop_ms_per_step'], c['step_graph'], c['step_replay_plan'], c['loss_last']))
PY

# replayed FineTune step: priorities of the weight-gradient and text lanes.  usage: bash tools/prio_sweep2.sh <outdir> [res]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5prio2}
RES=${2:-384}
mkdir -p $O
cd $R
B="--res $RES --steps 14 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1"
for v in "base:" "wgrad_low:wgrad:1" "wgrad_high:wgrad:-1" "text_high:text:-1" "wgrad_low_text_high:wgrad:1,text:-1" "all_high:wgrad:-1,text:-1"; do
  name=${v%%:*}; lp=${v#*:}
  EVK_REPLAY_LANE_PRIO="$lp" python3 bench.py $B > $O/$name.json 2> $O/$name.err || tail -3 $O/$name.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    c=d['config']
    print('%-22s %7.2f ms %8.1f studies/s host issue %5.1f loop %5.1f  graph %s' % (os.path.basename(f)[:-5], d['ms_per_step'], d['value'], c['host_launch_ms_per_step'], c['host_loop_ms_per_step'], c['step_graph']))
PY

# stream priorities of the FineTune step, eager and replayed (round 5).  usage: bash tools/prio_sweep.sh <outdir> [res]
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5prio}
RES=${2:-384}
mkdir -p $O
cd $R
B="--res $RES --steps 14 --warmup 4 --no-cpu-baseline --no-decode --no-prof"
python3 bench.py $B --graph 0 > $O/eager_main1_rm0.json 2> $O/e1.err
EVK_RM_STREAM_PRIO=-1 python3 bench.py $B --graph 0 > $O/eager_main1_rm1.json 2> $O/e2.err
EVK_RM_STREAM_PRIO=-1 EVK_MAIN_PRIO=0 python3 bench.py $B --graph 0 > $O/eager_main0_rm1.json 2> $O/e3.err
EVK_MAIN_PRIO=0 python3 bench.py $B --graph 1 > $O/replay_main0_rm0.json 2> $O/r1.err
EVK_RM_STREAM_PRIO=-1 EVK_MAIN_PRIO=0 python3 bench.py $B --graph 1 > $O/replay_main0_rm1.json 2> $O/r2.err
EVK_RM_STREAM_PRIO=-1 EVK_MAIN_PRIO=1 python3 bench.py $B --graph 1 > $O/replay_main1_rm1.json 2> $O/r3.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    c=d['config']
    print('%-22s %7.2f ms %8.1f studies/s host issue %5.1f loop %5.1f  graph %s loss %.4f' % (os.path.basename(f)[:-5], d['ms_per_step'], d['value'], c['host_launch_ms_per_step'], c['host_loop_ms_per_step'], c['step_graph'], c['loss_last']))
PY

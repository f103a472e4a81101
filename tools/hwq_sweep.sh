# does the number of hardware queues the HIP runtime multiplexes streams onto (GPU_MAX_HW_QUEUES, default 4 per priority) explain the erratic
# priority results?  eager and replayed steps at 4 / 8 / 16.   usage: bash tools/hwq_sweep.sh <outdir> [res]
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5hwq}
RES=${2:-384}
mkdir -p $O
cd $R
B="--res $RES --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof"
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=1 python3 bench.py $B --graph 0 > $O/q${q}_eager_main1.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 python3 bench.py $B --graph 0 > $O/q${q}_eager_main0.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --graph 1 > $O/q${q}_replay_main0_lane0.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=1 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --graph 1 > $O/q${q}_replay_main1_lane0.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=1 EVK_REPLAY_RM_PRIO=-1 python3 bench.py $B --graph 1 > $O/q${q}_replay_main1_lane-1.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=-1 python3 bench.py $B --graph 1 > $O/q${q}_replay_main0_lane-1.json 2> $O/e.err || tail -3 $O/e.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-32s %7.2f ms' % (os.path.basename(f)[:-5], d['ms_per_step']))
PY

# repeatability of the replayed step at several hardware-queue counts, default lane priorities.  usage: bash tools/hwq_sweep2.sh <outdir>
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5hwq2}
mkdir -p $O
cd $R
B="--steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof"
for rep in 1 2 3; do
  for q in 4 6 8 12; do
    for mp in 0 1; do
      GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=$mp EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --graph 1 > $O/ft384_q${q}_main${mp}_rep${rep}.json 2> $O/e.err || tail -3 $O/e.err
    done
  done
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --res 224 --graph 1 > $O/ft224_q${q}_main0.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --workload pretrain --res 224 --graph 1 > $O/pt224_q${q}_main0.json 2> $O/e.err || tail -3 $O/e.err
  GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --config 4 --graph 1 > $O/cfg4_q${q}_main0.json 2> $O/e.err || tail -3 $O/e.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-32s %7.2f ms' % (os.path.basename(f)[:-5], d['ms_per_step']))
PY

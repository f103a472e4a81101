# the replayed step with the hardware-queue check (lanes that share a queue are replaced) over GPU_MAX_HW_QUEUES and stream priorities
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5hwq3}
mkdir -p $O
cd $R
B="--steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1"
for q in 4 6 8 12; do
  for mp in 0 1; do
    GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=$mp EVK_REPLAY_RM_PRIO=0 python3 bench.py $B > $O/ft384_q${q}_main${mp}_sep.json 2> $O/e.err || tail -3 $O/e.err
  done
done
EVK_MAIN_PRIO=1 EVK_REPLAY_RM_PRIO=-1 python3 bench.py $B > $O/ft384_q4_main1_rm-1_sep.json 2> $O/e.err || tail -3 $O/e.err
EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=-1 python3 bench.py $B > $O/ft384_q4_main0_rm-1_sep.json 2> $O/e.err || tail -3 $O/e.err
EVK_EXPERIMENTAL=1 EVK_REPLAY_SEPARATE_LANES=0 EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B > $O/ft384_q4_main0_nosep.json 2> $O/e.err || tail -3 $O/e.err
EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --res 224 > $O/ft224_q4_main0_sep.json 2> $O/e.err || tail -3 $O/e.err
EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --workload pretrain --res 224 > $O/pt224_q4_main0_sep.json 2> $O/e.err || tail -3 $O/e.err
EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 python3 bench.py $B --config 4 > $O/cfg4_q4_main0_sep.json 2> $O/e.err || tail -3 $O/e.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-32s %7.2f ms  plan %s' % (os.path.basename(f)[:-5], d['ms_per_step'], {k:v for k,v in (d['config']['step_replay_plan'] or {}).items() if k in ('lanes','lane_streams_replaced')}))
PY

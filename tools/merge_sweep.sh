# replayed FineTune step: which capture streams share an in-order lane?  usage: bash tools/merge_sweep.sh <outdir> [res]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5merge}
RES=${2:-384}
mkdir -p $O
cd $R
B="--res $RES --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1"
for q in 4 8; do
  for v in "none:" "rm_wgrad:rm:wgrad" "rm_main:rm:main" "text_main:text:main" "rm_text:rm:text" "text_wgrad:text:wgrad" "rm_wgrad_text_main:rm:wgrad,text:main" "rm_text_wgrad:rm:wgrad,text:wgrad" "wgrad_main:wgrad:main"; do
    name=${v%%:*}; mg=${v#*:}
    GPU_MAX_HW_QUEUES=$q EVK_MAIN_PRIO=0 EVK_REPLAY_RM_PRIO=0 EVK_REPLAY_MERGE="$mg" python3 bench.py $B > $O/q${q}_$name.json 2> $O/e.err || tail -3 $O/e.err
  done
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-32s %7.2f ms  lanes %s' % (os.path.basename(f)[:-5], d['ms_per_step'], (d['config']['step_replay_plan'] or {}).get('lanes')))
PY

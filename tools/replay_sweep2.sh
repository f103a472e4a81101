# eager vs replayed step (lanes = capture streams) on the host-bound configurations.   usage: bash tools/replay_sweep2.sh <outdir>
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5sweep2}
mkdir -p $O
cd $R
for cfg in "ft224:--res 224" "pt224:--workload pretrain --res 224" "cfg4:--config 4" "pt384:--workload pretrain --res 384"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python3 bench.py $args --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 0 > $O/${name}_eager.json 2> $O/${name}_eager.err || tail -5 $O/${name}_eager.err
  python3 bench.py $args --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1 > $O/${name}_replay.json 2> $O/${name}_replay.err || tail -5 $O/${name}_replay.err
  EVK_MAIN_PRIO=0 python3 bench.py $args --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1 > $O/${name}_replay_eqprio.json 2> $O/${name}_replay_eqprio.err || tail -5 $O/${name}_replay_eqprio.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    c=d['config']
    print('%-22s %7.2f ms %8.1f studies/s host issue %5.1f loop %5.1f  graph %s %s loss %.4f' % (os.path.basename(f)[:-5], d['ms_per_step'], d['value'], c['host_launch_ms_per_step'], c['host_loop_ms_per_step'], c['step_graph'], (c['step_replay_plan'] or {}).get('lanes'), c['loss_last']))
PY
grep -h "Warning\|warn" $O/*.err | sort | uniq -c | head

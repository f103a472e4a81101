# stream separation: the default (replayed) line three times, the eager step with separated side streams, the decode workload with / without
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5sep}
mkdir -p $O
cd $R
B="--steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof"
for i in 1 2 3; do python3 bench.py $B > $O/default_rep$i.json 2> $O/e.err || tail -3 $O/e.err; done
python3 bench.py $B --graph 0 > $O/eager.json 2> $O/e.err || tail -3 $O/e.err
EVK_SEPARATE_SIDE_STREAMS=1 python3 bench.py $B --graph 0 > $O/eager_sepside.json 2> $O/e.err || tail -3 $O/e.err
EVK_SEPARATE_SIDE_STREAMS=1 EVK_MAIN_PRIO=0 python3 bench.py $B --graph 0 > $O/eager_sepside_main0.json 2> $O/e.err || tail -3 $O/e.err
python3 bench.py $B --res 224 > $O/ft224_default.json 2> $O/e.err || tail -3 $O/e.err
python3 bench.py --workload decode --steps 6 --warmup 2 --no-cpu-baseline > $O/decode_sep.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_SEPARATE_STREAMS=0 python3 bench.py --workload decode --steps 6 --warmup 2 --no-cpu-baseline > $O/decode_nosep.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_DEPTH=3 python3 bench.py --workload decode --steps 6 --warmup 2 --no-cpu-baseline > $O/decode_sep_depth3.json 2> $O/e.err || tail -3 $O/e.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-26s %9.2f %s  %7.2f ms  %s' % (os.path.basename(f)[:-5], d['value'], d['unit'], d['ms_per_step'], (d['config'].get('step_replay_plan') or '')))
PY

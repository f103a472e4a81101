# replayed step: priority of the main stream x the relational-memory LANE x the eager relational-memory STREAM (does an idle high-priority stream matter?)
set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5prio3}
mkdir -p $O
cd $R
B="--res 384 --steps 12 --warmup 4 --no-cpu-baseline --no-decode --no-prof --graph 1"
for mp in 1 0; do for lp in -1 0; do for sp in 0 -1; do
  EVK_MAIN_PRIO=$mp EVK_REPLAY_RM_PRIO=$lp EVK_RM_STREAM_PRIO=$sp python3 bench.py $B > $O/main${mp}_lane${lp}_stream${sp}.json 2> $O/e.err || tail -3 $O/e.err
done; done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-28s %7.2f ms' % (os.path.basename(f)[:-5], d['ms_per_step']))
PY

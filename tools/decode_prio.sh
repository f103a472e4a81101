set -x
export EVK_EXPERIMENTAL=1          # the switches below select measured alternatives: honoured only under this flag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5dprio}
mkdir -p $O
cd $R
B="--workload decode --steps 6 --warmup 2 --no-cpu-baseline"
python3 bench.py $B > $O/prio_hi_hi.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_PRIO=0,0 python3 bench.py $B > $O/prio_0_0.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_PRIO=-1,0 python3 bench.py $B > $O/prio_hi_0.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_PRIO=0,0,0 EVK_DECODE_DEPTH=3 python3 bench.py $B > $O/prio_0_0_0_depth3.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_PRIO=0 EVK_DECODE_DEPTH=1 python3 bench.py $B > $O/prio_0_depth1.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_DEPTH=1 python3 bench.py $B > $O/prio_hi_depth1.json 2> $O/e.err || tail -3 $O/e.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-26s %9.1f %s  %7.2f ms/batch  per-search step %.3f ms' % (os.path.basename(f)[:-5], d['value'], d['unit'], d['ms_per_step'], d['roofline']['per_search_step_ms']))
PY

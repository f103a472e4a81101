set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5dprio2}
mkdir -p $O
cd $R
B="--workload decode --steps 8 --warmup 2 --no-cpu-baseline"
python3 bench.py $B > $O/depth3_a.json 2> $O/e.err || tail -3 $O/e.err
python3 bench.py $B > $O/depth3_b.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_DEPTH=4 python3 bench.py $B > $O/depth4.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_DEPTH=2 python3 bench.py $B > $O/depth2.json 2> $O/e.err || tail -3 $O/e.err
EVK_DECODE_DEPTH=3 EVK_DECODE_THREADS=0 python3 bench.py $B > $O/depth3_nothreads.json 2> $O/e.err || tail -3 $O/e.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-26s %9.1f %s  %7.2f ms/batch  per-search step %.3f ms  emitted %.1f' % (os.path.basename(f)[:-5], d['value'], d['unit'], d['ms_per_step'], d['roofline']['per_search_step_ms'], d.get('emitted_tokens_per_s', 0)))
PY

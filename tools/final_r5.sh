# round-5 final measurement pass on the committed sources: the default line, the eager line of the same box, the other BASELINE configs in both
# builds, the forced-dist lines, then the profiling set.   usage: bash tools/final_r5.sh <outdir>
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5final}
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/default.json 2> $O/default.err
python3 bench.py --steps 20 --warmup 5 --graph 0 --no-decode --no-cpu-baseline > $O/default_eager.json 2> $O/default_eager.err
bash tools/bench_configs_r5.sh ${1:-r5final}/cfg > $O/cfg.log 2>&1
bash tools/forced_dist_bench.sh ${1:-r5final}/fd > $O/fd.log 2>&1
python3 - <<PY
import json,glob,os
for f in ['$O/default.json','$O/default_eager.json']+sorted(glob.glob('$O/cfg/*.json'))+sorted(glob.glob('$O/fd/*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    c=d['config']
    print('%-26s %8.1f %s %7.2f ms  %s frac %.4f  graph %s  %s' % (os.path.basename(f)[:-5], d['value'], d['unit'], d['ms_per_step'], d['dtype'], d.get('roofline',{}).get('frac',0), c.get('step_graph'), (c.get('grad_sync') or '')[:12]))
    if 'decode' in d: print('    decode', round(d['decode']['value'],1), d['decode']['ms_per_batch'], d['decode'].get('parity',{}).get('identical_sequences'))
PY

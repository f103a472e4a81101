set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5dsw4}
mkdir -p $O
cd $R
B="--workload decode --steps 8 --warmup 2 --no-cpu-baseline"
for v in "d4_nothreads_a:EVK_DECODE_THREADS=0 EVK_DECODE_DEPTH=4" "d5_nothreads:EVK_DECODE_THREADS=0 EVK_DECODE_DEPTH=5" "d6_nothreads:EVK_DECODE_THREADS=0 EVK_DECODE_DEPTH=6" "d4_nothreads_b:EVK_DECODE_THREADS=0 EVK_DECODE_DEPTH=4" "d5_threads:EVK_DECODE_DEPTH=5" "d4_threads_a1:EVK_DECODE_DEPTH=4 EVK_DECODE_AHEAD=1"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py $B > $O/$name.json 2> $O/e.err || tail -3 $O/e.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json'), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable'); continue
    print('%-22s %9.1f %s  %7.2f ms/batch  per-search step %.3f ms' % (os.path.basename(f)[:-5], d['value'], d['unit'], d['ms_per_step'], d['roofline']['per_search_step_ms']))
PY

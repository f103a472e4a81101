set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5dsw3}
mkdir -p $O
cd $R
B="--workload decode --steps 8 --warmup 2 --no-cpu-baseline"
for v in "d3_b8_a2:EVK_X=1" "d3_b16_a2:EVK_DECODE_BURST=16" "d3_b4_a2:EVK_DECODE_BURST=4" "d3_b8_a1:EVK_DECODE_AHEAD=1" "d3_b8_a4:EVK_DECODE_AHEAD=4" "d4_b8_a2:EVK_DECODE_DEPTH=4" "d3_nothreads:EVK_DECODE_THREADS=0" "d4_nothreads:EVK_DECODE_THREADS=0 EVK_DECODE_DEPTH=4" "d3_b8_a2_again:EVK_X=2"; do
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

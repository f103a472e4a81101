# the default workload with the reducer ACTIVE (1-rank RCCL group, EVK_FORCE_DIST=1: every bucket's collective, the
# update-mask exchange) next to the default line, in the three gradient-sum modes.   usage: bash tools/forced_dist_bench.sh <outdir>
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5fd}
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline > $O/default.json 2> $O/default.err && \
EVK_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline > $O/forced_allreduce.json 2> $O/forced_allreduce.err && \
EVK_FORCE_DIST=1 EVK_GRAD_SYNC=direct python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline > $O/forced_direct.json 2> $O/forced_direct.err && \
EVK_FORCE_DIST=1 EVK_GRAD_SYNC=16bit python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline > $O/forced_16bit.json 2> $O/forced_16bit.err && \
EVK_FORCE_DIST=1 python3 bench.py --workload pretrain --res 224 --steps 20 --warmup 5 --no-cpu-baseline > $O/forced_pt224.json 2> $O/forced_pt224.err
echo rc=$?
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    c=d['config']
    print(f.split('/')[-1], round(d['value'],1), 'studies/s', round(d['ms_per_step'],2), 'ms; host', round(c['host_launch_ms_per_step'],1), round(c['host_loop_ms_per_step'],1), '|', c['grad_sync'][:40], '|', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (c.get('comm') or {}).items() if k!='bucket_bytes'})
PY
tail -3 $O/*.err

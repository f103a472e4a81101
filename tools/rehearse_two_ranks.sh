# bench.py's N = 2 branch on ONE GPU: two ranks share the device, gloo carries the collectives (EVK_DIST_BACKEND=gloo).  A rehearsal of the code
# path the driver's scaling run takes (torch.distributed.run, barriers, max-over-ranks, reducer active, comm statistics, replicated decode);
# the numbers mean nothing.   usage: bash tools/rehearse_two_ranks.sh <outdir>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5two}
mkdir -p $O
cd $R
EVK_DIST_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 6 --warmup 3 > $O/ft384_two_ranks.json 2> $O/ft384.err
echo rc=$?
EVK_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 6 --warmup 3 --config 4 > $O/cfg4_two_ranks.json 2> $O/cfg4.err
echo rc=$?
python3 - <<PY
import json
for n in ('ft384_two_ranks','cfg4_two_ranks'):
    try:
        d=json.load(open('$O/%s.json' % n)); c=d['config']
        print(n, 'n_gpus', d['n_gpus'], round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms | sync', c['grad_sync'], '| backend', c['backend'], '| comm', {k:(round(v,2) if isinstance(v,float) else v) for k,v in (c['comm'] or {}).items() if k!='bucket_bytes'}, '| decode', (d.get('decode') or {}).get('value'))
    except Exception as e:
        print(n, 'unreadable:', e)
PY
tail -5 $O/ft384.err | cut -c1-300

set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5rmp}
mkdir -p $O
cd $R
python3 tools/rm_probe.py 32 100 5 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/tools/rm_probe.py 32 100 3 > $O/t.log 2>&1
python3 - <<PY
import csv, re
from collections import defaultdict
agg=defaultdict(lambda:[0,0,10**12,0])
for r in csv.DictReader(open('$O/t/t_kernel_trace.csv')):
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'\(.*','',n)[:40]
    key=(n, r.get('Grid_Size_X') or r.get('Grid_Size') or '?', r.get('Workgroup_Size_X') or '?')
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    a=agg[key]; a[0]+=1; a[1]+=d; a[2]=min(a[2],d); a[3]=max(a[3],d)
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:24]:
    print('%-42s grid %8s wg %4s  n=%5d  avg %7.2f us  min %7.2f  max %7.2f' % (k[0],k[1],k[2],a[0],a[1]/a[0]/1e3,a[2]/1e3,a[3]/1e3))
PY
rm -rf $O/t

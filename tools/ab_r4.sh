mkdir -p gpurun_out/r4a
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  EVK_GEMM_TN=$1 EVK_RM_F32=$2 python bench.py --no-decode --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r4a/ab_$1_$2.json 2>/dev/null
  python -c "
import json,sys; d=json.load(open('gpurun_out/r4a/ab_$1_$2.json')); r=d['roofline']; print('TN=$1 RMF32=$2', round(d['ms_per_step'],2), round(d['config']['host_launch_ms_per_step'],1), round(r['gemm_ms_per_step'],2), r['launches_per_step'])"
done

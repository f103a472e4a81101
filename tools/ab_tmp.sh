set -e
timeout -k 10 500 python -m pytest tests/test_hip_gemm.py -x -q -m gpu -k "weight_stationary" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "trunk or golden or finetune or pretrain" 2>&1 | tail -3
for g in 0 1 0 1; do
  echo "== bench EVK_BN_XSTATS=$g"
  EVK_BN_XSTATS=$g python bench.py --no-cpu-baseline --no-decode --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss_last'])"
done

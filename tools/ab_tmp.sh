set -e
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "beam or decode or gpt2" 2>&1 | tail -3
for t in 0 1 0 1; do
  echo "== EVK_DECODE_FUSED_APPEND=$t"
  EVK_DECODE_FUSED_APPEND=$t python bench.py --workload decode --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_batch'], d['roofline']['step_ms'])"
done

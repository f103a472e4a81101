for e in "A=1" "EVK_DECODE_SPLIT_CLN=0" "EVK_DECODE_SPLIT_CLN=0 EVK_DECODE_FUSED_APPEND=0"; do
  echo "== $e"
  env $e python bench.py --workload decode --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('parity'))"
done

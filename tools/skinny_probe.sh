R=$GRAFT_REPO_ROOT
cd $R
B="--steps 12 --warmup 4 --no-decode --no-cpu-baseline --no-prof"
for v in "default:EVK_X=0" "deep0_256_32_128:EVK_EXPERIMENTAL=1 EVK_SKINNY_DEEP=0" "tm128_512_16_128:EVK_EXPERIMENTAL=1 EVK_SKINNY_TM64=0" "default_again:EVK_X=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-20s %7.2f ms/step  graph %s' % ('$name', d['ms_per_step'], d['config']['step_graph']))"
done
for v in "default:EVK_X=0" "deep0_256_32_128:EVK_EXPERIMENTAL=1 EVK_SKINNY_DEEP=0" "tm128_512_16_128:EVK_EXPERIMENTAL=1 EVK_SKINNY_TM64=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py --workload decode --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('decode %-20s %9.1f tokens/s' % ('$name', d['value']))"
done

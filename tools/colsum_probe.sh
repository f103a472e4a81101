R=$GRAFT_REPO_ROOT
cd $R
B="--steps 12 --warmup 4 --no-decode --no-cpu-baseline --no-prof"
for v in "blocks768:EVK_X=0" "blocks256:EVK_EXPERIMENTAL=1 EVK_COLSUM_BLOCKS=256" "blocks96:EVK_EXPERIMENTAL=1 EVK_COLSUM_BLOCKS=96" "blocks32:EVK_EXPERIMENTAL=1 EVK_COLSUM_BLOCKS=32" "blocks768_again:EVK_X=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %7.2f ms/step  graph %s' % ('$name', d['ms_per_step'], d['config']['step_graph']))"
done

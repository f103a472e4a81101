R=$GRAFT_REPO_ROOT
cd $R
B="--steps 12 --warmup 4 --no-decode --no-cpu-baseline --no-prof"
for v in "full:EVK_X=0" "no_splitk_reduce:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_SPLITK_REDUCE=1" "no_colsum:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_COLSUM=1" "no_reduce_no_colsum:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_SPLITK_REDUCE=1 EVK_PROBE_SKIP_COLSUM=1" "full_again:EVK_X=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %7.2f ms/step  graph %s' % ('$name', d['ms_per_step'], d['config']['step_graph']))"
done

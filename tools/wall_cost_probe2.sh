R=$GRAFT_REPO_ROOT
cd $R
B="--steps 12 --warmup 4 --no-decode --no-cpu-baseline --no-prof"
for v in "full:EVK_X=0" "no_rm_fwd:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_RM_FWD=1" "no_rm_fwd_bwd:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_RM_FWD=1 EVK_PROBE_SKIP_RM_BWD=1" "persist_bwd:EVK_EXPERIMENTAL=1 EVK_RM_PERSIST=1" "full_again:EVK_X=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %7.2f ms/step  graph %s loss %s' % ('$name', d['ms_per_step'], d['config']['step_graph'], d['config']['loss_last']))"
done

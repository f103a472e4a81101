# what do the side lanes cost in WALL time?  the replayed default with the trunk's weight gradients / the recurrence's BPTT skipped (timing only)
R=$GRAFT_REPO_ROOT
cd $R
B="--steps 12 --warmup 4 --no-decode --no-cpu-baseline --no-prof"
for v in "full:EVK_X=0" "no_trunk_wgrad:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_WGRAD=1" "no_rm_bwd:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_RM_BWD=1" "neither:EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_WGRAD=1 EVK_PROBE_SKIP_RM_BWD=1"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %7.2f ms/step  graph %s' % ('$name', d['ms_per_step'], d['config']['step_graph']))"
  env $envs python3 bench.py $B --graph 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %7.2f ms/step  graph %s' % ('$name', d['ms_per_step'], d['config']['step_graph']))"
done

# round 5: the replayed default against the eager step on ONE box, alternating, three runs each (bench.py --steps 20 --warmup 5, no decode / CPU legs)
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do
  for g in 1 0; do
    python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline --no-prof --graph $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('run $i  %-8s %7.2f ms/step  %7.1f studies/s  host issue %5.1f ms  guard %s' % ('replayed' if c['step_graph'] else 'eager', d['ms_per_step'], d['value'], c['host_launch_ms_per_step'], c['step_graph_guard']))"
  done
done

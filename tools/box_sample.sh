# one sample of the default line (replayed) and the eager step on THIS box (each gpurun call lands on a fresh box)
R=$GRAFT_REPO_ROOT
cd $R
for g in 1 0; do
  python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline --no-prof --graph $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('%-8s %7.2f ms/step  %7.1f studies/s' % ('replayed' if c['step_graph'] else 'eager', d['ms_per_step'], d['value']))"
done

"""cProfile of the host side of the training step (where the Python launch time goes)."""
import cProfile
import io
import pstats
import sys

sys.argv = ['bench.py', '--steps', '6', '--warmup', '3', '--no-prof', '--no-cpu-baseline']
sys.path.insert(0, '.')
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats('cumulative').print_stats(60)
st.sort_stats('tottime').print_stats(45)
print(s.getvalue())

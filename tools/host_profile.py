"""cProfile of the host side of a bench workload: python tools/host_profile.py [finetune|pretrain] [res]"""
import cProfile
import io
import pstats
import sys

wl = sys.argv[1] if len(sys.argv) > 1 else 'finetune'
res = sys.argv[2] if len(sys.argv) > 2 else '384'
sys.argv = ['bench.py', '--workload', wl, '--res', res, '--steps', '20', '--warmup', '3', '--no-prof', '--no-cpu-baseline']
sys.path.insert(0, '.')
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats('tottime').print_stats(70)
st.print_callers("method 'to' of")
st.print_callers("method 'contiguous' of")
print(s.getvalue())

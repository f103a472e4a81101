"""Run bench.py with most of the HBM taken by a ballast tensor (how does the serving loop behave on a GPU that is short of memory?):
usage: python tools/ballast_run.py <free GB to leave> <bench.py arguments...>"""
import os
import runpy
import sys
import torch

leave = float(sys.argv[1])
free, total = torch.cuda.mem_get_info(0)
ballast = torch.empty(max(0, int(free - leave * (1 << 30))), dtype=torch.uint8, device='cuda:0')
print('[ballast] %.1f GB held, %.1f GB left of %.1f' % (ballast.numel() / 2**30, torch.cuda.mem_get_info(0)[0] / 2**30, total / 2**30), file=sys.stderr)
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')

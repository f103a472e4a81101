"""profiles/*_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1 --no-prof
--no-cpu-baseline`: usage traffic_json.py fetch.csv write.csv steps > out.json  (FETCH doubled on gfx950, KB -> bytes).
The file is stamped with evoke_amd.build.source_fingerprint(): bench.py reports it as `roofline.traffic` only while it matches."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from traffic_agg import load
from evoke_amd.build import source_fingerprint

fe, cnt = load(sys.argv[1], 2.0)
wr, _ = load(sys.argv[2], 1.0)
steps = float(sys.argv[3])
# the GEMM / convolution family: the tile kernels of gemm.hip, the strip GEMM, the halo-tile 3x3 and stem kernels (forward / data gradient and
# weight gradient: they took over launches of gemm_kernel in round 3) and the split-K reductions
FAMILY = ('void gemm_', 'gemm_strip_kernel', 'gemm_tn_kernel', 'gemm_f32_kernel', 'void conv3x3_halo_kernel', 'conv3x3_halo_kernel', 'wgk::conv3x3_wgrad_halo_kernel', 'stem_fwd_kernel', 'stem_wgrad_kernel', 'splitk_reduce')
fam = lambda n: n.startswith(FAMILY)
names = sorted(set(fe) | set(wr), key=lambda n: -(fe.get(n, 0) + wr.get(n, 0)))
out = {
    'fingerprint': source_fingerprint(), 'workload': ['finetune', 384, 32, 2], 'store': os.environ.get('EVK_STORE', 'f16').lower(),
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python bench.py --steps 1 --warmup 1 '
              '--no-prof --no-cpu-baseline (finetune 384^2, 32 studies); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies '
              '128-B requests at 64 B); KB -> bytes x1024; divided by the %g steps of the run (includes one-off initialisation)' % steps,
    'gemm_family': list(FAMILY),
    'gemm_family_hbm_bytes_per_step': sum(fe.get(n, 0) + wr.get(n, 0) for n in names if fam(n)) / steps,
    'gemm_family_fetch_bytes_per_step': sum(fe.get(n, 0) for n in names if fam(n)) / steps,
    'gemm_family_write_bytes_per_step': sum(wr.get(n, 0) for n in names if fam(n)) / steps,
    'all_kernels_hbm_bytes_per_step': sum(fe.get(n, 0) + wr.get(n, 0) for n in names) / steps,
    'per_kernel_GB_per_step': {n: {'calls': cnt.get(n, 0) / steps, 'fetch': round(fe.get(n, 0) / 1e9 / steps, 3),
                                   'write': round(wr.get(n, 0) / 1e9 / steps, 3)} for n in names[:40]},
}
print(json.dumps(out, indent=1))

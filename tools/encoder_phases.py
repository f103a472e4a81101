"""The encoders of the decode workload (visual extractor + fusion + indication text encoder of one 64-study batch) alone on the GPU: wall
time per call and the library's per-launch records (family, ms, GEMM shape) for one call.   usage: python tools/encoder_phases.py [dump.csv]"""
import collections
import csv
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench                                       # noqa: E402
from evoke_amd import hip as H                     # noqa: E402
from evoke_amd.model_pretrain_finetune import FineTune          # noqa: E402
from evoke_amd.config import load_default_tokenizer            # noqa: E402

dump = sys.argv[1] if len(sys.argv) > 1 else '/tmp/enc_launches.csv'
dev = torch.device('cuda:0')
torch.manual_seed(9233)
model = FineTune(bench.make_args('test'), load_default_tokenizer(), 'mimic_cxr').to(dev)
model.train()
wb = bench.synth_batch('finetune', 16, 2, 384, 100, 30, dev, 3000)
with torch.no_grad():
    for _ in range(40):
        model.visual_extractor(wb['images'])
model.eval()
b = bench.synth_batch('finetune', 64, 2, 384, 100, 30, dev, 2000)
with torch.no_grad():
    for _ in range(3):
        model.encoder_states(b['images'], b['pids'], 64, b['inc'], b['inc_masks'])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        model.encoder_states(b['images'], b['pids'], 64, b['inc'], b['inc_masks'])
    e1.record()
    torch.cuda.synchronize()
    print('encoders of one batch (%d images): %.2f ms' % (b['images'].shape[0], e0.elapsed_time(e1) / 10))
    e0.record()
    for _ in range(10):
        model.visual_extractor(b['images'])
    e1.record()
    torch.cuda.synchronize()
    print('  of which visual extractor: %.2f ms' % (e0.elapsed_time(e1) / 10))
    H.prof_enable(True)
    model.encoder_states(b['images'], b['pids'], 64, b['inc'], b['inc_masks'])
    torch.cuda.synchronize()
    H._dump_path = dump.encode()
    H.check(H.lib.evk_prof_dump_to(H._dump_path))
    fam, _ = H.prof_collect()
    H.prof_enable(False)
print('families (ms, launches):', {k: (round(v[0], 2), v[1]) for k, v in fam.items() if v[1]} if isinstance(fam, dict) else fam)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(dump)):
    key = (r['family'], r['M'], r['N'], r['K'], r['batch'], r['a_mode'], r['b_mode'])
    agg[key][0] += 1
    agg[key][1] += float(r['ms'])
print('family M N K batch a_mode b_mode : launches, ms')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    fl = 2.0 * int(k[1]) * int(k[2]) * int(k[3]) * max(1, int(k[4]))
    print('  %-44s n=%3d %7.3f ms  %6.0f TF/s' % (' '.join(k), v[0], v[1], fl * v[0] / max(v[1], 1e-9) / 1e9))

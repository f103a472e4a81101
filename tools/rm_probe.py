"""The relational-memory recurrence of the training step alone on the GPU: B samples x L tokens, forward (f32 recurrence) and backward (16-bit BPTT)
through RelationalMemory.run, HIP-event timed.  usage: python tools/rm_probe.py [B] [L] [reps]
Under `rocprofv3 --kernel-trace --stats` the per-kernel table shows what one token costs."""
import sys
import time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from evoke_amd import ops
from evoke_amd.layers import RelationalMemory

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
rm = RelationalMemory(3, 512, 8).cuda().train()
ops.set_dropout_enabled(False)
emb = (torch.randn(B, L, 512, device='cuda') * 0.5).to(ops.BF16).requires_grad_(True)
emb.evk_f32 = emb.detach().float()


def once():
    out = rm(emb)
    return out


for _ in range(2):
    o = once()
    o.float().sum().backward()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(reps):
    ev[0].record()
    o = once()
    ev[1].record()
    g = torch.ones_like(o)
    torch.cuda.synchronize()
    ev[1].record()
    o.backward(g)
    ev[2].record()
    torch.cuda.synchronize()
    tb += ev[1].elapsed_time(ev[2])
    # forward timed separately (the launches of the hoisted projections included)
    ev[0].record()
    with torch.no_grad():
        o2 = once()
    ev[1].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1])
print('relational memory B=%d L=%d: forward %.3f ms (%.1f us / token), backward %.3f ms (%.1f us / token)' % (
    B, L, tf / reps, 1e3 * tf / reps / L, tb / reps, 1e3 * tb / reps / L))

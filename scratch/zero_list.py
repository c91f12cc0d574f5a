"""Which gradient buffers of a training plan still take the per-step zero pass (engine.Plan.zero_grads)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from common import batch_args, esf_module
from egne_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "baseline_edge"
B = 8
b = synth.make_batch(B, seed=1)
m = esf_module(cfg).cuda().train()
edge = torch.rand(B, 1, 240, 320)
args = [a.cuda() if torch.is_tensor(a) else a for a in batch_args(b, edge)]
m(*args)[3].sum().backward()
pl = m._last_plan
tot = free = 0
rows = []
for bid, t in pl.gtwins.items():
    n = t.numel() * 4
    tot += n
    z = bid in pl._zero_free
    free += n if z else 0
    ents = pl._touched.get(bid, [])
    rows.append((n, tuple(t.shape), z, len(ents), [e for e in ents if not e[2]][:3]))
rows.sort(reverse=True)
for r in rows[:28]:
    print("%8.1f MB %-22s zero-free=%-5s touches=%d first non-store=%s" % (r[0] / 1e6, r[1], r[2], r[3], r[4]))
print("total %.1f MB, zero-free %.1f MB (%.0f %%)" % (tot / 1e6, free / 1e6, 100 * free / tot))

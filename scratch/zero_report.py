"""Which gradient twins of the B=64 bf16 training plan still take the zero pass (egne_zero_many), and their accesses."""
import sys, os, types
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from common import batch_args, esf_module
from egne_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = sys.argv[2] if len(sys.argv) > 2 else "baseline_edge"
DEV = "cuda:0"
from egne_amd import engine as _E
_sites, _buf0 = {}, _E.Plan.buf
def _buf(self, *a):
    t = _buf0(self, *a)
    f = sys._getframe(1)
    _sites[id(t)] = "%s:%d" % (os.path.basename(f.f_code.co_filename), f.f_lineno)
    if f.f_code.co_name in ("concat_members", "slices", "<listcomp>"):
        f2 = f.f_back.f_back if f.f_code.co_name == "<listcomp>" else f.f_back
        _sites[id(t)] += " <- %s:%d" % (os.path.basename(f2.f_code.co_filename), f2.f_lineno)
    return t
_E.Plan.buf = _buf
b = synth.make_batch(B, seed=1)
edge = torch.rand(B, 1, 240, 320)
m = esf_module(cfg, seed=11).to(DEV).to(torch.bfloat16).train()
loss = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])[3]
loss.sum().backward(); torch.cuda.synchronize()
pl = m._last_plan
tot = 0
names = {}
for k, v in pl.dbg.items():
    if k == "D":
        for i, d in enumerate(v):
            for kk, pc in d.items(): names[id(pc.buf)] = "D[%d].%s" % (i, kk)
    elif torch.is_tensor(v): names[id(v)] = k
for bid, t in pl.gtwins.items():
    free = bid in pl._zero_free
    nb = t.numel() * t.element_size()
    if not free:
        tot += nb
        print("%-14s %-34s %-28s %8.1f MB  accesses %s" % (names.get(bid, "?"), _sites.get(bid, "?"), tuple(t.shape), nb / 1e6, pl._touched.get(bid)))
print("zeroed per step: %.1f MB of %.1f MB of twins" % (tot / 1e6, sum(t.numel() * t.element_size() for t in pl.gtwins.values()) / 1e6))

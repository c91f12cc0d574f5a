"""f16 storage under a weight update and under an overflow: the plan of a module whose weights were changed in place (x3: inside the head-room;
x200: beyond it, sticky overflow word -> recalibration) against a fresh module with the same weights."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import egne_amd
from egne_amd import engine, synth
from common import bdcn_module
DEV = 'cuda:0'
B = 64
x = torch.cat((synth.make_batch(B, seed=5)["img"],) * 3, 1).to(DEV)
bd = bdcn_module().to(DEV); bd.f16_products = 1
o0 = bd.forward_fuse(x); torch.cuda.synchronize()
print("first run: overflowed", bd.overflowed(), "storage level", bd._last_plan.f16_storage)
for fac in (3.0, 200.0):
    with torch.no_grad():
        bd.features.conv1_1.weight.mul_(fac); bd.features.conv1_1.bias.mul_(fac)
    o1 = bd.forward_fuse(x); torch.cuda.synchronize()
    ov = bd.overflowed()
    tries = 0
    while ov and tries < 3:
        o1 = bd.forward_fuse(x); torch.cuda.synchronize(); ov = bd.overflowed(); tries += 1
    fresh = bdcn_module().to(DEV); fresh.f16_products = 1
    fresh.load_state_dict(bd.state_dict())
    o2 = fresh.forward_fuse(x); torch.cuda.synchronize()
    print("weights x%g: reruns after overflow %d, finite %s, equal to a fresh module's plan %s (max diff %.3e)" % (
        fac, tries, bool(torch.isfinite(o1).all()), torch.equal(o1, o2), (o1 - o2).abs().max().item()))
# frames far beyond the calibrated head-room: stored halves overflow -> sticky word -> the next call re-calibrates
bd2 = bdcn_module().to(DEV); bd2.f16_products = 1
bd2.forward_fuse(x); torch.cuda.synchronize(); bd2.overflowed()
xl = x * 3000.0
o1 = bd2.forward_fuse(xl); torch.cuda.synchronize()
ov, tries = bd2.overflowed(), 0
while ov and tries < 3:
    o1 = bd2.forward_fuse(xl); torch.cuda.synchronize(); ov = bd2.overflowed(); tries += 1
fresh = bdcn_module().to(DEV); fresh.f16_products = 1
o2 = fresh.forward_fuse(xl); torch.cuda.synchronize()
print("frames x3000: reruns after overflow %d, finite %s, equal to a fresh module's plan %s (max diff %.3e)" % (tries, bool(torch.isfinite(o1).all()), torch.equal(o1, o2), (o1 - o2).abs().max().item()))

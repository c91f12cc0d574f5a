"""Is the 3e-3..9e-3 deviation of the deepest gradients fp32 noise?  Compare reference-fp32 (golden), HIP-fp32 and the
oracle run in float64 (taken as truth)."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np, types
from common import *
from egne_amd import synth
from egne_amd.utils import calc_edge
from oracle import esfnet as oesf, bdcn as obdcn
torch.set_num_threads(32)
DEV='cuda:0'
bd = bdcn_module()
name = "esf_edge_b2_absent1"
cfg, variant, kw = ESF_CASES[name]; kw=dict(kw)
g = gold(name)
b = synth.make_batch(kw.pop("B"), **kw)
edge = obdcn.calc_edge(bd.state_dict(), b["img"])
m = esf_module(cfg, variant)
sd = {k: v.double().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
a64 = [a.double() if (torch.is_tensor(a) and a.dtype.is_floating_point) else a for a in batch_args(b, edge)]
out = oesf.esf_forward(sd, setting(cfg), *a64, variant=variant, training=True)
out[3].sum().backward()
m = m.to(DEV).train()
args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
o = m(*args); o[3].sum().backward(); torch.cuda.synchronize()
params = dict(m.named_parameters())
for k in ("enc.head.conv1.weight", "enc.down_block1.conv21.weight", "dec.final.conv2.weight"):
    t64 = sd[k].grad.numpy(); ref32 = g["grad::"+k]; hip = params[k].grad.cpu().numpy()
    sc = np.abs(t64).max()
    print('%-32s ref32-vs-f64 %.2e   hip-vs-f64 %.2e   hip-vs-ref32 %.2e' % (k, np.abs(ref32-t64).max()/sc, np.abs(hip-t64).max()/sc, np.abs(hip-ref32).max()/sc))
names = [str(n) for n in g["grad_names"]]
t = np.array([sd[n].grad.norm().item() for n in names]); r = g["grad_l2"]; h = np.array([params[n].grad.double().norm().item() for n in names])
print('L2 norms: max rel  ref32-vs-f64 %.2e   hip-vs-f64 %.2e' % (np.max(np.abs(r-t)/t), np.max(np.abs(h-t)/t)))

import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from common import ESF_CASES, batch_args, esf_module, gold, setting, bdcn_module
from egne_amd import synth
from oracle import bdcn as obdcn, esfnet as oesf
torch.set_num_threads(8)
name = "esf_edge_b2_absent1"
cfg, variant, kw = ESF_CASES[name]; kw = dict(kw)
b = synth.make_batch(kw.pop("B"), **kw)
edge = obdcn.calc_edge({k: v.cpu() for k, v in bdcn_module().state_dict().items()}, b["img"])
m = esf_module(cfg, variant)
def run(dt):
    sd = {k: (v.to(dt) if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    a = [x.to(dt) if (torch.is_tensor(x) and x.dtype.is_floating_point) else x for x in batch_args(b, edge)]
    t0 = time.time()
    out = oesf.esf_forward(sd, setting(cfg), *a, variant=variant, training=True)
    out[3].sum().backward()
    print(dt, "loss", float(out[3].sum()), "%.1fs" % (time.time() - t0), flush=True)
    return {k: v.grad.double() for k, v in sd.items() if v.grad is not None}
g64 = run(torch.float64)
names = list(g64)
ft = torch.cat([g64[n].reshape(-1) for n in names])
for dt in (torch.float32, torch.bfloat16, torch.float16):
    try:
        g = run(dt)
    except Exception as e:
        print(dt, "FAILED", repr(e)[:300]); continue
    fh = torch.cat([g[n].reshape(-1) for n in names])
    print(dt, "whole rel L2 %.3e cos %.6f" % (float((fh - ft).norm() / ft.norm()), float(torch.dot(fh, ft) / (fh.norm() * ft.norm()))), flush=True)

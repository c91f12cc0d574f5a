"""|mean| / std per (sample, channel) of the tensors ESF-Net instance-normalises (block inputs x, block outputs `out`): the factor by
which InstanceNorm amplifies the relative rounding noise of a stored tensor."""
import sys, types
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import batch_args, esf_module, bdcn_module
from egne_amd import synth
from egne_amd.utils import calc_edge
DEV = "cuda:0"
bd = bdcn_module().to(DEV)
b = synth.make_batch(4, seed=4321)
edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)
outs = {}
for st in (torch.float32, torch.bfloat16):
    m = esf_module("baseline_edge").to(DEV).to(st).train()
    args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
    m(*args)
    torch.cuda.synchronize()
    pl = m._last_plan
    outs[st] = pl
    if st == torch.float32:
        for i, d in enumerate(pl.dbg["D"]):
            for k in ("x", "out"):
                p = d[k]
                t = p.buf[..., p.off:p.off + p.C].float()
                mu, sd = t.mean((1, 2)), t.std((1, 2))
                r = (mu.abs() / sd.clamp_min(1e-12)).flatten().cpu().numpy()
                print("level %d %-3s C=%3d  |mean|/std: median %.2f  p90 %.2f  max %.1f" % (i, k, p.C, np.median(r), np.percentile(r, 90), r.max()))
f, h = outs[torch.float32], outs[torch.bfloat16]
for i in range(5):
    for k in ("x", "x1", "x22", "out"):
        a, c = f.dbg["D"][i][k], h.dbg["D"][i][k]
        ta, tc = a.buf[..., a.off:a.off + a.C].float(), c.buf[..., c.off:c.off + c.C].float()
        print("level %d %-3s forward deviation bf16 vs fp32 plan: rel L2 %.4f" % (i, k, float((ta - tc).norm() / ta.norm())))
tb, tcb = f.dbg["bott"].float(), h.dbg["bott"].float()
print("bottleneck rel L2 %.4f" % float((tb - tcb).norm() / tb.norm()))

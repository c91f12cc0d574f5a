"""Gradient noise of the bf16-storage training plan against the fp32-storage plan (HIP, same weights and batch) as a function
of the batch size: per-parameter relative L2 deviation."""
import sys, types
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import batch_args, esf_module, bdcn_module
from egne_amd import synth
from egne_amd.utils import calc_edge
DEV = "cuda:0"
bd = bdcn_module().to(DEV)
for B in (2, 8, 32):
    b = synth.make_batch(B, seed=4321)
    edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)
    grads = {}
    for st in (torch.float32, torch.bfloat16):
        m = esf_module("baseline_edge").to(DEV).to(st).train()
        args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
        m(*args)[3].sum().backward()
        torch.cuda.synchronize()
        grads[st] = {n: p.grad.double().clone() for n, p in m.named_parameters() if p.grad is not None}
    devs = {n: float((grads[torch.bfloat16][n] - g).norm() / max(g.norm(), 1e-30)) for n, g in grads[torch.float32].items() if g.norm() > 0}
    v = np.array(list(devs.values()))
    worst = sorted(devs.items(), key=lambda kv: -kv[1])[:6]
    print("B=%d: per-tensor rel L2 deviation bf16 vs fp32 storage: median %.3f mean %.3f max %.3f; worst %s"
          % (B, np.median(v), v.mean(), v.max(), [(n, round(d, 3)) for n, d in worst]))
    if B == 2:
        for n, d in devs.items():
            print("   %-40s %.3f" % (n, d))

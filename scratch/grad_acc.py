"""bf16-storage against fp32-storage gradients of one training step over several seeded batches: whole-vector relative L2, cosine, and the
median / 90th percentile of the per-tensor gradient-norm errors (the statistics of tests/test_gpu_distinct.py).  usage: [EGNE_LIB=...] python scratch/grad_acc.py [B]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from common import batch_args, esf_module
from egne_amd import synth
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for seed in (77, 78, 79, 80):
    b = synth.make_batch(B, seed=seed)
    edge = torch.rand(B, 1, 240, 320, generator=torch.Generator().manual_seed(seed))
    g = {}
    for st in (torch.float32, torch.bfloat16):
        m = esf_module("baseline_edge", seed=7).to(DEV).to(st).train()
        out = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])
        out[3].sum().backward(); torch.cuda.synchronize()
        g[st] = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
        del m
    gf, gh = g[torch.float32], g[torch.bfloat16]
    names = list(gf)
    rel = np.array([abs(gh[n].norm().item() - gf[n].norm().item()) / max(gf[n].norm().item(), 1e-30) for n in names])
    keep = np.array([gf[n].norm().item() for n in names]); rel = rel[keep > 1e-6 * keep.max()]
    fh, ff = torch.cat([gh[n].reshape(-1) for n in names]), torch.cat([gf[n].reshape(-1) for n in names])
    print("seed %d: whole rel L2 %.3f cosine %.4f | norm errors median %.3f p90 %.3f" % (seed, float((fh - ff).norm() / ff.norm()), float(torch.dot(fh, ff) / (fh.norm() * ff.norm())), np.median(rel), np.sort(rel)[int(0.9 * len(rel))]), flush=True)

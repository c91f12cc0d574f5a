"""Parameter gradients of one bf16-storage training step (B frames, seeded) saved for a comparison between two library builds (EGNE_LIB).
usage: EGNE_LIB=... python scratch/grad_diff.py out.pt [B] ; python scratch/grad_diff.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    rows = []
    for k in a:
        d = (a[k].double() - b[k].double()).norm().item(); n = a[k].double().norm().item()
        rows.append((d / max(n, 1e-30), k, n))
    for r, k, n in sorted(rows, reverse=True)[:40]:
        print("%-44s rel diff %.3e (norm %.3e)" % (k, r, n))
    sys.exit(0)
from common import batch_args, esf_module, bdcn_module
from egne_amd import synth
DEV = "cuda:0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = synth.make_batch(B, seed=77)
edge = torch.rand(B, 1, 240, 320, generator=torch.Generator().manual_seed(3))
m = esf_module("baseline_edge", seed=7).to(DEV).to(torch.bfloat16).train()
out = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])
out[3].sum().backward()
torch.cuda.synchronize()
g = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}
g["__loss"] = out[3].detach().float().cpu(); g["__op"] = out[0].detach().float().cpu()
torch.save(g, sys.argv[1])

"""Permutation equivariance of one bf16-storage training step, per tensor (debugging aid for tests/test_gpu_b256.py)."""
import sys, os, types, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from common import bdcn_module, batch_args, esf_module
from egne_amd import synth
from egne_amd.utils import calc_edge
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
storage = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
DEV = "cuda:0"
NS = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
b = synth.make_batch(B, seed=2025)
net = bdcn_module().to(DEV)
edge = torch.cat([calc_edge(NS, b["img"][i:i + 64].to(DEV), net, DEV) for i in range(0, B, 64)])
del net
def take(b, idx):
    return {k: (v[idx] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in b.items()}
def step(b, edge):
    gc.collect(); torch.cuda.empty_cache()
    m = esf_module("baseline_edge", seed=11).to(DEV).to(storage).train()
    op, _, latent, loss, elOut = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])
    loss.sum().backward(); torch.cuda.synchronize()
    g = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    del m, op, latent, loss, elOut
    gc.collect(); torch.cuda.empty_cache()
    return g
perm = torch.from_numpy(np.random.RandomState(17).permutation(B))
g0 = step(b, edge); g1 = step(take(b, perm), edge[perm.to(edge.device)])
g2 = step(b, edge)
rows = []
for n in g0:
    d = (g1[n] - g0[n]).norm().item() / max(g0[n].norm().item(), 1e-30)
    d2 = (g2[n] - g0[n]).norm().item() / max(g0[n].norm().item(), 1e-30)
    rows.append((d, d2, n, g0[n].norm().item()))
a = torch.cat([g0[n].reshape(-1) for n in g0]); c = torch.cat([g1[n].reshape(-1) for n in g0]); e = torch.cat([g2[n].reshape(-1) for n in g0])
print("B=%d %s env=%s: whole perm %.3e  rerun %.3e" % (B, storage, {k: v for k, v in os.environ.items() if k.startswith("EGNE_")}, (a - c).norm() / a.norm(), (a - e).norm() / a.norm()))
order = list(g0)
for d, d2, n, nn in rows:
    if d > 1e-3 or d2 > 1e-3:
        print("  %-40s perm %.3e rerun %.3e  |g| %.3e" % (n, d, d2, nn))

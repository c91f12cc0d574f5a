"""bf16 training step with and without the statistics from the 3x3's epilogue: per-tensor gradient differences (debugging aid)."""
import sys, os, types, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from common import bdcn_module, batch_args, esf_module
from egne_amd import synth, engine
from egne_amd.utils import calc_edge
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = "cuda:0"
NS = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
b = synth.make_batch(B, seed=int(os.environ.get("BSEED", "2025")))
net = bdcn_module().to(DEV)
edge = calc_edge(NS, b["img"].to(DEV), net, DEV)
del net
def step(flag, storage=torch.bfloat16):
    engine.STATS_FUSED_BF16 = flag
    gc.collect(); torch.cuda.empty_cache()
    m = esf_module("baseline_edge", seed=int(os.environ.get("MSEED", "11"))).to(DEV).to(storage).train()
    op, _, latent, loss, elOut = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])
    loss.sum().backward(); torch.cuda.synchronize()
    g = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    rs = {n: t.detach().double().cpu().clone() for n, t in m.named_buffers() if "running" in n}
    return g, float(loss), rs
g1, l1, r1 = step(True); g0, l0, r0 = step(False); gf, lf, rf = step(False, torch.float32)
print("loss fused %.6f plain %.6f fp32 %.6f" % (l1, l0, lf))
for n in r1:
    print("  %-40s fused-vs-plain %.3e  plain-vs-fp32 %.3e" % (n, (r1[n] - r0[n]).abs().max() / r0[n].abs().max(), (r0[n] - rf[n]).abs().max() / rf[n].abs().max()))
for n in g0:
    d = (g1[n] - g0[n]).norm().item() / max(g0[n].norm().item(), 1e-30)
    e1 = (g1[n] - gf[n]).norm().item() / max(gf[n].norm().item(), 1e-30)
    e0 = (g0[n] - gf[n]).norm().item() / max(gf[n].norm().item(), 1e-30)
    n1, n0 = abs(g1[n].norm().item() - gf[n].norm().item()) / gf[n].norm().item(), abs(g0[n].norm().item() - gf[n].norm().item()) / gf[n].norm().item()
    if n1 > 0.1 or n0 > 0.1:
        print("  %-40s NORM err vs fp32: fused %.3e plain %.3e" % (n, n1, n0))
    if e1 > 1.5 * e0 + 0.02:
        print("  %-40s fused-vs-plain %.3e | vs fp32: fused %.3e plain %.3e" % (n, d, e1, e0))

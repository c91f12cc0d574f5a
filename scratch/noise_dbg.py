"""How far apart are independent bf16 'realisations' of one training step?  Relative L2 error of decoder / encoder / whole gradients against the
fp32-storage plan for variants that differ only in rounding points (debugging aid, round 5)."""
import sys, os, types, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from common import bdcn_module, batch_args, esf_module
from egne_amd import synth, engine, esf_engine
from egne_amd.utils import calc_edge
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DEV = "cuda:0"
NS = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
b = synth.make_batch(B, seed=int(os.environ.get("BSEED", "777")))
net = bdcn_module().to(DEV)
edge = calc_edge(NS, b["img"].to(DEV), net, DEV)
del net
def step(storage=torch.bfloat16, **flags):
    old = {}
    for k, v in flags.items():
        mod = esf_engine if hasattr(esf_engine, k) and not hasattr(engine, k) else engine
        old[k] = (mod, getattr(mod, k)); setattr(mod, k, v)
    gc.collect(); torch.cuda.empty_cache()
    m = esf_module("baseline_edge", seed=int(os.environ.get("MSEED", "7"))).to(DEV).to(storage).train()
    loss = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])[3]
    loss.sum().backward(); torch.cuda.synchronize()
    g = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    for k, (mod, v) in old.items(): setattr(mod, k, v)
    return g, float(loss.detach())
gf, lf = step(torch.float32)
def grp(g, pref):
    names = [n for n in gf if n.startswith(pref)]
    a = torch.cat([g[n].reshape(-1) for n in names]); c = torch.cat([gf[n].reshape(-1) for n in names])
    return float((a - c).norm() / c.norm())
def norms(g):
    names = list(gf)
    rel = np.array([abs(g[n].norm().item() - gf[n].norm().item()) / max(gf[n].norm().item(), 1e-30) for n in names])
    return float(np.median(rel)), float(np.sort(rel)[int(0.9 * len(rel))])
for label, fl in (("default", {}), ("STATS_FUSED_BF16=0", dict(STATS_FUSED_BF16=False)), ("FOLD_UP_TRAIN=0", dict(FOLD_UP_TRAIN=False)),
                  ("MULTI_DGRAD=0", dict(MULTI_DGRAD=False)), ("NORM_FUSE=0", dict(NORM_FUSE=False)), ("STATS0+FOLD0", dict(STATS_FUSED_BF16=False, FOLD_UP_TRAIN=False)),
                  ("BF16_FAST1X1=0", dict(BF16_FAST1X1=False))):
    g, l = step(**fl)
    print("%-22s loss %.5f (fp32 %.5f) | rel L2 vs fp32: enc %.3f dec %.3f elReg %.3f whole %.3f | norms median %.3f p90 %.3f" % ((label, l, lf, grp(g, "enc."), grp(g, "dec."), grp(g, "elReg"), grp(g, "")) + norms(g)), flush=True)

"""Experiment (round 5): how much of the bf16-storage gradient error (0.21 relative L2 of the whole gradient vector against float64)
comes from the convBlock head alone?  An fp32-storage training plan whose head tensors (t0 = conv1 output, pre = conv2 output) and /
or their gradients are rounded to bf16 in place between launches, against the float64 oracle gradient."""
import os, sys
os.environ["EGNE_WGRAD_SIDE"] = "0"; os.environ["EGNE_PAIR_BIAS_SIDE"] = "0"; os.environ["EGNE_ZERO_AHEAD"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import ESF_CASES, batch_args, esf_module, bdcn_module, setting
from egne_amd import synth
from oracle import bdcn as obdcn, esfnet as oesf
DEV = "cuda:0"
name = "esf_edge_b2_absent1"
cfg, variant, kw = ESF_CASES[name]; kw = dict(kw)
b = synth.make_batch(kw.pop("B"), **kw)
edge = obdcn.calc_edge({k: v.cpu() for k, v in bdcn_module().state_dict().items()}, b["img"])
m0 = esf_module(cfg, variant)
sd = {k: v.double().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m0.state_dict().items()}
a64 = [a.double() if (torch.is_tensor(a) and a.dtype.is_floating_point) else a for a in batch_args(b, edge)]
oesf.esf_forward(sd, setting(cfg), *a64, variant=variant, training=True)[3].sum().backward()
names = [n for n, v in sd.items() if v.grad is not None]
flat_t = torch.cat([sd[n].grad.reshape(-1) for n in names])

def rounder(t):
    def f(st):
        t.copy_(t.to(torch.bfloat16).to(t.dtype)); return 0
    return f

def insert_after(plan, prefix_names, fn, tag):
    # (no index shifts: the plan keeps tables keyed by call index) -- wrap the launch itself
    idx = max(i for i, c in enumerate(plan.calls) if c[2] in prefix_names)
    f0, a0, n0 = plan.calls[idx]
    def both(*a, f0=f0, fn=fn):
        rc = f0(*a)
        fn(a[-1])
        return rc
    plan.calls[idx] = (both, a0, n0)

def run(storage, fwd_round=False, bwd_round=False, blocks=(), which=("t0", "pre")):
    mm = esf_module(cfg, variant).to(DEV).to(storage).train()
    args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
    out = mm(*args)      # builds the plan
    pl = mm._last_plan
    t0, pre = pl.dbg["t0"], pl.dbg["head_pre"]
    if fwd_round:
        if "t0" in which: insert_after(pl, {"enc.head.conv1"}, rounder(t0), "round.t0")
        if "pre" in which: insert_after(pl, {"enc.head.conv2"}, rounder(pre), "round.pre")
        if "x" in which: insert_after(pl, {"enc.head.bn.apply", "enc.head.bn.edge.apply"}, rounder(pl.dbg["D"][0]["x"].buf), "round.x")
    if bwd_round:
        bw = pl.bw
        insert_after(bw, {"enc.head.bn.bwd", "enc.head.bn.edge.bwd"}, rounder(pl.gbuf(pre)), "round.gpre")
        insert_after(bw, {"enc.head.conv2.act_bwd"}, rounder(pl.gbuf(pre)), "round.gz_pre")
        insert_after(bw, {"enc.head.conv2.dgrad0"}, rounder(pl.gbuf(t0)), "round.gt0")
        insert_after(bw, {"enc.head.conv1.act_bwd"}, rounder(pl.gbuf(t0)), "round.gz_t0")
    for p in mm.parameters():
        if p.grad is not None: p.grad.zero_()
    mm.load_state_dict({k: v for k, v in m0.state_dict().items()})
    mm(*args)[3].sum().backward()
    torch.cuda.synchronize()
    params = dict(mm.named_parameters())
    flat_h = torch.cat([params[n].grad.double().cpu().reshape(-1) for n in names])
    whole = float((flat_h - flat_t).norm() / flat_t.norm()); cos = float(torch.dot(flat_h, flat_t) / (flat_h.norm() * flat_t.norm()))
    per = {n: float((params[n].grad.double().cpu() - sd[n].grad).norm() / max(sd[n].grad.norm(), 1e-30)) for n in names}
    # contribution of groups of tensors to the whole error
    def grp(pref):
        sel = [n for n in names if n.startswith(pref)]
        d = torch.cat([(params[n].grad.double().cpu() - sd[n].grad).reshape(-1) for n in sel]); return float(d.norm() / flat_t.norm())
    return whole, cos, per, {p: grp(p) for p in ("enc.head", "enc.down_block1", "enc.down_block2", "enc.down_block3", "enc.down_block4", "enc.bottleneck", "dec.", "elReg")}

for label, a, kw in (("fp32 storage", (torch.float32, False, False), {}), ("fp32 + t0 rounded", (torch.float32, True, False), dict(which=("t0",))),
                     ("fp32 + pre rounded", (torch.float32, True, False), dict(which=("pre",))),
                     ("fp32 + block-0 input x rounded", (torch.float32, True, False), dict(which=("x",))),
                     ("fp32 + t0, pre rounded", (torch.float32, True, False), {}), ("bf16 storage", (torch.bfloat16, False, False), {})):
    w, c, per, g = run(*a, **kw)
    top = sorted(per, key=per.get)[-4:]
    print("%-42s whole %.3e cos %.5f | share of the whole error by group: %s | worst tensors: %s"
          % (label, w, c, {k: "%.3f" % v for k, v in g.items()}, {k: "%.2f" % per[k] for k in top}), flush=True)

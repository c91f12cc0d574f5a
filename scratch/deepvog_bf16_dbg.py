"""(Needs a local patch that lets DeepVOG_pytorch.to(torch.bfloat16) through -- round 4 tried it and did not ship it)
Where the bf16-storage DeepVOG plan drifts from the fp32-storage one: activated output of every conv-bn-relu, relative to its largest value."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import egne_amd
from egne_amd import synth
from egne_amd.modelSummary import get_model
from test_oracle_golden import _deepvog_case
DEV = "cuda:0"
b = _deepvog_case("b3")
args = [a.to(DEV) if torch.is_tensor(a) else a for a in
        (b["img"], torch.zeros_like(b["img"]), b["label"], b["pupil_center"], b["elNorm"], b["spatWts"], b["distMap"], b["cond"], b["ID"], b["alpha"])]
outs = {}
for st in (torch.float32, torch.bfloat16):
    m = get_model("deepvog", None)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=1, kind="esf"))
    m = m.to(DEV).to(st).train()
    out = m(*args)
    torch.cuda.synchronize()
    pl = m._last_plan
    outs[st] = {k: v.buf[..., v.off:v.off + v.C].float().clone() for k, v in pl.dbg.items()}
    outs[st]["logits"] = out[0].detach().float()
    print(st, "kinds", sorted({k for k, _ in pl.meta}))
for k in outs[torch.float32]:
    a, c = outs[torch.float32][k], outs[torch.bfloat16][k]
    print("%-16s max |diff| / max |ref| = %.3e   (rms diff / rms ref %.3e)" % (k, (a - c).abs().max().item() / a.abs().max().item(), ((a - c).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item()))

# gradients: bf16 vs fp32 storage, ratio of norms and cosine per parameter
grads = {}
for st in (torch.float32, torch.bfloat16):
    m = get_model("deepvog", None)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=1, kind="esf"))
    m = m.to(DEV).to(st).train()
    out = m(*args)
    out[3].sum().backward()
    torch.cuda.synchronize()
    grads[st] = {k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}
for k in grads[torch.float32]:
    a, c = grads[torch.float32][k].flatten(), grads[torch.bfloat16][k].flatten()
    if a.norm() > 0:
        print("%-28s norm ratio %.3f  cosine %.4f  n=%d" % (k, (c.norm() / a.norm()).item(), (a @ c / (a.norm() * c.norm() + 1e-300)).item(), a.numel()))

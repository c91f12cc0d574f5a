"""bf16 against fp32 storage at other widths (chz 16 / 48): whole-gradient relative L2 and loss (sanity check of the round-5 paths)."""
import sys, os, types, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from common import bdcn_module, batch_args, setting
from egne_amd import synth
from egne_amd.utils import calc_edge
from egne_amd.models.RITnet_v2 import DenseNet2D
DEV = "cuda:0"
NS = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
B = 8
b = synth.make_batch(B, seed=5)
net = bdcn_module().to(DEV)
edge = calc_edge(NS, b["img"].to(DEV), net, DEV)
del net
for chz in [int(c) for c in os.environ.get("CHZ", "16,48,64").split(",")]:
    res = {}
    for st in (torch.float32, torch.bfloat16):
        torch.manual_seed(3)
        m = DenseNet2D(setting("baseline_edge"), chz=chz).to(DEV).to(st).train()
        loss = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)])[3]
        loss.sum().backward(); torch.cuda.synchronize()
        res[st] = (float(loss.detach()), torch.cat([p.grad.detach().double().reshape(-1).cpu() for p in m.parameters() if p.grad is not None]))
        kinds = {mm[0] for mm in m._last_plan.bw.meta}
        del m; gc.collect(); torch.cuda.empty_cache()
    gf, gh = res[torch.float32][1], res[torch.bfloat16][1]
    print("chz %d: loss fp32 %.5f bf16 %.5f | gradient rel L2 %.3f cos %.4f | finite %s" % (chz, res[torch.float32][0], res[torch.bfloat16][0], float((gh - gf).norm() / gf.norm()),
          float(torch.dot(gh, gf) / (gh.norm() * gf.norm())), bool(torch.isfinite(gh).all())), flush=True)

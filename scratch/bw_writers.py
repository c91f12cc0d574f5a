"""Which launch of the backward plan precedes each act_bwd (candidates for applying the activation mask + bias sums in that
launch's epilogue instead of a separate pass)."""
import sys, torch, yaml, os
sys.path.insert(0, '.')
import egne_amd
from egne_amd import synth, engine
from egne_amd.models.RITnet_v2 import DenseNet2D
setting = yaml.safe_load(open(os.path.join(os.path.dirname(egne_amd.__file__), "configs", "baseline_edge.yaml")))
dev = torch.device("cuda:0")
net = DenseNet2D(dict(setting))
net.load_state_dict(synth.seeded_state_dict(net.state_dict(), kind="esf"))
net.to(dev).to(torch.bfloat16); net.train()
net._ensure_grad_arena()
pl = net._build_plan(2, 240, 320, dev, True, torch.bfloat16)
bw = pl.bw
print(len(pl.calls), "forward launches,", len(bw.calls), "backward launches")
meta = bw.meta
import collections
prev = collections.Counter()
for i, m in enumerate(meta):
    if 'act_bwd' in m[0]:
        print("%-30s <- %s" % (m[0], " | ".join("%s (%s)" % (mm[0], mm[1]) for mm in meta[max(0, i - 2):i])))
        prev[meta[i - 1][1]] += 1
print(prev)

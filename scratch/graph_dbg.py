import os, sys, argparse, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egne_amd
from egne_amd import synth, _entry
from egne_amd.evaluate import _seg_and_fit, graphed_runner
from egne_amd.utils import calc_edge
DEV = "cuda:0"
with open(os.path.join(os.path.dirname(egne_amd.__file__), "configs", "baseline_edge.yaml")) as f:
    bd, net = _entry.seeded_networks(yaml.safe_load(f))
bd, net = bd.to(DEV).eval(), net.to(DEV).eval()
ns = argparse.Namespace(prec=torch.float32, edge_thres=0)
warm = synth.make_batch(2, seed=5)["img"].to(DEV)
run = graphed_runner(warm, net, bd)
for seed in (6, 7, 6):
    x = synth.make_batch(2, seed=seed)["img"].to(DEV)
    got = [t.clone() for t in run(x)]
    with torch.no_grad():
        want = _seg_and_fit(x, net)(calc_edge(ns, x, bd, DEV))
        torch.cuda.synchronize()
        want2 = _seg_and_fit(x, net)(calc_edge(ns, x, bd, DEV))
    torch.cuda.synchronize()
    for i, (a, b, c) in enumerate(zip(got, want, want2)):
        print(seed, i, tuple(a.shape), a.dtype, "graph==eager", torch.equal(a, b), "eager==eager", torch.equal(b, c),
              (a.double() - b.double()).abs().max().item())

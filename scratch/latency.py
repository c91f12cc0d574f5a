"""Per-call latency of edge + seg (+ mask) at small batch, with / without hipGraph replay (EGNE_GRAPH=0/1)."""
import sys, time, types, torch
sys.path.insert(0, "/root/repo")
import egne_amd
from egne_amd import _entry, synth
from egne_amd.utils import calc_edge
import yaml, os
dev = torch.device("cuda:0")
with open(os.path.join(os.path.dirname(egne_amd.__file__), "configs", "baseline_edge.yaml")) as f:
    setting = yaml.safe_load(f)
bd, net = _entry.seeded_networks(setting)
bd, net = bd.to(dev).eval(), net.to(dev).eval()
args = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
for B in (1, 2, 4, 8):
    b = synth.make_batch(B, seed=1)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    def step():
        with torch.no_grad():
            edge = calc_edge(args, t["img"], bd, dev)
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"], t["ID"], t["alpha"])
            return net.predictions()
    for _ in range(3): m = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): m = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("B=%d  %.2f ms per call  %.0f frames/s  mask sum %d" % (B, dt * 1e3, B / dt, int(m.sum())), flush=True)

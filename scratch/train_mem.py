import sys, time, torch
sys.path.insert(0, ".")
import egne_amd
from egne_amd import _entry, synth
from egne_amd.utils import calc_edge
dev = torch.device("cuda:0")
cfg = sys.argv[2] if len(sys.argv) > 2 else "baseline_edge"
for B in [int(x) for x in sys.argv[1].split(",")]:
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    args, bd, net = _entry.seeded_networks("configs/%s.yaml" % cfg, dev)
    net.train()
    opt = torch.optim.Adam([p for n, p in net.named_parameters() if "dsIdentify" not in n], lr=5e-4)
    b = synth.make_batch(B, seed=1)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    def step():
        with torch.no_grad():
            edge = calc_edge(args, t["img"], bd, dev)
        opt.zero_grad(set_to_none=False)
        out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"], t["ID"], t["alpha"])
        out[3].sum().backward(); opt.step()
        return out[3]
    for _ in range(2): l = step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): l = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print("B=%d  %.1f ms/step  %.1f frames/s  peak %.1f GB  loss %.4f" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30, l.item()), flush=True)
    del net, bd, opt

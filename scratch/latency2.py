"""Per-call latency of edge + seg + fit at B = 1 / 2 (the head-mounted-display case: two eyes per video frame, evaluate.py:235-249):
wall time per call with a synchronisation after every call, host time to queue a call, launch counts and the largest launches."""
import sys, time, types, os
import torch
sys.path.insert(0, ".")
import egne_amd
from egne_amd import _entry, synth, engine
from egne_amd.utils import calc_edge, fit_ellipses_from_pred
import yaml
dev = torch.device("cuda:0")
with open(os.path.join(os.path.dirname(egne_amd.__file__), "configs", "baseline_edge.yaml")) as f:
    setting = yaml.safe_load(f)
bd, net = _entry.seeded_networks(setting)
bd, net = bd.to(dev).eval(), net.to(dev).eval()
args = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
for B in (1, 2):
    b = synth.make_batch(B, seed=1)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    def step(fit=True):
        with torch.no_grad():
            edge = calc_edge(args, t["img"], bd, dev)
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"], t["ID"], t["alpha"])
            m = net.predictions()
            return fit_ellipses_from_pred(m, out[1]) if fit else m
    for _ in range(5): r = step()
    torch.cuda.synchronize()
    for fit in (False, True):
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            r = step(fit); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n): r = step(fit)
        th = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        print("B=%d fit=%s: %.3f ms per call (synchronised), host queues a call in %.3f ms" % (B, fit, dt * 1e3, th * 1e3), flush=True)
    pb, pe = list(bd._plans.values())[-1], net._last_plan
    print("   launches: BDCN %d, ESF-Net %d" % (len(pb.calls), len(pe.calls)))
    ev = []
    bd._events = net._events = ev
    engine.EVENT_KINDS = None
    step(False); torch.cuda.synchronize()
    bd._events = net._events = None
    rows = sorted(((e0.elapsed_time(e1) * 1e3, name, kind) for kind, fl, e0, e1, name in ev), reverse=True)
    tot = sum(r[0] for r in rows)
    print("   sum of launch durations %.3f ms over %d launches; top: %s" % (tot / 1e3, len(rows), [(n_, round(us)) for us, n_, k in rows[:12]]))
    if os.environ.get("LAT_DUMP"):
        for kind, fl, e0, e1, name in ev:
            print("      %-34s %-22s %7.1f us  %8.3f GF" % (name, kind, e0.elapsed_time(e1) * 1e3, fl / 1e9))
    import collections
    fam = collections.defaultdict(lambda: [0.0, 0])
    for us, n_, k in rows:
        fam[k][0] += us; fam[k][1] += 1
    print("   by kind:", [(k, round(v[0]), v[1]) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:14]])

# the same call as one hipGraph replay (egne_amd.pipeline.GraphedFrames)
from egne_amd.evaluate import graphed_runner, _seg_and_fit
for B in (1, 2):
    x = synth.make_batch(B, seed=1)["img"].to(dev)
    run = graphed_runner(x, net, bd)
    with torch.no_grad():
        want = _seg_and_fit(x, net)(calc_edge(args, x, bd, dev))
    got = run(x)
    torch.cuda.synchronize()
    print("   captured with branches:", [getattr(p_, "branch_runs", 0) for p_ in bd._plans.values()], flush=True)
    same = all(torch.equal(a, b) for a, b in zip(want, got))
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        r = run(x); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B=%d graph replay (edge + seg + fit): %.3f ms per call (synchronised), identical to the eager call: %s" % (B, dt * 1e3, same), flush=True)

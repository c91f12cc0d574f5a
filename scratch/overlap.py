"""Experiment: the frozen edge network of batch i+1 on a second stream while ESF-Net works on batch i (two-stage pipeline across
batches) against the sequential loop.  Prints frames/s of both."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from common import batch_args, bdcn_module, esf_module
from egne_amd import synth
from egne_amd.utils import calc_edge
B, K = 64, 30
dev = torch.device("cuda:0")
b = synth.make_batch(8, seed=1)
rep = B // 8
t = {k: (v.repeat(*([rep] + [1] * (v.dim() - 1))).to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
bd = bdcn_module().to(dev)
m = esf_module("baseline_edge").to(dev).eval()
ns = types.SimpleNamespace(prec=torch.float32, edge_thres=0)

def esf(edge):
    with torch.no_grad():
        return m(*[a for a in batch_args(t, edge)])

def seq(n):
    for _ in range(n):
        e = calc_edge(ns, t["img"], bd, dev)
        out = esf(e)
    return out

def pipe(n):
    pa, pb = [int(v) for v in os.environ.get('PRIO', '0,0').split(',')]
    sa, sb = torch.cuda.Stream(priority=pa), torch.cuda.Stream(priority=pb)
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    edges = [torch.empty(B, 1, 240, 320, device=dev) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    out = None
    for i in range(n + 1):
        if i < n:
            with torch.cuda.stream(sa):
                if i >= 2:
                    sa.wait_event(freed[i & 1])
                e = calc_edge(ns, t["img"], bd, dev)
                edges[i & 1].copy_(e)
                ready[i & 1].record(sa)
        if i >= 1:
            j = (i - 1) & 1
            with torch.cuda.stream(sb):
                sb.wait_event(ready[j])
                out = esf(edges[j])
                freed[j].record(sb)
    cur.wait_stream(sa); cur.wait_stream(sb)
    return out

for fn in (seq, pipe, seq, pipe, seq, pipe):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-5s %8.1f frames/s  %.2f ms/step  loss %.5f" % (fn.__name__, B * K / dt, 1e3 * dt / K, float(out[3].sum())))

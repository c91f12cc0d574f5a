"""How the two halves of a training step scale with the number of CUs they may use (hipExtStreamCreateWithCUMask): the frozen edge
network (MFMA bound) and ESF-Net forward / backward / Adam (HBM bound), bf16 storage -- is a CU partition between the two pipeline
stages worth building?  usage: python3 scratch/cu_split.py [B]"""
import ctypes as C, sys, time, os
import torch
sys.path.insert(0, ".")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sys.argv = ["bench.py", "--mode", "train", "--train-batch", str(B), "--train-storage", "bf16", "--no-cpu-baseline", "--no-pipeline"]
import bench as bench_mod
a = bench_mod.parse()
bn = bench_mod.Bench(a)
dev = bn.dev
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= (1 << (i % 32)); n += 1
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev), n

from egne_amd.utils import calc_edge
t, bd, net, args = bn.batch(B), bn.bd, bn.net, bn.args
net.to(torch.bfloat16); net.train()
opt = torch.optim.Adam([p for n, p in net.named_parameters() if "dsIdentify" not in n], lr=5e-4, fused=True)
def edge():
    with torch.no_grad():
        return calc_edge(args, t["img"], bd, dev)
def rest(e):
    opt.zero_grad()
    out = net(t["img"], e, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"], t["ID"], t["alpha"])
    out[3].backward()
    opt.step()
    return out
e = edge()
for _ in range(2): rest(e)
torch.cuda.synchronize()
def timeit(fn, stream, n=3):
    with torch.cuda.stream(stream):
        fn(); 
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
full = torch.cuda.Stream(device=dev)
print("B=%d  all CUs: edge %.1f ms, ESF train %.1f ms" % (B, timeit(edge, full), timeit(lambda: rest(e), full)), flush=True)
for per32 in (8, 12, 16, 20, 24):
    s_lo, n_lo = masked_stream(lambda i: (i % 32) < per32)
    s_hi, n_hi = masked_stream(lambda i: (i % 32) >= per32)
    te, tr = timeit(edge, s_lo), timeit(lambda: rest(e), s_hi)
    # both at once: edge of the next batch on its partition next to a training step on the other
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        with torch.cuda.stream(s_lo): edge()
        with torch.cuda.stream(s_hi): rest(e)
    torch.cuda.synchronize()
    both = (time.perf_counter() - t0) / n * 1e3
    print("edge on %3d CUs: %.1f ms | ESF train on %3d CUs: %.1f ms | both at once: %.1f ms per step = %.0f frames/s" % (n_lo, te, n_hi, tr, both, B / both * 1e3), flush=True)

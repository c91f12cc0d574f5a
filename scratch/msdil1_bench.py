"""Dilated group of an MSBlock on plain f16 operands: strip form (msblock_dil_ps_f16.hip, NP = 1) against the ring form
(msblock_dil1_f16.hip).  usage: python scratch/msdil1_bench.py [B]"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, SplitScale
DEV = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, H, W in (("stage 1", 240, 320), ("stage 2", 120, 160), ("stage 3", 60, 80)):
    res = []
    for ring in ("0", "1"):
        os.environ["EGNE_MSDIL1"] = ring
        torch.manual_seed(1)
        pl = Plan(DEV)
        pl.f16_products = 1
        xb = pl.buf(B, H, W, 64); xb.normal_().relu_()
        l0 = ConvLayer([torch.nn.Parameter(torch.randn(32, 64, 3, 3, device=DEV) / 24)], [torch.nn.Parameter(torch.randn(32, device=DEV))], [(64, 64)], pad=(1, 1), act=1)
        lg = ConvLayer([torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV) / 17) for _ in range(3)],
                       [torch.nn.Parameter(torch.randn(32, device=DEV)) for _ in range(3)], [(32, 32)], pad=(1, 1), dils=(4, 8, 12), act=1)
        l0.split = lg.split = True
        po = Piece(pl.buf(B, H, W, 32), 0, 32)
        po.presplit = SplitScale()
        pl.conv(l0, [Piece(xb, 0, 64)], po, B, H, W)
        s0, s1 = pl.vec(B, H, W), pl.vec(B, H, W)
        cw, cc = torch.randn(2, 32, device=DEV) / 6, torch.tensor([0.7, -1.3], device=DEV)
        pl.keep += [cw, cc]
        pl.conv(lg, [po], po, B, H, W, residual=po, scores=(cw, cc, s0, s1, False))
        for _ in range(3): pl.run()
        torch.cuda.synchronize()
        fn, args, nm = pl.calls[-1]
        st = egne_amd._lib.stream_ptr()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): fn(*args, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * W * 32 * 32 * 27
        res.append((ms, fl / ms / 1e9, s0.clone()))
        del pl
    print("%-8s B=%d  strips %7.3f ms %6.1f TFLOP/s   ring %7.3f ms %6.1f TFLOP/s   x%.2f  bit-identical %s" % (
        name, B, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] / res[1][0], torch.equal(res[0][2], res[1][2])), flush=True)

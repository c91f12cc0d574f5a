"""Role-split 3x3 kernel against the halo kernel, layer by layer (B=64).  usage: rs_bench.py"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
B = 64
for (Cin, Cout, H, W) in [(64, 64, 240, 320), (64, 128, 120, 160), (64, 32, 240, 320), (32, 32, 240, 320), (32, 64, 240, 320), (128, 32, 120, 160), (32,32,120,160)]:
    res = []
    for rs in (0, 1, 2, 1, 2):
        engine.RS_ENABLED = bool(rs)
        engine.RW_ENABLED = rs == 2
        pl = Plan(DEV)
        xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
        w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
        b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
        layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
        layer.split = True
        ob = pl.buf(B, H, W, pad8(Cout))
        pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
        kind = pl.meta[-1][0]
        for _ in range(3): pl.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): pl.run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2 * B * H * W * Cin * Cout * 9
        res.append("%s %.0f us %.0f TF/s" % (kind.split(':')[-1], us, fl / us / 1e6))
    print(Cin, Cout, H, W, " | ".join(res), flush=True)

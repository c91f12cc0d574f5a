"""Micro-benchmark of the bf16-storage 3x3 kernel (conv3x3_bf16.hip) on the shapes of the training plans.
usage: [EGNE_B3_MB=1] python scratch/b3_bench.py [res]      (res: with an accumulated residual = a data gradient)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import egne_amd  # noqa
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
BF = torch.bfloat16
CASES = [  # name, B, Cin, Cout, H, W
    ("32->32 240x320 (b0, chz32)", 128, 32, 32, 240, 320),
    ("64->64 240x320 (b0, chz64)", 128, 64, 64, 240, 320),
    ("64->64 120x160 (b1)", 128, 64, 64, 120, 160),
    ("38->64 120x160 (b1.conv1)", 128, 38, 64, 120, 160),
    ("128->128 120x160 (b1 chz64)", 128, 128, 128, 120, 160),
    ("96->96 60x80 (b2)", 128, 96, 96, 60, 80),
    ("192->192 60x80 (b2 chz64)", 128, 192, 192, 60, 80),
    ("128->128 30x40 (b3)", 128, 128, 128, 30, 40),
    ("256->256 30x40 (b3 chz64)", 128, 256, 256, 30, 40),
    ("180->180 30x40 (up4)", 64, 180, 180, 30, 40),
    ("100->100 60x80 (up3)", 64, 100, 100, 60, 80),
    ("62->62 120x160 (up2)", 64, 62, 62, 120, 160),
]
res = len(sys.argv) > 1 and sys.argv[1] == "res"
only = os.environ.get("B3_ONLY")
if only:
    CASES = [c for c in CASES if only in c[0]]
for name, B, Cin, Cout, H, W in CASES:
    pl = Plan(DEV, dtype=BF)
    xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
    b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=0 if res else 2)
    ob = pl.buf(B, H, W, pad8(Cout))
    rb = None
    if res:
        rbuf = pl.buf(B, H, W, pad8(Cout)); rbuf.normal_()
        rb = Piece(rbuf, 0, Cout)
    pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W, residual=rb)
    assert pl.meta[0][0] == "conv_bf16:3x3", pl.meta
    for _ in range(2):
        pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        pl.run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * Cout * Cin * 9
    by = 2.0 * B * H * W * (pad8(Cin) + pad8(Cout) * (2 if res else 1))
    print("%-30s %8.3f ms  %7.1f TFLOP/s  %5.2f TB/s" % (name, ms, fl / ms / 1e9, by / ms / 1e9), flush=True)
    del pl, xb, ob
    torch.cuda.empty_cache()

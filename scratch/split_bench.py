"""Micro-benchmark of single split-f16 conv layers.  usage: python scratch/split_bench.py [case-substring]"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
CASES = [  # name, B, Cin, Cout, H, W, dil
    ("esf32 128x240x320", 128, 32, 32, 240, 320, 1),
    ("vgg1_2 64x240x320", 64, 64, 64, 240, 320, 1),
    ("ms1 64->32 240x320", 64, 64, 32, 240, 320, 1),
    ("ms2 128->32 120x160", 64, 128, 32, 120, 160, 1),
    ("ms3 256->32 60x80", 64, 256, 32, 60, 80, 1),
    ("enc.b1 64->64", 128, 64, 64, 120, 160, 1),
    ("ms4 512->32 30x40", 64, 512, 32, 30, 40, 1),
    ("ms5 512->32 30x40 d2?", 64, 512, 32, 30, 40, 1),
    ("enc.b3 128->128 30x40", 128, 128, 128, 30, 40, 1),
    ("dec.up3 62->62 60x80", 64, 64, 64, 60, 80, 1),
    ("vgg2_2 64x120x160", 64, 128, 128, 120, 160, 1),
    ("vgg3_2 64x60x80", 64, 256, 256, 60, 80, 1),
    ("vgg4_2 64x30x40", 64, 512, 512, 30, 40, 1),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, B, Cin, Cout, H, W, d in CASES:
    if flt not in name:
        continue
    pl = Plan(DEV)
    xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
    b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), dils=(d,), act=1)
    layer.split = True
    ob = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
    for _ in range(2): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): pl.run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * Cout * Cin * 9
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(xb[:2, :, :, :Cin].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=d, dilation=d).relu()
        err = (ob[:2, :, :, :Cout].permute(0, 3, 1, 2).double() - ref).abs().max().item()
    print("%-22s %-28s %8.3f ms  %6.1f TFLOP/s  err %.1e" % (pl.meta[-1][0], name, ms, fl / ms / 1e9, err), flush=True)
    del pl, xb, ob
    torch.cuda.empty_cache()

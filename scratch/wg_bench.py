"""Micro-benchmark of the bf16 3x3 weight-gradient kernels (wgrad_bf16.hip) on the shapes of the training plans.
usage: [EGNE_WGRAD3_WIDE=0] python scratch/wg_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import egne_amd  # noqa
from egne_amd import _lib
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0'); BF = torch.bfloat16
CASES = [("32->32 240x320", 128, 32, 32, 240, 320), ("64->64 240x320 (chz64)", 128, 64, 64, 240, 320), ("64->64 120x160", 128, 64, 64, 120, 160),
         ("128->128 120x160 (chz64)", 128, 128, 128, 120, 160), ("96->96 60x80", 128, 96, 96, 60, 80), ("192->192 60x80 (chz64)", 128, 192, 192, 60, 80),
         ("128->128 30x40", 128, 128, 128, 30, 40), ("256->256 30x40 (chz64)", 128, 256, 256, 30, 40), ("180->180 30x40", 64, 180, 180, 30, 40),
         ("100->100 60x80", 64, 100, 100, 60, 80), ("62->62 120x160", 64, 62, 62, 120, 160)]
for name, B, Cin, Cout, H, W in CASES:
    pl = Plan(DEV, dtype=BF); pl.train = True
    xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5)); b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
    w.grad, b.grad = torch.zeros_like(w), torch.zeros_like(b)
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
    ob = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
    bw = pl.build_backward()
    pl.run(); pl.zero_grads(); pl.gbuf(ob).normal_(); bw.run()
    fn, args, _ = [c for c in bw.calls if c[2].endswith(".wgrad")][0]
    st = _lib.stream_ptr()
    for _ in range(2):
        fn(*args, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * Cout * Cin * 9
    by = 2.0 * B * H * W * (pad8(Cin) + pad8(Cout))
    print("%-28s %8.3f ms  %7.1f TFLOP/s  %5.2f TB/s" % (name, ms, fl / ms / 1e9, by / ms / 1e9), flush=True)
    del pl, xb, ob, bw
    torch.cuda.empty_cache()

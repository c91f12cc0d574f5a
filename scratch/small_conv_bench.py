"""Per-launch time of the small-problem form of the flat split-f16 kernel on the layers that dominate a one- or two-frame call."""
import sys, torch
sys.path.insert(0, ".")
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
dev = torch.device("cuda:0")
cases = [("vgg.conv4_2", 2, 512, 512, 30, 40, 1), ("vgg.conv5_2", 2, 512, 512, 30, 40, 2), ("vgg.conv3_2", 2, 256, 256, 60, 80, 1), ("vgg.conv4_1", 2, 256, 512, 30, 40, 1),
         ("ms4.conv", 2, 512, 32, 30, 40, 1), ("ms3.conv", 2, 256, 32, 60, 80, 1), ("ms2.conv", 2, 128, 32, 120, 160, 1), ("enc.b2.conv2.b", 2, 45, 45, 60, 80, 1),
         ("enc.b3.conv1", 2, 56, 64, 30, 40, 1), ("dec.up4", 2, 153, 153, 30, 40, 1), ("vgg.conv4_2@B1", 1, 512, 512, 30, 40, 1)]
g = torch.Generator().manual_seed(0)
for name, B, Cin, Cout, H, W, d in cases:
    pl = Plan(dev)
    x = pl.buf(B, H, W, pad8(Cin)); x.normal_()
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)); b = torch.nn.Parameter(torch.randn(Cout, device=dev))
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), dils=(d,), act=1); layer.split = True
    out = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [Piece(x, 0, Cin)], Piece(out, 0, Cout), B, H, W)
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): pl.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * B * H * W * Cout * Cin * 9
    print("%-16s %-18s %7.1f us  %6.1f TFLOP/s" % (name, pl.meta[-1][0], us, fl / us / 1e6), flush=True)

"""Deep trunk kernel on plain f16 operands: two-stage (conv_f16x3_big.hip, NP = 1) against the four-stage form (conv_f16_big1.hip).
usage: python scratch/big1_bench.py [B]"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan
DEV = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CASES = [("conv3_1", 128, 256, 60, 80, 1), ("conv3_2", 256, 256, 60, 80, 1), ("conv4_1", 256, 512, 30, 40, 1),
         ("conv4_2", 512, 512, 30, 40, 1), ("conv5_1", 512, 512, 30, 40, 2)]
engine.BIG_SPLIT_TAIL = False
for name, Cin, Cout, H, W, d in CASES:
    res = []
    for big1 in (False, True):
        engine.BIG1_ENABLED = big1
        pl = Plan(DEV)
        pl.f16_products = 1
        xb = pl.buf(B, H, W, Cin); xb.normal_().relu_()
        w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
        b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
        layer = ConvLayer([w], [b], [(Cin, Cin)], pad=(1, 1), dils=(d,), act=1)      # (padding counts taps)
        layer.split = True
        ob = pl.buf(B, H, W, Cout)
        pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
        for _ in range(3): pl.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): pl.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * W * Cout * Cin * 9
        res.append((ms, fl / ms / 1e9, ob.clone()))
        del pl
    same = torch.equal(res[0][2], res[1][2])
    print("%-8s B=%d  two-stage %7.3f ms %6.1f TFLOP/s   four-stage %7.3f ms %6.1f TFLOP/s   x%.2f  bit-identical %s" % (
        name, B, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] / res[1][0], same), flush=True)

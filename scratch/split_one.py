"""One split-f16 conv layer, 3 launches (for rocprofv3 --pmc).  usage: split_one.py B Cin Cout H W"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
B, Cin, Cout, H, W = map(int, sys.argv[1:6])
DEV = torch.device('cuda:0')
pl = Plan(DEV)
xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
layer.split = True
ob = pl.buf(B, H, W, pad8(Cout))
pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
for _ in range(3): pl.run()
torch.cuda.synchronize()

"""One conv layer, few launches (for rocprofv3 --pmc).  usage: conv_one.py halo|flat B Cin Cout H W"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
mode, B, Cin, Cout, H, W = sys.argv[1], *map(int, sys.argv[2:7])
DEV = torch.device('cuda:0')
engine.HALO_ENABLED = (mode == 'halo')
pl = Plan(DEV)
xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
ob = pl.buf(B, H, W, pad8(Cout))
pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
for _ in range(3): pl.run()
torch.cuda.synchronize()

"""Time attribution of the role-split 3x3 kernel (64 -> 64, 240x320, B=64) with parts switched off (EGNE_RS_DBG bits:
1 no weight refills, 2 no LDS operand reads, 4 no conversion, 16 no halo loads, 8 no output stores, 32 clock stamps)."""
import os, sys, subprocess
if len(sys.argv) == 1:
    for m16, dbg in ((0, 32), (0, 33)):
        env = dict(os.environ, EGNE_RS_DBG=str(dbg), EGNE_RS_M16=str(m16))
        print("m16=%d dbg=%d" % (m16, dbg), subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
sys.path.insert(0, '/root/repo')
import torch, ctypes as C
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
B, Cin, Cout, H, W = 64, 64, 64, 240, 320
pl = Plan(DEV)
xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5))
b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
layer.split = True
ob = pl.buf(B, H, W, pad8(Cout))
pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
stamps = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=DEV)
d = pl.calls[-1][1][0]._obj
d.stats_ws = stamps.data_ptr()
for _ in range(3): pl.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(1500): pl.run()          # ~2 s of back-to-back launches
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 1500
s = stamps.view(-1, 8).cpu().double()
clk = (s[:, 0] / s[:, 1] * 100).median().item()
cyc = (s[:, 0] / s[:, 2]).median().item()
print("phases per tile: mfma %.0f epilogue %.0f barrier %.0f" % tuple((s[:, k] / s[:, 2]).median().item() for k in (3, 4, 5)))
print(pl.meta[-1][0], "%.0f us %.0f TF/s  clock %.0f MHz  %.0f cycles per tile (MFMA alone 6912)" % (us, 2 * B * H * W * Cin * Cout * 9 / us / 1e6, clk, cyc))

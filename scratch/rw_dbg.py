"""Producer / consumer time per job of the resident-weights 3x3 kernel (64 -> 64, 240x320, B=64)."""
import sys
sys.path.insert(0, '/root/repo')
import ctypes as C
import numpy as np
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
B, Cin, Cout, H, W = 64, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 240, 320
pl = Plan(DEV)
xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5)); b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=1); layer.split = True
ob = pl.buf(B, H, W, pad8(Cout))
pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
print(pl.meta[-1][0])
L = pl.L
L.egne_rw_debug.restype = C.c_int; L.egne_rw_debug.argtypes = [C.c_int, C.c_void_p]
for dbg in (0, 64):
    L.egne_rw_debug(dbg, None)
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): pl.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    st = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L.egne_rw_debug(dbg, st.ctypes.data)
    st = st.reshape(256, 8, 4).astype(np.float64)
    jobs = np.maximum(st[:, :, 2], 1)
    print("dbg %d: %.0f us | per job: producer work %.0f wait %.0f | consumer work %.0f wait %.0f cycles (108 MFMAs = 3456)" % (dbg, us,
          np.median(st[:, :4, 0] / jobs[:, :4]), np.median(st[:, :4, 1] / jobs[:, :4]), np.median(st[:, 4:, 0] / jobs[:, 4:]), np.median(st[:, 4:, 1] / jobs[:, 4:])), flush=True)

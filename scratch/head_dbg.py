"""enc.head shape: convBlock 3x3 (1 -> 32, leaky) -> 3x3 (32 -> 32, leaky) at 240x320, B=128, statistics on."""
import os, sys
sys.path.insert(0, '/root/repo')
import ctypes as C
import numpy as np
import torch
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
B, H, W = 128, 240, 320
pl = Plan(DEV)
xb = pl.buf(B, H, W, 8); xb.normal_()
w1 = torch.nn.Parameter(torch.randn(32, 1, 3, 3, device=DEV) / 3); b1 = torch.nn.Parameter(torch.randn(32, device=DEV))
w2 = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV) / 17); b2 = torch.nn.Parameter(torch.randn(32, device=DEV))
l1 = ConvLayer([w1], [b1], [(1, 8)], pad=(1, 1), act=2)
l2 = ConvLayer([w2], [b2], [(32, 32)], pad=(1, 1), act=2)
l1.split = l2.split = True
ob = pl.buf(B, H, W, 128)
pl.conv_pair(l1, [Piece(xb, 0, 1, 8)], l2, Piece(ob, 32, 32), B, H, W, stats=(os.environ.get("STATS", "1") == "1"))
print([m[0] for m in pl.meta])
L = pl.L
L.egne_fused_debug.restype = C.c_int
L.egne_fused_debug.argtypes = [C.c_int, C.c_void_p]
for dbg in (0, 64, 0, 64):
    L.egne_fused_debug(dbg, None)
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 100
    for _ in range(n): pl.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    st = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L.egne_fused_debug(dbg, st.ctypes.data)
    raw = st.reshape(256, 8, 4)
    mma = np.median((raw[:, 4:, 2] >> np.uint64(32)).astype(np.float64))
    raw[:, :, 2] &= np.uint64(0xffffffff)
    st = raw.astype(np.float64)
    tiles = np.maximum(st[:, :, 2], 1)
    print("dbg %d: %.0f us | per tile: producer work %.0f wait %.0f | consumer work %.0f (mfma %.0f) wait %.0f" % (dbg, us,
          np.median(st[:, :4, 0] / tiles[:, :4]), np.median(st[:, :4, 1] / tiles[:, :4]), np.median(st[:, 4:, 0] / tiles[:, 4:]) + mma, mma,
          np.median(st[:, 4:, 1] / tiles[:, 4:])), flush=True)

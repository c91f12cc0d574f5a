"""Where the fused convBlock-head kernel (3x3 on 1 channel -> 3x3, utils.py:1039-1050) spends its time at the benchmarked shape
(128 frames = 2B, 240x320): per-wave s_memtime stamps, producers (waves 0-3) vs consumers (4-7)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import numpy as np
import torch
import egne_amd
from egne_amd import engine, _lib
from egne_amd.engine import ConvLayer, Piece, PlanarPiece, Plan, pad8
DEV = torch.device('cuda:0')
B, H, W = 128, 240, 320
pl = Plan(DEV)
x = torch.randn(B, 1, H, W, device=DEV)
pl.keep.append(x)
w1 = torch.nn.Parameter(torch.randn(32, 1, 3, 3, device=DEV) / 3)
b1 = torch.nn.Parameter(torch.randn(32, device=DEV))
w2 = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV) / 17)
b2 = torch.nn.Parameter(torch.randn(32, device=DEV))
l1 = ConvLayer([w1], [b1], [(1, 8)], pad=(1, 1), act=2)
l2 = ConvLayer([w2], [b2], [(32, 32)], pad=(1, 1), act=2)
l1.split = l2.split = True
ob = pl.buf(B, H, W, 32)
pl.conv_pair(l1, [PlanarPiece(x)], l2, Piece(ob, 0, 32), B, H, W, name="head")
print([m[0] for m in pl.meta])
L = pl.L
L.egne_fused_debug.restype = C.c_int
L.egne_fused_debug.argtypes = [C.c_int, C.c_void_p]
for dbg in (0, 0, 64, 65, 0, 0):
    L.egne_fused_debug(dbg, None)
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 100
    for _ in range(n): pl.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    if not dbg:
        print("plain: %.0f us" % us); continue
    st = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L.egne_fused_debug(dbg, st.ctypes.data)
    raw = st.reshape(256, 8, 4)
    mma = np.median((raw[:, 4:, 2] >> np.uint64(32)).astype(np.float64))
    raw[:, :, 2] &= np.uint64(0xffffffff)
    st = raw.astype(np.float64)
    tiles = st[:, :, 2]
    pw, pwait = np.median(st[:, :4, 0] / tiles[:, :4]), np.median(st[:, :4, 1] / tiles[:, :4])
    cw, cwait = np.median(st[:, 4:, 0] / tiles[:, 4:]), np.median(st[:, 4:, 1] / tiles[:, 4:])
    print("dbg %d: %.0f us | per tile: producer work %.0f wait %.0f | consumer work %.0f (mfma loop %.0f) wait %.0f cycles" % (dbg, us, pw, pwait, cw + mma, mma, cwait), flush=True)

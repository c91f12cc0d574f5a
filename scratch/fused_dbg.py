"""Where the fused 1x1 -> 3x3 kernel spends its time (b0.conv3 shape: 96 -> 32 -> 32 at 240x320, B=64): per-wave s_memtime stamps
of work vs barrier wait, producers (waves 0-3) and consumers (4-7); dbg bit 1 = producers issue no loads."""
import os, sys, subprocess
sys.path.insert(0, '/root/repo')
import ctypes as C
import torch
import egne_amd
from egne_amd import engine, _lib
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
B, H, W = [int(v) for v in os.environ.get('BHW', '64,240,320').split(',')]
chans = [int(c) for c in (sys.argv[1].split(',') if len(sys.argv) > 1 else "32,32,32".split(','))]
C1 = C2 = int(os.environ.get('C12', '32'))
pl = Plan(DEV)
pieces = []
uni = os.environ.get("UNI", "1") == "1"
if uni:      # slices of one buffer, as in a dense block (out | x | x1 | x22)
    xb = pl.buf(B, H, W, 32 + sum(pad8(c) for c in chans)); xb.normal_()
    o = 32
    for c in chans:
        pieces.append(Piece(xb, o, c)); o += pad8(c)
else:
    for c in chans:
        xb = pl.buf(B, H, W, pad8(c)); xb.normal_()
        pieces.append(Piece(xb, 0, c))
w1 = torch.nn.Parameter(torch.randn(C1, sum(chans), 1, 1, device=DEV) / 8)
b1 = torch.nn.Parameter(torch.randn(C1, device=DEV))
w2 = torch.nn.Parameter(torch.randn(C2, C1, 3, 3, device=DEV) / 17)
b2 = torch.nn.Parameter(torch.randn(C2, device=DEV))
l1 = ConvLayer([w1], [b1], [(p.C, p.Cp) for p in pieces])
l2 = ConvLayer([w2], [b2], [(C1, pad8(C1))], pad=(1, 1), act=2)
l1.split1 = l2.split = True
ob = pl.buf(B, H, W, pad8(C2))
pl.conv_pair(l1, pieces, l2, Piece(ob, 0, C2), B, H, W)
print([m[0] for m in pl.meta])
L = pl.L
L.egne_fused_debug.restype = C.c_int
L.egne_fused_debug.argtypes = [C.c_int, C.c_void_p]
import numpy as np
for dbg in (64, 65, 64, 65):
    L.egne_fused_debug(dbg, None)
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 300
    for _ in range(n): pl.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    st = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L.egne_fused_debug(dbg, st.ctypes.data)
    raw = st.reshape(256, 8, 4)
    mma = np.median((raw[:, 4:, 2] >> np.uint64(32)).astype(np.float64))
    raw[:, :, 2] &= np.uint64(0xffffffff)
    st = raw.astype(np.float64)
    tiles = st[:, :, 2]
    pw, pwait = np.median(st[:, :4, 0] / tiles[:, :4]), np.median(st[:, :4, 1] / tiles[:, :4])
    cw, cwait = np.median(st[:, 4:, 0] / tiles[:, 4:]), np.median(st[:, 4:, 1] / tiles[:, 4:])
    clk = np.median((st[:, :, 0] + st[:, :, 1]) / st[:, :, 3]) * 100
    print("dbg %d: %.0f us  clock %.0f MHz | per tile: producer work %.0f wait %.0f | consumer work %.0f (mfma loop %.0f) wait %.0f cycles" % (dbg, us, clk, pw, pwait, cw + mma, mma, cwait), flush=True)

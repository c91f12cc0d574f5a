"""Producer / consumer time of the one-launch dilated group (B=64, 240x320): s_memtime stamps of work vs barrier wait per strip;
dbg bit 1 = producers issue no loads, bit 2 = no conversion / LDS writes."""
import sys
sys.path.insert(0, '/root/repo')
import ctypes as C
import numpy as np
import torch
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan
DEV = torch.device('cuda:0')
B, H, W = 64, 240, 320
g = torch.Generator().manual_seed(0)
pl = Plan(DEV)
ob = pl.buf(B, H, W, 32); ob.copy_(torch.relu(torch.randn(B, H, W, 32, generator=g)).to(DEV))
ws = [torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=g) / 17).to(DEV)) for _ in range(3)]
bs = [torch.nn.Parameter(torch.randn(32, generator=g).to(DEV)) for _ in range(3)]
layer = ConvLayer(ws, bs, [(32, 32)], pad=(1, 1), dils=(4, 8, 12), act=1)
layer.split = True
out = pl.buf(B, H, W, 32)
pl.conv(layer, [Piece(ob, 0, 32)], Piece(out, 0, 32), B, H, W, residual=Piece(ob, 0, 32))
L = pl.L
L.egne_msdil_debug.restype = C.c_int
L.egne_msdil_debug.argtypes = [C.c_int, C.c_void_p]
for dbg in (64, 66, 67):
    L.egne_msdil_debug(dbg, None)
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 100
    for _ in range(n): pl.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    st = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L.egne_msdil_debug(dbg, st.ctypes.data)
    raw = st.reshape(256, 8, 4)
    pwait = np.median((raw[:, :4, 2] >> np.uint64(32)).astype(np.float64))
    raw[:, :, 2] &= np.uint64(0xffffffff)
    st = raw.astype(np.float64)
    tiles = np.maximum(st[:, :, 2], 1)
    print("   producer per tile: setup+weights issue %.0f | item loop %.0f | weights to LDS %.0f | barrier wait %.0f" % (
          np.median(st[:, :4, 0] / tiles[:, :4]), np.median(st[:, :4, 1] / tiles[:, :4]), np.median(st[:, :4, 3] / tiles[:, :4]), pwait))
    clk = np.median((st[:, :, 0] + st[:, :, 1]) / np.maximum(st[:, :, 3], 1)) * 100
    print("dbg %d: %.0f us clock %.0f MHz | per tile: producer work %.0f wait %.0f | consumer work %.0f wait %.0f cycles" % (dbg, us, clk,
          np.median(st[:, :4, 0] / tiles[:, :4]), np.median(st[:, :4, 1] / tiles[:, :4]), np.median(st[:, 4:, 0] / tiles[:, 4:]),
          np.median(st[:, 4:, 1] / tiles[:, 4:])), flush=True)

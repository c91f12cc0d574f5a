"""Where the waves of the one-launch dilated-group kernel spend their cycles (egne_msdil_debug stamps, dbg bit 64): per producer wave the
cycles in weight loads / gather+convert / weight stores and at barriers, per consumer wave MFMA work vs waiting at barriers."""
import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch, numpy as np
import egne_amd
from egne_amd import _lib
from egne_amd.engine import ConvLayer, Piece, Plan
DEV = torch.device('cuda:0')
B, H, W = 64, 240, 320
g = torch.Generator().manual_seed(0)
pl = Plan(DEV)
ob = pl.buf(B, H, W, 32); ob.copy_(torch.relu(torch.randn(B, H, W, 32, generator=g)).to(DEV))
ws = [torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=g) / 17).to(DEV)) for _ in range(3)]
bs = [torch.nn.Parameter(torch.randn(32, generator=g).to(DEV)) for _ in range(3)]
layer = ConvLayer(ws, bs, [(32, 32)], pad=(1, 1), dils=(4, 8, 12), act=1); layer.split = True
out = pl.buf(B, H, W, 32)
pl.conv(layer, [Piece(ob, 0, 32)], Piece(out, 0, 32), B, H, W, residual=Piece(ob, 0, 32))
print(pl.meta[-1][0])
L = _lib.lib()
L.egne_msdil_debug.restype = C.c_int
L.egne_msdil_debug.argtypes = [C.c_int, C.c_void_p]
for _ in range(3): pl.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [pl.run() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
print("launch (both column classes): %.1f us" % (e0.elapsed_time(e1) * 100))
for dbg, what in ((64, "stamps"), (64 | 2, "stamps, conversion off"), (64 | 1, "stamps, loads off")):
    assert L.egne_msdil_debug(dbg, None) == 0
    pl.run(); torch.cuda.synchronize()
    st = np.zeros(256 * 8 * 4, np.uint64)
    assert L.egne_msdil_debug(0, st.ctypes.data_as(C.c_void_p)) == 0
    st = st.reshape(256, 8, 4).astype(np.int64)
    ntile = st[:, 4:, 2].max()
    prod = st[:, :4]; cons = st[:, 4:]
    print("%s: tiles per workgroup %d" % (what, ntile))
    print("  producers (mean cycles per workgroup): weights+issue %d, gather/convert %d, weight stores %d, barrier wait per tile %d" % (
        prod[..., 0].mean(), prod[..., 1].mean(), prod[..., 3].mean(), (prod[..., 2] >> 32).mean()))
    print("  consumers: work %d, wait %d" % (cons[..., 0].mean(), cons[..., 1].mean()))

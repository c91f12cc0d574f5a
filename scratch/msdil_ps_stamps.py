"""msdil_stamps.py for the split-pair input form (egne_seg.presplit): where producer / consumer waves spend their cycles when a
producer item is a 16-byte copy.  usage: python3 scratch/msdil_ps_stamps.py [H W]"""
import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch, numpy as np
import egne_amd
from egne_amd import _lib
from egne_amd.engine import ConvLayer, Piece, Plan, SplitScale
DEV = torch.device('cuda:0')
B = 64
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 320)
g = torch.Generator().manual_seed(0)
pl = Plan(DEV)
o = torch.relu(torch.randn(B, H, W, 32, generator=g)).to(DEV)
s = 256.0
hi = (o * s).half()
lo = (o * s - hi.float()).half()
ob = pl.buf(B, H, W, 32)
ob.view(torch.float16).reshape(B, H, W, 2, 32)[..., 0, :] = hi
ob.view(torch.float16).reshape(B, H, W, 2, 32)[..., 1, :] = lo
ws = [torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=g) / 17).to(DEV)) for _ in range(3)]
bs = [torch.nn.Parameter(torch.randn(32, generator=g).to(DEV)) for _ in range(3)]
layer = ConvLayer(ws, bs, [(32, 32)], pad=(1, 1), dils=(4, 8, 12), act=1); layer.split = True
if os.environ.get("PERM"):
    layer.k_perm = torch.tensor([16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3) for p in range(32)])
out = pl.buf(B, H, W, 32)
po = Piece(ob, 0, 32); po.presplit = SplitScale(); po.presplit.value = s
pl.conv(layer, [po], Piece(out, 0, 32), B, H, W, residual=po)
print(pl.meta[-1][0])
L = _lib.lib()
DBG = L.egne_msdil_debug if os.environ.get('EGNE_MSDIL_PS_OLD') else L.egne_msdil_ps_debug
DBG.restype = C.c_int
DBG.argtypes = [C.c_int, C.c_void_p]
for _ in range(3): pl.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [pl.run() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
print("launch (both column classes): %.1f us" % (e0.elapsed_time(e1) * 100))
for dbg, what in ((64, "stamps"), (64 | 2, "stamps, LDS copies off"), (64 | 1, "stamps, loads off")):
    assert DBG(dbg, None) == 0
    pl.run(); torch.cuda.synchronize()
    st = np.zeros(256 * 8 * 4, np.uint64)
    assert DBG(0, st.ctypes.data_as(C.c_void_p)) == 0
    st = st.reshape(256, 8, 4).astype(np.int64)
    ntile = st[:, 4:, 2].max()
    prod = st[:, :4]; cons = st[:, 4:]
    if not os.environ.get('EGNE_MSDIL_PS_OLD'):
        print("%s: tiles per workgroup %d; symmetric waves: work %d, wait at barriers %d cycles per workgroup" % (what, st[..., 2].max(), st[..., 0].mean(), st[..., 1].mean()))
        continue
    print("%s: tiles per workgroup %d" % (what, ntile))
    print("  producers (mean cycles per workgroup): weights+issue %d, gather/copy %d, weight stores %d, barrier wait per tile %d" % (
        prod[..., 0].mean(), prod[..., 1].mean(), prod[..., 3].mean(), (prod[..., 2] >> 32).mean()))
    print("  consumers: work %d, wait %d" % (cons[..., 0].mean(), cons[..., 1].mean()))

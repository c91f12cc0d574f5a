import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ["EGNE_MSDIL_DBG"] = "16"
import torch, numpy as np
import egne_amd
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
for _ in range(3): pl.run()
torch.cuda.synchronize()
st = out.view(torch.int64).reshape(-1)[:8 * 9 * 8].cpu().numpy().reshape(8, 9, 8)
t0 = st[0, 0, 0]
for w in range(8):
    if w < 4:
        print("P%d" % w, " ".join("[%d +%d +%d]" % (st[w, s, 0] - t0, st[w, s, 1] - st[w, s, 0], st[w, s, 2] - st[w, s, 1]) for s in range(9)))
    else:
        print("C%d" % w, " ".join("[%d wait %d]" % (st[w, s, 0] - t0, st[w, s, 1] - st[w, s, 0]) for s in range(9)))

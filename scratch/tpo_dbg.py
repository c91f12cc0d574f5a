import sys
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import torch.nn.functional as F
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B, Cin, Cout, H, W = 8, 32, 32, 240, 320
x = torch.randn(B, Cin, H, W, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / 17; b = torch.randn(Cout, generator=g)
truth = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
pl = Plan(DEV)
xb = pl.buf(B, H, W, 32); xb.copy_(x.permute(0, 2, 3, 1).to(DEV))
layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(32, 32)], pad=(1, 1), act=2); layer.split = True
ob = pl.buf(B, H, W, 32)
pl.conv(layer, [Piece(xb, 0, 32)], Piece(ob, 0, 32), B, H, W)
pl.run(); torch.cuda.synchronize()
got = ob.cpu().permute(0, 3, 1, 2).double()
err = (got - truth).abs()
print(pl.meta[-1][0], "max err", err.max().item())
bad = (err > 1e-3)
print("bad fraction", bad.float().mean().item())
bb = bad.any(1)   # [B,H,W]
for bi in range(B):
    ys, xs = np.nonzero(bb[bi].numpy())
    if len(ys): print("frame", bi, "bad px", len(ys), "y range", ys.min(), ys.max(), "x range", xs.min(), xs.max(), "rows mod 8:", sorted(set((ys % 8).tolist())), "cols mod 32 count", len(set((xs % 32).tolist())))
cb = bad.any(0).any(1).any(1)
print("bad channels", np.nonzero(cb.numpy())[0].tolist())

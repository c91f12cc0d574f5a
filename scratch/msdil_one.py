"""One MSBlock dilated group (B=64, 240x320 unless argv says otherwise), a few launches: profiling target.
usage: python scratch/msdil_one.py [B H W] [reps]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan
DEV = torch.device('cuda:0')
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 240, 320)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator().manual_seed(0)
pl = Plan(DEV)
ob = pl.buf(B, H, W, 32); ob.copy_(torch.relu(torch.randn(B, H, W, 32, generator=g)).to(DEV))
ws = [torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=g) / 17).to(DEV)) for _ in range(3)]
bs = [torch.nn.Parameter(torch.randn(32, generator=g).to(DEV)) for _ in range(3)]
layer = ConvLayer(ws, bs, [(32, 32)], pad=(1, 1), dils=(4, 8, 12), act=1)
layer.split = True
out = pl.buf(B, H, W, 32)
pl.conv(layer, [Piece(ob, 0, 32)], Piece(out, 0, 32), B, H, W, residual=Piece(ob, 0, 32))
for _ in range(2): pl.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): pl.run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("msdil %dx%dx%d: %.3f ms  %.1f TFLOP/s  (%s)" % (B, H, W, ms, 2.0 * B * H * W * 32 * 32 * 27 / ms / 1e9, pl.meta[-1][0]), flush=True)

"""rs kernel: transposed-output path against the plain one on ESF-like shapes (env EGNE_RS_TPO toggled per subprocess)."""
import os, sys, subprocess, pickle
if len(sys.argv) == 1:
    outs = {}
    for tpo in ("0", "1"):
        r = subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, EGNE_RS_TPO=tpo, EGNE_RW="0"), capture_output=True)
        outs[tpo] = pickle.loads(r.stdout)
    for k in outs["0"]:
        a, b = outs["0"][k], outs["1"][k]
        print(k, "max diff %.3e" % float(abs(a - b).max()), "max |a| %.3e" % float(abs(a).max()))
    sys.exit(0)
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
res = {}
g = torch.Generator().manual_seed(3)
for (B, Cin, Cout, H, W, norm, off, stride) in [(3, 32, 32, 240, 320, True, 64, 128), (3, 32, 32, 240, 320, False, 0, 32), (2, 62, 62, 120, 160, False, 0, 64),
                                                 (2, 38, 64, 120, 160, True, 104, 232), (4, 32, 32, 61, 83, True, 32, 96)]:
    pl = Plan(DEV)
    xb = pl.buf(B, H, W, pad8(Cin)); xb.copy_(torch.randn(B, H, W, pad8(Cin), generator=g).to(DEV)); xb[..., Cin:] = 0
    w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV)); b = torch.nn.Parameter(torch.randn(Cout, generator=g).to(DEV))
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=2); layer.split = True
    ob = pl.buf(B, H, W, stride)
    px = Piece(xb, 0, Cin)
    if norm:
        sc = torch.rand(B, pad8(Cin), generator=g).to(DEV) + 0.5; sh = torch.randn(B, pad8(Cin), generator=g).to(DEV)
        pl.keep += [sc, sh]
        px = px.with_norm(sc, sh, 2)
    pl.conv(layer, [px], Piece(ob, off, Cout), B, H, W)
    for _ in range(2): pl.run()
    torch.cuda.synchronize()
    res["%dx%d->%d %dx%d norm=%d off=%d %s" % (B, Cin, Cout, H, W, norm, off, pl.meta[-1][0])] = ob.cpu().numpy()
sys.stdout.buffer.write(pickle.dumps(res))

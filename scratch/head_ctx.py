"""Does enc.head slow down when it follows MFMA-heavy launches (clock state) or cold caches?"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import egne_amd
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0')
B, H, W = 128, 240, 320
pl = Plan(DEV)
xb = pl.buf(B, H, W, 8); xb.normal_()
w1 = torch.nn.Parameter(torch.randn(32, 1, 3, 3, device=DEV) / 3); b1 = torch.nn.Parameter(torch.randn(32, device=DEV))
w2 = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV) / 17); b2 = torch.nn.Parameter(torch.randn(32, device=DEV))
l1 = ConvLayer([w1], [b1], [(1, 8)], pad=(1, 1), act=2); l2 = ConvLayer([w2], [b2], [(32, 32)], pad=(1, 1), act=2)
l1.split = l2.split = True
ob = pl.buf(B, H, W, 128)
pl.conv_pair(l1, [Piece(xb, 0, 1, 8)], l2, Piece(ob, 32, 32), B, H, W, stats=True)
# a heavy neighbour: 512 -> 512 at 30x40 (deep trunk kernel) and a big streaming copy to flush caches
pb = Plan(DEV)
hx = pb.buf(64, 30, 40, 512); hx.normal_()
wh = torch.nn.Parameter(torch.randn(512, 512, 3, 3, device=DEV) / 68); bh = torch.nn.Parameter(torch.randn(512, device=DEV))
lh = ConvLayer([wh], [bh], [(512, 512)], pad=(1, 1), act=1); lh.split = True
ho = pb.buf(64, 30, 40, 512)
pb.conv(lh, [Piece(hx, 0, 512)], Piece(ho, 0, 512), 64, 30, 40)
big = torch.empty(1 << 28, device=DEV)      # 1 GiB
def timed(pre):
    for _ in range(2):
        pre(); pl.run()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(10):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); pl.run(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 100
src = torch.randn_like(xb)
print("after rewriting its input (315 MB copy) %.0f us" % timed(lambda: xb.copy_(src)))
print("after rewriting input + 5 trunk convs before %.0f us" % timed(lambda: ([pb.run() for _ in range(5)], xb.copy_(src))))
print("back to back            %.0f us" % timed(lambda: None))
print("after 5 trunk convs     %.0f us" % timed(lambda: [pb.run() for _ in range(5)]))
print("after a 1 GiB fill      %.0f us" % timed(lambda: big.fill_(1.0)))
print("after both              %.0f us" % timed(lambda: ([pb.run() for _ in range(5)], big.fill_(1.0))))

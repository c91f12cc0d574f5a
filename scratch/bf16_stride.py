"""Does a 32-channel bf16 slice (64 contiguous bytes per pixel) inside a wider NHWC buffer stream slower than the same tensor in a
buffer of its own?  3x3 32->32 and act_bwd_bias on 240x320, 128 frames."""
import sys
import torch
sys.path.insert(0, ".")
import egne_amd  # noqa
from egne_amd.engine import ConvLayer, Piece, Plan
from egne_amd import _lib
DEV = torch.device("cuda:0")
B, H, W = 128, 240, 320
for dt in (torch.bfloat16, torch.float32):
    for ctot in (32, 128):
        pl = Plan(DEV, dtype=dt)
        if dt == torch.float32:
            pl.dyn_scales = True
        xb, yb = pl.buf(B, H, W, ctot), pl.buf(B, H, W, ctot)
        xb.normal_(); 
        off = 0 if ctot == 32 else 32
        w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV) / 17)
        b = torch.nn.Parameter(torch.zeros(32, device=DEV))
        layer = ConvLayer([w], [b], [(32, 32)], pad=(1, 1), act=2)
        layer.split = True
        pl.conv(layer, [Piece(xb, off, 32)], Piece(yb, off, 32), B, H, W)
        L = _lib.lib()
        ws = torch.zeros(int(L.egne_act_bwd_bias_workspace_bytes(B * H * W, 32)) // 8 + 1, dtype=torch.float64, device=DEV)
        fn = L.egne_act_bwd_bias_bf16 if dt == torch.bfloat16 else L.egne_act_bwd_bias
        es = 2 if dt == torch.bfloat16 else 4
        for _ in range(3):
            pl.run()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        st = _lib.stream_ptr()
        e0.record()
        for _ in range(10):
            pl.run()
        e1.record()
        for _ in range(10):
            _lib.check(fn(xb.data_ptr() + es * off, ctot, 0, yb.data_ptr() + es * off, ctot, 0, 2, 32, B * H * W, None, 32, 0, ws.data_ptr(), st))
        e2.record()
        torch.cuda.synchronize()
        npx = B * H * W
        tc, ta = e0.elapsed_time(e1) / 10, e1.elapsed_time(e2) / 10
        print("%s buffer of %3d channels: conv3x3 32->32 %.3f ms = %.2f TB/s; act_bwd_bias %.3f ms = %.2f TB/s (3 passes over the slice)"
              % (str(dt)[6:], ctot, tc, npx * 64 * es / tc / 1e9, ta, npx * 96 * es / ta / 1e9))
        del pl, xb, yb
        torch.cuda.empty_cache()

"""Cycle stamps of the resident-weights 3x3 kernel on plain f16 operands (conv1_2 class: 64 -> 64 at 240x320, f16 tensors in and out):
per wave, cycles between barriers (work) and at barriers (wait), per job.  usage: python scratch/rw1_stamps.py [B]"""
import ctypes as C, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import egne_amd
from egne_amd import _lib, engine
from egne_amd.engine import ConvLayer, Piece, Plan, SplitScale
DEV = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = _lib.lib()
L.egne_rw_debug.restype = C.c_int; L.egne_rw_debug.argtypes = [C.c_int, C.c_void_p]
for name, Cin, Cout, H, W, f16io in (("conv1_2 f16", 64, 64, 240, 320, True), ("conv1_2 fp32", 64, 64, 240, 320, False), ("ms1 conv f16 in", 64, 32, 240, 320, True)):
    pl = Plan(DEV); pl.f16_products = 1
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5)); b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
    layer = ConvLayer([w], [b], [(Cin, Cin)], pad=(1, 1), act=1); layer.split = True
    if f16io:
        xb = pl.buf16(B, H, W, Cin); xb.copy_((torch.randn(B, H, W, Cin, device=DEV).relu() * 64).half())
        pin = Piece(xb, 0, Cin); pin.f16s = SplitScale(); pin.f16s.value, pin.f16s.vmax = 64.0, 5.0
        ob = pl.buf16(B, H, W, Cout); pout = Piece(ob, 0, Cout); pout.f16s = SplitScale()
    else:
        xb = pl.buf(B, H, W, Cin); xb.normal_().relu_(); pin = Piece(xb, 0, Cin)
        ob = pl.buf(B, H, W, Cout); pout = Piece(ob, 0, Cout)
    pl.conv(layer, [pin], pout, B, H, W)
    assert pl.meta[0][0] == "conv_f16x3:rw", pl.meta
    for _ in range(3): pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): pl.run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    L.egne_rw_debug(64, None)
    pl.run(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L.egne_rw_debug(0, buf.ctypes.data_as(C.c_void_p))
    st = buf.reshape(256, 8, 4).astype(np.float64)
    jobs = st[:, :, 2]
    print("   consumers: epilogue part %.0f cycles per job" % (st[:, 4:, 3] / np.maximum(st[:, 4:, 2], 1)).mean())
    prod, cons = st[:, :4], st[:, 4:]
    print("%-18s B=%d  %.3f ms;  jobs per workgroup %.0f;  per job: producers work %.0f wait %.0f | consumers work %.0f wait %.0f cycles (100 MHz stamps x clock ratio)" % (
        name, B, ms, jobs.mean(), (prod[:, :, 0] / np.maximum(prod[:, :, 2], 1)).mean(), (prod[:, :, 1] / np.maximum(prod[:, :, 2], 1)).mean(),
        (cons[:, :, 0] / np.maximum(cons[:, :, 2], 1)).mean(), (cons[:, :, 1] / np.maximum(cons[:, :, 2], 1)).mean()), flush=True)

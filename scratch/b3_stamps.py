"""Cycle stamps of the bf16 3x3 kernel (a diagnostic build: make -C csrc clean conv3x3_bf16.o EXTRA=-DEGNE_B3_STAMPS && make -C csrc):
per wave the cycles spent in its job loop / at the barrier / in the hand-over.  usage: python scratch/b3_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import egne_amd  # noqa
from egne_amd import _lib
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0'); BF = torch.bfloat16
L = _lib.lib()
for name, B, Cin, Cout, H, W in [("64->64 240x320", 128, 64, 64, 240, 320), ("128->128 120x160", 128, 128, 128, 120, 160), ("32->32 240x320", 128, 32, 32, 240, 320)]:
    pl = Plan(DEV, dtype=BF)
    xb = pl.buf(B, H, W, pad8(Cin)); xb.normal_()
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) / (3 * Cin ** 0.5)); b = torch.nn.Parameter(torch.randn(Cout, device=DEV))
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
    ob = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [Piece(xb, 0, Cin)], Piece(ob, 0, Cout), B, H, W)
    for _ in range(3):
        pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); pl.run(); e1.record(); torch.cuda.synchronize()
    host = (C.c_ulonglong * (256 * 8 * 4))()
    L.egne_b3_read_stamps.argtypes = [C.c_void_p]
    assert L.egne_b3_read_stamps(host) == 0
    a = np.ctypeslib.as_array(host).reshape(256, 8, 4).astype(np.float64)
    ms = e0.elapsed_time(e1)
    tiles = B * ((H + 7) // 8) * ((W + 31) // 32)
    nrun = (Cout + 63) // 64 if Cout >= 64 and os.environ.get("EGNE_B3_MB", "2") != "1" else (Cout + 31) // 32
    per_wg = tiles / (256 // nrun)
    print("%s: %.3f ms, %.0f tiles per workgroup; median cycles per TILE:" % (name, ms, per_wg))
    p, c = np.median(a[:, :4], axis=(0, 1)) / per_wg, np.median(a[:, 4:], axis=(0, 1)) / per_wg
    print("   producers: step %.0f  barrier wait %.0f  total %.0f" % (p[0], p[1], p[3]))
    print("   consumers: jobs %.0f  barrier wait %.0f  hand-over %.0f  total %.0f   (MFMA issue: %d cycles per tile)" % (c[0], c[1], c[2], c[3], 9 * ((Cin + 31) // 32) * 8 * min(2, (Cout + 31) // 32) * 16))
    del pl, xb, ob

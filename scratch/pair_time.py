import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egne_amd import _lib
L = _lib.lib(); st = _lib.stream_ptr(); DEV = "cuda:0"
B, H, W, Cs, Ca = 64, 240, 320, 32, 32
g = torch.randn(B, H, W, Cs, device=DEV).bfloat16()
w = torch.randn(Cs, Ca, 3, 3, device=DEV)
ws = torch.zeros(int(L.egne_act_bwd_bias_workspace_bytes(B * H * W, Cs)) // 8 + 1, dtype=torch.float64, device=DEV)
wsp = torch.zeros(int(L.egne_pair_bias_bwd_workspace_bytes(B, Cs)) // 8 + 1, dtype=torch.float64, device=DEV)
da, db = torch.zeros(Ca, device=DEV), torch.zeros(Cs, device=DEV)
_lib.check(L.egne_act_bwd_bias_bf16(g.data_ptr(), Cs, 0, None, 0, 0, 0, Cs, B * H * W, None, Cs, 1, ws.data_ptr(), st))
for dbg in (1, 2, 3, 0):
    os.environ["EGNE_PAIR_DBG"] = str(dbg)
    for _ in range(3):
        _lib.check(L.egne_pair_bias_bwd_bf16(g.data_ptr(), Cs, 0, Cs, B, H, W, ws.data_ptr(), w.data_ptr(), Cs, Ca, db.data_ptr(), da.data_ptr(), wsp.data_ptr(), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _lib.check(L.egne_pair_bias_bwd_bf16(g.data_ptr(), Cs, 0, Cs, B, H, W, ws.data_ptr(), w.data_ptr(), Cs, Ca, db.data_ptr(), da.data_ptr(), wsp.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
    print("dbg", dbg, "%.1f us per call (border + final)" % (e0.elapsed_time(e1) * 1e3 / 50))

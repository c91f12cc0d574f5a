"""Time per evaluation round of the ellipse search: one golden case at a time (eight-wave form) and 16 copies of it (pair form)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import egne_amd
from egne_amd import _lib
from egne_amd.utils import _mesh_axes
g = np.load("tests/golden/fit_cases.npz")
H, W = 240, 320
masks = torch.from_numpy(np.stack([np.unpackbits(m).reshape(H, W) for m in g["masks"]]).astype(np.int64)).cuda()
L = _lib.lib()
xs, ys = _mesh_axes(H, W, masks.device)
for case in (10, 0, 18, 11):
    for n in (1, 16):
        fo = torch.full((n,), case, dtype=torch.int32, device="cuda")
        cl = torch.ones(n, dtype=torch.int32, device="cuda")
        ini = torch.from_numpy(np.repeat(g["inits"][case][None], n, 0)).cuda()
        out = torch.empty((n, 5), dtype=torch.float64, device="cuda"); ev = torch.zeros(n, dtype=torch.int32, device="cuda")
        def run():
            _lib.check(L.egne_ellipse_fit(masks.data_ptr(), len(masks), fo.data_ptr(), cl.data_ptr(), n, H, W, xs.data_ptr(), ys.data_ptr(),
                                          ini.data_ptr(), out.data_ptr(), ev.data_ptr(), _lib.stream_ptr()), "fit")
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("case %2d n=%2d: %7.1f us, %3d reference evaluations -> %.2f us per reference evaluation" % (case, n, us, int(ev[0]), us / int(ev[0])), flush=True)

"""Where the 3 ms of the with-fit step go: the fit launch alone, and the inference step with / without the fit stage, with the
host-side pieces (pinned copy, event synchronisation) switched on one at a time."""
import sys, time, types
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
a = types.SimpleNamespace(gpus=1, batch=64, train_batch=256, config="baseline_edge", chz=32, no_pipeline=False, layers=False, fit=False)
bn = bench.Bench(a)
torch = bn.torch
from egne_amd.utils import fit_ellipses_from_pred
for fit in (False, True):
    B, dt, _ = bn.leg_infer(20, 5, fit=fit, events=False)
    print("fit=%s pipelined: %.3f ms/step" % (fit, 1e3 * dt / 20))
    B, dt, _ = bn.leg_infer(20, 5, fit=fit, events=False, pipeline=False)
    print("fit=%s back to back: %.3f ms/step" % (fit, 1e3 * dt / 20))
# the fit launch alone on the last mask / elPred
t = bn.batch(64)
from egne_amd.utils import calc_edge
with torch.no_grad():
    e = calc_edge(bn.args, t["img"], bn.bd, bn.dev)
    out = bn.net(t["img"], e, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"], t["ID"], t["alpha"])
mask, elp = bn.net.predictions(), out[1]
for _ in range(3):
    r = fit_ellipses_from_pred(mask, elp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    r = fit_ellipses_from_pred(mask, elp)
e1.record()
torch.cuda.synchronize()
print("fit launch alone (seeds + 128 searches): %.3f ms" % (e0.elapsed_time(e1) / 10))
t0 = time.perf_counter()
for _ in range(10):
    r = fit_ellipses_from_pred(mask, elp)
print("host time to queue it: %.3f ms" % ((time.perf_counter() - t0) * 100))
torch.cuda.synchronize()
# host launch time of one inference step (no sync)
step = bn.infer_step(64, False, pipeline=False)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
th = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print("host time to queue one inference step: %.3f ms" % (1e3 * th))

"""-m gpu: the entry points (test.py / train.py / evaluate.py mirrors) end to end on synthetic frames."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_test_py_synthetic(capsys):
    from egne_amd import test as T
    miou, pd, idist, loss = T.main(["--synthetic", "4", "--batchsize", "2", "--setting", "configs/baseline_edge.yaml"])
    assert np.isfinite(miou) and 0 <= miou <= 1 and np.isfinite(loss)
    assert "mIoU" in capsys.readouterr().out


def _amplified_eyes(first, factor=3000.0):
    """_entry.SyntheticEyes whose frames are multiplied by ``factor`` from the ``first``-th FETCHED sample on (whatever its index: the
    training loader shuffles): far beyond the 32x head-room of pre-scales calibrated on the first batch (engine.Plan: the split-f16
    kernels then produce non-finite values and flag it)."""
    from egne_amd import _entry
    base = _entry.SyntheticEyes

    class Amplified(base):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.fetched = 0

        def __getitem__(self, i):
            t = super().__getitem__(i)
            self.fetched += 1
            return ((t[0] * factor,) + tuple(t[1:])) if self.fetched > first else t
    return Amplified


def _count_overflows(monkeypatch):
    """Counts the True answers of BDCN.overflowed (each one clears the word and marks the plan for re-calibration)."""
    from egne_amd.bdcn_new import BDCN
    seen = []
    orig = BDCN.overflowed

    def spy(self):
        r = orig(self)
        if r:
            seen.append(1)
        return r
    monkeypatch.setattr(BDCN, "overflowed", spy)
    return seen


def test_test_py_redoes_batches_beyond_the_calibrated_prescales(monkeypatch, capsys):
    """test.py's loop (calc_acc, test.py:31-252 of the reference) on a set whose later frames are 3000x the calibration batch: the edge
    network's plan overflows its f16 pre-scales, BOTH plans' words are read (round-5 advisor finding: a short-circuit `or` left the edge
    plan's word set and its stale scales in place, and the rerun raised the false "input frames are not finite"), the batch and the one
    queued behind it run again, the metrics are finite."""
    from egne_amd import _entry
    from egne_amd import test as T
    monkeypatch.setattr(_entry, "SyntheticEyes", _amplified_eyes(first=4))
    seen = _count_overflows(monkeypatch)
    miou, pd, idist, loss = T.main(["--synthetic", "8", "--batchsize", "2", "--setting", "configs/baseline_edge.yaml"])
    assert seen, "no overflow was reported: the amplified frames went through on the first batch's scales"
    assert np.isfinite(miou) and 0 <= miou <= 1 and np.isfinite(loss)


@pytest.mark.parametrize("pipeline", ["0", "1"])
def test_train_py_recomputes_an_overflowed_edge_map(tmp_path, monkeypatch, pipeline):
    """train.py's loop with frames beyond the calibrated pre-scales of the frozen edge network (round-5 advisor finding: the loop never
    looked at the edge plan's overflow word -- up to 16 optimizer steps on NaN edge maps, NaN BatchNorm statistics, Adam state and
    weights): the step reads the word where it has the edge map, recomputes the map on the re-calibrated plan (and the next batch's under
    the pipeline), and every parameter and BatchNorm statistic is finite after the epoch."""
    monkeypatch.chdir(tmp_path)
    from egne_amd import _entry
    from egne_amd import train as TR
    monkeypatch.setattr(_entry, "SyntheticEyes", _amplified_eyes(first=2))
    seen = _count_overflows(monkeypatch)
    m = TR.main(["--synthetic", "8", "--batchsize", "2", "--epochs", "1", "--setting", "configs/baseline_edge.yaml", "--expname", "o" + pipeline,
                 "--pipeline", pipeline])
    assert seen, "no overflow was reported"
    bad = [k for k, v in m.state_dict().items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
    assert not bad, "non-finite parameters / statistics after an overflowed edge map: %s" % bad[:5]


def test_train_py_synthetic_loss_decreases(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from egne_amd import train as TR
    m = TR.main(["--synthetic", "4", "--batchsize", "2", "--epochs", "2", "--setting", "configs/baseline_edge.yaml",
                 "--disentangle", "1", "--expname", "t"])
    ck = torch.load(os.path.join("logs", "ritnet_v2", "t", "weights", "ritnet_v2_1.pkl"), map_location="cpu")
    assert set(ck) == {"state_dict", "epoch"} and ck["epoch"] == 1
    assert not any("dsIdentify" in k for k in ck["state_dict"])


def test_train_py_device_prep(tmp_path, monkeypatch):
    """--device_prep 1: distance maps from egne_amd.dataprep inside the training loop."""
    monkeypatch.chdir(tmp_path)
    from egne_amd import train as TR
    TR.main(["--synthetic", "4", "--batchsize", "2", "--epochs", "1", "--setting", "configs/baseline_edge.yaml",
             "--device_prep", "1", "--expname", "p"])
    assert os.path.exists(os.path.join("logs", "ritnet_v2", "p", "weights", "ritnet_v2_0.pkl"))


def test_train_py_pipeline_and_device_weights(tmp_path, monkeypatch):
    """--pipeline 1 (edge network of a batch next to the previous batch's training step, egne_amd.pipeline) and --device_prep 2
    (distance maps and boundary weights on the device): the same checkpoints appear, and with the pipeline the weights after an
    epoch are THE SAME BITS as without it (same batches, same order, same arithmetic)."""
    monkeypatch.chdir(tmp_path)
    from egne_amd import train as TR
    base = ["--synthetic", "4", "--batchsize", "2", "--epochs", "1", "--setting", "configs/baseline_edge.yaml", "--device_prep", "2"]
    TR.main(base + ["--expname", "a", "--pipeline", "0"])
    TR.main(base + ["--expname", "b", "--pipeline", "1"])
    sa = torch.load(os.path.join("logs", "ritnet_v2", "a", "weights", "ritnet_v2_0.pkl"), map_location="cpu")["state_dict"]
    sb = torch.load(os.path.join("logs", "ritnet_v2", "b", "weights", "ritnet_v2_0.pkl"), map_location="cpu")["state_dict"]
    assert set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_evaluate_py_synthetic():
    from egne_amd import evaluate as E
    pup, iri = E.main(["--synthetic", "2"])
    assert pup.shape == (2, 5) and iri.shape == (2, 5) and np.isfinite(pup).all() and np.isfinite(iri).all()


def test_evaluate_py_model_flag():
    """--model (evaluate.py:37,362-367): the DeepVOG comparator runs through the same edge -> seg -> fit path (its two-class output has
    no iris class: the ellipses are whatever the search makes of the seeds, as in the reference); anything else is refused."""
    from egne_amd import evaluate as E
    pup, iri = E.main(["--synthetic", "2", "--model", "deepvog"])
    assert pup.shape == (2, 5) and iri.shape == (2, 5)
    with pytest.raises(SystemExit):
        E.main(["--synthetic", "2", "--model", "ritnet_v9"])


def test_evaluate_on_real_video_frames():
    """Real frames of the reference's videos/example1.avi (fixture: 4 eye crops) through the evaluate
    path: logits within 1e-3 of the reference, identical masks, and -- given the same (mask, initial
    ellipse) -- a bit-identical fit.  Fitted ellipses of the full pipeline agree whenever the mask and
    the network's initial ellipse agree (the search is a discontinuous hill climb)."""
    import types
    from common import bdcn_module, esf_module, gold
    from egne_amd import evaluate as E
    from egne_amd.utils import fit_ellipses
    g = gold("evaluate_real_frames")
    dev = "cuda:0"
    bd, net = bdcn_module().to(dev), esf_module("baseline_edge").to(dev).eval()
    H, W = 240, 320
    xs = torch.stack([E.preprocess_frame(e, (240, 320), True)[0] for e in g["eyes"]]).to(dev)
    edge, seg, pup, iri = E.evaluate_ellseg_on_image(xs, net, bd)
    for k in range(4):
        # identical masks, except at pixels whose top-2 logit gap in the reference is below 2e-3 (counted in the fixture)
        r1 = np.unpackbits(g["masks"][2 * k]).reshape(H, W).astype(bool)
        r2 = np.unpackbits(g["masks"][2 * k + 1]).reshape(H, W).astype(bool)
        ndiff = int(np.count_nonzero(np.asarray(seg[k]) != (r1.astype(np.int64) + 2 * r2.astype(np.int64))))
        print("real video frame %d: %d mask pixels differ (near-tie pixels in the fixture: %d)" % (k, ndiff, int(g["gap_lt_2e3"][k])))
        assert ndiff <= int(g["gap_lt_2e3"][k]), "mask %d differs in %d pixels (near-tie budget %d)" % (k, ndiff, g["gap_lt_2e3"][k])
    # fit stage alone: reference masks + reference initial ellipses -> bit-identical result
    masks = np.stack([np.unpackbits(g["masks"][2 * k]).reshape(H, W).astype(np.int64)
                      + 2 * np.unpackbits(g["masks"][2 * k + 1]).reshape(H, W).astype(np.int64) for k in range(4)])
    init = g["inits"].reshape(8, 5)
    fit = fit_ellipses(torch.from_numpy(masks).to(dev), [0, 0, 1, 1, 2, 2, 3, 3], [1, 2] * 4, init)
    np.testing.assert_array_equal(fit, g["fits"].reshape(8, 5))
    # end to end: initial ellipses come from the network (1e-3 parity), so the climb may take another path;
    # centres are never moved by the search and must match to 1e-3 * image size
    np.testing.assert_allclose(iri[:, :2], g["fits"][:, 0, :2], atol=0.35)
    np.testing.assert_allclose(pup[:, :2], g["fits"][:, 1, :2], atol=0.35)
    same = sum(np.allclose(iri[k], g["fits"][k, 0], atol=1e-2) for k in range(4)) + \
        sum(np.allclose(pup[k], g["fits"][k, 1], atol=1e-2) for k in range(4))
    print("end-to-end fitted ellipses equal to the reference's within 1e-2: %d / 8" % same)
    assert same >= 6


def test_evaluate_video_end_to_end(tmp_path):
    """evaluate.py on a Motion-JPEG video built from the real example frames (two eyes per 640x240 frame): overlay and edge
    videos are written and can be read back, the ellipse dictionary holds (iris, pupil) per frame and per eye."""
    from common import bdcn_module, esf_module, gold
    from egne_amd import evaluate as E
    g = gold("evaluate_real_frames")
    vid = tmp_path / "clip.avi"
    w = E.MJPEGWriter(str(vid), 30, (640, 240))
    for k in range(2):
        fr = np.concatenate([g["eyes"][2 * k], g["eyes"][2 * k + 1]], axis=1)
        for _ in range(2):
            w.write(np.stack([fr] * 3, axis=2))
    w.release()
    dev = torch.device("cuda:0")
    bd, net = bdcn_module().to(dev), esf_module("baseline_edge").to(dev).eval()
    args = E.parse_args(["--path2data", str(tmp_path)])
    res = E.evaluate_ellseg_per_video(str(vid), args, net, bd, dev)
    assert set(k for k in res if isinstance(k, int)) == {0, 1, 2, 3} and all((j, i) in res for j in range(4) for i in range(2))
    iri, pup = res[(0, 0)]
    assert iri.shape == (5,) and pup.shape == (5,) and np.isfinite(iri).all() and np.isfinite(pup).all()
    np.testing.assert_allclose(iri[:2], g["fits"][0, 0, :2], atol=2.5)       # JPEG re-encoding moves the frame a little
    for name in ("clip_result_baseline.avi", "clip_edge_baseline.avi"):
        frames = list(E.mjpeg_frames(str(tmp_path / name)))
        assert len(frames) == 4 and frames[0].shape == (240, 640)
    assert os.path.exists(tmp_path / "clip_pred2_baseline.npy")
    # --low_latency 1: a frame's two eyes per call through the hipGraph replay; the same ellipses up to what the small-batch kernels'
    # summation order moves (initial ellipses within 1e-3, the search may take another step)
    args2 = E.parse_args(["--path2data", str(tmp_path), "--low_latency", "1", "--method", "live"])
    res2 = E.evaluate_ellseg_per_video(str(vid), args2, net, bd, dev)
    assert set(res2) == set(res)
    close = sum(np.allclose(res2[k][0], res[k][0], atol=1e-2) and np.allclose(res2[k][1], res[k][1], atol=1e-2) for k in res)
    assert close >= len(res) - 2, "%d of %d results differ between the batched and the per-frame path" % (len(res) - close, len(res))
    for k in res:
        np.testing.assert_allclose(res2[k][0][:2], res[k][0][:2], atol=0.35)


_DP_TRAIN_WORKER = r"""
import os, sys, json, hashlib
sys.path.insert(0, %(root)r)
import torch
import egne_amd
from egne_amd import parallel, train, _entry
out = sys.argv[1]
model = train.main(["--synthetic", "16", "--batchsize", "4", "--epochs", "1", "--setting", "configs/baseline_edge.yaml", "--curObj", "x",
                    "--expname", "dp2_rank" + os.environ["RANK"], "--pipeline", sys.argv[2]] + sys.argv[3:])
rank, world = parallel.rank(), parallel.world_size()
h = hashlib.sha256()
for n, p in sorted(model.state_dict().items()):
    if "num_batches_tracked" in n or "running_" in n or "dsIdentify" in n:
        continue            # BatchNorm statistics are per rank between validations (DataParallel keeps replica 0's), the identity head is outside the optimiser
    h.update(p.detach().float().cpu().numpy().tobytes())
ts, _ = parallel.samplers(_entry.SyntheticEyes(16, seed=1234), _entry.SyntheticEyes(4, seed=99), rank, world)
ts.set_epoch(0)
json.dump({"rank": rank, "world": world, "sha": h.hexdigest(), "shard": list(ts),
           "w0": float(model.enc.head.conv1.weight.detach().double().sum())}, open(os.path.join(out, "rank%%d.json" %% rank), "w"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("pipeline,extra", [("0", []), ("1", ["--prec", "16"])])
def test_train_py_two_ranks_one_epoch(tmp_path, pipeline, extra):
    """train.py --synthetic for one epoch on TWO ranks (torch.distributed, gloo backend with both ranks on this GPU: RCCL refuses two
    ranks on one device, and the pool's boxes have one): disjoint shards of one per-epoch permutation, one gradient all-reduce per
    step, identical weights on both ranks afterwards, different from the initial weights (train.py:205,262-287 with nn.DataParallel
    replaced by one process per GPU).  Second case: the default pipeline and --prec 16 (bf16 activation storage)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_train_worker.py"
    script.write_text(_DP_TRAIN_WORKER % dict(root=root))
    port = 29700 + (os.getpid() % 200) + (0 if pipeline == "0" else 1)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_RANK="0", EGNE_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, str(script), str(tmp_path), pipeline] + extra, env=dict(env, RANK=str(r)), cwd=str(tmp_path),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [x["world"] for x in res] == [2, 2]
    assert res[0]["sha"] == res[1]["sha"], "the ranks ended the epoch with different weights"
    assert not set(res[0]["shard"]) & set(res[1]["shard"]) and len(res[0]["shard"]) == len(res[1]["shard"]) == 8
    from common import esf_module
    w_init = float(esf_module("baseline_edge").enc.head.conv1.weight.detach().double().sum())
    assert abs(res[0]["w0"] - w_init) > 1e-6, "the weights did not move"


_RCCL_WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch
import torch.distributed as dist
import egne_amd
from egne_amd import parallel, synth
from common import batch_args, bdcn_module, esf_module
import types
from egne_amd.utils import calc_edge

mode = sys.argv[1]                    # "none" | "sync" | "async" | "overlap" (two buckets, the first issued inside the backward pass)
dev = "cuda:0"
torch.cuda.set_device(0)
rank, world = parallel.init()
out = {"active": parallel.active(), "backend": dist.get_backend() if dist.is_initialized() else None, "world": world}
b = synth.make_batch(2, seed=1234)
edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(dev), bdcn_module().to(dev), dev)
m = esf_module("baseline_edge").to(dev).train()
parallel.broadcast_state(m)
opt = torch.optim.Adam([p for n, p in m.named_parameters() if "dsIdentify" not in n], lr=5e-4)
args = [a.to(dev) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
if mode == "overlap":
    parallel.overlap_grads(m)
    issued = []
    orig = m.grad_comm.tail_ready
    def spy():
        orig()
        issued.append(bool(m.grad_comm.pending))
    m.grad_comm.tail_ready = spy
ms = []
for step in range(2):                 # train.py:284-287 with the gradient exchange of egne_amd.parallel in place of nn.DataParallel
    opt.zero_grad()
    loss = m(*args)[3]
    loss.backward()
    if mode != "none":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = parallel.allreduce_grads(m, async_op=(mode == "async"))
        if mode == "async":
            work, finish = r
            work.wait()
            finish()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    opt.step()
torch.cuda.synchronize()
out["loss_mean"] = float(parallel.mean_loss(loss.detach()).item())
out["sum"] = parallel.sum_over_ranks([1.5, 2.5], device=dev)
h = hashlib.sha256()
for k, v in sorted(m.state_dict().items()):
    h.update(v.detach().cpu().contiguous().numpy().tobytes())
out.update(sha=h.hexdigest(), allreduce_ms=ms, loss=float(loss.item()))
if mode == "overlap":
    out["issued_inside_backward"] = issued
json.dump(out, open(sys.argv[2], "w"))
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
'''


def test_rccl_one_rank_training_matches_the_plain_run(tmp_path):
    """EGNE_FORCE_DIST=1: an nccl (= RCCL) process group of ONE rank on this box's GPU.  Two Adam steps whose gradient arena goes
    through parallel.allreduce_grads on the HIP stream (blocking and async_op forms), after the parameter broadcast, must leave exactly
    the weights of the run without a process group (SUM over one rank, / 1); the all-reduce takes measurable time; the scalar
    all-reduces (mean_loss, the device branch of sum_over_ranks) return their inputs.  train.py:205,285 with nn.DataParallel replaced
    by one process per GPU -- the scaling run itself needs an 8-GPU node, the code path does not."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER % dict(root=root))
    res = {}
    for i, mode in enumerate(("none", "sync", "async", "overlap")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + (os.getpid() % 90) + i), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.pop("EGNE_DIST_BACKEND", None)
        if mode != "none":
            env["EGNE_FORCE_DIST"] = "1"
        else:
            env.pop("EGNE_FORCE_DIST", None)
        outp = tmp_path / ("%s.json" % mode)
        p = subprocess.run([sys.executable, str(script), mode, str(outp)], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, "%s run failed:\n%s" % (mode, p.stdout.decode()[-3000:])
        res[mode] = json.load(open(outp))
    assert not res["none"]["active"] and res["none"]["backend"] is None
    # the two-bucket form: the tail of the arena went out from inside both backward passes (parallel.GradOverlap, engine.Plan.tail_hook)
    assert res["overlap"]["issued_inside_backward"] == [True, True], res["overlap"]
    for mode in ("sync", "async", "overlap"):
        r = res[mode]
        assert r["active"] and r["backend"] == "nccl" and r["world"] == 1
        assert r["sha"] == res["none"]["sha"], "%s: weights differ from the run without a process group" % mode
        assert r["loss"] == res["none"]["loss"] and abs(r["loss_mean"] - r["loss"]) < 1e-6 * abs(r["loss"])
        assert r["sum"] == [1.5, 2.5]
        assert len(r["allreduce_ms"]) == 2 and all(t > 0 for t in r["allreduce_ms"])
    print("RCCL, one rank: all-reduce of the gradient arena %.3f ms (blocking), %.3f ms (async_op)"
          % (res["sync"]["allreduce_ms"][-1], res["async"]["allreduce_ms"][-1]))

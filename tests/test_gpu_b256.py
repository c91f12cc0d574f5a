"""-m gpu: the B=256 plans (BASELINE.json configs[2]: 256 frames per GPU) on 256 DISTINCT frames (round-4 verdict, "What's weak" 2).

tests/test_gpu_batch.py and tests/test_gpu_bf16.py feed the B=256 plans the golden two-frame batch tiled x128, so a frame-indexing
or 32-bit-offset bug that only bites above 64 frames and maps frame j onto an identical frame k is invisible there; the distinct-
frame checks of tests/test_gpu_distinct.py stop at B=64 (inference) and B=32 (training: the oracle's autograd graph needs ~1 GB of
host memory per frame).  Here every one of the 256 frames differs, and three size-independent properties stand in for the oracle:

* inference (per-sample InstanceNorm, folded BatchNorm): frame i of a B=256 call equals frame i of the B=64 call that holds it --
  the B=64 plan is the one checked against the live oracle on distinct frames;
* training (batch-statistic BatchNorm ties the frames together, so no sub-batch reproduces it): the step is EQUIVARIANT under a
  permutation of the batch -- logits follow their frames, the loss and every parameter gradient stay (sums in another order).  A
  kernel that reads or writes the wrong frame's data breaks this unless the mistake itself commutes with an arbitrary permutation;
* bf16 against fp32 storage on the same 256 frames: per-frame logits and per-tensor gradient cosines.
* the frozen edge network (round-5 verdict, "What's weak" 1): bench.py and train.py call ``calc_edge`` on the whole 256-frame shard
  (utils.py:645-656 has no chunking), so its B=256 plan -- other rounds of the deep trunk kernel, other ``.tail`` splits, stage-1
  tensors of 5 GB -- runs here in ONE call, in both product modes, against four B=64 calls (that plan is the oracle-checked one)
  and against the live oracle on the first and last frames, all 11 outputs.
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
NS = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
B = 256


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _dev(args):
    return [a.to(DEV) if torch.is_tensor(a) else a for a in args]


def _take(b, idx):
    return {k: (v[idx] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in b.items()}


@pytest.fixture(scope="module")
def frames256():
    from common import bdcn_module
    from egne_amd import synth
    from egne_amd.utils import calc_edge
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    b = synth.make_batch(B, seed=2025)
    assert len({bytes(f.numpy().tobytes()[:4096]) for f in b["img"]}) == B, "the frames must all differ"
    net = bdcn_module().to(DEV)
    edge = torch.cat([calc_edge(NS, b["img"][i:i + 64].to(DEV), net, DEV) for i in range(0, B, 64)])
    del net
    _free()
    return b, edge


def test_inference_b256_distinct_frames_vs_four_b64_calls(frames256):
    """Eval mode: InstanceNorm is per sample and BatchNorm is folded, so every frame of the B=256 call must reproduce the frame of
    the B=64 call that holds it to round-off (the kernels choose tiles by B, so not bit for bit), masks included."""
    from common import batch_args, esf_module
    b, edge = frames256
    m = esf_module("baseline_edge", seed=11).to(DEV).eval()
    with torch.no_grad():
        op, elPred, latent, _, elOut = [t.clone() for t in m(*_dev(batch_args(b, edge)))]
        mask = m.predictions().clone()
        worst = 0.0
        for i in range(0, B, 64):
            idx = torch.arange(i, i + 64)
            o4, p4, l4, _, e4 = m(*_dev(batch_args(_take(b, idx), edge[i:i + 64])))
            per = (op[i:i + 64] - o4).abs().flatten(1).max(1)[0] / o4.abs().flatten(1).max(1)[0]
            worst = max(worst, per.max().item())
            assert per.max().item() < 1e-4, "frame %d of the B=256 call differs from its B=64 call by %.2e of its largest logit" % (i + int(per.argmax()), per.max())
            assert (elOut[i:i + 64] - e4).abs().max().item() < 1e-4 and (latent[i:i + 64] - l4).abs().max().item() < 1e-4
            nd = int((mask[i:i + 64] != m.predictions()).sum())
            assert nd <= 64, "%d mask pixels of frames %d..%d differ between the two calls (near-ties only: <= 1 per frame)" % (nd, i, i + 63)
    del m
    _free()
    print("inference, 256 distinct frames: worst per-frame logit difference to the B=64 calls %.2e of the frame's largest logit" % worst)


def _free():
    """Plans hold closures that refer back to their model (reference cycles): collect them before the next B=256 plan is built --
    the fp32-storage one alone takes 222 of the 288 GB."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _train_step(b, edge, storage):
    from common import batch_args, esf_module
    _free()
    m = esf_module("baseline_edge", seed=11).to(DEV).to(storage).train()
    op, _, latent, loss, elOut = m(*_dev(batch_args(b, edge)))
    loss.sum().backward()
    torch.cuda.synchronize()
    pl = m._last_plan
    assert pl.bf16 == (storage == torch.bfloat16) and pl.in_img.shape[0] == B
    out = dict(loss=loss.item(), op=op.detach().cpu(), elOut=elOut.detach().cpu(),
               grads={n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None})
    del m, pl, op, latent, loss, elOut
    _free()
    return out


@pytest.fixture(scope="module")
def steps256(frames256):
    b, edge = frames256
    perm = torch.from_numpy(np.random.RandomState(17).permutation(B))
    assert (perm != torch.arange(B)).sum() > B - 8
    res = {"perm": perm}
    for st in (torch.float32, torch.bfloat16):
        res[st] = _train_step(b, edge, st)
        res[st, "perm"] = _train_step(_take(b, perm), edge[perm.to(edge.device)], st)
    return res


def _whole(ga, gb):
    names = [n for n in ga if n in gb]
    a, c = torch.cat([ga[n].reshape(-1) for n in names]), torch.cat([gb[n].reshape(-1) for n in names])
    return float((a - c).norm() / c.norm()), float(torch.dot(a, c) / (a.norm() * c.norm()))


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_training_b256_distinct_frames_permutation_equivariance(steps256, storage):
    """One training step on 256 distinct frames and on the same frames in another order: per-frame logits follow the frames, loss and
    gradients agree.  fp32 storage differs by summation order only; bf16 storage re-rounds tensors whose BatchNorm statistics moved
    in the last bits, so single elements move by a bf16 ulp and the gradients by their storage noise."""
    r, rp, perm = steps256[storage], steps256[storage, "perm"], steps256["perm"]
    bf = storage == torch.bfloat16
    per = (rp["op"] - r["op"][perm]).abs().flatten(1).max(1)[0] / r["op"][perm].abs().flatten(1).max(1)[0]
    whole, cos = _whole(rp["grads"], r["grads"])
    print("%s storage, B=256 distinct frames, permuted batch: loss %.6f vs %.6f, logits worst frame %.2e (median %.2e) of its largest logit, "
          "gradient relative L2 %.2e, cosine %.6f" % ("bf16" if bf else "fp32", rp["loss"], r["loss"], per.max(), per.median(), whole, cos))
    assert abs(rp["loss"] - r["loss"]) < (2e-3 if bf else 1e-5) * abs(r["loss"])
    assert per.max().item() < (6e-2 if bf else 1e-4), "frame %d does not follow its permutation" % int(per.argmax())
    assert (rp["elOut"] - r["elOut"][perm]).abs().max().item() < (5e-2 if bf else 1e-4)
    assert whole < (1.5e-1 if bf else 1e-3) and cos > (0.985 if bf else 0.999999)


def test_training_b256_distinct_frames_bf16_vs_fp32_storage(steps256):
    """bf16 against fp32 storage on the same 256 distinct frames: loss, per-frame logits, per-tensor gradient cosines."""
    rf, rh = steps256[torch.float32], steps256[torch.bfloat16]
    per = (rh["op"] - rf["op"]).abs().flatten(1).max(1)[0] / rf["op"].abs().flatten(1).max(1)[0]
    names = [n for n in rf["grads"] if n in rh["grads"]]
    scale = max(rf["grads"][n].norm().item() for n in names)
    cosines = {n: float(torch.dot(rh["grads"][n].reshape(-1), rf["grads"][n].reshape(-1)) / (rh["grads"][n].norm() * rf["grads"][n].norm()))
               for n in names if rf["grads"][n].norm().item() > 1e-5 * scale}
    cs = np.array(sorted(cosines.values()))
    whole, cos = _whole(rh["grads"], rf["grads"])
    low = sorted(cosines, key=cosines.get)[:4]
    print("bf16 vs fp32 storage, B=256 distinct frames: loss %.5f vs %.5f, logits per frame median %.2e worst %.2e of the largest logit; "
          "gradient cosines per tensor: min %.4f (%s), 10th percentile %.4f, median %.4f; whole vector relative L2 %.2e, cosine %.4f"
          % (rh["loss"], rf["loss"], per.median(), per.max(), cs[0], ", ".join(low), cs[len(cs) // 10], np.median(cs), whole, cos))
    assert abs(rh["loss"] - rf["loss"]) < 1e-2 * abs(rf["loss"])
    assert per.median().item() < 8e-2 and per.max().item() < 2.5e-1
    assert cs[len(cs) // 10] > 0.93 and np.median(cs) > 0.975 and whole < 0.15 and cos > 0.99        # measured 0.963 / 0.986 / 0.108 / 0.9949


def _kinds(pl):
    return {k for k, _ in pl.meta}


@pytest.mark.parametrize("products", [0, 1], ids=["split3", "plain_f16"])
def test_bdcn_b256_distinct_frames_one_call(products):
    """``BDCN.forward_fuse`` / ``forward`` on 256 DISTINCT frames in ONE call (what utils.calc_edge does with a configs[2..4] shard),
    with the 22-bit split products and with plain f16 operands (``f16_products = 1``: the plan next to a bf16-storage training step).
    (1) every frame of the fused map against the B=64 call that holds it (1e-5: the plans choose tiles / frame tails by B, products
    are summed in another order); (2) the live oracle on frames 0, 1, 254, 255: the fused map and the 10 side outputs at 1e-3
    (split) / 2^-9 (plain f16: the bound of test_bdcn_plain_f16_operands_next_to_a_bf16_training_plan); (3) the plan's kernel kinds."""
    from common import bdcn_module
    from egne_amd import synth
    from oracle import bdcn as obdcn
    _free()
    b = synth.make_batch(B, seed=2025)
    x = torch.cat((b["img"],) * 3, 1)
    assert len({bytes(f.numpy().tobytes()[:4096]) for f in b["img"]}) == B
    bd = bdcn_module().to(DEV)
    bd.f16_products = products
    xd = x.to(DEV)
    got = bd.forward_fuse(xd)
    pl = bd._last_plan
    assert pl.x_in.shape[0] == B and pl.f16_products == products
    assert not bd.overflowed()
    kinds, names = _kinds(pl), [n for _, _, n in pl.calls]
    assert {"conv_f16x3:big", "conv_f16x3:msdil", "conv_f16x3:first"} <= kinds and kinds & {"conv_f16x3:rs", "conv_f16x3:rw"}, kinds
    tails = [n for n in names if n.endswith(".tail")]
    worst = 0.0
    for i in range(0, B, 64):
        g4 = bd.forward_fuse(xd[i:i + 64])
        per = (got[i:i + 64] - g4).abs().flatten(1).max(1)[0]
        worst = max(worst, per.max().item())
        assert per.max().item() < 1e-5, "frame %d of the B=256 call differs from its B=64 call by %.2e" % (i + int(per.argmax()), per.max())
    assert bd._last_plan is not pl and bd._last_plan.x_in.shape[0] == 64
    sub = [0, 1, B - 2, B - 1]
    outs = [o[sub].cpu() for o in bd(xd)]                      # all 11 maps of the B=256 plan (only_fuse=False: a third plan)
    assert bd._last_plan.x_in.shape[0] == B
    with torch.no_grad():
        ref = obdcn.bdcn_forward(bdcn_module().state_dict(), x[sub])
    # plain f16 operands: the FUSED map -- the one output utils.calc_edge takes (utils.py:648) -- at bf16's half ulp below 1; the ten side
    # outputs (deep-supervision maps that nothing consumes in this mode: single-stage sums without the fuse layer's averaging) at 2^-8
    tol, tol_side = (2.0 ** -9, 2.0 ** -8) if products else (1e-3, 1e-3)
    errs = [float((o - r).abs().max()) for o, r in zip(outs, ref)]
    print("BDCN B=256 distinct frames, %s: worst frame vs its B=64 call %.2e; vs the oracle on frames %s: fused %.2e, side outputs max %.2e; "
          "tail launches %s" % ("plain f16 operands" if products else "split products", worst, sub, errs[10], max(errs[:10]), tails or "none"))
    assert errs[10] < tol and max(errs[:10]) < tol_side, errs
    assert float((outs[10] - got[sub].cpu()).abs().max()) < 1e-6            # forward()[-1] and forward_fuse agree
    del bd, got, xd
    _free()

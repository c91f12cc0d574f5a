"""-m gpu: the benchmarked plans (B=64 / B=40 inference, a B=32 training step) on batches of DISTINCT frames against the live CPU
oracle (round-3 verdict, "What's weak" 1).

tests/test_gpu_batch.py feeds these plans the golden B=2 batch tiled x32: every even frame is the same frame, so a kernel that
mis-indexes frames with period 2 (256-row tiles that span frame boundaries on 30x40 maps, the frame-tail split, per-(n, c)
InstanceNorm tables) would still reproduce the fixture.  Here every frame differs (``synth.make_batch(64, seed=...)``, no tiling)
and each is compared with the oracle's result for THAT frame: test.py:75-92 runs the path exactly like this.

The oracle runs once per module (64 frames of the edge network + ESF-Net on the host cores, in chunks of 8 frames to bound memory;
eval-mode results are per-frame independent -- InstanceNorm per sample, BatchNorm folded -- and the loss head is evaluated over
the whole batch afterwards).
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3          # BASELINE.json north_star: logits / edge maps within 1e-3
DEV = "cuda:0"
NS = types.SimpleNamespace(prec=torch.float32, edge_thres=0)


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _host_mem_gb():
    """Memory this process may use: the smaller of MemAvailable and the cgroup limit."""
    import psutil
    avail = psutil.virtual_memory().available
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(p).read().strip()
            if v.isdigit():
                avail = min(avail, int(v))
        except OSError:
            pass
    return avail / 2 ** 30


def _kinds(pl):
    return {k for k, _ in pl.meta}


def _names(pl):
    return [n for _, _, n in pl.calls]


def _dev(args):
    return [a.to(DEV) if torch.is_tensor(a) else a for a in args]


def _sub(b, n):
    return {k: (v[:n] if torch.is_tensor(v) else v) for k, v in b.items()}


@pytest.fixture(scope="module")
def frames64():
    """64 distinct synthetic frames (one in eight without a mask, as SURVEY.md section 8d) and the oracle's edge maps, logits,
    ellipse heads and latents for each of them."""
    from common import batch_args, bdcn_module, esf_module, setting
    from egne_amd import synth
    from oracle import bdcn as obdcn, esfnet as oesf
    b = synth.make_batch(64, seed=20264)
    bd, m = bdcn_module(), esf_module("baseline_edge", seed=11)
    bsd = {k: v for k, v in bd.state_dict().items()}
    msd = {k: v.clone() for k, v in m.state_dict().items()}
    st = setting("baseline_edge")
    edge, op, elOut, latent = [], [], [], []
    with torch.no_grad():
        for i in range(0, 64, 8):
            bi = {k: (v[i:i + 8] if torch.is_tensor(v) else v) for k, v in b.items()}
            e = obdcn.calc_edge(bsd, bi["img"])
            r = oesf.esf_forward(msd, st, *batch_args(bi, e))
            edge.append(e), op.append(r[0]), latent.append(r[2]), elOut.append(r[4])
    return dict(batch=b, bdcn=bd, esf_sd=msd, edge=torch.cat(edge), op=torch.cat(op), elOut=torch.cat(elOut), latent=torch.cat(latent))


def _oracle_loss(f, n):
    from oracle import losses
    b = f["batch"]
    with torch.no_grad():
        total, _, _ = losses.all_loss(f["op"][:n], f["elOut"][:n], b["label"][:n], b["pupil_center"][:n], b["elNorm"][:n],
                                      b["spatWts"][:n], b["distMap"][:n], b["cond"][:n], b["alpha"])
    return float(torch.as_tensor(total).reshape(-1)[0])


@pytest.mark.parametrize("B", [64, 40])
def test_bdcn_distinct_frames_vs_oracle(frames64, B):
    """The B=64 / B=40 edge-network plans (deep trunk kernel, frame tail, one-launch dilated groups, role-split 3x3): every frame
    against the oracle's edge map of that frame."""
    f = frames64
    bd = f["bdcn"].to(DEV)
    x = torch.cat((f["batch"]["img"][:B],) * 3, 1).to(DEV)
    got = bd.forward_fuse(x).cpu()
    per_frame = (got - f["edge"][:B]).abs().flatten(1).max(1)[0]
    worst = int(per_frame.argmax())
    print("BDCN B=%d distinct frames: max err %.2e (frame %d), median over frames %.2e" % (B, per_frame.max(), worst, per_frame.median()))
    assert per_frame.max().item() < TOL, "frame %d: fused edge map off by %.2e" % (worst, per_frame.max())
    pl = next(p for k, p in bd._plans.items() if k[0] == B and k[4])
    kinds = _kinds(pl)
    assert {"conv_f16x3:big", "conv_f16x3:msdil", "conv_f16x3:first"} <= kinds and kinds & {"conv_f16x3:rs", "conv_f16x3:rw"}, kinds
    if B == 40:
        assert any(n.endswith(".tail") for n in _names(pl)), "no .tail launch in the B=40 plan"
    # frames must not leak into each other: two frames swapped at the input swap at the output.  Not bit for bit -- the last frames
    # of the batch take the frame-tail kernel, whose products are summed in another order than the deep trunk kernel's -- but to
    # rounding: a frame's result may not depend on its neighbours
    perm = torch.arange(B)
    perm[[1, B - 2]] = perm[[B - 2, 1]]
    got2 = bd.forward_fuse(x[perm.to(DEV)]).cpu()
    moved = (got2[perm] - got).abs().max().item()
    print("   frames 1 and %d swapped: results move by %.2e" % (B - 2, moved))
    assert moved < 1e-5, "edge maps depend on the position of a frame in the batch: %.2e" % moved
    keep = [i for i in range(B) if i not in (1, B - 2)]
    assert torch.equal(got2[keep], got[keep]), "frames that did not move changed"


@pytest.mark.parametrize("B", [64, 40])
def test_esf_eval_distinct_frames_vs_oracle(frames64, B):
    """The B=64 / B=40 ESF-Net inference plans fed with the edge maps of the B-frame edge plan, as test.py:75-92: logits, ellipse
    head, latent and loss against the oracle per frame; argmax masks identical except where the oracle's two largest logits are
    within 2e-3 (counted)."""
    from common import batch_args, esf_module
    from egne_amd.utils import calc_edge
    f = frames64
    b = _sub(f["batch"], B)
    bd = f["bdcn"].to(DEV)
    edge = calc_edge(NS, b["img"].to(DEV), bd, DEV)
    m = esf_module("baseline_edge", seed=11).to(DEV).eval()
    with torch.no_grad():
        op, elPred, latent, loss, elOut = m(*_dev(batch_args(b, edge)))
    ref = f["op"][:B]
    per_frame = (op.cpu() - ref).abs().flatten(1).max(1)[0]
    worst = int(per_frame.argmax())
    assert per_frame.max().item() < TOL, "frame %d: logits off by %.2e" % (worst, per_frame.max())
    np.testing.assert_allclose(elOut.cpu().numpy(), f["elOut"][:B].numpy(), atol=TOL)
    np.testing.assert_allclose(latent.cpu().numpy(), f["latent"][:B].numpy(), atol=TOL)
    np.testing.assert_allclose(float(loss.cpu().reshape(-1)[0]), _oracle_loss(f, B), rtol=1e-3)
    # mask identity, pixel by pixel: a differing pixel must be a near tie of the oracle's logits
    mask = m.predictions().cpu()
    top2 = ref.topk(2, dim=1)[0]
    near = (top2[:, 0] - top2[:, 1]) < 2e-3
    diff = mask != ref.max(1)[1]
    assert not (diff & ~near).any(), "%d mask pixels differ away from logit ties" % int((diff & ~near).sum())
    kinds = _kinds(m._last_plan)
    assert {"conv_f16x3:fused1x1", "conv_f16x3:fused3x3c4", "conv_f16x3:tdpool1x1"} <= kinds and kinds & {"conv_f16x3:rs", "conv_f16x3:rw"}, kinds
    print("ESF-Net B=%d distinct frames: logits err %.2e (frame %d), %d of %d mask pixels differ (all among %d near ties)"
          % (B, per_frame.max(), worst, int(diff.sum()), diff.numel(), int(near.sum())))
    try:
        import json
        import os
        from common import ROOT
        with open(os.path.join(ROOT, "gpurun_out", "mask_mismatch.jsonl"), "a") as fh:
            fh.write(json.dumps({"case": "distinct frames, B=%d eval plan vs live oracle" % B, "pixels": int(diff.numel()),
                                 "mismatch_pixels": int(diff.sum()), "near_tie_pixels_lt_2e-3": int(near.sum())}) + "\n")
    except OSError:
        pass


def test_a_batch_beyond_the_calibrated_f16_range_is_never_silent(frames64):
    """Round-4 verdict, "What's weak" 3.  The split-f16 kernels of an inference plan scale their fp32 operands by powers of two
    calibrated on the FIRST batch (32x of head-room, engine.Plan); a later batch beyond that turns f16 operands into inf.  Every
    split-f16 epilogue now tests what it stores and sets the plan's sticky overflow word (egne_conv_desc.ovf_flag):

    * both B=64 plans are calibrated on the 64 frames ATTENUATED 200x, then fed the frames themselves: raw activations 200x the
      calibration maxima, beyond the 32-64x of head-room (frames64 holds the oracle's results for exactly these frames);
    * ``overflowed()`` (what test.py / evaluate.py / train.py's validation ask where they synchronise) reports it, the plan
      re-calibrates on the next call, and THAT call's edge maps / logits are within 1e-3 of the oracle for all 64 frames, masks
      identical away from logit ties;
    * a caller that never asks gets an exception from the next run instead of going on with invalid results."""
    from common import batch_args, esf_module
    from egne_amd.utils import calc_edge
    f = frames64
    b = dict(f["batch"])
    dim = dict(b, img=b["img"] * 0.005)
    bd = f["bdcn"].to(DEV)
    m = esf_module("baseline_edge", seed=11).to(DEV).eval()
    m.load_state_dict(f["esf_sd"])

    def run(batch, edge=None):
        with torch.no_grad():
            e = calc_edge(NS, batch["img"].to(DEV), bd, DEV) if edge is None else edge
            return e, m(*_dev(batch_args(batch, e)))[0]

    for pl in bd._plans.values():           # (the module-scoped edge network carries plans calibrated by the tests above)
        pl.calibrated = False
    e_dim, _ = run(dim)                     # calibrates both plans on the attenuated frames
    assert not bd.overflowed() and not m.overflowed()
    # the frames themselves on the stale scales: the edge network reports, and its next call is right
    e1 = calc_edge(NS, b["img"].to(DEV), bd, DEV)
    assert bd.overflowed(), "200x the calibration maxima went through the edge network unnoticed"
    e2 = calc_edge(NS, b["img"].to(DEV), bd, DEV)
    assert not bd.overflowed()
    per_e = (e2.cpu() - f["edge"]).abs().flatten(1).max(1)[0]
    assert per_e.max().item() < TOL, "frame %d: edge map off by %.2e after re-calibration" % (int(per_e.argmax()), per_e.max())
    # ESF-Net on the stale scales: either it reports, or its results are right
    _, op_a = run(b, e2)
    if not m.overflowed():
        per = (op_a.cpu() - f["op"]).abs().flatten(1).max(1)[0]
        assert per.max().item() < TOL, "no overflow reported at 200x, yet frame %d is off by %.2e" % (int(per.argmax()), per.max())
    # (ESF-Net's raw tensors do not follow the input's scale: the head's bias and BatchNorm put a floor under the calibration
    #  maxima -- tests/test_gpu_batch.py::test_esf_stale_scales_after_a_batchnorm_update_are_reported overflows its plan)
    _, op2 = run(b, e2)
    assert not m.overflowed()
    per = (op2.cpu() - f["op"]).abs().flatten(1).max(1)[0]
    assert per.max().item() < TOL, "frame %d: logits off by %.2e after re-calibration" % (int(per.argmax()), per.max())
    top2 = f["op"].topk(2, dim=1)[0]
    near = (top2[:, 0] - top2[:, 1]) < 2e-3
    assert not ((m.predictions().cpu() != f["op"].max(1)[1]) & ~near).any()
    bad1 = int((~torch.isfinite(e1)).sum()) + int(((e1.cpu() - f["edge"]).abs() > TOL).sum())
    print("frames at 200x the calibration batch, stale scales: %d of %d edge-map values non-finite or off by more than 1e-3 (reported); "
          "after re-calibration worst frame %.2e (edge), %.2e (logits)" % (bad1, e1.numel(), per_e.max(), per.max()))
    # a caller that never asks: calibrate on the attenuated frames again, overflow, and go on without looking
    bd._last_plan.calibrated = False
    calc_edge(NS, dim["img"].to(DEV), bd, DEV)
    with pytest.raises(RuntimeError, match="overflowed the f16 range"):
        for _ in range(2 * bd._last_plan.OVF_MIRROR_EVERY + 2):       # (the host mirror of the word is refreshed every 16th run)
            calc_edge(NS, b["img"].to(DEV), bd, DEV)
            torch.cuda.synchronize()
    e4 = calc_edge(NS, b["img"].to(DEV), bd, DEV)       # (the refused call marked the plan: this one re-calibrates and is right)
    assert (e4.cpu() - f["edge"]).abs().max().item() < TOL and not bd.overflowed()


def _train_batch():
    """B for the distinct-frame training step: 32 (bench.py's shape family) when the host can hold the oracle's autograd graph
    (~1 GB per frame), else 16."""
    mem = _host_mem_gb()
    if mem >= 52:
        return 32
    if mem >= 28:
        return 16
    pytest.skip("the oracle's autograd graph needs ~1 GB of host memory per frame; %.0f GB available" % mem)


@pytest.fixture(scope="module")
def train_ref():
    """One training step of the oracle (autograd, batch-statistic BatchNorm over ALL frames) on distinct frames."""
    from common import batch_args, bdcn_module, esf_module, setting
    from egne_amd import engine, synth
    from egne_amd.utils import calc_edge
    from oracle import esfnet as oesf
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    B = _train_batch()
    b = synth.make_batch(B, seed=777)
    old, engine.F16X3_ENABLED = engine.F16X3_ENABLED, False        # exact-fp32 edge maps, as the gradient fixtures' tests
    try:
        edge = calc_edge(NS, b["img"].to(DEV), bdcn_module().to(DEV), DEV)
    finally:
        engine.F16X3_ENABLED = old
    m = esf_module("baseline_edge", seed=7)
    sdg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    out = oesf.esf_forward(sdg, setting("baseline_edge"), *batch_args(b, edge.cpu()), training=True)
    out[3].sum().backward()
    grads = {k: v.grad.clone() for k, v in sdg.items() if v.grad is not None}
    return dict(B=B, batch=b, edge=edge, loss=float(out[3].detach().reshape(-1)[0]), op=out[0].detach(), grads=grads)


def test_esf_train_distinct_frames_vs_oracle(train_ref):
    """fp32-storage training plan, B=32 distinct frames: loss, logits and every parameter's gradient against the oracle's autograd
    (gradient norms at the 1e-2 of the fixture tests; three full tensors)."""
    from common import batch_args, esf_module
    r = train_ref
    m = esf_module("baseline_edge", seed=7).to(DEV).train()
    op, elPred, latent, loss, elOut = m(*_dev(batch_args(r["batch"], r["edge"])))
    np.testing.assert_allclose(loss.item(), r["loss"], rtol=1e-3)
    per_frame = (op.detach().cpu() - r["op"]).abs().flatten(1).max(1)[0]
    assert per_frame.max().item() < TOL, "frame %d: training-mode logits off by %.2e" % (int(per_frame.argmax()), per_frame.max())
    loss.sum().backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    scale = max(g.norm().item() for g in r["grads"].values())
    worst, wname = 0.0, ""
    for n, g in r["grads"].items():
        rn, hn = g.double().norm().item(), params[n].grad.double().norm().item()
        e = abs(rn - hn) / max(rn, 1e-6 * scale)
        if e > worst:
            worst, wname = e, n
    print("training step, B=%d distinct frames: loss %.5f vs %.5f, worst gradient-norm deviation %.2e (%s)"
          % (r["B"], loss.item(), r["loss"], worst, wname))
    assert worst < 1e-2, "gradient norm of %s differs by %.2e" % (wname, worst)
    for k in ("dec.final.conv2.weight", "enc.head.conv1.weight", "enc.down_block1.conv21.weight", "elReg.l2.weight"):
        g = r["grads"][k]
        e = (params[k].grad.cpu() - g).abs().max().item()
        assert e <= 1.5e-2 * g.abs().max().item() + 1e-7, "%s: max err %.3e (scale %.3e)" % (k, e, g.abs().max().item())
    kinds = _kinds(m._last_plan) | _kinds(m._last_plan.bw)
    assert any(k.startswith("conv_f16x3:") for k in kinds), kinds


def test_esf_train_distinct_frames_bf16_vs_fp32_storage(train_ref):
    """The bf16-storage training plan on the same distinct frames against the fp32-storage HIP plan (itself checked against the
    oracle above) AND the oracle: loss, logits, gradient norms at the tolerances of tests/test_gpu_bf16.py."""
    from common import batch_args, esf_module
    r = train_ref
    res = {}
    for st in (torch.float32, torch.bfloat16):
        m = esf_module("baseline_edge", seed=7).to(DEV).to(st).train()
        op, _, _, loss, _ = m(*_dev(batch_args(r["batch"], r["edge"])))
        loss.sum().backward()
        torch.cuda.synchronize()
        res[st] = (loss.item(), op.detach().cpu(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None})
        if st == torch.bfloat16:
            pl = m._last_plan
            assert pl.bf16 and pl.bw.bf16
            kinds = _kinds(pl) | _kinds(pl.bw)
            assert "conv_bf16:3x3" in kinds and "conv_bf16:1x1" in kinds, kinds
        del m
    lf, of, gf = res[torch.float32]
    lh, oh, gh = res[torch.bfloat16]
    lerr = abs(lh - lf) / abs(lf)
    operr = ((oh - of).abs().flatten(1).max(1)[0] / of.abs().flatten(1).max(1)[0])
    names = [n for n in gf if n in r["grads"]]
    rel = np.array([abs(gh[n].norm().item() - gf[n].norm().item()) / max(gf[n].norm().item(), 1e-30) for n in names])
    keep = np.array([gf[n].norm().item() for n in names])
    rel = rel[keep > 1e-6 * keep.max()]
    flat_h, flat_f = torch.cat([gh[n].reshape(-1) for n in names]), torch.cat([gf[n].reshape(-1) for n in names])
    flat_o = torch.cat([r["grads"][n].double().reshape(-1) for n in names])
    whole = float((flat_h - flat_f).norm() / flat_f.norm())
    cos = float(torch.dot(flat_h, flat_o) / (flat_h.norm() * flat_o.norm()))
    print("bf16 vs fp32 storage, B=%d distinct frames: loss rel %.2e, logits rel-to-max worst frame %.2e, gradient norms median %.2e p90 %.2e, "
          "whole vector rel L2 %.2e, cosine to the oracle's gradient %.4f" % (r["B"], lerr, operr.max(), np.median(rel),
                                                                             np.sort(rel)[int(0.9 * len(rel))], whole, cos))
    # per frame: max |logit error| / max |logit|; a frame-indexing bug would put single frames at O(1), bf16 rounding noise spreads
    # (tests/test_gpu_bf16.py bounds the golden frames at 8e-2; over 32 distinct frames the worst one measured 1.4e-1)
    print("   logits rel-to-max per frame: median %.2e, worst %.2e" % (operr.median(), operr.max()))
    assert lerr < 1e-2 and operr.median().item() < 8e-2 and operr.max().item() < 2.5e-1
    # ONE realisation of the storage noise: two correct builds whose forward activations differ by single bf16 ulps draw gradient-norm
    # errors (median / p90) of 0.012-0.047 / 0.07-0.22 and whole-vector errors of 0.33-0.41 over four batches of 32 frames
    # (profiles/r06_bf16_noise_realisations.txt: round 5's and round 6's libraries side by side; this batch drew 0.016 / 0.086 / 0.28 with
    # the former and 0.027 / 0.28 / 0.35 with the latter).  The bounds are that spread; an addressing bug moves the vector by O(1)
    assert np.median(rel) < 6e-2 and np.sort(rel)[int(0.9 * len(rel))] < 3.5e-1
    assert whole < 4.5e-1 and cos > 0.90

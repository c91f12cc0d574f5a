"""-m gpu: the plans bench.py measures (B=64 inference, B=40 for the frame-tail split, a B=32 training plan) against the
reference fixtures.  Kernel selection keys on the batch size (deep trunk kernel, streaming / LDS 1x1 kernels, tail
launches), so the golden B=2 batch is tiled: InstanceNorm is per sample, eval BatchNorm is folded and every loss term is
a mean over valid samples, hence EVERY pair of frames of the big batch must reproduce the B=2 fixture.
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _kinds(pl):
    return {k for k, _ in pl.meta}


def _names(pl):
    return [n for _, _, n in pl.calls]


@pytest.mark.parametrize("rep", [32, 20])
def test_bdcn_big_batch_vs_reference(rep):
    from common import bdcn_module, gold
    from egne_amd import synth
    g = gold("bdcn_b2_240x320")
    bd = bdcn_module().to(DEV)
    b = synth.make_batch(2, seed=1234)
    x = torch.cat((b["img"],) * 3, 1).to(DEV).repeat(rep, 1, 1, 1)
    got = bd.forward_fuse(x).cpu().numpy().reshape(rep, 2, 1, 240, 320)
    err = np.abs(got - g["fuse"][None]).max()
    assert err < TOL, "B=%d fused edge map off by %.2e" % (2 * rep, err)
    pl = next(iter(bd._plans.values()))
    kinds = _kinds(pl)
    assert "conv_f16x3:big" in kinds, kinds            # the deep trunk kernel is part of this plan
    # ... and so are the role-split 3x3 (conv1_2 with pool1 as its second output), the one-launch dilated groups and the halo kernel
    assert {"conv_f16x3:msdil", "conv_f16x3:halo", "conv_f16x3:first"} <= kinds and kinds & {"conv_f16x3:rs", "conv_f16x3:rw"}, kinds
    assert sum(n == "vgg.pool" for n in _names(pl)) == 2, "pool1 / pool2 should come from the epilogues of conv1_2 / conv2_2"
    if rep == 20:                                      # B=40: ragged last round -> frame tail on the flat kernel
        assert any(n.endswith(".tail") for n in _names(pl)), "no .tail launch in the B=40 plan"
    print("B=%d: edge err %.2e, kernels %s" % (2 * rep, err, sorted(kinds)))


def test_bdcn_plain_f16_operands_next_to_a_bf16_training_plan():
    """``BDCN.f16_products = 1`` (train.py --prec 16, bench.py's bf16 training leg): the kernels that know egne_conv_desc.f16_products
    multiply plain f16 operands (11-bit significand, fp32 accumulation) instead of the split's hi / lo pairs.  The edge map feeds a
    network that rounds it to bf16 on entry, so the bound is bf16's resolution on (0, 1): half an ulp = 2^-9 below 1 -- measured
    far inside it -- against the reference-generated fixture, whose edge values span 0.12 .. 0.99.  The plan is a separate one (the key
    carries the product count): the split plan of the same module, used by validation and the inference entry points, still meets
    1e-3 afterwards."""
    from common import bdcn_module, gold
    from egne_amd import synth
    g = gold("bdcn_b2_240x320")
    bd = bdcn_module().to(DEV)
    b = synth.make_batch(2, seed=1234)
    x = torch.cat((b["img"],) * 3, 1).to(DEV).repeat(32, 1, 1, 1)
    bd.f16_products = 1
    got1 = bd.forward_fuse(x).cpu().numpy().reshape(32, 2, 1, 240, 320)
    pl1 = bd._last_plan
    assert pl1.f16_products == 1 and "conv_f16x3:big" in _kinds(pl1)
    bd.f16_products = 0
    got3 = bd.forward_fuse(x).cpu().numpy().reshape(32, 2, 1, 240, 320)
    assert bd._last_plan is not pl1 and bd._last_plan.f16_products == 0
    e1, e3 = np.abs(got1 - g["fuse"][None]).max(), np.abs(got3 - g["fuse"][None]).max()
    spread = float(g["fuse"].max() - g["fuse"].min())
    print("edge map, B=64: plain f16 operands %.2e, split products %.2e from the reference (edge values span %.3f)" % (e1, e3, spread))
    assert e3 < TOL and e1 < 2.0 ** -9, "plain-f16 edge map off by %.2e" % e1
    assert np.abs(got1 - got3).max() > 0, "the single-product plan ran the split kernels"


def test_bdcn_f16_storage_recalibrates_after_an_overflow_and_a_weight_update():
    """The storage scales of a plain-f16 plan (egne_conv_desc.out_split = 2) are calibrated like every operand scale, with 32x of head-room.
    Frames 3000x the calibration batch push stored halves beyond the f16 range: the sticky word must be set (BDCN.overflowed), the next call
    re-calibrates -- bound first, measured maximum second (engine.Plan.post_cal) -- and then reproduces a FRESH module's plan bit for bit;
    weights changed in place (the plan's version guards) likewise, without a rerun."""
    from common import bdcn_module
    from egne_amd import synth
    x = torch.cat((synth.make_batch(64, seed=5)["img"],) * 3, 1).to(DEV)
    bd = bdcn_module().to(DEV)
    bd.f16_products = 1
    bd.forward_fuse(x)
    assert not bd.overflowed() and bd._last_plan.f16_storage == 2
    xl = x * 3000.0
    bd.forward_fuse(xl)
    assert bd.overflowed(), "stored halves beyond the f16 range went unnoticed"
    o1 = bd.forward_fuse(xl)
    assert not bd.overflowed() and bool(torch.isfinite(o1).all())
    fresh = bdcn_module().to(DEV)
    fresh.f16_products = 1
    assert torch.equal(o1, fresh.forward_fuse(xl)), "re-calibrated plan differs from a fresh one"
    with torch.no_grad():
        for m in (bd, fresh):
            m.features.conv1_1.weight.mul_(200.0)
            m.features.conv1_1.bias.mul_(200.0)
    o2 = bd.forward_fuse(x)
    assert not bd.overflowed() and bool(torch.isfinite(o2).all())
    fresh2 = bdcn_module().to(DEV)
    fresh2.f16_products = 1
    fresh2.load_state_dict(bd.state_dict())
    assert torch.equal(o2, fresh2.forward_fuse(x)), "plan after a weight update differs from a fresh one"


@pytest.mark.parametrize("B,H,W", [(64, 240, 320), (6, 240, 320), (8, 101, 150)])
def test_bdcn_plain_f16_plan_keeps_its_stage1_tensors_as_f16(B, H, W):
    """Round 6: in a plain-f16 plan conv1_1, conv1_2, pool1 and every trunk tensor from conv3_1 on (with pool3 / pool4) are STORED as f16 (egne_conv_desc.out_split = 2, egne_seg.presplit = 2;
    vgg16_c.py:66-70 under utils.py:646).  Their consumers round every operand to exactly the stored value while staging it, so the plan
    must reproduce the fp32-storage plan of the same arithmetic BIT for bit -- all 11 outputs, distinct frames -- at half the bytes; the
    storage scales come out of the calibration (bound first, measured maximum second) with the usual 32x of head-room."""
    from common import bdcn_module
    from egne_amd import engine, synth
    bd = bdcn_module().to(DEV)
    bd.f16_products = 1
    x = torch.cat((synth.make_batch(B, seed=77)["img"],) * 3, 1).to(DEV)
    if (H, W) != (240, 320):           # ragged tiles / columns in every kernel of the plan
        x = x[:, :, :H, :W].contiguous()
    outs, plans = [], []
    for on in (True, False):
        old, old_tail = engine.F16_STORAGE, engine.BIG_SPLIT_TAIL
        engine.F16_STORAGE = on
        engine.BIG_SPLIT_TAIL = False      # (the f16 plan has no fp32 frame tails on the flat kernel, whose sums run in another order: compare like with like)
        try:
            bd._plans.clear()
            o = bd(x)
            torch.cuda.synchronize()
            o2 = bd(x)          # replay with the stored scales
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(o, o2))
            outs.append([t.clone() for t in o])
            plans.append(bd._last_plan)
        finally:
            engine.F16_STORAGE, engine.BIG_SPLIT_TAIL = old, old_tail
    p16, p32 = plans
    # B = 64: conv1_1, conv1_2, conv3_1 .. conv5_3; B = 6: the deep layers are too small for the deep trunk kernel -> stage 1 only
    level, nf16 = (2, 11) if B == 64 else (1, 2)
    assert p16.f16_storage == level and not p32.f16_storage and len(p16.post_cal) == nf16 and not p32.post_cal
    for (d, dst, fs, npix) in p16.post_cal.values():
        m = float(dst.buf.float().abs().max())
        assert dst.buf.dtype == torch.float16 and 1024.0 <= m < 2048.0 and d.out_split == 2 and d.out_split_scale == fs.value, (m, fs.value)
    assert [k for k, _ in p16.meta] == [k for k, _ in p32.meta]
    for k, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), "output %d differs between f16 and fp32 storage: %.3e" % (k, (a - b).abs().max().item())
    bytes16 = sum(t.numel() * t.element_size() for t in p16.keep if torch.is_tensor(t))
    bytes32 = sum(t.numel() * t.element_size() for t in p32.keep if torch.is_tensor(t))
    print("B=%d: f16 storage of conv1_1 / conv1_2 / pool1: plan buffers %.2f -> %.2f GB, all 11 outputs bit-identical" % (B, bytes32 / 1e9, bytes16 / 1e9))
    bd._plans.clear()


def test_bdcn_side_outputs_big_batch():
    """All 11 maps at B=64 (the 10 side outputs only at the fixture's 8x8 sub-grid)."""
    from common import bdcn_module, gold
    from egne_amd import synth
    g = gold("bdcn_b2_240x320")
    bd = bdcn_module().to(DEV)
    b = synth.make_batch(2, seed=1234)
    outs = bd(torch.cat((b["img"],) * 3, 1).to(DEV).repeat(32, 1, 1, 1))
    for i in range(10):
        o = outs[i][:, :, ::8, ::8].cpu().numpy().reshape(32, 2, 1, 30, 40)
        assert np.abs(o - g["map%d_sub" % i][None]).max() < TOL, "side output %d" % i


def _tiled_args(b, edge, rep):
    from common import batch_args
    out = []
    for a in batch_args(b, edge):
        if torch.is_tensor(a):
            a = a.to(DEV)
            out.append(a.repeat(*([rep] + [1] * (a.dim() - 1))))
        else:
            out.append(a)
    return out


@pytest.mark.parametrize("name", ["esf_edge_b2", "esf_edge_b2_absent1", "esf_adain_edge_b2"])
def test_esf_eval_b64_vs_reference(name):
    from common import ESF_CASES, bdcn_module, esf_module, gold
    from egne_amd import synth
    from egne_amd.utils import calc_edge
    cfg, variant, kw = ESF_CASES[name]
    kw = dict(kw)
    g = gold(name)
    rep = 32
    bd = bdcn_module().to(DEV)
    b = synth.make_batch(kw.pop("B"), **kw)
    x = b["img"].to(DEV).repeat(rep, 1, 1, 1)
    edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), x, bd, DEV)     # B=64 edge maps from the B=64 plan
    m = esf_module(cfg, variant).to(DEV).eval()
    args = _tiled_args(b, edge[:2], rep)
    args[1] = edge
    with torch.no_grad():
        op, elPred, latent, loss, elOut = m(*args)
    assert tuple(op.shape) == (64, 3, 240, 320)
    opc = op.cpu().numpy().reshape(rep, 2, 3, 240, 320)
    ref = g["op"]
    got = opc if ref.shape[-1] == 320 else opc[..., ::4, ::4]
    err = np.abs(got - ref[None]).max()
    assert err < TOL, "logits off by %.2e" % err
    for t, k in ((elOut, "elOut"), (elPred, "elPred"), (latent, "latent")):
        v = t.cpu().numpy()
        np.testing.assert_allclose(v.reshape((rep, 2) + v.shape[1:]), np.broadcast_to(g[k][None], (rep,) + g[k].shape), atol=TOL)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=1e-3)
    from common import mask_mismatch
    mask = m.predictions().cpu().numpy().astype(np.uint8).reshape(rep, 2, 240, 320)
    worst = max(range(rep), key=lambda r: int(np.count_nonzero(mask[r] != mask[0])))        # (every tile holds the same two frames)
    bad = mask_mismatch(mask[0], g, name + " (eval, B=64 plan, first tile)")
    if worst != 0:
        bad = max(bad, mask_mismatch(mask[worst], g, name + " (eval, B=64 plan, most deviating tile %d)" % worst))
    kinds = _kinds(m._last_plan)
    assert any(k.startswith("conv_f16x3:") for k in kinds), kinds
    if name == "esf_edge_b2":        # the benchmarked configuration: fused pairs, the convBlock head, one-pass Transition_down
        assert {"conv_f16x3:fused1x1", "conv_f16x3:fused3x3c4", "conv_f16x3:tdpool1x1"} <= kinds and kinds & {"conv_f16x3:rs", "conv_f16x3:rw"}, kinds
    print("%s at B=64: logits err %.2e, worst mask pixel diff %d, kernels %s" % (name, err, bad, sorted(kinds)))


@pytest.mark.parametrize("name", ["esf_edge_b2", "esf_edge_b2_absent1"])
def test_esf_train_b32_vs_reference(name):
    """Training plan at B=32 (bench.py's shape family) on the golden batch tiled x16: batch-statistic BatchNorm sees the
    same mean / biased variance, every loss term is a mean over valid samples, so loss and parameter gradients equal the
    B=2 reference values."""
    _train_tiled_vs_reference(name, 16)


def test_esf_train_b256_vs_reference():
    """BASELINE.json configs[2]'s shape as bench.py's train leg runs it: the training plan at B=256 (golden batch tiled x128,
    ~220 GB of HBM) against the B=2 reference loss, logits, BatchNorm statistics and parameter gradients.  Skipped when the card
    does not have the memory free (other tests' cached plans) - never a failure for lack of memory."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0]
    if free < 240e9:
        pytest.skip("needs ~225 GB of free HBM, %.0f GB available" % (free / 1e9))
    try:
        _train_tiled_vs_reference("esf_edge_b2", 128)
    except torch.cuda.OutOfMemoryError:
        pytest.skip("out of HBM while building the B=256 training plan")
    finally:
        gc.collect()
        torch.cuda.empty_cache()


def _train_tiled_vs_reference(name, rep):
    from common import ESF_CASES, bdcn_module, esf_module, gold
    from egne_amd import engine, synth
    from egne_amd.utils import calc_edge
    cfg, variant, kw = ESF_CASES[name]
    kw = dict(kw)
    g = gold(name)
    b = synth.make_batch(kw.pop("B"), **kw)
    old = engine.F16X3_ENABLED
    engine.F16X3_ENABLED = False          # exact-fp32 edge maps: the gradient fixtures are sensitive to 1e-6 input changes
    try:
        bd = bdcn_module().to(DEV)
        edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)
    finally:
        engine.F16X3_ENABLED = old
    m = esf_module(cfg, variant).to(DEV).train()
    args = _tiled_args(b, edge, rep)
    op, elPred, latent, loss, elOut = m(*args)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["t_loss"], rtol=1e-3)
    o = op.detach().cpu()[:, :, ::4, ::4].numpy().reshape(rep, 2, 3, 60, 80)
    assert np.abs(o - g["t_op_sub"][None]).max() < TOL
    np.testing.assert_allclose(m.enc.head.bn.running_mean.cpu().numpy(), g["t_head_rm"], rtol=1e-4, atol=1e-5)
    loss.sum().backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    got = np.array([params[n].grad.double().norm().item() for n in names])
    ref = g["grad_l2"]
    rel = np.abs(got - ref) / np.maximum(ref, 1e-6 * ref.max())
    worst = int(np.argmax(rel))
    assert rel.max() < 1e-2, "grad L2 of %s: %.6e vs %.6e" % (names[worst], got[worst], ref[worst])
    for k in ("dec.final.conv2.weight", "enc.head.conv1.weight", "enc.down_block1.conv21.weight"):
        r = g["grad::" + k]
        e = np.abs(params[k].grad.cpu().numpy() - r).max()
        assert e <= 1.5e-2 * np.abs(r).max() + 1e-7, "%s: max err %.3e (scale %.3e)" % (k, e, np.abs(r).max())


def test_predictions_follow_the_last_forward(capsys):
    """Cached plans of several batch sizes: predictions() must be the mask of the forward that ran last."""
    from common import batch_args, bdcn_module, esf_module
    from egne_amd import synth
    from egne_amd.utils import calc_edge
    bd = bdcn_module().to(DEV)
    m = esf_module("baseline_edge").to(DEV).eval()
    ns = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
    masks = {}
    for B in (2, 3, 2, 1, 3):
        b = synth.make_batch(B, seed=10 + B)
        edge = calc_edge(ns, b["img"].to(DEV), bd, DEV)
        args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
        with torch.no_grad():
            op = m(*args)[0]
        pred = m.predictions()
        assert tuple(pred.shape) == (B, 240, 320)
        assert torch.equal(pred, op.max(1)[1])
        held = masks.setdefault(B, pred)
        assert torch.equal(held, pred)          # same input -> same mask, and the copy handed out earlier was not overwritten


def test_fit_rejects_frames_it_was_not_given():
    import ctypes as C
    from egne_amd import _lib
    from egne_amd.utils import _mesh_axes, fit_ellipses
    mask = torch.zeros((2, 240, 320), dtype=torch.int64, device=DEV)
    mask[:, 100:140, 130:190] = 1
    init = np.array([[160.0, 120.0, 30.0, 20.0, 0.0]] * 2)
    with pytest.raises(ValueError):
        fit_ellipses(mask, [0, 2], [1, 1], init)
    # straight through the C-ABI: the out-of-range fit reports NaN and reads nothing, the valid one is unaffected
    L = _lib.lib()
    fo = torch.tensor([0, 7], dtype=torch.int32, device=DEV)
    cl = torch.tensor([1, 1], dtype=torch.int32, device=DEV)
    ini = torch.from_numpy(init).to(DEV)
    out = torch.zeros((2, 5), dtype=torch.float64, device=DEV)
    xs, ys = _mesh_axes(240, 320, mask.device)
    _lib.check(L.egne_ellipse_fit(mask.data_ptr(), 2, fo.data_ptr(), cl.data_ptr(), 2, 240, 320, xs.data_ptr(), ys.data_ptr(),
                                  ini.data_ptr(), out.data_ptr(), None, _lib.stream_ptr()))
    o = out.cpu().numpy()
    assert np.isnan(o[1]).all() and np.array_equal(o[0], fit_ellipses(mask, [0], [1], init[:1])[0])


def test_ellipse_seeds_on_device_vs_reference():
    """egne_ellipse_init_from_pred against my_ellipse.transform of the reference (fixture) -- float64 conic algebra."""
    from common import gold
    from egne_amd.utils import ellipse_seeds_from_pred
    g = gold("ellipse_transform")
    prm = g["params"].astype(np.float32)                     # the regression head emits float32
    from oracle import fit as ofit
    Hm = np.array([[160.0, 0, 160.0], [0, 120.0, 120.0], [0, 0, 1]])
    want = np.stack([ofit.transform(p.astype(np.float64), Hm) for p in prm])     # oracle == reference on the fixture (CPU test)
    el = torch.from_numpy(prm.reshape(8, 10)).to(DEV)
    init, fo, cl = ellipse_seeds_from_pred(el, 240, 320)
    np.testing.assert_allclose(init.cpu().numpy(), want, rtol=1e-11, atol=1e-11)
    assert fo.cpu().tolist() == [i // 2 for i in range(16)] and cl.cpu().tolist() == [1, 2] * 8
    # and the float64 fixture itself within float32 input rounding
    np.testing.assert_allclose(init.cpu().numpy(), g["out"], rtol=2e-5, atol=2e-5)


def test_data_parallel_two_shards_vs_reference():
    """DP arithmetic with real gradients on one GPU: two shards run one after the other (own BatchNorm statistics, own
    loss normalisation), their gradient arenas combined exactly as parallel.allreduce_grads does (SUM over ranks, then
    / world) -- against the reference run as two DataParallel replicas (fixture dp_two_shards; train.py:205,285)."""
    from common import batch_args, bdcn_module, esf_module, gold
    from egne_amd import engine, synth
    from egne_amd.utils import calc_edge
    g = gold("dp_two_shards")
    old = engine.F16X3_ENABLED
    engine.F16X3_ENABLED = False
    try:
        bd = bdcn_module().to(DEV)
        shards = []
        for kw in (dict(seed=1234), dict(seed=4321, mask_absent_every=2)):
            b = synth.make_batch(2, **kw)
            shards.append((b, calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)))
    finally:
        engine.F16X3_ENABLED = old
    arenas, rm = [], None
    for i, (b, edge) in enumerate(shards):
        m = esf_module("baseline_edge").to(DEV).train()          # every rank starts from the broadcast parameters
        args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
        loss = m(*args)[3]
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss"][i], rtol=1e-3)
        loss.sum().backward()
        torch.cuda.synchronize()
        arenas.append(m._ensure_grad_arena().clone())
        if i == 0:
            rm = m.enc.head.bn.running_mean.cpu().numpy()
    flat = m._ensure_grad_arena()
    flat.copy_(arenas[0] + arenas[1])        # dist.all_reduce(SUM)
    flat.div_(2)                             # / world  (parallel.allreduce_grads.finish)
    params = dict(m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    got = np.array([params[n].grad.double().norm().item() for n in names])
    ref = g["grad_l2"]
    rel = np.abs(got - ref) / np.maximum(ref, 1e-6 * ref.max())
    worst = int(np.argmax(rel))
    assert rel.max() < 1e-2, "averaged grad L2 of %s: %.6e vs %.6e" % (names[worst], got[worst], ref[worst])
    for k in ("elReg.l2.weight", "dec.final.conv2.weight", "enc.head.conv1.weight", "enc.down_block1.conv21.weight"):
        r = g["grad::" + k]
        e = np.abs(params[k].grad.cpu().numpy() - r).max()
        assert e <= 1.5e-2 * np.abs(r).max() + 1e-7, "%s: max err %.3e (scale %.3e)" % (k, e, np.abs(r).max())
    np.testing.assert_allclose(rm, g["head_rm"], rtol=1e-4, atol=1e-5)       # rank 0's running statistics are the ones kept


@pytest.mark.parametrize("mag", [1e-4, 1.0, 3e3, 1e6])
@pytest.mark.parametrize("shape", [(2, 64, 64, 64, 80), (1, 128, 256, 150, 223), (2, 32, 32, 120, 160)])
def test_split_kernels_any_activation_magnitude(mag, shape):
    """The split-f16 kernels pre-scale their input by a power of two chosen from its measured max (calibration pass): the
    error against float64 stays at the fp32 level whether activations are 1e-4 or 1e6 (round 1's fixed scale of 16
    overflowed the f16 range above 4094)."""
    import torch.nn.functional as F
    from gpu_util import to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(int(mag * 7) % 1000 + Cin)
    x = F.relu(torch.randn(B, Cin, H, W, generator=g)) * 3 * mag
    w, b = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5), torch.randn(Cout, generator=g) * mag
    truth = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
    layer.split = True
    out = pl.buf(B, H, W, Cout)
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
    assert pl.meta[-1][0].startswith("conv_f16x3") and pl.cal
    for _ in range(2):          # calibrating run, then the replay with the stored scale
        pl.run()
        torch.cuda.synchronize()
        got = out.cpu().permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 2e-6, "mag %g: relative error %.2e" % (mag, err)
    a_scale = pl.calls[-1][1][pl.cal[len(pl.calls) - 1][0]]
    assert 1024 <= a_scale * x.abs().max().item() < 2048, a_scale
    # the same plan fed 1000x the batch it was calibrated on (beyond the 32x of head-room): the kernel's epilogue reports the f16
    # overflow (egne_conv_desc.ovf_flag), the plan answers by re-calibrating and running again (Plan.check_overflow)
    assert not pl.overflowed()
    if mag < 1e6:
        px.buf.mul_(1000.0)
        pl.run()
        if pl.meta[-1][0] == "conv_f16x3:halo":
            # the LDS-halo kernel is at its register limit and carries no test of its own (conv_halo_f16.hip): what it stores non-finite
            # is reported by the kernel that reads it next -- in a plan of one layer, nobody: check the stored values themselves
            torch.cuda.synchronize()
            assert not torch.isfinite(out).all(), "1000x the calibration batch left finite values behind"
            pl.calibrated = False
            pl.run()
        else:
            assert pl.check_overflow(), "1000x the calibration batch went through unnoticed"
        torch.cuda.synchronize()
        truth2 = F.relu(F.conv2d(x.double() * 1000.0, w.double(), b.double(), padding=1))
        err = (out.cpu().permute(0, 3, 1, 2).double() - truth2).abs().max().item() / truth2.abs().max().item()
        assert err < 2e-6 and not pl.overflowed(), "after re-calibration: relative error %.2e" % err


def test_esf_stale_scales_after_a_batchnorm_update_are_reported():
    """A calibrated inference plan whose BatchNorm parameters change under it (a checkpoint with the same convolutions and another head
    BatchNorm: no convolution is re-packed, so nothing re-calibrates): the raw block-0 tensors grow 3000x, beyond the head-room of
    the stored f16 pre-scales.  The fused 1x1 -> 3x3 kernels report it (DenseNet2D.overflowed), the next call re-calibrates and
    matches the oracle."""
    from common import batch_args, bdcn_module, esf_module, setting
    from egne_amd import synth
    from egne_amd.utils import calc_edge
    from oracle import esfnet as oesf
    bd = bdcn_module().to(DEV)
    b = synth.make_batch(2, seed=77)
    edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)
    m = esf_module("baseline_edge", seed=2).to(DEV).eval()
    args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
    with torch.no_grad():
        m(*args)
        assert not m.overflowed()
        m.enc.head.bn.weight.mul_(3000.0)
        m.enc.head.bn.bias.add_(0.3).mul_(3000.0)
        ref = oesf.esf_forward({k: v.cpu() for k, v in m.state_dict().items()}, setting("baseline_edge"), *batch_args(b, edge.cpu()))
        m(*args)
        assert m.overflowed(), "raw activations 3000x the calibration batch's went through unnoticed"
        op = m(*args)[0]
        assert not m.overflowed()
    scale = ref[0].abs().max().item()
    err = (op.cpu() - ref[0]).abs().max().item()
    assert err < 1e-3 * max(1.0, scale), "after re-calibration: logits off by %.3g (scale %.3g)" % (err, scale)


def test_esf_large_raw_activations_vs_oracle():
    """A checkpoint whose activations are ~3000x those of the seeded weights (head BatchNorm gamma / beta x 3000: the raw
    skip / dense-block tensors grow by that factor, the normalised ones do not): eval forward against the CPU oracle."""
    from common import batch_args, bdcn_module, esf_module, setting
    from egne_amd import synth
    from egne_amd.utils import calc_edge
    from oracle import esfnet as oesf
    bd = bdcn_module().to(DEV)
    b = synth.make_batch(2, seed=77)
    edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)
    m = esf_module("baseline_edge", seed=2)
    with torch.no_grad():
        m.enc.head.bn.weight.mul_(3000.0)
        m.enc.head.bn.bias.add_(0.3).mul_(3000.0)
        ref = oesf.esf_forward(m.state_dict(), setting("baseline_edge"), *batch_args(b, edge.cpu()))
    m = m.to(DEV).eval()
    args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
    with torch.no_grad():
        op, elPred, latent, loss, elOut = m(*args)
    assert torch.isfinite(op).all() and torch.isfinite(loss).all()
    scale = ref[0].abs().max().item()
    err = (op.cpu() - ref[0]).abs().max().item()
    print("logit scale %.3g, err %.3g (relative %.2e)" % (scale, err, err / scale))
    assert err < 1e-3 * max(1.0, scale)
    assert (m.predictions().cpu() != ref[0].max(1)[1]).float().mean().item() < 1e-4

"""-m gpu: every C-ABI op against the CPU oracle / plain torch fp32 on seeded inputs.

Tolerances: fp32 MFMA accumulates a k-ordered fmaf chain (exact fp32), the CPU reference sums in a
different order -> relative 1e-5 of the output scale for convs; byte-exact for argmax / fit.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(g, *shape):
    return torch.randn(*shape, generator=g)


def _close(a, b, rel=2e-5):
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= rel * scale, "max err %.3e vs scale %.3e" % (err, scale)


@pytest.fixture(autouse=True)
def _one_kernel_per_test(request):
    """The op tests use small shapes to address ONE kernel each (halo, role-split, resident-weights ...); the small-problem form of
    the flat split-f16 kernel, which the engine prefers for such shapes, is only left on in its own tests (and is what the one- and
    two-frame network tests run on)."""
    from egne_amd import engine
    old = engine.SMALL_ENABLED
    engine.SMALL_ENABLED = any(k in request.node.name for k in ("small_problem", "split_precision"))
    yield
    engine.SMALL_ENABLED = old


@pytest.fixture(scope="module")
def G():
    from gpu_util import conv_hip  # noqa: F401  (imports torch.cuda)
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.Generator().manual_seed(2024)


@pytest.mark.parametrize("B,Cin,Cout,H,W,d,act", [
    (2, 3, 64, 29, 39, 1, 1),      # VGG first layer shape class, odd spatial, M tail
    (1, 64, 64, 24, 40, 1, 1),
    (2, 38, 38, 30, 40, 2, 2),     # odd channels -> padded K and N, dilation 2 (VGG conv5 class)
    (1, 128, 256, 15, 20, 1, 0),   # wide N tile (128), no activation
    (3, 32, 3, 16, 24, 1, 2),      # Cout=3 (final conv)
    (1, 512, 512, 9, 11, 2, 1),    # deep K
    # W >= 60 -> LDS-halo kernel (conv_halo.hip): ragged tiles in x and y, padded K/N, dilation 2, deep K
    (2, 32, 32, 21, 70, 1, 2),
    (1, 64, 128, 16, 96, 1, 1),
    (1, 38, 64, 13, 64, 2, 0),
    (2, 3, 64, 19, 100, 1, 1),
    (1, 160, 100, 9, 65, 1, 2),
])
def test_conv3x3(G, B, Cin, Cout, H, W, d, act):
    from gpu_util import conv_hip
    x, w, b = _rand(G, B, Cin, H, W), _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    ref = F.conv2d(x, w, b, padding=d, dilation=d)
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref) if act == 2 else ref)
    _close(conv_hip([x], [w], [b], pad=(1, 1), dils=(d,), act=act), ref)


def test_conv1x1_concat_free_with_fused_instancenorm(G):
    """1x1 conv reading three slices (would-be torch.cat), InstanceNorm + LeakyReLU fused on load."""
    from gpu_util import conv_hip
    B, H, W = 2, 17, 23
    xs = [_rand(G, B, 38, H, W) * 2 + 1, _rand(G, B, 32, H, W), _rand(G, B, 64, H, W)]
    w, b = _rand(G, 76, 134, 1, 1) / 11, _rand(G, 76)
    mean = xs[0].mean((2, 3))
    rstd = 1 / torch.sqrt(xs[0].var((2, 3), unbiased=False) + 1e-5)
    xin = torch.cat([F.leaky_relu(F.instance_norm(xs[0])), xs[1], xs[2]], 1)
    ref = F.conv2d(xin, w, b)
    got = conv_hip(xs, [w], [b], norm={0: (rstd, -mean * rstd, 2)})
    _close(got, ref, 5e-5)


def test_conv3x3_halo_with_fused_instancenorm(G):
    """Down-block conv1(IN(x)) on the halo path: zero padding applies AFTER the normalisation."""
    from gpu_util import conv_hip
    B, C, H, W = 2, 38, 18, 72
    x = _rand(G, B, C, H, W) * 2 + 1
    w, b = _rand(G, 64, C, 3, 3) / 18, _rand(G, 64)
    mean, rstd = x.mean((2, 3)), 1 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
    ref = F.leaky_relu(F.conv2d(F.instance_norm(x), w, b, padding=1))
    _close(conv_hip([x], [w], [b], pad=(1, 1), act=2, norm={0: (rstd, -mean * rstd, 0)}), ref, 5e-5)


def test_msblock_fused_dilated_group(G):
    """bdcn_new.py:49-55: o + relu(conv_d4(o)) + relu(conv_d8(o)) + relu(conv_d12(o)) in one launch."""
    from gpu_util import conv_hip
    B, H, W = 2, 31, 45
    o = F.relu(_rand(G, B, 32, H, W))
    ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
    bs = [_rand(G, 32) for _ in range(3)]
    ref = o.clone()
    for w, b, d in zip(ws, bs, (4, 8, 12)):
        ref = ref + F.relu(F.conv2d(o, w, b, padding=d, dilation=d))
    _close(conv_hip([o], ws, bs, pad=(1, 1), dils=(4, 8, 12), act=1, residual=o), ref)


def test_conv_misc_geometries(G):
    from gpu_util import conv_hip
    # regressionModule c1: 2x3 kernel, no padding, two slices (utils.py:991-994)
    xa, xb = _rand(G, 2, 153, 15, 20), _rand(G, 2, 153, 15, 20)
    w, b = _rand(G, 128, 306, 2, 3) / 40, _rand(G, 128)
    _close(conv_hip([xa, xb], [w], [b], act=2), F.leaky_relu(F.conv2d(torch.cat([xa, xb], 1), w, b)))
    # StyleEncoder: reflect pad 3 + 7x7, then reflect pad 1 + 4x4 stride 2 (RITnet_v2.py:95-101)
    x = _rand(G, 2, 3, 20, 28)
    w, b = _rand(G, 64, 3, 7, 7) / 12, _rand(G, 64)
    _close(conv_hip([x], [w], [b], pad=(3, 3), act=1, pad_mode=1), F.relu(F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), w, b)))
    x = _rand(G, 2, 64, 20, 28)
    w, b = _rand(G, 128, 64, 4, 4) / 32, _rand(G, 128)
    _close(conv_hip([x], [w], [b], stride=2, pad=(1, 1), act=1, pad_mode=1),
           F.relu(F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w, b, stride=2)))
    # Linear(480,256) as a 3x5 "valid" conv over the NHWC map (utils.py:1020)
    x = _rand(G, 4, 32, 3, 5)
    wl, bl = _rand(G, 256, 480) / 22, _rand(G, 256)
    got = conv_hip([x], [wl], [bl], kernel_hw=(3, 5))
    _close(got.reshape(4, 256), F.linear(x.reshape(4, -1), wl, bl))
    # eval-mode BatchNorm folded behind the activation (utils.py:1047-1049)
    x = _rand(G, 2, 32, 12, 16)
    w, b = _rand(G, 3, 32, 3, 3) / 17, _rand(G, 3)
    ps, pt = torch.rand(3, generator=G) + 0.5, _rand(G, 3)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1)) * ps[None, :, None, None] + pt[None, :, None, None]
    _close(conv_hip([x], [w], [b], pad=(1, 1), act=2, post=(ps, pt)), ref)


def test_norm_pool_upsample(G):
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import Piece, Plan
    B, C, H, W = 3, 38, 30, 40
    x = _rand(G, B, C, H, W) * 3 + 0.5
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    sc, sh, _, _ = pl.norm_stats(px, B, H * W)
    bsc, bsh, bm, bv = pl.norm_stats(px, B, H * W, per_sample=False, want_moments=True)
    ap = pl.buf(B, H // 2, W // 2, px.Cp)
    pl.avgpool2(px, Piece(ap, 0, C), B, H, W)
    mp2 = pl.buf(B, 15, 20, px.Cp)
    pl.maxpool2(px, Piece(mp2, 0, C), B, H, W, 2)
    mp1 = pl.buf(B, 29, 39, px.Cp)
    pl.maxpool2(px, Piece(mp1, 0, C), B, H, W, 1)
    up = pl.buf(B, 2 * H, 2 * W, px.Cp)
    pl.upsample2x(px, Piece(up, 0, C), B, H, W)
    pl.run()
    torch.cuda.synchronize()
    nchw = lambda t: t.cpu()[..., :C].permute(0, 3, 1, 2)  # noqa: E731
    mean, var = x.mean((2, 3)), x.var((2, 3), unbiased=False)
    np.testing.assert_allclose(sc.cpu()[:, :C], 1 / torch.sqrt(var + 1e-5), rtol=1e-5)
    np.testing.assert_allclose(sh.cpu()[:, :C], -mean / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bm.cpu()[0, :C], x.mean((0, 2, 3)), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bv.cpu()[0, :C], x.var((0, 2, 3), unbiased=False), rtol=1e-5)
    np.testing.assert_allclose(nchw(ap), F.avg_pool2d(x, 2), rtol=1e-6, atol=1e-6)
    assert torch.equal(nchw(mp2), F.max_pool2d(x, 2, 2, ceil_mode=True))
    assert torch.equal(nchw(mp1), F.max_pool2d(x, 2, 1, ceil_mode=True))
    np.testing.assert_allclose(nchw(up), F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False),
                               rtol=1e-5, atol=1e-6)
    # odd sizes: ceil_mode windows hanging over the border (vgg16_c.py pools at 25 -> 13)
    x2 = _rand(G, 1, 8, 25, 13)
    pl2 = Plan(torch.device(DEV))
    (p2,) = to_nhwc_buf(pl2, [x2], 1, 25, 13)
    o2 = pl2.buf(1, 13, 7, 8)
    pl2.maxpool2(p2, Piece(o2, 0, 8), 1, 25, 13, 2)
    pl2.run()
    assert torch.equal(o2.cpu().permute(0, 3, 1, 2), F.max_pool2d(x2, 2, 2, ceil_mode=True))


@pytest.mark.parametrize("H,W,absent", [(48, 64, "none"), (240, 320, "some"), (30, 40, "all")])
def test_loss_head_vs_oracle(G, H, W, absent):
    import ctypes as C
    from gpu_util import DEV
    from egne_amd import _lib
    from oracle import losses as oloss
    B = 5
    op = 2 * _rand(G, B, 3, H, W)
    tgt = torch.randint(0, 3, (B, H, W), generator=G)
    tgt[1][tgt[1] == 2] = 1           # a sample without pupil pixels (one class absent)
    sw = 1 + 20 * (torch.rand(B, H, W, generator=G) > 0.9).float()
    dist = _rand(G, B, 3, H, W)
    pc = torch.rand(B, 2, generator=G) * torch.tensor([W, H])
    eln = torch.rand(B, 2, 5, generator=G) * 2 - 1
    elOut = torch.rand(B, 10, generator=G) * 2 - 1
    cond = torch.zeros(B, 4)
    if absent == "some":
        cond[2, 1:] = 1
        cond[4, 1:] = 1
    elif absent == "all":
        cond[:, 1:] = 1
    alpha = 0.3
    total, pred_c, terms = oloss.all_loss(op, elOut, tgt, pc, eln, sw, dist, cond, alpha)
    L = _lib.lib()
    d = _lib.LossDesc()
    t = {k: v.to(DEV).contiguous() for k, v in dict(tgt=tgt, sw=sw, dist=dist, pc=pc, eln=eln, elOut=elOut, cond=cond).items()}
    logits = torch.zeros(B, H, W, 8, device=DEV)
    logits[..., :3] = op.permute(0, 2, 3, 1).to(DEV)
    part = torch.zeros(int(L.egne_loss_workspace_floats(B, H, W)), device=DEV)
    out_terms, pcd, elp = torch.zeros(8, device=DEV), torch.zeros(B, 2, 2, device=DEV), torch.zeros(B, 10, device=DEV)
    mask = torch.zeros(B, H, W, dtype=torch.int64, device=DEV)
    opn = torch.zeros(B, 3, H, W, device=DEV)
    gx, gy = torch.linspace(-1, 1, W).to(DEV), torch.linspace(-1, 1, H).to(DEV)
    d.B, d.H, d.W = B, H, W
    d.logits, d.pix_stride, d.ch_off = logits.data_ptr(), 8, 0
    d.target, d.spatWts, d.distMap, d.cond = t["tgt"].data_ptr(), t["sw"].data_ptr(), t["dist"].data_ptr(), t["cond"].data_ptr()
    d.pupil_center, d.elNorm, d.elOut, d.alpha = t["pc"].data_ptr(), t["eln"].data_ptr(), t["elOut"].data_ptr(), alpha
    d.grid_x, d.grid_y = gx.data_ptr(), gy.data_ptr()
    d.partials, d.out_terms, d.pred_c, d.elPred = part.data_ptr(), out_terms.data_ptr(), pcd.data_ptr(), elp.data_ptr()
    d.mask, d.op_nchw = mask.data_ptr(), opn.data_ptr()
    _lib.check(L.egne_loss_fwd(C.byref(d), _lib.stream_ptr()), "loss")
    torch.cuda.synchronize()
    ot = out_terms.cpu().numpy()
    np.testing.assert_allclose(ot[0], float(total), rtol=2e-5)
    for i, k in enumerate(["l_seg2pt", "l_seg", "l_pt", "l_ellipse"]):
        np.testing.assert_allclose(ot[1 + i], float(terms[k]), rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(pcd.cpu().numpy(), pred_c.numpy(), atol=2e-6)
    assert torch.equal(mask.cpu(), op.max(1)[1]), "argmax mask must be identical (first max on ties)"
    assert torch.equal(opn.cpu(), op)
    ref_elp = torch.cat([pred_c[:, 0], elOut[:, 2:5], pred_c[:, 1], elOut[:, 7:10]], 1)
    np.testing.assert_allclose(elp.cpu().numpy(), ref_elp.numpy(), atol=2e-6)


def test_loss_two_absent_classes_flag(G):
    """wCE raises in the reference when two classes are absent (loss.py:132); the device path cannot
    raise without a sync, it reports the sample count in out_terms[6] instead."""
    import ctypes as C
    from gpu_util import DEV
    from egne_amd import _lib
    B, H, W = 1, 16, 16
    L = _lib.lib()
    d = _lib.LossDesc()
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
    logits, tgt = z(B, H, W, 8), z(B, H, W, dt=torch.int64)
    bufs = [z(B, H, W), z(B, 3, H, W), z(B, 4), z(B, 2), z(B, 2, 5), z(B, 10), z(int(L.egne_loss_workspace_floats(B, H, W))),
            z(8), z(B, 2, 2), z(B, 10)]
    d.B, d.H, d.W, d.logits, d.pix_stride, d.ch_off, d.target = B, H, W, logits.data_ptr(), 8, 0, tgt.data_ptr()
    (d.spatWts, d.distMap, d.cond, d.pupil_center, d.elNorm, d.elOut, d.partials, d.out_terms, d.pred_c,
     d.elPred) = [b.data_ptr() for b in bufs]
    _lib.check(L.egne_loss_fwd(C.byref(d), _lib.stream_ptr()), "loss")
    torch.cuda.synchronize()
    assert bufs[7][6].item() == 1.0


def test_fit_bit_exact_vs_golden(G):
    """Device ellipse fit == reference search (fixtures from utils.py:450-486 run on CPU)."""
    from common import gold
    from gpu_util import DEV
    from egne_amd.utils import fit_ellipses, search_proper_parameter_iou_for_our_data
    g = gold("fit_cases")
    H, W = 240, 320
    n = len(g["masks"])
    masks = np.stack([np.unpackbits(m).reshape(H, W) for m in g["masks"]]).astype(np.int64)
    out, ev = fit_ellipses(torch.from_numpy(masks).to(DEV), list(range(n)), [1] * n, g["inits"], return_evals=True)
    bad = [i for i in range(n) if not np.array_equal(out[i], g["outs"][i])]
    assert not bad, "fit differs from the reference for cases %s" % bad
    one = search_proper_parameter_iou_for_our_data(torch.from_numpy(masks[3] == 1).to(DEV), g["inits"][3])
    np.testing.assert_array_equal(one, g["outs"][3])


def test_fit_large_batch_form_is_bit_exact_too(G):
    """n >= 64 searches take the dense form of the fit kernel: four searches per workgroup whose wave pairs meet through LDS flags of
    their own instead of the workgroup barrier (fit.hip, LOCAL).  The 24 golden cases in a shuffled batch of 96 (every case four times,
    neighbours of different length in one workgroup) must reproduce the reference bit for bit, and the evaluation counts of the
    small-batch form."""
    from common import gold
    from gpu_util import DEV
    from egne_amd.utils import fit_ellipses
    g = gold("fit_cases")
    H, W = 240, 320
    n = len(g["masks"])
    masks = torch.from_numpy(np.stack([np.unpackbits(m).reshape(H, W) for m in g["masks"]]).astype(np.int64)).to(DEV)
    _, ev_small = fit_ellipses(masks, list(range(n)), [1] * n, g["inits"], return_evals=True)
    order = np.random.RandomState(3).permutation(np.tile(np.arange(n), 4))
    out, ev = fit_ellipses(masks, order.tolist(), [1] * len(order), g["inits"][order], return_evals=True)
    assert len(order) >= 64
    bad = [int(i) for i, c in enumerate(order) if not np.array_equal(out[i], g["outs"][c])]
    assert not bad, "dense batch form differs from the reference at positions %s" % bad[:8]
    np.testing.assert_array_equal(ev, ev_small[order])


def test_fit_large_masks_in_a_large_batch(G):
    """384x512 class maps: two searches fit the default 64 KB of dynamic LDS, four do not, so a batch of n >= 64 searches must fall
    back from the four-per-workgroup form to the two-per-workgroup one (fit.hip, egne_ellipse_fit) instead of failing the launch.
    Four rendered ellipses, each sixteen times in a shuffled batch, against the oracle's search (utils.py:450-486 restated)."""
    from gpu_util import DEV
    from egne_amd.utils import fit_ellipses
    from oracle import fit as ofit
    H, W = 384, 512
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    cases = [(250.0, 190.0, 90.0, 60.0, 0.3), (200.0, 150.0, 40.0, 55.0, -0.5), (300.0, 220.0, 120.0, 80.0, 1.0), (256.0, 192.0, 30.0, 30.0, 0.0)]
    masks, inits, want = [], [], []
    for cx, cy, a, b, t in cases:
        X = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
        Y = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
        m = ((X / a) ** 2 + (Y / b) ** 2 <= 1.0)
        init = np.array([cx, cy, a * 0.9 + 2.0, b * 1.1 - 1.5, t + 0.07])
        masks.append(m.astype(np.int64))
        inits.append(init)
        want.append(ofit.fit_ellipse(m, list(init)))
    order = np.random.RandomState(5).permutation(np.tile(np.arange(4), 16))
    assert len(order) >= 64
    out = fit_ellipses(torch.from_numpy(np.stack(masks)).to(DEV), order.tolist(), [1] * len(order), np.stack(inits)[order])
    bad = [int(i) for i, c in enumerate(order) if not np.array_equal(out[i], want[c])]
    assert not bad, "fit of 384x512 masks differs from the oracle at positions %s" % bad[:8]


def test_errors_are_reported(G):
    """Host-side validation returns an error code + message instead of launching."""
    import ctypes as C
    from egne_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc()
    rc = L.egne_conv2d_fwd(C.byref(d), None)
    assert rc == -1 and b"conv" in L.egne_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "conv")


@pytest.mark.parametrize("B,Cin,Cout,H,W,d", [(2, 64, 64, 33, 47, 1), (1, 128, 256, 20, 24, 1), (1, 512, 512, 9, 13, 2),
                                              (2, 64, 128, 128, 131, 1), (1, 96, 256, 150, 223, 2), (3, 128, 384, 101, 110, 1)])
def test_conv_f16x3_split_precision(G, B, Cin, Cout, H, W, d):
    """Split-f16 MFMA convolution (conv_f16x3.hip): error against a float64 convolution must be at the fp32
    level (the fp32 CPU conv is measured against the same truth for comparison)."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = F.relu(_rand(G, B, Cin, H, W)) * 3          # post-ReLU activations, O(1-10)
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    truth = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=d, dilation=d))
    ref32 = F.relu(F.conv2d(x, w, b, padding=d, dilation=d))
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1),
                      dils=(d,), act=1)
    layer.split = True
    out = pl.buf(B, H, W, Cout)
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
    assert pl.meta[-1][0].startswith("conv_f16x3")
    if B * H * W >= 256 * 128 and Cout >= 256:      # the last two cases run the deep 256-wide kernel (ragged M, 128- and 256-wide N tiles)
        assert pl.calls[-1][0] is pl.L.egne_conv2d_f16x3_big_fwd
    pl.run()
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    scale = truth.abs().max().item()
    e_split, e_f32 = (got - truth).abs().max().item() / scale, (ref32.double() - truth).abs().max().item() / scale
    print("f16x3 err %.2e  (fp32 CPU conv err %.2e)" % (e_split, e_f32))
    assert e_split < 2e-6


@pytest.mark.parametrize("B,Cin,Cout,H,W,d,k,extras", [
    (2, 512, 512, 30, 40, 1, 3, ""),                 # VGG conv4 at two frames: 64 x 64 tiles, K split 3 ways
    (1, 512, 32, 30, 40, 1, 3, "res+post"),          # MSBlock conv at one frame: 128 x 32 tiles, deep split; residual and post affine
    (2, 256, 21, 15, 20, 2, 3, "norm"),              # odd Cout (21 of 32 stored), dilation 2, normalisation + LeakyReLU fused into the load
    (1, 136, 96, 29, 39, 1, 3, "leaky"),             # K tail (136 = 4 * 32 + 8), ragged M
    (2, 64, 64, 33, 47, 1, 3, ""),                   # short K loop: small tiles, no split
    (2, 320, 128, 15, 20, 1, 5, ""),                 # 5x5 taps
])
def test_conv_f16x3_small_problem_form(G, B, Cin, Cout, H, W, d, k, extras):
    """Small-problem form of the flat split-f16 kernel (conv_f16x3.hip small_plan: 64-wide tiles, K range split over gridDim.z, partial
    sums through a workspace, second launch for bias / activation / post affine / residual) against a float64 convolution at the fp32
    level, and bit-compatible in its epilogue features with the standard launch (same result up to the summation order)."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd import engine
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, Cin, H, W) * 2
    w, b = _rand(G, Cout, Cin, k, k) / (k * Cin ** 0.5), _rand(G, Cout)
    act = 2 if "leaky" in extras or "norm" in extras else 1
    xin = x
    sc = sh = None
    if "norm" in extras:
        sc, sh = 0.5 + torch.rand(B, Cin, generator=G), 0.2 * _rand(G, B, Cin)
        xin = F.leaky_relu(x * sc[:, :, None, None] + sh[:, :, None, None])
    pd = d * (k // 2)
    y = F.conv2d(xin.double(), w.double(), b.double(), padding=pd, dilation=d)
    y = F.relu(y) if act == 1 else F.leaky_relu(y)
    res = ps = pt = None
    if "post" in extras:
        ps, pt = 0.5 + torch.rand(Cout, generator=G), _rand(G, Cout)
        y = y * ps.double()[None, :, None, None] + pt.double()[None, :, None, None]
    if "res" in extras:
        res = _rand(G, B, Cout, H, W)
        y = y + res.double()
    outs = []
    for small in (True, False):
        engine.SMALL_ENABLED = small
        try:
            pl = Plan(torch.device(DEV))
            (px,) = to_nhwc_buf(pl, [x], B, H, W)
            if sc is not None:
                scp, shp = torch.zeros(B, px.Cp, device=DEV), torch.zeros(B, px.Cp, device=DEV)
                scp[:, :Cin], shp[:, :Cin] = sc.to(DEV), sh.to(DEV)
                pl.keep += [scp, shp]
                px = px.with_norm(scp, shp, 2)
            layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(k // 2, k // 2),
                              dils=(d,), act=act)
            layer.split = True
            if ps is not None:
                a, c = torch.zeros(layer.CoutP, device=DEV), torch.zeros(layer.CoutP, device=DEV)
                a[:Cout], c[:Cout] = ps.to(DEV), pt.to(DEV)
                layer.post = (a, c)
            rp = to_nhwc_buf(pl, [res], B, H, W)[0] if res is not None else None
            out = pl.buf(B, H, W, pad8(Cout) + 8)
            out.fill_(777.0)
            pl.conv(layer, [px], Piece(out, 8, Cout), B, H, W, residual=rp)
            kind = pl.meta[-1][0]
            assert kind == ("conv_f16x3:small" if small else kind) and kind.startswith("conv_f16x3"), kind
            pl.run()
            pl.run()                                   # the workspace is reused: a second run must give the same answer
            torch.cuda.synchronize()
            o = out.cpu()
            assert (o[..., :8] == 777.0).all() and (o[..., 8 + pad8(Cout):] == 777.0).all(), "wrote outside the output slice"
            outs.append(o[..., 8:8 + Cout].permute(0, 3, 1, 2).double())
        finally:
            engine.SMALL_ENABLED = True
    scale = y.abs().max().item()
    e_small, e_std = (outs[0] - y).abs().max().item() / scale, (outs[1] - y).abs().max().item() / scale
    print("small form err %.2e, standard launch %.2e" % (e_small, e_std))
    assert e_small < 2e-6 and (outs[0] - outs[1]).abs().max().item() / scale < 2e-6


def test_conv_f16x3_grouped_msblock(G):
    """Split-f16 kernel, 256x32 tile, fused dilated group (bdcn_new.py:49-55) against float64."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan
    B, H, W = 2, 37, 53
    o = F.relu(_rand(G, B, 32, H, W)) * 2
    ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
    bs = [_rand(G, 32) for _ in range(3)]
    truth = o.double()
    for w, b, d in zip(ws, bs, (4, 8, 12)):
        truth = truth + F.relu(F.conv2d(o.double(), w.double(), b.double(), padding=d, dilation=d))
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [o], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV)) for w in ws], [torch.nn.Parameter(b.to(DEV)) for b in bs], [(32, 32)],
                      pad=(1, 1), dils=(4, 8, 12), act=1)
    layer.split = True
    out = pl.buf(B, H, W, 32)
    pl.conv(layer, [px], Piece(out, 0, 32), B, H, W, residual=px)
    assert pl.meta[-1][0].startswith("conv_f16x3")
    pl.run()
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    err = (got - truth).abs().max().item() / truth.abs().max().item()
    print("grouped f16x3 err %.2e" % err)
    assert err < 2e-6


@pytest.mark.parametrize("chans,Cout,B,H,W,act", [((32, 32), 32, 2, 37, 53, 0), ((38, 64, 24), 64, 1, 61, 35, 2),
                                                  ((115, 8, 40), 100, 1, 19, 23, 1), ((32, 32, 32), 30, 3, 16, 16, 0),
                                                  ((76, 96), 96, 2, 30, 41, 0), ((306, 128, 115), 180, 1, 30, 40, 2),
                                                  ((100, 64, 38, 62), 62, 2, 33, 17, 1)])
def test_conv1x1_streaming_split(G, chans, Cout, B, H, W, act):
    """Streaming split-f16 1x1 kernel (conv1x1_f16.hip) over several slices of one buffer and of a second buffer
    (ragged slice widths: 8-channel tail groups, padded channels, pixel count not a multiple of 32)."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd import engine
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    xs = [_rand(G, B, c, H, W) * 2 for c in chans]
    Cin = sum(chans)
    w, b = _rand(G, Cout, Cin, 1, 1) / Cin ** 0.5, _rand(G, Cout)
    t = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double())
    truth = F.relu(t) if act == 1 else (F.leaky_relu(t, 0.01) if act == 2 else t)
    pl = Plan(torch.device(DEV))
    pieces = to_nhwc_buf(pl, xs[:-1], B, H, W) + to_nhwc_buf(pl, xs[-1:], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(p.C, p.Cp) for p in pieces], act=act)
    layer.split1 = True
    out = pl.buf(B, H, W, pad8(Cout) + 16)
    out.fill_(777.0)
    old = engine.S1X1_MIN_PIX, engine.MS1X1_MIN_PIX
    engine.S1X1_MIN_PIX = engine.MS1X1_MIN_PIX = 0
    try:
        pl.conv(layer, pieces, Piece(out, 8, Cout), B, H, W)
    finally:
        engine.S1X1_MIN_PIX, engine.MS1X1_MIN_PIX = old
    # Cout 96 / K 549 do not fit the streaming kernel's LDS weight image: LDS-staged multi-slice GEMM (conv1x1_ms_f16.hip)
    staged = Cout in (96, 180)
    assert pl.calls[-1][0] is (pl.L.egne_conv1x1_ms_f16x3_fwd if staged else pl.L.egne_conv1x1_f16x3_fwd)
    pl.run()
    torch.cuda.synchronize()
    o = out.cpu()
    assert (o[..., :8] == 777.0).all() and (o[..., 8 + pad8(Cout):] == 777.0).all(), "wrote outside the output slice"
    got = o[..., 8:8 + Cout].permute(0, 3, 1, 2).double()
    err = (got - truth).abs().max().item() / truth.abs().max().item()
    print("conv1x1 streaming err %.2e" % err)
    assert err < 2e-6


@pytest.mark.parametrize("B,Cin,Cout,H,W,norm", [(2, 32, 32, 16, 64, False), (2, 40, 24, 37, 53, True), (1, 64, 70, 9, 33, False)])
def test_conv3x3_backward_halo_wgrad(G, B, Cin, Cout, H, W, norm):
    """Backward of a 3x3 conv through the training plan (act_bwd + halo weight-gradient kernel wgrad_halo.hip +
    dgrad) against torch autograd: ragged tiles, channel tails, fused InstanceNorm affine + LeakyReLU on the input."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, Cin, H, W)
    w = (_rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5)).requires_grad_(True)
    b = _rand(G, Cout).requires_grad_(True)
    gy = _rand(G, B, Cout, H, W)
    sc, sh = 0.5 + _rand(G, B, Cin).abs(), _rand(G, B, Cin)
    xin = x.clone().requires_grad_(True)
    xe = F.leaky_relu(xin * sc[:, :, None, None] + sh[:, :, None, None], 0.01) if norm else xin
    y = F.conv2d(xe, w, b, padding=1)
    y.backward(gy)
    pl = Plan(torch.device(DEV), train=True)
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    if norm:
        scp, shp = torch.zeros(B, px.Cp, device=DEV), torch.zeros(B, px.Cp, device=DEV)
        scp[:, :Cin], shp[:, :Cin] = sc.to(DEV), sh.to(DEV)
        pl.keep += [scp, shp]
        px = px.with_norm(scp, shp, 2)
        px.nograd = True     # the norm backward needs its statistics; only the weight path is under test here
    wd, bd = torch.nn.Parameter(w.detach().to(DEV)), torch.nn.Parameter(b.detach().to(DEV))
    wd.grad, bd.grad = torch.zeros_like(wd), torch.zeros_like(bd)
    layer = ConvLayer([wd], [bd], [(Cin, pad8(Cin))], pad=(1, 1))
    out = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
    bw = pl.build_backward()
    pl.run()
    pl.gbuf(out)[..., :Cout] = gy.permute(0, 2, 3, 1).to(DEV)
    bw.run()
    torch.cuda.synchronize()
    ew = (wd.grad.cpu() - w.grad).abs().max().item() / w.grad.abs().max().item()
    eb = (bd.grad.cpu() - b.grad).abs().max().item() / b.grad.abs().max().item()
    print("wgrad err %.2e  bias grad err %.2e" % (ew, eb))
    assert ew < 2e-5 and eb < 2e-5
    if not norm:
        gx = pl.gbuf(px.buf).cpu()[..., :Cin].permute(0, 3, 1, 2)
        ex = (gx - xin.grad).abs().max().item() / xin.grad.abs().max().item()
        assert ex < 2e-5


def test_msblock_lattice_groups_wide_and_tall_tiles(G):
    """Dilated group of an MSBlock (bdcn_new.py:49-55) on a 240x320 frame: three lattice-halo launches (dilation 4 / 8 / 12
    as 16 / 64 / 144 ordinary 3x3 convs on sub-lattices) accumulating through the residual; the dilation-4 and -8 lattices
    (60x80, 30x40 points) are walked transposed ("tall" tiles, transposed 3x3 taps), dilation 12 (20x27) is not."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan
    B, H, W = 1, 240, 320
    o = F.relu(_rand(G, B, 32, H, W)) * 2
    ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
    bs = [_rand(G, 32) for _ in range(3)]
    truth = o.double()
    for w, b, d in zip(ws, bs, (4, 8, 12)):
        truth = truth + F.relu(F.conv2d(o.double(), w.double(), b.double(), padding=d, dilation=d))
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [o], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV)) for w in ws], [torch.nn.Parameter(b.to(DEV)) for b in bs], [(32, 32)],
                      pad=(1, 1), dils=(4, 8, 12), act=1)
    layer.split = True
    out = pl.buf(B, H, W, 32)
    from egne_amd import engine
    old = engine.MSDIL_ENABLED
    engine.MSDIL_ENABLED = False          # round 1's path, still what other dilation sets take
    try:
        pl.conv(layer, [px], Piece(out, 0, 32), B, H, W, residual=px)
    finally:
        engine.MSDIL_ENABLED = old
    assert len(pl.calls) == 3 and all(c[0] is pl.L.egne_conv3x3_halo_f16_fwd for c in pl.calls)
    pl.run()
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    err = (got - truth).abs().max().item() / truth.abs().max().item()
    print("lattice group err %.2e" % err)
    assert err < 2e-6


@pytest.mark.parametrize("Cin,Cout,B,H,W,post", [(3, 64, 2, 37, 53, False), (1, 32, 3, 16, 48, True), (2, 30, 1, 41, 33, True)])
def test_first_layer_streaming_split(G, Cin, Cout, B, H, W, post):
    """Streaming split-f16 first-layer kernel (conv3x3_c4_f16.hip: 9 taps folded into K = 48, operands straight from HBM)
    against a float64 convolution; with the folded eval-mode BatchNorm (post affine) of ESF-Net's head."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, Cin, H, W) * 2
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    truth = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.01)
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
    layer.split = True
    if post:
        ps, pt = 0.5 + _rand(G, Cout).abs(), _rand(G, Cout)
        truth = truth * ps.double()[None, :, None, None] + pt.double()[None, :, None, None]
        psd, ptd = torch.zeros(layer.CoutP, device=DEV), torch.zeros(layer.CoutP, device=DEV)
        psd[:Cout], ptd[:Cout] = ps.to(DEV), pt.to(DEV)
        layer.post = (psd, ptd)
    out = pl.buf(B, H, W, pad8(Cout) + 16)
    out.fill_(777.0)
    from egne_amd import engine
    old_mode, engine.C4H_MODE = engine.C4H_MODE, "all"
    try:
        pl.conv(layer, [px], Piece(out, 8, Cout), B, H, W)
    finally:
        engine.C4H_MODE = old_mode
    assert pl.calls[-1][0] is pl.L.egne_conv3x3_smallcin_f16_fwd
    pl.run()
    torch.cuda.synchronize()
    o = out.cpu()
    assert (o[..., :8] == 777.0).all() and (o[..., 8 + pad8(Cout):] == 777.0).all(), "wrote outside the output slice"
    got = o[..., 8:8 + Cout].permute(0, 3, 1, 2).double()
    err = (got - truth).abs().max().item() / truth.abs().max().item()
    print("first-layer streaming err %.2e" % err)
    assert err < 2e-6


def test_dataprep_dist_maps_bit_exact_and_zscore(G):
    """Device-side batch preparation (dataprep.hip) against the fixture produced by the reference's own
    helperfunctions.one_hot2dist: the exact EDT must agree bit for bit (classes absent, a class filling the frame, thin
    structures, a single-pixel class); z-score within one float32 ulp."""
    from common import gold
    from egne_amd import dataprep
    g = gold("dataprep")
    lab = torch.from_numpy(g["label"].astype(np.int64)).cuda()
    got = dataprep.dist_maps(lab).cpu().numpy()
    assert np.array_equal(got, g["dist"]), "dist maps differ in %d values" % (got != g["dist"]).sum()
    z = dataprep.zscore(torch.from_numpy(g["img"].astype(np.float32)).cuda()).cpu().numpy()
    np.testing.assert_allclose(z, g["z"], rtol=2e-7, atol=1e-7)
    # throughput at the bench batch (printed, not asserted)
    big = lab[:1].repeat(64, 1, 1)
    dataprep.dist_maps(big)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dataprep.dist_maps(big)
    e1.record()
    torch.cuda.synchronize()
    print("dist_maps: %.0f frames/s" % (64 * 5 / (e0.elapsed_time(e1) * 1e-3)))


def test_augment_batch_vs_reference(G):
    """Batched device augmentation (dataprep.hip egne_augment, egne_amd.data_augment) against the reference's own outputs
    (tests/golden/augment.npz: flip / exposure / noise / none, branch given or drawn) bit for bit, one frame per launch as the
    reference does it and all frames in ONE launch; the gamma branch against the oracle (its table is the reference's, the look-up
    is unpinned); OpenCV branches are refused."""
    import hashlib
    from test_oracle_golden import _augment_cases
    from egne_amd import data_augment as DA
    from oracle import data_augment as oaug
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    cases = list(_augment_cases())
    for n, choice, npseed, (base, mask, pc, el), exp, _ in cases:
        np.random.seed(npseed)
        ob, om, opc, (op_, oi) = DA.augment(base, mask, pc, el, choice if choice >= 0 else None)
        assert ob.dtype == np.uint8 and sha(ob) == exp["img_sha"], "case %d: %d pixels differ in the sampled rows" % (n, (ob[::16] != exp["rows"]).sum())
        assert sha(om.astype(np.int64)) == exp["mask_sha"], "case %d mask" % n
        assert np.array_equal(opc, exp["pc"]) and np.array_equal(np.stack([op_, oi]), exp["el"]), "case %d geometry" % n
    # one launch over a batch with mixed branches (explicit choices, host-drawn noise in the reference's order per frame)
    ex = [c for c in cases if c[1] >= 0]
    img = torch.from_numpy(np.stack([c[3][0] for c in ex])).cuda()
    lab = torch.from_numpy(np.stack([c[3][1] for c in ex])).cuda()
    pcs = torch.from_numpy(np.stack([c[3][2] for c in ex]))
    els = torch.from_numpy(np.stack([c[3][3] for c in ex]))
    np.random.seed(11)
    oi_, ol_, pc_, el_, ch_ = DA.augment_batch(img, lab, pcs, els, choices=[c[1] for c in ex], host_noise=True)
    np.random.seed(11)
    for k, c in enumerate(ex):
        wb, wm, wpc, (wp, wi) = oaug.augment(*c[3], c[1])
        assert np.array_equal(oi_[k].cpu().numpy(), wb) and np.array_equal(ol_[k].cpu().numpy(), wm), "batched frame %d" % k
        assert np.array_equal(pc_[k].numpy(), wpc) and np.array_equal(el_[k].numpy(), np.stack([wp, wi]))
    # gamma: table of the reference, look-up on the device
    base, mask, pc, el = cases[0][3]
    for s in range(4):
        np.random.seed(40 + s)
        got = DA.augment(base, mask, pc, el, 2)
        np.random.seed(40 + s)
        want = oaug.augment(base, mask, pc, el, 2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # device-drawn noise: right statistics (mean 0, the drawn standard deviation), untouched label
    flat = torch.full((4, 240, 320), 128, dtype=torch.uint8).cuda()
    np.random.seed(3)
    o, l2, _, _, _ = DA.augment_batch(flat, lab[:1].repeat(4, 1, 1), pcs[:4], els[:4], choices=[4] * 4)
    np.random.seed(3)
    stds = [14 * np.random.rand() + 2 for _ in range(4)]
    d = o.double() - 128.0
    for k in range(4):
        assert abs(d[k].mean().item() + 0.5) < 0.2 and abs(d[k].std().item() / stds[k] - 1) < 0.05    # truncation shifts the mean by -0.5
    assert torch.equal(l2, lab[:1].repeat(4, 1, 1))
    for c in (1, 5, 6):
        with pytest.raises(NotImplementedError):
            DA.augment(base, mask, pc, el, c)


def test_deep_trunk_kernel_with_frame_tail(G):
    """conv4-like layer at a batch where the 256x256 launch is cut to whole rounds of workgroups and the remaining frames go to
    the 128x128 kernel (engine.BIG_SPLIT_TAIL): both launches together must equal the convolution of the whole batch
    (reference here: torch's own fp32 convolution on the GPU, the split kernels' error bound is 2e-6 of the output scale)."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan
    B, Cin, Cout, H, W = 40, 256, 512, 30, 40
    x = F.relu(_rand(G, B, Cin, H, W))
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, Cin)], pad=(1, 1), act=1)
    layer.split = True
    out = pl.buf(B, H, W, Cout)
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
    assert [c[0] for c in pl.calls] == [pl.L.egne_conv2d_f16x3_big_fwd, pl.L.egne_conv2d_f16x3_fwd], "expected the big + tail launches"
    pl.run()
    ref = F.relu(F.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), padding=1))
    torch.cuda.synchronize()
    err = (out.permute(0, 3, 1, 2) - ref).abs().max().item() / ref.abs().max().item()
    print("big + tail err vs torch fp32 %.2e" % err)
    assert err < 1e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W,d", [
    (28, 128, 256, 30, 40, 1),     # conv3_1 class: four chunks, 36 K steps (the planner takes the deep kernel from 32 768 pixels on)
    (47, 256, 512, 23, 31, 1),     # conv4_1 class, ragged last tile (33 511 pixels), two output tiles
    (110, 512, 512, 15, 20, 2),    # conv5 class: dilation 2, 144 K steps
    (102, 256, 384, 17, 19, 1),    # 128-wide output tiles (Cout % 256 != 0), ragged last tile
])
def test_deep_trunk_kernel_plain_f16_four_stage_form(G, B, Cin, Cout, H, W, d):
    """egne_conv2d_f16_big1_fwd (the frozen edge network's wide layers next to a bf16-storage training plan, vgg16_c.py:70-88): same
    operands, same accumulation order as egne_conv2d_f16x3_big_fwd with f16_products = 1 -> BIT-identical; and against float64 on
    the f16-rounded operands (what "plain f16 products, fp32 accumulate" means) to 4e-6 of the output scale."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd import engine
    from egne_amd.engine import ConvLayer, Piece, Plan
    x = F.relu(_rand(G, B, Cin, H, W))
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    outs = []
    for big1 in (True, False):
        old = engine.BIG1_ENABLED
        engine.BIG1_ENABLED = big1
        try:
            pl = Plan(torch.device(DEV))
            pl.f16_products = 1
            (px,) = to_nhwc_buf(pl, [x], B, H, W)
            layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, Cin)], pad=(1, 1), dils=(d,), act=1)      # (padding counts taps)
            layer.split = True
            out = pl.buf(B, H, W, Cout + 8)
            out.fill_(777.0)
            pl.conv(layer, [px], Piece(out, 8, Cout), B, H, W)
            want = pl.L.egne_conv2d_f16_big1_fwd if big1 else pl.L.egne_conv2d_f16x3_big_fwd
            assert pl.calls[0][0] == want, "expected the %s launch" % ("four-stage" if big1 else "two-stage")
            pl.run()
            pl.run()
            torch.cuda.synchronize()
            assert (out[..., :8] == 777.0).all(), "wrote outside its output slice"
            outs.append((out[..., 8:].clone(), pl.calls[0][1][2], layer.w_scale_big))
        finally:
            engine.BIG1_ENABLED = old
    (o1, a_s, w_s), (o0, _, _) = outs
    assert torch.equal(o1, o0), "four-stage and two-stage forms differ: max %.3e" % (o1 - o0).abs().max().item()
    xh = (x.to(DEV) * a_s).half().double() / a_s
    wh = (w.to(DEV) * w_s).half().double() / w_s
    ref = F.relu(F.conv2d(xh, wh, b.to(DEV).double(), padding=d, dilation=d))
    err = (o1.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    print("big1 vs float64 on f16 operands %.2e" % err)
    assert err < 4e-6          # (fp32 accumulation over up to 4 608 products: measured 4e-7 .. 7.4e-7)


@pytest.mark.parametrize("B,Cin,Cout,H,W,d", [
    (47, 256, 512, 23, 31, 1),     # ragged last tile (33 511 pixels), two output tiles
    (110, 512, 512, 15, 20, 2),    # dilation 2, 144 K steps
    (102, 256, 384, 17, 19, 1),    # 128-wide output tiles, ragged last tile
])
def test_deep_trunk_kernel_f16_storage_both_operands_by_dma(G, B, Cin, Cout, H, W, d):
    """conv_f16_big1_h_kernel (egne_seg.presplit = 2, egne_conv_desc.out_split = 2; the frozen edge network's conv3_2 .. conv5_3 in a
    plain-f16 plan, vgg16_c.py:72-88 under utils.py:646): input held as f16 halves of x s, activations staged by LDS-DMA like the weights,
    output stored as halves -- against the fp32-tensor form of the same kernel on the de-quantised input: the stored halves must be
    exactly f16(out s_out) of that result, with the output's scale calibrated to put its maximum in [1024, 2048), and nothing outside
    the slice is touched."""
    from gpu_util import DEV
    from egne_amd.engine import ConvLayer, Piece, Plan, SplitScale
    s_in = 64.0
    xh = (F.relu(_rand(G, B, Cin, H, W)) * s_in).half()                 # the tensor AS STORED
    x = xh.float() / s_in
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, Cin)], pad=(1, 1), dils=(d,), act=1)
    layer.split = True
    # fp32 tensors in and out
    pl0 = Plan(torch.device(DEV))
    pl0.f16_products = 1
    xb = pl0.buf(B, H, W, Cin)
    xb.copy_(x.permute(0, 2, 3, 1).to(DEV))
    o0 = pl0.buf(B, H, W, Cout)
    pl0.conv(layer, [Piece(xb, 0, Cin)], Piece(o0, 0, Cout), B, H, W)
    assert pl0.calls[0][0] == pl0.L.egne_conv2d_f16_big1_fwd
    pl0.run()
    # f16 tensors in and out
    pl1 = Plan(torch.device(DEV))
    pl1.f16_products = 1
    xb16 = pl1.buf16(B, H, W, Cin + 8)
    xb16.fill_(777.0)
    xb16[..., 8:].copy_(xh.permute(0, 2, 3, 1).to(DEV))
    pin = Piece(xb16, 8, Cin)
    pin.f16s = SplitScale()
    pin.f16s.value, pin.f16s.vmax = s_in, float(x.abs().max())
    o1 = pl1.buf16(B, H, W, Cout + 8)
    o1.fill_(777.0)
    pout = Piece(o1, 8, Cout)
    pout.f16s = SplitScale()
    pl1.conv(layer, [pin], pout, B, H, W)
    assert pl1.calls[0][0] == pl1.L.egne_conv2d_f16_big1_fwd and len(pl1.calls) == 1 and len(pl1.post_cal) == 1
    pl1.run()
    pl1.run()
    torch.cuda.synchronize()
    s_out = pout.f16s.value
    m = float(o1[..., 8:].float().abs().max())
    assert 1024.0 <= m < 2048.0, (m, s_out)
    assert (o1[..., :8] == 777.0).all(), "stores outside the output slice"
    want = (o0 * s_out).half()
    assert torch.equal(o1[..., 8:], want), "stored halves differ from f16(fp32 result * scale): %d elements" % (o1[..., 8:] != want).sum().item()
    print("big1 f16 in / out: %dx%dx%dx%d -> %d, scale %g, max stored %.1f" % (B, Cin, H, W, Cout, s_out, m))


@pytest.mark.parametrize("chans,C1,C2,B,H,W,res,post", [
    ((32, 32), 32, 32, 2, 61, 83, False, False),            # ESF block 0 conv21 + conv22 class: ragged tiles in x and y
    ((32, 32, 32), 32, 32, 3, 24, 64, True, False),         # conv31 + conv32 class, residual add in the epilogue
    ((38, 64), 64, 64, 2, 37, 70, False, True),             # block 1: 8-channel tail group, two LDS chunks, 64-wide 3x3, 4-row tiles
    ((38, 64, 64), 64, 64, 1, 120, 160, False, False),
    ((62, 32, 32), 32, 32, 2, 50, 96, False, False),        # up block 1 conv11 + conv12 class (62 logical of 64 channels)
    ((100, 64, 38, 62), 62, 62, 1, 33, 65, False, False),   # up block 2 class: 62-channel intermediate and output (padded to 64)
    ((64, 32), 64, 32, 1, 20, 100, False, False),           # two chunks in, 32 out
    ((32, 40), 30, 64, 2, 17, 61, False, False),            # one (padded) chunk in, 64 out
])
def test_conv1x1_3x3_fused(G, chans, C1, C2, B, H, W, res, post):
    """conv_fused_1x1_3x3_f16.hip: 3x3(1x1(cat(slices)) + b1) with the intermediate in LDS only, against float64.  The 3x3's
    zero padding applies to the 1x1 OUTPUT (b1 must not leak into the border), LeakyReLU + optional eval-BatchNorm affine
    and residual in the epilogue."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd import engine
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    xs = [_rand(G, B, c, H, W) * 2 for c in chans]
    Cin = sum(chans)
    w1, b1 = _rand(G, C1, Cin, 1, 1) / Cin ** 0.5, _rand(G, C1)
    w2, b2 = _rand(G, C2, C1, 3, 3) / (3 * C1 ** 0.5), _rand(G, C2)
    t = F.conv2d(torch.cat(xs, 1).double(), w1.double(), b1.double())
    truth = F.leaky_relu(F.conv2d(t, w2.double(), b2.double(), padding=1), 0.01)
    ps = pt = rz = None
    if post:
        ps, pt = _rand(G, C2).abs() + 0.5, _rand(G, C2)
        truth = truth * ps.double()[None, :, None, None] + pt.double()[None, :, None, None]
    if res:
        rz = _rand(G, B, C2, H, W)
        truth = truth + rz.double()
    pl = Plan(torch.device(DEV))
    pieces = to_nhwc_buf(pl, xs[:-1], B, H, W) + to_nhwc_buf(pl, xs[-1:], B, H, W)
    l1 = ConvLayer([torch.nn.Parameter(w1.to(DEV))], [torch.nn.Parameter(b1.to(DEV))], [(p.C, p.Cp) for p in pieces])
    l2 = ConvLayer([torch.nn.Parameter(w2.to(DEV))], [torch.nn.Parameter(b2.to(DEV))], [(C1, pad8(C1))], pad=(1, 1), act=2)
    l1.split1 = l2.split = True
    if post:
        a, b_ = torch.zeros(l2.CoutP, device=DEV), torch.zeros(l2.CoutP, device=DEV)
        a[:C2], b_[:C2] = ps.to(DEV), pt.to(DEV)
        l2.post = (a, b_)
    out = pl.buf(B, H, W, pad8(C2) + 16)
    out.fill_(777.0)
    rp = to_nhwc_buf(pl, [rz], B, H, W)[0] if res else None
    old = engine.FUSE_1X1_MIN_W
    engine.FUSE_1X1_MIN_W = 0
    try:
        pl.conv_pair(l1, pieces, l2, Piece(out, 8, C2), B, H, W, residual=rp)
    finally:
        engine.FUSE_1X1_MIN_W = old
    fusable = sum((c + 15) // 16 for c in [(x + 7) // 8 * 8 for x in chans]) <= 12
    assert len(pl.calls) == (1 if fusable else 2) and (not fusable or pl.calls[0][0] is pl.L.egne_conv1x1_3x3_fused_f16_fwd)
    for _ in range(2):      # calibrating run, replay
        pl.run()
        torch.cuda.synchronize()
        o = out.cpu()
        assert (o[..., :8] == 777.0).all() and (o[..., 8 + pad8(C2):] == 777.0).all(), "stores outside the output slice"
        got = o[..., 8:8 + C2].permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 3e-6, "relative error %.2e" % err
    # the unfused pair (two launches through a temporary) is the same function
    pl2 = Plan(torch.device(DEV))
    out2 = pl2.buf(B, H, W, pad8(C2))
    old = engine.FUSE_1X1
    engine.FUSE_1X1 = False
    try:
        pl2.conv_pair(l1, pieces, l2, Piece(out2, 0, C2), B, H, W, residual=rp)
    finally:
        engine.FUSE_1X1 = old
    assert len(pl2.calls) == 2
    pl2.run()
    torch.cuda.synchronize()
    got2 = out2.cpu()[..., :C2].permute(0, 3, 1, 2).double()
    assert (got2 - truth).abs().max().item() / truth.abs().max().item() < 3e-6


@pytest.mark.parametrize("B,H,W,stride_pad", [(2, 240, 320, 0), (3, 37, 53, 0), (2, 120, 160, 8), (5, 30, 40, 0), (4, 29, 39, 0), (1, 9, 33, 0)])
def test_msblock_dilated_group_one_launch(G, B, H, W, stride_pad):
    """msblock_dil_f16.hip: out = o + sum_g relu(conv_{dil 4,8,12}(o) + b_g) in ONE launch (bdcn_new.py:51-54), against float64:
    every BDCN stage size (240x320 ... 29x39: maps smaller than the 12-pixel reach of the widest dilation), ragged tiles,
    an input slice inside a wider buffer."""
    from gpu_util import DEV
    from egne_amd.engine import ConvLayer, Piece, Plan
    o = F.relu(_rand(G, B, 32, H, W)) * 2
    ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
    bs = [_rand(G, 32) for _ in range(3)]
    truth = o.double()
    for w, b, d in zip(ws, bs, (4, 8, 12)):
        truth = truth + F.relu(F.conv2d(o.double(), w.double(), b.double(), padding=d, dilation=d))
    pl = Plan(torch.device(DEV))
    buf = pl.buf(B, H, W, 32 + stride_pad)
    buf.fill_(55.0)
    buf[..., stride_pad:] = o.permute(0, 2, 3, 1).to(DEV)
    px = Piece(buf, stride_pad, 32)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV)) for w in ws], [torch.nn.Parameter(b.to(DEV)) for b in bs], [(32, 32)],
                      pad=(1, 1), dils=(4, 8, 12), act=1)
    layer.split = True
    out = pl.buf(B, H, W, 48)
    out.fill_(777.0)
    pl.conv(layer, [px], Piece(out, 8, 32), B, H, W, residual=px)
    assert len(pl.calls) == 1 and pl.calls[0][0] is pl.L.egne_msblock_dil_f16_fwd
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        oc = out.cpu()
        assert (oc[..., :8] == 777.0).all() and (oc[..., 40:] == 777.0).all()
        got = oc[..., 8:40].permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 2e-6, "relative error %.2e" % err


@pytest.mark.parametrize("chans,Cout,B,H,W", [((64, 38), 62, 2, 24, 40), ((32,), 32, 3, 10, 18), ((96, 76), 100, 1, 30, 44)])
def test_conv1x1_stream_with_upsampled_addend(G, chans, Cout, B, H, W):
    """conv1x1_f16.hip in its `up_add` mode (engine.Plan.conv(..., up_add=)): W_skip skip + b + bilinear-x2(P) with P at half resolution --
    conv11(cat(up(x), skip)) of an up block (models/RITnet_v2.py:84-86) without the up-sampled tensor -- against float64 with
    F.interpolate(scale_factor=2, mode="bilinear", align_corners=False); frames that end inside a 32-pixel block included."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    xs = [_rand(G, B, c, H, W) * 2 for c in chans]
    P = _rand(G, B, Cout, H // 2, W // 2) * 3
    w, b = _rand(G, Cout, sum(chans), 1, 1) / sum(chans) ** 0.5, _rand(G, Cout)
    truth = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double()) + F.interpolate(P.double(), scale_factor=2, mode="bilinear", align_corners=False)
    pl = Plan(torch.device(DEV))
    pieces = to_nhwc_buf(pl, xs, B, H, W)
    ocp = pad8(Cout)
    Pb = pl.buf(B, H // 2, W // 2, 2 * ocp)
    Pb.fill_(55.0)
    Pb[..., ocp:ocp + Cout] = P.permute(0, 2, 3, 1).to(DEV)           # second half of a two-addend buffer, as the decoder lays it out
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(p.C, p.Cp) for p in pieces])
    layer.split1 = True
    from egne_amd import engine
    old = engine.S1X1_MIN_PIX
    engine.S1X1_MIN_PIX = 0
    try:
        out = pl.buf(B, H, W, ocp + 8)
        out.fill_(777.0)
        pl.conv(layer, pieces, Piece(out, 8, Cout), B, H, W, name="a", up_add=(Piece(Pb, ocp, Cout, ocp), H // 2, W // 2))
    finally:
        engine.S1X1_MIN_PIX = old
    assert [m[0] for m in pl.meta] == ["conv_f16x3:stream1x1"]
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        o = out.cpu()
        assert (o[..., :8] == 777.0).all(), "stores outside the output slice"
        got = o[..., 8:8 + Cout].permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 3e-6, "relative error %.2e" % err


@pytest.mark.parametrize("Cin,Cout,B,H,W,act", [
    (32, 3, 2, 61, 83, 0),        # ESF-Net's logits layer class (models/RITnet_v2.py:249): ragged tiles in x and y
    (32, 3, 1, 240, 320, 0),
    (64, 4, 2, 24, 40, 2),        # two 32-channel chunks, four outputs, LeakyReLU
    (40, 1, 3, 17, 33, 1),        # a padded chunk (40 of 64 channels), one output, ReLU
    (62, 2, 1, 9, 70, 0),
])
def test_conv3x3_narrow_output_is_exact_fp32(G, Cin, Cout, B, H, W, act):
    """conv_narrow_f32.hip (egne_conv3x3_narrow_fwd): a 3x3 / pad 1 convolution with <= 4 output channels on the vector ALU in exact
    fp32 -- against float64 at the rounding level of a sequential fp32 sum, the engine routes such a layer of an
    inference plan to it, and nothing outside the destination slice is touched."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, Cin, H, W) * 3
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    truth = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if act == 1:
        truth = F.relu(truth)
    elif act == 2:
        truth = F.leaky_relu(truth, 0.01)
    pl = Plan(torch.device(DEV))
    (piece,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(piece.C, piece.Cp)], pad=(1, 1), act=act)
    layer.split = True
    if act == 2:          # eval BatchNorm behind the activation (utils.py:1049)
        a_, b_ = _rand(G, Cout).abs() + 0.5, _rand(G, Cout)
        truth = truth * a_.double()[None, :, None, None] + b_.double()[None, :, None, None]
        pa, pb = torch.zeros(layer.CoutP, device=DEV), torch.zeros(layer.CoutP, device=DEV)
        pa[:Cout], pb[:Cout] = a_.to(DEV), b_.to(DEV)
        layer.post = (pa, pb)
    out = pl.buf(B, H, W, 16)
    out.fill_(777.0)
    pl.conv(layer, [piece], Piece(out, 4, Cout), B, H, W, name="narrow")
    assert [m[0] for m in pl.meta] == ["conv3x3_narrow"] and pl.calls[0][0] is pl.L.egne_conv3x3_narrow_fwd
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        o = out.cpu()
        assert (o[..., :4] == 777.0).all(), "stores outside the output slice"
        assert (o[..., 4 + Cout:12] == 0.0).all() and (o[..., 12:] == 777.0).all()       # the slice's padding channels are written as zeros
        got = o[..., 4:4 + Cout].permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 1.5e-6, "relative error %.2e" % err       # 288-576 sequential fp32 multiply-adds per output
    # a training plan keeps the matrix path (its backward plan indexes the split packs)
    pl2 = Plan(torch.device(DEV), train=True)
    (p2,) = to_nhwc_buf(pl2, [x], B, H, W)
    pl2.conv(layer, [p2], Piece(pl2.buf(B, H, W, 8), 0, Cout), B, H, W, name="narrow")
    assert pl2.meta[0][0] != "conv3x3_narrow"


@pytest.mark.parametrize("B,H,W,Cin,mag", [(2, 240, 320, 64, 1.0), (3, 120, 160, 128, 1.0), (2, 75, 101, 64, 3e3), (2, 60, 80, 256, 1e-3), (1, 30, 40, 512, 1.0)])
def test_msblock_with_split_pair_storage(G, B, H, W, Cin, mag):
    """A whole MSBlock (bdcn_new.py:49-55) the way the edge network's plan runs it: the 3x3 convolution writes `o` in SPLIT-PAIR
    storage (egne_conv_desc.out_split: hi / lo f16 halves of o * s, s from a bound taken at calibration), the one-launch dilated
    group copies the halves into its operand image and recovers o itself for the 4-way sum -- against float64, at activation
    magnitudes from 1e-3 to 3e3, on maps with ragged tiles, and bit-identical on replay.  Where the producer is not the
    resident-weights kernel the plan must fall back to plain fp32 storage (asserted either way)."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, SplitScale, pad8
    x = F.relu(_rand(G, B, Cin, H, W)) * 2 * mag
    w0, b0 = _rand(G, 32, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, 32) * mag
    ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
    bs = [_rand(G, 32) * mag for _ in range(3)]
    o = F.relu(F.conv2d(x.double(), w0.double(), b0.double(), padding=1))
    truth = o.clone()
    for w, b, d in zip(ws, bs, (4, 8, 12)):
        truth = truth + F.relu(F.conv2d(o, w.double(), b.double(), padding=d, dilation=d))
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    l0 = ConvLayer([torch.nn.Parameter(w0.to(DEV))], [torch.nn.Parameter(b0.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
    lg = ConvLayer([torch.nn.Parameter(w.to(DEV)) for w in ws], [torch.nn.Parameter(b.to(DEV)) for b in bs], [(32, 32)],
                   pad=(1, 1), dils=(4, 8, 12), act=1)
    l0.split = lg.split = True
    obuf, out = pl.buf(B, H, W, 32), pl.buf(B, H, W, 32)
    po = Piece(obuf, 0, 32)
    assert pl.msdil_ok(lg, po, H, W)
    po.presplit = SplitScale()
    pl.conv(l0, [px], po, B, H, W)
    split = pl.last_presplit
    if not split:
        po.presplit = None
    pl.conv(lg, [po], Piece(out, 0, 32), B, H, W, residual=po)
    kinds = [k for k, _ in pl.meta]
    assert kinds[-1] == "conv_f16x3:msdil" and split == (kinds[0] == "conv_f16x3:rw"), kinds
    firsts = []
    for _ in range(2):          # calibrating run, replay
        pl.run()
        torch.cuda.synchronize()
        got = out.cpu().permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 3e-6, "relative error %.2e (split-pair storage %s)" % (err, split)
        firsts.append(out.clone())
    assert torch.equal(firsts[0], firsts[1])
    if split:
        s = po.presplit.value
        omax = o.abs().max().item()
        assert 1.0 < s * omax < 2048.0, (s, omax)        # the bound holds (nothing overflows f16) and is within 2^10 of the maximum
        # the stored halves ARE the split of o: hi + lo reproduces o to 2^-21 of the maximum
        raw = obuf.view(torch.float16).reshape(B, H, W, 2, 32).float().cpu()
        from egne_amd.engine import SPLIT_PAIR_PERM
        pos = raw[..., 0, :] + raw[..., 1, :]                  # position p of a plane holds channel SPLIT_PAIR_PERM[p]
        rec = torch.empty_like(pos)
        rec[..., SPLIT_PAIR_PERM] = pos
        rec = rec.permute(0, 3, 1, 2).double() / s
        assert (rec - o).abs().max().item() < 2.0 ** -20 * omax
    print("MSBlock %dx%dx%d Cin %d mag %g: err %.2e, split-pair storage %s (%s)" % (B, H, W, Cin, mag, err, split, kinds[0]))


@pytest.mark.parametrize("B,H,W,Cin,scores", [(2, 240, 320, 64, False), (3, 120, 160, 128, True), (2, 75, 101, 64, True), (5, 60, 80, 64, False),
                                               (40, 64, 96, 64, True)])
def test_msblock_plain_f16_ring_form(G, B, H, W, Cin, scores, monkeypatch):
    """msblock_dil1_f16.hip (the dilated group of an MSBlock on plain f16 operands, a ring of 32 rows in LDS; bdcn_new.py:49-55 under
    utils.py:646 next to a bf16-storage training plan): BIT-identical to the strip form of msblock_dil_ps_f16.hip with one product --
    block output or fused score maps, ragged maps, several segments per column, more work items than workgroups -- and within the
    plain-f16 rounding of float64."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, SplitScale, pad8
    x = F.relu(_rand(G, B, Cin, H, W)) * 2
    w0, b0 = _rand(G, 32, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, 32)
    ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
    bs = [_rand(G, 32) for _ in range(3)]
    cw, cc = _rand(G, 2, 32) / 6, torch.tensor([0.7, -1.3])
    o = F.relu(F.conv2d(x.double().to(DEV), w0.double().to(DEV), b0.double().to(DEV), padding=1))
    truth = o.clone()
    for w, b, d in zip(ws, bs, (4, 8, 12)):
        truth = truth + F.relu(F.conv2d(o, w.double().to(DEV), b.double().to(DEV), padding=d, dilation=d))
    if scores:
        truth = torch.stack([(truth * cw[h].double().to(DEV)[None, :, None, None]).sum(1) + float(cc[h]) for h in range(2)], 1)
    res = []
    for ring in ("1", "0"):
        monkeypatch.setenv("EGNE_MSDIL1", ring)
        pl = Plan(torch.device(DEV))
        pl.f16_products = 1
        (px,) = to_nhwc_buf(pl, [x], B, H, W)
        l0 = ConvLayer([torch.nn.Parameter(w0.to(DEV))], [torch.nn.Parameter(b0.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=1)
        lg = ConvLayer([torch.nn.Parameter(w.to(DEV)) for w in ws], [torch.nn.Parameter(b.to(DEV)) for b in bs], [(32, 32)],
                       pad=(1, 1), dils=(4, 8, 12), act=1)
        l0.split = lg.split = True
        obuf = pl.buf(B, H, W, 32)
        po = Piece(obuf, 0, 32)
        assert pl.msdil_ok(lg, po, H, W)
        po.presplit = SplitScale()
        pl.conv(l0, [px], po, B, H, W)
        assert pl.last_presplit, "the resident-weights 3x3 writes split-pair storage"
        if scores:
            s0, s1 = pl.vec(B, H, W), pl.vec(B, H, W)
            s0.fill_(123.0); s1.fill_(-5.0)
            cwd, ccd = cw.to(DEV), cc.to(DEV)
            pl.keep += [cwd, ccd]
            pl.conv(lg, [po], po, B, H, W, residual=po, scores=(cwd, ccd, s0, s1, False))
        else:
            out = pl.buf(B, H, W, 40)
            out.fill_(777.0)
            pl.conv(lg, [po], Piece(out, 8, 32), B, H, W, residual=po)
        assert [k for k, _ in pl.meta][-1] == "conv_f16x3:msdil"
        for _ in range(2):
            pl.run()
        torch.cuda.synchronize()
        if scores:
            res.append(torch.stack([s0, s1], 1).clone())
        else:
            assert (out[..., :8] == 777.0).all(), "stores outside the output slice"
            res.append(out[..., 8:].permute(0, 3, 1, 2).clone())
    assert torch.equal(res[0], res[1]), "ring and strip forms differ: max %.3e" % (res[0] - res[1]).abs().max().item()
    err = (res[0].double() - truth).abs().max().item() / truth.abs().max().item()
    print("MSBlock plain f16, ring form, %dx%dx%d scores %s: err %.2e vs float64" % (B, H, W, scores, err))
    assert err < 3e-3          # operands rounded to 11 bits, twice (o itself, then the dilated taps)


@pytest.mark.parametrize("kind,B,H,W,C1,C2", [("halo", 3, 61, 83, 32, 32), ("halo", 2, 120, 160, 64, 96), ("pair", 3, 61, 83, 32, 32),
                                              ("pair", 2, 37, 70, 64, 64), ("halo", 2, 60, 80, 32, 32), ("halo", 2, 64, 96, 32, 64)])
def test_instance_norm_statistics_from_the_conv_epilogue(G, kind, B, H, W, C1, C2):
    """stats=True: the producing kernel writes per-tile partial sums of the values it stores and egne_norm_stats_finish turns
    them into the consumer's InstanceNorm (scale, shift) (models/RITnet_v2.py:40,57: eps 1e-5, biased variance) -- compared
    with the statistics of the stored tensor itself.  60x80 is walked transposed by the halo kernel: separate pass."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd import engine
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    pl = Plan(torch.device(DEV))
    out = pl.buf(B, H, W, pad8(C2))
    if kind == "halo":
        x = _rand(G, B, C1, H, W)
        (px,) = to_nhwc_buf(pl, [x], B, H, W)
        w, b = _rand(G, C2, C1, 3, 3) / (3 * C1 ** 0.5), _rand(G, C2)
        layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(C1, pad8(C1))], pad=(1, 1), act=2)
        layer.split = True
        pl.conv(layer, [px], Piece(out, 0, C2), B, H, W, stats=True)
    else:
        xs = [_rand(G, B, 32, H, W), _rand(G, B, 40, H, W)]
        pieces = to_nhwc_buf(pl, xs, B, H, W)
        w1, b1 = _rand(G, C1, 72, 1, 1) / 8, _rand(G, C1)
        w2, b2 = _rand(G, C2, C1, 3, 3) / (3 * C1 ** 0.5), _rand(G, C2)
        l1 = ConvLayer([torch.nn.Parameter(w1.to(DEV))], [torch.nn.Parameter(b1.to(DEV))], [(p.C, p.Cp) for p in pieces])
        l2 = ConvLayer([torch.nn.Parameter(w2.to(DEV))], [torch.nn.Parameter(b2.to(DEV))], [(C1, pad8(C1))], pad=(1, 1), act=2)
        l1.split1 = l2.split = True
        old = engine.FUSE_1X1_MIN_W
        engine.FUSE_1X1_MIN_W = 0
        try:
            pl.conv_pair(l1, pieces, l2, Piece(out, 0, C2), B, H, W, stats=True)
        finally:
            engine.FUSE_1X1_MIN_W = old
    scale, shift = pl.last_stats
    fused = pl.calls[-1][0] is pl.L.egne_norm_stats_finish
    tall = ((H + 31) // 32) * ((W + 7) // 8) < ((W + 31) // 32) * ((H + 7) // 8)      # the halo kernel walks such maps transposed
    rs = any(m[0].endswith(":rs") for m in pl.meta)                                    # the role-split kernel never does
    assert fused == (kind == "pair" or rs or not tall)
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        y = out.cpu().double()[..., :C2]
        mean, var = y.mean((1, 2)), y.var((1, 2), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        np.testing.assert_allclose(scale.cpu().numpy()[:, :C2], rstd.numpy(), rtol=2e-6)
        np.testing.assert_allclose(shift.cpu().numpy()[:, :C2], (-mean * rstd).numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("B,Cin,Cout,H,W,mode", [(2, 64, 64, 61, 83, "plain"), (1, 64, 128, 120, 160, "plain"), (2, 64, 32, 61, 70, "res"),
                                                  (2, 32, 32, 61, 83, "norm"), (1, 38, 64, 120, 160, "norm"), (3, 56, 30, 33, 64, "plain"),
                                                  (2, 32, 2, 64, 96, "norm"),
                                                  # several tiles per workgroup (more than 256 tiles): the multi-tile paths of every form
                                                  (6, 32, 32, 240, 320, "plain"), (5, 64, 64, 120, 160, "plain"), (4, 64, 32, 240, 320, "plain"),
                                                  # wider inputs: weights streamed chunk by chunk into LDS (vgg conv2_2, MSBlock convs of
                                                  # stages 2-3, the 60x80 dense block); 96 channels = three output blocks, one of four idle
                                                  (3, 128, 32, 120, 160, "plain"), (2, 128, 32, 120, 130, "res"),
                                                  (1, 256, 32, 120, 160, "plain"), (2, 80, 32, 120, 128, "norm_nostats")])
def test_conv3x3_role_split(G, B, Cin, Cout, H, W, mode):
    """conv3x3_rs_f16.hip (producer / consumer waves; vgg16_c.py:66-69, bdcn_new.py:50, models/RITnet_v2.py:57, utils.py:1047)
    against a float64 convolution: plain, with the residual addend, and with the InstanceNorm affine + LeakyReLU applied while
    the halo is staged (zero padding AFTER the normalisation) plus statistics of the stored output."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, Cin, H, W) * 2 + 0.5
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
    layer.split = True
    out = pl.buf(B, H, W, pad8(Cout))
    xin, residual = x.double(), None
    if mode.startswith("norm"):
        mean, rstd = x.mean((2, 3)), 1 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        sc, sh = torch.zeros(B, px.Cp, device=DEV), torch.zeros(B, px.Cp, device=DEV)
        sc[:, :Cin], sh[:, :Cin] = rstd.to(DEV), (-mean * rstd).to(DEV)
        pl.keep += [sc, sh]
        px = px.with_norm(sc, sh, 2)
        xin = F.leaky_relu(F.instance_norm(x.double()))
    truth = F.leaky_relu(F.conv2d(xin, w.double(), b.double(), padding=1))
    if mode == "res":                      # the residual joins AFTER the activation (bdcn_new.py:55: o + relu(conv(o)))
        r = _rand(G, B, Cout, H, W)
        (pr,) = to_nhwc_buf(pl, [r], B, H, W)
        residual = pr
        truth = truth + r.double()
    from egne_amd import engine
    # (a 33..48-channel slice takes the halo kernel by default -- it skips the zero half of the last chunk's k-steps, engine.TAIL16_HALO;
    #  here the role-split kernels are what is tested, the halo route is asserted below)
    tail16, engine.TAIL16_HALO = engine.TAIL16_HALO, False
    try:
        pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W, residual=residual, stats=(mode == "norm"))
    finally:
        engine.TAIL16_HALO = tail16
    assert any(m[0] in ("conv_f16x3:rs", "conv_f16x3:rw") for m in pl.meta)     # role-split: register-ring or resident-weights form
    if 32 < pad8(Cin) <= 48 and tail16:
        pl2 = Plan(torch.device(DEV))
        (p2,) = to_nhwc_buf(pl2, [x], B, H, W)
        pl2.conv(layer, [p2], Piece(pl2.buf(B, H, W, pad8(Cout)), 0, Cout), B, H, W)
        assert [m[0] for m in pl2.meta] == ["conv_f16x3:halo"]
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        got = out.cpu().permute(0, 3, 1, 2).double()
        assert (got[:, :Cout] - truth).abs().max().item() / truth.abs().max().item() < 2e-6
        assert (got[:, Cout:] == 0).all()
        if mode == "norm":
            scale, shift = pl.last_stats
            y = got[:, :Cout]
            rstd = 1.0 / torch.sqrt(y.var((2, 3), unbiased=False) + 1e-5)
            np.testing.assert_allclose(scale.cpu().numpy()[:, :Cout], rstd.numpy(), rtol=2e-6)
            np.testing.assert_allclose(shift.cpu().numpy()[:, :Cout], (-y.mean((2, 3)) * rstd).numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("chans,Cout,B,H,W", [((32, 32), 38, 3, 24, 40), ((64, 38), 76, 2, 30, 50), ((32, 24), 30, 3, 10, 30), ((32, 32), 32, 2, 240, 320)])
def test_transition_down_pool_folded_into_the_1x1(G, chans, Cout, B, H, W):
    """models/RITnet_v2.py:32-44 in an eval plan: avg_pool2d(conv1x1(leaky(IN(cat(out, x)))), 2) as ONE launch whose operand is the
    2x2 window average of the normalised, activated slices (pooling and 1x1 commute) -- against float64; blocks that span two
    frames (H/2 * W/2 not a multiple of 32) included."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    xs = [_rand(G, B, c, H, W) * 2 + 0.3 for c in chans]
    w, b = _rand(G, Cout, sum(chans), 1, 1) / sum(chans) ** 0.5, _rand(G, Cout)
    xin = torch.cat([F.leaky_relu(F.instance_norm(x.double())) for x in xs], 1)
    truth = F.avg_pool2d(F.conv2d(xin, w.double(), b.double()), 2)
    pl = Plan(torch.device(DEV))
    pieces = to_nhwc_buf(pl, xs, B, H, W)
    normed = []
    for x, pc in zip(xs, pieces):
        mean, rstd = x.mean((2, 3)), 1 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        sc, sh = torch.zeros(B, pc.Cp, device=DEV), torch.zeros(B, pc.Cp, device=DEV)
        sc[:, :pc.C], sh[:, :pc.C] = rstd.to(DEV), (-mean * rstd).to(DEV)
        pl.keep += [sc, sh]
        normed.append(pc.with_norm(sc, sh, 2))
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(p.C, p.Cp) for p in pieces])
    layer.split1 = True
    out = pl.buf(B, H // 2, W // 2, pad8(Cout))
    dst = Piece(out, 0, Cout)
    assert pl.td_pool_fusable(layer, normed, dst)
    pl.conv1x1_pooled(layer, normed, dst, B, H, W)
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        got = out.cpu().permute(0, 3, 1, 2).double()[:, :Cout]
        assert (got - truth).abs().max().item() / truth.abs().max().item() < 2e-6


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (2, 61, 83), (1, 240, 320)])
def test_conv3x3_role_split_pooled_second_output(G, B, H, W):
    """vgg16_c.py:69-70: relu(conv1_2(x)) and its 2x2 / stride 2 / ceil-mode max pooling from ONE launch (the pooled tensor is a
    second output of the role-split kernel's epilogue) -- both against a float64 convolution, odd sizes included."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = F.relu(_rand(G, B, 64, H, W)) * 3
    w, b = _rand(G, 64, 64, 3, 3) / 24, _rand(G, 64)
    truth = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    pooled = F.max_pool2d(truth, 2, stride=2, ceil_mode=True)
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(64, 64)], pad=(1, 1), act=1)
    layer.split = True
    out, pout = pl.buf(B, H, W, 64), pl.buf(B, (H + 1) // 2, (W + 1) // 2, 64)
    pl.conv(layer, [px], Piece(out, 0, 64), B, H, W, pool=Piece(pout, 0, 64))
    assert pl.last_pooled and any(m[0] in ("conv_f16x3:rs", "conv_f16x3:rw") for m in pl.meta)
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        got, gp = out.cpu().permute(0, 3, 1, 2).double(), pout.cpu().permute(0, 3, 1, 2).double()
        scale = truth.abs().max().item()
        assert (got - truth).abs().max().item() / scale < 2e-6
        assert (gp - pooled).abs().max().item() / scale < 2e-6


@pytest.mark.parametrize("B,H,W,Cout,act", [(2, 120, 160, 128, 1), (2, 61, 83, 128, 2), (1, 40, 96, 64, 1), (1, 9, 33, 192, 1)])
def test_conv3x3_halo_pooled_second_output(G, B, H, W, Cout, act):
    """vgg16_c.py:71-72: relu(conv2_2(x)) (128 input channels: the LDS-halo kernel) and its 2x2 / stride 2 / ceil-mode max pooling from ONE
    launch -- both against a float64 convolution; odd sizes (clipped last window), ragged tiles, LeakyReLU (negative values in the
    pool), poison around both output slices."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, 128, H, W) * 2
    w, b = _rand(G, Cout, 128, 3, 3) / 34, _rand(G, Cout)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    truth = F.relu(y) if act == 1 else F.leaky_relu(y)
    pooled = F.max_pool2d(truth, 2, stride=2, ceil_mode=True)
    pl = Plan(torch.device(DEV))
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(128, 128)], pad=(1, 1), act=act)
    layer.split = True
    out, pout = pl.buf(B, H, W, Cout + 8), pl.buf(B, (H + 1) // 2, (W + 1) // 2, Cout + 16)
    out.fill_(777.0); pout.fill_(777.0)
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W, pool=Piece(pout, 8, Cout))
    assert pl.last_pooled and pl.meta[-1][0] == "conv_f16x3:halo", pl.meta
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        o, q = out.cpu(), pout.cpu()
        assert (o[..., Cout:] == 777.0).all() and (q[..., :8] == 777.0).all() and (q[..., 8 + Cout:] == 777.0).all()
        got, gp = o[..., :Cout].permute(0, 3, 1, 2).double(), q[..., 8:8 + Cout].permute(0, 3, 1, 2).double()
        scale = truth.abs().max().item()
        assert (got - truth).abs().max().item() / scale < 2e-6
        assert (gp - pooled).abs().max().item() / scale < 2e-6


@pytest.mark.parametrize("Cin,B,H,W,post", [(1, 3, 61, 83, True), (2, 2, 240, 320, True), (3, 2, 33, 64, False)])
def test_convblock_pair_fused(G, Cin, B, H, W, post):
    """utils.py:1047-1048 convBlock: conv2(leaky(conv1(x))) with conv1 on 1-3 channels, as ONE launch (the first conv's 9 taps
    folded into K on each tile's halo, its result kept in LDS), LeakyReLU + eval-BatchNorm affine in the epilogue, against float64."""
    from gpu_util import DEV
    from egne_amd import engine
    from egne_amd.engine import ConvLayer, Piece, Plan
    x = _rand(G, B, Cin, H, W) * 2
    w1, b1 = _rand(G, 32, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, 32)
    w2, b2 = _rand(G, 32, 32, 3, 3) / (3 * 32 ** 0.5), _rand(G, 32)
    t = F.leaky_relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1), 0.01)
    truth = F.leaky_relu(F.conv2d(t, w2.double(), b2.double(), padding=1), 0.01)
    pl = Plan(torch.device(DEV))
    xin = pl.buf(B, H, W, 8)
    xin[..., :Cin] = x.permute(0, 2, 3, 1).to(DEV)
    l1 = ConvLayer([torch.nn.Parameter(w1.to(DEV))], [torch.nn.Parameter(b1.to(DEV))], [(Cin, 8)], pad=(1, 1), act=2)
    l2 = ConvLayer([torch.nn.Parameter(w2.to(DEV))], [torch.nn.Parameter(b2.to(DEV))], [(32, 32)], pad=(1, 1), act=2)
    l1.split = l2.split = True
    if post:
        ps, pt = _rand(G, 32).abs() + 0.5, _rand(G, 32)
        truth = truth * ps.double()[None, :, None, None] + pt.double()[None, :, None, None]
        l2.post = (ps.to(DEV), pt.to(DEV))
    out = pl.buf(B, H, W, 48)
    out.fill_(777.0)
    old = engine.FUSE_1X1_MIN_W
    engine.FUSE_1X1_MIN_W = 0
    try:
        pl.conv_pair(l1, [Piece(xin, 0, Cin, 8)], l2, Piece(out, 8, 32), B, H, W, stats=True)
    finally:
        engine.FUSE_1X1_MIN_W = old
    assert pl.calls[0][0] is pl.L.egne_conv3x3c4_3x3_fused_f16_fwd and len(pl.calls) == 2
    scale, shift = pl.last_stats
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        o = out.cpu()
        assert (o[..., :8] == 777.0).all() and (o[..., 40:] == 777.0).all()
        got = o[..., 8:40].permute(0, 3, 1, 2).double()
        err = (got - truth).abs().max().item() / truth.abs().max().item()
        assert err < 3e-6, "relative error %.2e" % err
        y = o[..., 8:40].double()
        rstd = 1.0 / torch.sqrt(y.var((1, 2), unbiased=False) + 1e-5)
        np.testing.assert_allclose(scale.cpu().numpy(), rstd.numpy(), rtol=2e-6)


@pytest.mark.parametrize("B,H,W", [(2, 240, 320), (3, 29, 39), (2, 61, 83)])
def test_msblock_dilated_group_with_fused_scores(G, B, H, W):
    """egne_msblock_dil_scores_f16_fwd: the two score maps of a BDCN stage accumulated from the epilogues of its blocks
    (s = sum_k w_k . out_k + c), block outputs never stored, against float64."""
    from gpu_util import DEV
    from egne_amd.engine import ConvLayer, Piece, Plan
    pl = Plan(torch.device(DEV))
    s0, s1 = pl.vec(B, H, W), pl.vec(B, H, W)
    s0.fill_(123.0); s1.fill_(-5.0)            # stale contents must be overwritten by the first block
    cc = torch.tensor([0.7, -1.3])
    ccd = cc.to(DEV)
    pl.keep.append(ccd)
    want0, want1 = torch.full((B, H, W), 0.7, dtype=torch.float64), torch.full((B, H, W), -1.3, dtype=torch.float64)
    for k in range(2):
        o = F.relu(_rand(G, B, 32, H, W)) * 2
        ws = [_rand(G, 32, 32, 3, 3) / 17 for _ in range(3)]
        bs = [_rand(G, 32) for _ in range(3)]
        out = o.double()
        for w, b, d in zip(ws, bs, (4, 8, 12)):
            out = out + F.relu(F.conv2d(o.double(), w.double(), b.double(), padding=d, dilation=d))
        cw = _rand(G, 2, 32) / 6
        want0 += (out * cw[0].double()[None, :, None, None]).sum(1)
        want1 += (out * cw[1].double()[None, :, None, None]).sum(1)
        buf = pl.buf(B, H, W, 32)
        buf.copy_(o.permute(0, 2, 3, 1).to(DEV))
        px = Piece(buf, 0, 32)
        layer = ConvLayer([torch.nn.Parameter(w.to(DEV)) for w in ws], [torch.nn.Parameter(b.to(DEV)) for b in bs], [(32, 32)],
                          pad=(1, 1), dils=(4, 8, 12), act=1)
        layer.split = True
        assert pl.msdil_ok(layer, px, H, W)
        cwd = cw.to(DEV)
        pl.keep.append(cwd)
        pl.conv(layer, [px], px, B, H, W, residual=px, scores=(cwd, ccd, s0, s1, k > 0))
    for _ in range(2):
        pl.run()
        torch.cuda.synchronize()
        for got, want in ((s0, want0), (s1, want1)):
            err = (got.cpu().double() - want).abs().max().item() / want.abs().max().item()
            assert err < 3e-6, "relative error %.2e" % err


@pytest.mark.parametrize("Cin,Cout,H,W,mag", [(32, 32, 64, 96, 1.0), (64, 32, 61, 83, 3.0e3), (64, 64, 120, 160, 2.0e-6), (128, 128, 30, 40, 40.0),
                                                (48, 40, 20, 24, 1.0e4)])
def test_device_side_prescale_matches_the_host_calibration(G, Cin, Cout, H, W, mag):
    """Training plans (engine.Plan.dyn_scales): the split-f16 pre-scale is derived inside the kernel from the bit pattern of
    max|x| (egne_conv_desc.dyn_scale, written by egne_absmax on the stream) instead of a host-side calibration.  Both choose the
    power of two that puts max|x| in [1024, 2048), so the two plans must produce THE SAME BITS at any activation magnitude
    (gradient tensors of 1e-6, activations of 1e4), on the role-split / resident-weights, halo and flat kernels."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    B = 2
    x = _rand(G, B, Cin, H, W) * mag
    w, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout) * mag
    outs = []
    for dyn in (False, True):
        pl = Plan(torch.device(DEV))
        pl.dyn_scales = dyn
        (px,) = to_nhwc_buf(pl, [x], B, H, W)
        layer = ConvLayer([torch.nn.Parameter(w.to(DEV))], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
        layer.split = True
        out = pl.buf(B, H, W, pad8(Cout))
        pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
        kinds = [m[0] for m in pl.meta]
        assert any(k.startswith("conv_f16x3:") for k in kinds), kinds
        assert ("absmax" in kinds) == dyn and bool(pl.cal) == (not dyn)
        for _ in range(2):
            pl.run()
        torch.cuda.synchronize()
        outs.append(out.cpu())
    truth = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    got = outs[1].permute(0, 3, 1, 2).double()[:, :Cout]
    assert torch.isfinite(got).all()
    assert (got - truth).abs().max().item() / truth.abs().max().item() < 2e-6
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dyn", [False, True])
@pytest.mark.parametrize("Cin,Cout,H,W", [(32, 32, 64, 96), (64, 64, 30, 40), (256, 256, 30, 40), (3, 64, 40, 64)])
def test_weight_scale_follows_a_repack(G, Cin, Cout, H, W, dyn):
    """The split-f16 launches carry the weight pack's power-of-two scale BY VALUE.  ensure_packed re-measures max |w| when the
    weights change (optimizer step, load_state_dict) and may repack with another power of two: Plan.run must then rewrite the
    baked argument (Plan._refresh_wscales), or the layer's output is off by 2^k for the rest of the run.  Weights are moved
    across several powers of two between runs of ONE plan and every run is compared with float64."""
    from gpu_util import DEV, to_nhwc_buf
    import egne_amd.engine as eng
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    if dyn and Cin < 8:
        pytest.skip("training plans run the first layer in exact fp32")
    B = 2
    x = _rand(G, B, Cin, H, W)
    w0, b = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), _rand(G, Cout)
    pl = Plan(torch.device(DEV))
    pl.dyn_scales = dyn
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    wp = torch.nn.Parameter(w0.to(DEV))
    layer = ConvLayer([wp], [torch.nn.Parameter(b.to(DEV))], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
    layer.split = True
    out = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
    assert any(k.startswith("conv_f16x3:") for k, _ in pl.meta) and pl.wscale_refs
    old_every, eng.WSCALE_EVERY = eng.WSCALE_EVERY, 1        # re-measure at every repack (training plans: every 16th otherwise)
    try:
        seen = set()
        for f in (1.0, 4.0, 0.03, 300.0):
            with torch.no_grad():
                wp.copy_(w0.to(DEV) * f)
            pl.run()
            torch.cuda.synchronize()
            seen.add(getattr(layer, pl.wscale_refs[0][3]))
            truth = F.leaky_relu(F.conv2d(x.double(), (w0 * f).double(), b.double(), padding=1))
            got = out.cpu().permute(0, 3, 1, 2).double()[:, :Cout]
            err = (got - truth).abs().max().item() / truth.abs().max().item()
            assert err < 3e-6, "weights x%g: relative error %.2e" % (f, err)
        assert len(seen) >= 3, seen          # the pack's scale did change
    finally:
        eng.WSCALE_EVERY = old_every


@pytest.mark.parametrize("Cin,Cout,H,W", [(32, 32, 64, 96), (38, 64, 61, 83), (96, 96, 30, 40)])
def test_split_data_gradient_layer(G, Cin, Cout, H, W):
    """engine.SplitDgradLayer (backward of models/RITnet_v2.py:57-62's 3x3 convs, train.py:285): the data gradient as an ordinary
    split-f16 3x3 over gz with flipped / transposed weights, accumulated onto what the slice already holds -- against
    float64 autograd; the pre-scale comes from the device-side max|gz| word (gradient magnitudes of 1e-4)."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, SplitDgradLayer, pad8
    B = 2
    w = _rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    gz, acc0 = _rand(G, B, Cout, H, W) * 1e-4, _rand(G, B, Cin, H, W) * 1e-4
    xd = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, w.double(), padding=1).backward(gz.double())
    truth = xd.grad + acc0.double()
    pl = Plan(torch.device(DEV))
    pl.dyn_scales = True
    fwd = ConvLayer([torch.nn.Parameter(w.to(DEV))], None, [(Cin, pad8(Cin))], pad=(1, 1))
    dl = SplitDgradLayer(fwd, 0, torch.device(DEV))
    pl.pre.append(dl.guard)
    pg, pa = to_nhwc_buf(pl, [gz, acc0], B, H, W)
    pl.conv(dl, [Piece(pg.buf, pg.off, Cout, pad8(Cout))], pa, B, H, W, residual=pa)
    assert any(k.startswith("conv_f16x3:") for k, _ in pl.meta)
    pl.run()
    torch.cuda.synchronize()
    got = pa.buf.cpu().permute(0, 3, 1, 2).double()[:, pa.off:pa.off + Cin]
    assert (got - truth).abs().max().item() / truth.abs().max().item() < 3e-6


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 24, 40), (40, 3, 30, 50), (64, 2, 120, 160)])
def test_norm_pool2_bwd(G, C, B, H, W):
    """egne_norm_pool2_bwd: backward of zp = avg_pool2d(leaky_relu(instance_norm(x)), 2) (Transition_down pooled in front of its
    1x1, models/RITnet_v2.py:32-44) w.r.t. x, accumulated onto gx, against float64 autograd."""
    from gpu_util import DEV
    from egne_amd import _lib
    L = _lib.lib()
    x, gzp, g0 = _rand(G, B, C, H, W) * 2 + 0.3, _rand(G, B, C, H // 2, W // 2), _rand(G, B, C, H, W)
    xd = x.double().requires_grad_(True)
    F.avg_pool2d(F.leaky_relu(F.instance_norm(xd)), 2).backward(gzp.double())
    truth = xd.grad + g0.double()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)       # noqa: E731
    xn, gn, gx = nhwc(x), nhwc(gzp), nhwc(g0)
    mean, rstd = x.mean((2, 3)), 1 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
    sc, sh = rstd.to(DEV).contiguous(), (-mean * rstd).to(DEV).contiguous()
    sums = torch.zeros(B * C * 2, device=DEV)
    ws = torch.zeros((int(L.egne_norm_bwd_workspace_bytes(B, H * W, C, 1)) + 7) // 8, dtype=torch.float64, device=DEV)
    _lib.check(L.egne_norm_pool2_bwd(xn.data_ptr(), C, 0, sc.data_ptr(), sh.data_ptr(), gn.data_ptr(), C, 0, 2, C, B, H, W,
                                     gx.data_ptr(), C, 0, 1, sums.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "norm_pool2_bwd")
    torch.cuda.synchronize()
    got = gx.cpu().permute(0, 3, 1, 2).double()
    assert (got - truth).abs().max().item() / truth.abs().max().item() < 5e-6
    # accumulate = 0: the first writer of a gradient slice stores (whatever the slice held before)
    gx.fill_(123.0)
    _lib.check(L.egne_norm_pool2_bwd(xn.data_ptr(), C, 0, sc.data_ptr(), sh.data_ptr(), gn.data_ptr(), C, 0, 2, C, B, H, W,
                                     gx.data_ptr(), C, 0, 0, sums.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "norm_pool2_bwd")
    torch.cuda.synchronize()
    got = gx.cpu().permute(0, 3, 1, 2).double()
    assert (got - xd.grad).abs().max().item() / xd.grad.abs().max().item() < 5e-6


@pytest.mark.parametrize("B,Cin,Cout,H,W,norm,mag", [(2, 32, 32, 64, 96, False, 1.0), (2, 38, 64, 61, 83, True, 1.0), (1, 96, 96, 30, 40, False, 1e-5),
                                                       (3, 64, 32, 120, 160, False, 300.0), (2, 32, 32, 240, 320, True, 1e-6)])
def test_conv3x3_weight_gradient_split_f16(G, B, Cin, Cout, H, W, norm, mag):
    """wgrad_halo.hip, split-f16 form (training plans; train.py:285-286's loss.backward() for models/RITnet_v2.py:57-62): the
    weight gradient of a 3x3 convolution on 3 f16 MFMAs per product, x pre-scaled by the forward launch's device word (or the
    fixed scale of inputs normalised on load), gz by the word of act_bwd_bias_absmax -- against float64 autograd at gradient
    magnitudes from 1e-6 to 300, ragged tiles and channel tails included."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    x = _rand(G, B, Cin, H, W) * 2 + 0.25
    w = (_rand(G, Cout, Cin, 3, 3) / (3 * Cin ** 0.5)).double().requires_grad_(True)
    b = _rand(G, Cout).double().requires_grad_(True)
    gy = _rand(G, B, Cout, H, W) * mag
    xe = x.double()
    if norm:
        mean, rstd = x.mean((2, 3)), 1 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        xe = F.leaky_relu(F.instance_norm(x.double()))
    y_lin = F.conv2d(xe, w, b, padding=1)
    pl = Plan(torch.device(DEV), train=True)
    assert pl.dyn_scales
    (px,) = to_nhwc_buf(pl, [x], B, H, W)
    px.nograd = True
    if norm:
        scp, shp = torch.zeros(B, px.Cp, device=DEV), torch.zeros(B, px.Cp, device=DEV)
        scp[:, :Cin], shp[:, :Cin] = rstd.to(DEV), (-mean * rstd).to(DEV)
        pl.keep += [scp, shp]
        px = px.with_norm(scp, shp, 2)
        px.nograd = True
    wd, bd = torch.nn.Parameter(w.detach().float().to(DEV)), torch.nn.Parameter(b.detach().float().to(DEV))
    wd.grad, bd.grad = torch.zeros_like(wd), torch.zeros_like(bd)
    layer = ConvLayer([wd], [bd], [(Cin, pad8(Cin))], pad=(1, 1), act=2)
    layer.split = True
    out = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, [px], Piece(out, 0, Cout), B, H, W)
    bw = pl.build_backward()
    assert any(k == "conv_f16x3:wgrad" for k, _ in bw.meta), [k for k, _ in bw.meta]
    for _ in range(2):                      # the second pass accumulates onto the first
        pl.run()
        pl.gbuf(out).zero_()
        pl.gbuf(out)[..., :Cout] = gy.permute(0, 2, 3, 1).to(DEV)
        bw.run()
    torch.cuda.synchronize()
    # the float64 gradient takes the LeakyReLU branch of every pixel from the DEVICE's forward output: a pre-activation within
    # round-off of zero may land on either side, and with gradients of random sign one such pixel moves the sums by ~1e-3 of their
    # size (seen with an unlucky draw) -- that is the activation's discontinuity, not the arithmetic under test
    y_dev = out.cpu()[..., :Cout].permute(0, 3, 1, 2).double()
    assert ((y_dev > 0) != (y_lin > 0)).double().mean().item() < 1e-4
    y_lin.backward(gy.double() * torch.where(y_dev > 0, 1.0, 0.01))
    ew = (wd.grad.cpu().double() / 2 - w.grad).abs().max().item() / w.grad.abs().max().item()
    eb = (bd.grad.cpu().double() / 2 - b.grad).abs().max().item() / b.grad.abs().max().item()
    assert ew < 1e-5 and eb < 1e-5, (ew, eb)       # fp32 accumulation over up to 150 000 pixels per weight


@pytest.mark.parametrize("chans,normed,Cout,B,H,W", [((32, 32), (), 32, 2, 64, 96), ((38, 64, 64), (0,), 64, 2, 61, 83),
                                                       ((76, 96, 96), (0, 2), 96, 3, 30, 50), ((64, 38), (0, 1), 76, 2, 60, 80),
                                                       ((32,), (), 38, 2, 24, 40)])
def test_conv1x1_weight_gradient_all_blocks_in_one_workgroup(G, chans, normed, Cout, B, H, W):
    """backward.hip conv1x1_wgrad_allpairs_kernel (weight gradient of the 1x1 convs over concatenated slices: models/RITnet_v2.py
    :38,59,61,86 under train.py:285-286): all (32 co, 32 k) blocks from one staging of the pixel chunk, InstanceNorm affine +
    LeakyReLU fused on the slices that carry one, channel tails (38, 76), chunk tails and the small-map fallback to the
    tile-per-workgroup kernel -- weight, bias and data gradients against float64 autograd."""
    from gpu_util import DEV, to_nhwc_buf
    from egne_amd.engine import ConvLayer, Piece, Plan, pad8
    xs = [_rand(G, B, c, H, W) * 1.5 + 0.2 for c in chans]
    w = (_rand(G, Cout, sum(chans), 1, 1) / sum(chans) ** 0.5).double().requires_grad_(True)
    b = _rand(G, Cout).double().requires_grad_(True)
    gy = _rand(G, B, Cout, H, W)
    xd = [x.double().requires_grad_(True) for x in xs]
    xe = [F.leaky_relu(F.instance_norm(x)) if i in normed else x for i, x in enumerate(xd)]
    F.conv2d(torch.cat(xe, 1), w, b).backward(gy.double())
    pl = Plan(torch.device(DEV), train=True)
    pieces = to_nhwc_buf(pl, xs, B, H, W)
    pin = []
    for i, (x, pc) in enumerate(zip(xs, pieces)):
        if i in normed:
            mean, rstd = x.mean((2, 3)), 1 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
            sc, sh = torch.zeros(B, pc.Cp, device=DEV), torch.zeros(B, pc.Cp, device=DEV)
            sc[:, :pc.C], sh[:, :pc.C] = rstd.to(DEV), (-mean * rstd).to(DEV)
            pl.keep += [sc, sh]
            pc = pc.with_norm(sc, sh, 2)
            pc.nograd = True        # (the normalisation backward is covered by the network fixtures)
        pin.append(pc)
    wd, bd = torch.nn.Parameter(w.detach().float().to(DEV)), torch.nn.Parameter(b.detach().float().to(DEV))
    wd.grad, bd.grad = torch.zeros_like(wd), torch.zeros_like(bd)
    layer = ConvLayer([wd], [bd], [(p.C, p.Cp) for p in pin])
    out = pl.buf(B, H, W, pad8(Cout))
    pl.conv(layer, pin, Piece(out, 0, Cout), B, H, W)
    bw = pl.build_backward()
    pl.run()
    pl.gbuf(out)[..., :Cout] = gy.permute(0, 2, 3, 1).to(DEV)
    bw.run()
    torch.cuda.synchronize()
    ew = (wd.grad.cpu().double() - w.grad).abs().max().item() / w.grad.abs().max().item()
    eb = (bd.grad.cpu().double() - b.grad).abs().max().item() / b.grad.abs().max().item()
    assert ew < 1e-5 and eb < 1e-5, (ew, eb)
    for i, (pc, x) in enumerate(zip(pieces, xd)):
        if i in normed:
            continue
        gx = pl.gbuf(pc.buf).cpu()[..., pc.off:pc.off + pc.C].permute(0, 3, 1, 2).double()
        assert (gx - x.grad).abs().max().item() / x.grad.abs().max().item() < 1e-5


@pytest.mark.parametrize("B,H,W", [(4, 240, 320), (2, 61, 83), (3, 100, 100)])
def test_dataprep_spatial_weights_matches_its_restatement(G, B, H, W):
    """dataprep.hip egne_spatial_weights (CurriculumLib.py:128-129) against oracle.dataprep.spatial_weights, bit
    for bit, on synthetic eye labels (nested ellipses, some classes absent) and on random blobs.  PARITY UNPINNED: the oracle
    restates OpenCV's published Canny / dilate and could not be run against OpenCV itself."""
    from gpu_util import DEV
    from egne_amd import dataprep, synth
    from oracle import dataprep as oprep
    labs = []
    if (H, W) == (240, 320):
        b = synth.make_batch(B, seed=7, mask_absent_every=3)
        labs.append(b["label"].numpy().astype(np.int64))
    rng = np.random.RandomState(5)
    yy, xx = np.mgrid[0:H, 0:W]
    blobs = np.zeros((B, H, W), np.int64)
    for i in range(B):
        for k in range(6):
            cy, cx, ry, rx = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(3, H / 3), rng.uniform(3, W / 3)
            blobs[i][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = rng.randint(0, 3)
    labs.append(blobs)
    labs.append((rng.rand(B, H, W) < 0.5).astype(np.int64) * 2)          # salt and pepper: every branch of the suppression, long chains
    for lab in labs:
        want = oprep.spatial_weights(lab)
        got = dataprep.spatial_weights(torch.from_numpy(lab).to(DEV)).cpu().numpy()
        assert got.dtype == np.float32 and np.array_equal(got, want)

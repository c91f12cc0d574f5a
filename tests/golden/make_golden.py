#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE itself (CPU) in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Needs /root/reference (read-only) -- it therefore only runs in the container, never on the GPU
box; the .npz files it writes are committed.  Fixtures hold data only (inputs' checksums and
expected outputs); weights and inputs are regenerated from seeds on the consuming side with
``egne_amd.synth`` (the same generator is used here to load the reference modules).
"""
import contextlib
import hashlib
import io
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_shim  # noqa: E402
import egne_amd  # noqa: E402,F401
from egne_amd import synth  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)
REF = _ref_shim.reference_modules()
CFG_DIR = os.path.join(ROOT, egne_amd.PKG_DIRNAME if hasattr(egne_amd, "PKG_DIRNAME") else
                       "edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd", "configs")


def sha(t):
    a = t.detach().cpu().contiguous().numpy() if torch.is_tensor(t) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-34s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def load_setting(name):
    with open(os.path.join(CFG_DIR, name + ".yaml")) as f:
        return yaml.safe_load(f)


def ref_bdcn(seed=0):
    m = quiet(REF["bdcn_new"].BDCN)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=seed, kind="bdcn"))
    return m.eval()


def ref_esf(setting, variant="v2", seed=0, disentangle=False, nsets=4):
    mod = REF["RITnet_v2"] if variant == "v2" else REF["RITnet_concat"]
    m = quiet(mod.DenseNet2D, dict(setting))
    if disentangle:
        m.disentangle = True
        m.setDatasetInfo(nsets)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=seed, kind="esf"))
    return m


def batch_args(b, edge):
    return (b["img"], edge, b["label"], b["pupil_center"], b["elNorm"], b["spatWts"],
            b["distMap"], b["cond"], b["ID"], b["alpha"])


# ------------------------------------------------------------------------------------------
def gold_bdcn():
    bd = ref_bdcn()
    b = synth.make_batch(2, seed=1234)
    x = b["img"]
    with torch.no_grad():
        x3 = torch.cat((x, x, x), 1)
        outs = bd(x3)
        feats = bd.features(x3)
    arrs = dict(img_sha=sha(x), fuse=npy(outs[-1]))
    for i, o in enumerate(outs[:-1]):
        arrs["map%d_sub" % i] = npy(o[:, :, ::8, ::8])
        arrs["map%d_sum" % i] = npy(o.double().sum((1, 2, 3)))
    arrs["feat_mean"] = np.array([f.double().mean().item() for f in feats])
    arrs["feat_absmax"] = np.array([f.abs().max().item() for f in feats])
    save("bdcn_b2_240x320", **arrs)

    # odd size + genuinely 3-channel input: exercises ceil_mode pools and crop offsets
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 3, 100, 100, generator=g)
    with torch.no_grad():
        outs = bd(x)
    save("bdcn_b1_100x100", x=npy(x), **{"map%d" % i: npy(o) for i, o in enumerate(outs)})
    return bd


def esf_case(name, cfg, variant, b, edge, full_op, train=True, disentangle=False, overrides=None):
    setting = load_setting(cfg)
    setting.update(overrides or {})
    m = ref_esf(setting, variant, disentangle=disentangle)
    args = batch_args(b, edge)
    m.eval()
    with torch.no_grad():
        op, elPred, latent, loss, elOut = quiet(m, *args)
    arrs = dict(cfg=cfg, variant=variant, B=b["img"].shape[0],
                img_sha=sha(b["img"]), edge_sha=sha(edge), dist_sha=sha(b["distMap"]),
                elOut=npy(elOut), elPred=npy(elPred), latent=npy(latent), loss=npy(loss),
                op_sum=npy(op.double().sum((2, 3))), op_abs=npy(op.double().abs().sum((2, 3))),
                mask=np.packbits(npy(op.max(1)[1]).astype(np.uint8) == 1),
                mask2=np.packbits(npy(op.max(1)[1]).astype(np.uint8) == 2))
    arrs["op"] = npy(op) if full_op else npy(op[:, :, ::4, ::4])
    # top-2 logit gap, to list near-tie pixels where argmax may legitimately differ
    srt = op.sort(dim=1, descending=True)[0]
    arrs["gap_lt_2e3"] = np.array(int(((srt[:, 0] - srt[:, 1]) < 2e-3).sum()))
    if train:
        m.train()
        m.zero_grad()
        op, elPred, latent, loss, elOut = quiet(m, *args)
        loss.sum().backward()
        arrs.update(t_loss=npy(loss), t_elOut=npy(elOut), t_latent=npy(latent),
                    t_op_sub=npy(op[:, :, ::4, ::4]),
                    t_head_rm=npy(m.enc.head.bn.running_mean), t_head_rv=npy(m.enc.head.bn.running_var),
                    t_final_rm=npy(m.dec.final.bn.running_mean), t_final_rv=npy(m.dec.final.bn.running_var))
        names, gsum, gl2 = [], [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            gsum.append(p.grad.double().sum().item())
            gl2.append(p.grad.double().norm().item())
        arrs.update(grad_names=np.array(names), grad_sum=np.array(gsum), grad_l2=np.array(gl2))
        for k in ("elReg.l2.weight", "dec.final.conv2.weight", "enc.head.conv1.weight",
                  "enc.down_block1.conv21.weight", "dec.up_block4.conv11.bias"):
            arrs["grad::" + k] = npy(dict(m.named_parameters())[k].grad)
        # one Adam step as train.py:148,285-287 (lr 5e-4, defaults); dsIdentify params excluded
        opt = torch.optim.Adam([p for n, p in m.named_parameters() if "dsIdentify" not in n], lr=5e-4)
        opt.step()
        arrs["adam_names"] = np.array([n for n, _ in m.named_parameters()])
        arrs["adam_sum"] = np.array([p.double().sum().item() for _, p in m.named_parameters()])
        arrs["adam::enc.head.conv1.weight"] = npy(m.enc.head.conv1.weight)
    save(name, **arrs)


def gold_esf(bd):
    b = synth.make_batch(2, seed=1234)
    with torch.no_grad():
        edge = bd(torch.cat((b["img"],) * 3, 1))[-1]
    esf_case("esf_edge_b2", "baseline_edge", "v2", b, edge, True)
    esf_case("esf_adain_edge_b2", "baseline_adain_edge", "v2", b, edge, False)
    esf_case("esf_baseline_b2", "baseline", "v2", b, edge, False)
    esf_case("esf_adain_b2", "baseline_adain", "v2", b, edge, False, train=False)
    esf_case("esf_input_concat_b2", "baseline_input_concat", "v2", b, edge, False, train=False)
    esf_case("esf_only_edge_b2", "baseline_only_edge", "v2", b, edge, False, train=False)
    esf_case("esf_concat_b2", "baseline_edge", "concat", b, edge, False)
    esf_case("esf_edge_disent_b2", "baseline_edge", "v2", b, edge, False, disentangle=True)
    # mask-absent sample in the batch (cond[:,1:4]=1 for sample 1) and an all-absent batch
    b2 = synth.make_batch(2, seed=4321, mask_absent_every=2)
    with torch.no_grad():
        e2 = bd(torch.cat((b2["img"],) * 3, 1))[-1]
    esf_case("esf_edge_b2_absent1", "baseline_edge", "v2", b2, e2, False)
    b3 = synth.make_batch(2, seed=99, mask_absent_every=1)
    with torch.no_grad():
        e3 = bd(torch.cat((b3["img"],) * 3, 1))[-1]
    esf_case("esf_edge_b2_absent_all", "baseline_edge", "v2", b3, e3, False)
    # B=1 exactly as evaluate.py:112-131 builds its arguments
    b1 = synth.make_batch(1, seed=555)
    H, W = b1["img"].shape[-2:]
    lab = torch.zeros((1, H, W))
    lab[..., 0, 2] = 1
    lab[..., 2, 2] = 2
    b1.update(label=lab.long(), pupil_center=torch.zeros(1, 2), elNorm=torch.zeros(1, 2, 5),
              spatWts=torch.zeros(1, H, W), distMap=torch.zeros(1, 3, H, W), cond=torch.zeros(1, 4),
              ID=0, alpha=0)
    with torch.no_grad():
        e1 = bd(torch.cat((b1["img"],) * 3, 1))[-1]
    esf_case("esf_edge_b1_eval", "baseline_edge", "v2", b1, e1, False, train=False)


def gold_esf_adain_train(bd):
    """Training fixtures of the AdaIN fusion path: image-only AdaIN (baseline_adain) and the seg_detach switch
    (RITnet_v2.py:291-292).  Separate target so that the other fixtures need no regeneration."""
    b = synth.make_batch(2, seed=1234)
    with torch.no_grad():
        edge = bd(torch.cat((b["img"],) * 3, 1))[-1]
    esf_case("esf_adain_b2_train", "baseline_adain", "v2", b, edge, False)
    esf_case("esf_adain_edge_detach_b2", "baseline_adain_edge", "v2", b, edge, False, overrides={"seg_detach": 1})


def gold_dp(bd):
    """DataParallel semantics of train.py:205,285 with two replicas, from the reference modules: the batch is scattered
    (here: shard 0 = seed 1234, shard 1 = seed 4321 with one mask-absent sample), every replica runs the forward with ITS
    OWN BatchNorm batch statistics and ITS OWN loss normalisation, the caller takes loss.mean() over replicas, so the
    parameter gradient is the AVERAGE of the per-shard gradients.  Replica 0's BatchNorm buffers are the ones kept."""
    setting = load_setting("baseline_edge")
    shards = [synth.make_batch(2, seed=1234), synth.make_batch(2, seed=4321, mask_absent_every=2)]
    grads, losses, rm = [], [], None
    for i, b in enumerate(shards):
        m = ref_esf(setting)            # replicas start from the same (seeded) parameters
        with torch.no_grad():
            edge = bd(torch.cat((b["img"],) * 3, 1))[-1]
        m.train()
        m.zero_grad()
        out = quiet(m, *batch_args(b, edge))
        (out[3].sum() / len(shards)).backward()          # loss.mean() over the replicas
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        losses.append(npy(out[3]))
        if i == 0:
            rm = (npy(m.enc.head.bn.running_mean), npy(m.enc.head.bn.running_var))
    names = sorted(grads[0])
    tot = {k: grads[0][k] + grads[1][k] for k in names}
    arrs = dict(loss=np.stack(losses), grad_names=np.array(names),
                grad_l2=np.array([tot[k].double().norm().item() for k in names]),
                grad_sum=np.array([tot[k].double().sum().item() for k in names]),
                head_rm=rm[0], head_rv=rm[1])
    for k in ("elReg.l2.weight", "dec.final.conv2.weight", "enc.head.conv1.weight", "enc.down_block1.conv21.weight"):
        arrs["grad::" + k] = npy(tot[k])
    save("dp_two_shards", **arrs)


def prep_labels():
    """Label maps for the data-prep fixture: synthetic eyes (one with the mask absent = all background), a frame filled by
    one class, random blobs with thin structures, a single-pixel class."""
    b = synth.make_batch(3, seed=2024, mask_absent_every=3)
    lab = [b["label"].numpy()[i].astype(np.int64) for i in range(3)]
    lab[2] = np.zeros_like(lab[2])                      # mask absent: classes 1 and 2 missing, class 0 everywhere
    rng = np.random.RandomState(7)
    blobs = (rng.rand(240, 320) > 0.97).astype(np.int64)
    blobs[100:103, :] = 2
    blobs[:, 17] = 1
    lab.append(blobs)
    one = np.ones((240, 320), np.int64)
    one[5, 300] = 2
    lab.append(one)
    return np.stack(lab)


def gold_prep():
    """helperfunctions.one_hot2dist (called per class from CurriculumLib.py:131-136) and the z-score of :139."""
    hf = REF["hf"]
    lab = prep_labels()
    dist = np.zeros((lab.shape[0], 3) + lab.shape[1:], np.float64)
    for i in range(lab.shape[0]):
        for c in range(3):
            dist[i, c] = hf.one_hot2dist(lab[i].astype(np.uint8) == c)
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=(3, 240, 320)).astype(np.uint8)
    img[1] = (img[1] // 4 + 20).astype(np.uint8)
    z = np.stack([(im - im.mean()) / im.std() for im in img])        # CurriculumLib.py:139 verbatim expression on uint8 frames
    save("dataprep", label=lab.astype(np.uint8), dist=dist.astype(np.float32), img=img, z=z.astype(np.float32))


def gold_losses():
    L = REF["loss"]
    g = torch.Generator().manual_seed(5)
    B, H, W = 4, 48, 64
    op = 2 * torch.randn(B, 3, H, W, generator=g)
    tgt = torch.randint(0, 3, (B, H, W), generator=g)
    tgt[1][tgt[1] == 2] = 1          # sample 1: pupil class absent
    tgt[2][tgt[2] == 0] = 2          # sample 2: background absent
    sw = 1 + 20 * (torch.rand(B, H, W, generator=g) > 0.9).float()
    dist = torch.randn(B, 3, H, W, generator=g)
    gt = torch.rand(B, 2, generator=g) * 2 - 1
    arrs = dict(op=npy(op), tgt=npy(tgt).astype(np.uint8), sw=npy(sw), dist=npy(dist), gt=npy(gt))
    l, p = L.get_seg2ptLoss(op[:, 2], gt, temperature=4)
    arrs.update(s2p_loss=npy(l), s2p_pts=npy(p))
    l, p = L.get_seg2ptLoss(-op[:, 0], gt, temperature=4)
    arrs.update(s2p_iri_loss=npy(l), s2p_iri_pts=npy(p))
    arrs["surface"] = np.array([L.SurfaceLoss(op[i:i + 1], dist[i:i + 1]).item() for i in range(B)])
    arrs["gdice"] = np.array([L.GDiceLoss(op[i:i + 1], tgt[i:i + 1], torch.nn.functional.softmax).item()
                              for i in range(B)])
    arrs["wce"] = np.array([L.wCE(op[i], tgt[i], sw[i]).item() for i in range(B)])
    for nm, cond in (("all", [1, 1, 1, 1]), ("some", [1, 0, 1, 0]), ("none", [0, 0, 0, 0])):
        c = torch.tensor(cond, dtype=torch.float32)
        v = L.get_segLoss(op, tgt, sw, dist, c, 0.3)
        arrs["segloss_" + nm] = np.array(float(v))
        v = L.get_ptLoss(op[:, :, 0, 0:10].reshape(B, -1)[:, :10], dist[:, 0, 0, :10], c)
        arrs["ptloss_" + nm] = np.array(float(v))
    x = torch.randn(6, 4, generator=g)
    arrs["conf_in"] = npy(x)
    arrs["conf_true"] = np.array(L.conf_Loss(x, torch.tensor([0, 1, 2, 3, 0, 1]), True).item())
    arrs["conf_false"] = np.array(L.conf_Loss(x, torch.tensor([0, 1, 2, 3, 0, 1]), False).item())
    save("loss_cases", **arrs)


def gold_fit():
    U = REF["utils"]
    rng = np.random.RandomState(3)
    H, W = 240, 320
    masks, inits, outs = [], [], []
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(24):
        cx, cy = rng.uniform(90, 230), rng.uniform(70, 170)
        a, b = rng.uniform(12, 70), rng.uniform(12, 70)
        th = rng.uniform(-1.4, 1.4)
        X = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        Y = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m = (X / a) ** 2 + (Y / b) ** 2 <= 1
        if i % 4 == 1:   # occlude the top (eyelid)
            m[: int(cy - 0.4 * b)] = False
        if i % 4 == 2:   # speckle
            m ^= rng.rand(H, W) > 0.995
        if i == 23:      # empty mask -> nan scores, search must stop after one sweep
            m[:] = False
        init = np.array([cx + rng.uniform(-3, 3), cy + rng.uniform(-3, 3), a * rng.uniform(0.8, 1.25),
                         b * rng.uniform(0.8, 1.25), th + rng.uniform(-0.3, 0.3)])
        res = U.search_proper_parameter_iou_for_our_data(torch.from_numpy(m), init.copy())
        masks.append(np.packbits(m))
        inits.append(init)
        outs.append(res)
    # calc_ell_iou scores for a few raw evaluations (pins the IoU map itself)
    mesh = U.create_meshgrid(H, W, normalized_coordinates=True)
    m0 = np.unpackbits(masks[0]).reshape(H, W).astype(bool)
    sc = []
    for i in range(8):
        el = np.concatenate([inits[i][:4], [inits[i][4] * 180. / 3.14159]])
        sc.append(U.calc_ell_iou(torch.from_numpy(m0), el.copy(), mesh, False, True))
    save("fit_cases", masks=np.stack(masks), inits=np.stack(inits), outs=np.stack(outs),
         iou0=np.array(sc))
    # my_ellipse.transform on its own (evaluate.py:141-146 un-normalisation)
    Hm = np.array([[W / 2, 0, W / 2], [0, H / 2, H / 2], [0, 0, 1]])
    prm = np.stack([[rng.uniform(-.5, .5), rng.uniform(-.5, .5), rng.uniform(.1, .5), rng.uniform(.1, .5),
                     rng.uniform(-1.5, 1.5)] for _ in range(16)])
    tr = np.stack([REF["hf"].my_ellipse(p).transform(Hm)[0][:-1] for p in prm])
    save("ellipse_transform", params=prm, out=tr)


def gold_metrics():
    U = REF["utils"]
    rng = np.random.RandomState(11)
    B, H, W = 4, 60, 80
    yt = rng.randint(0, 3, (B, H, W))
    yp = np.where(rng.rand(B, H, W) < 0.8, yt, rng.randint(0, 3, (B, H, W)))
    yt[2][yt[2] == 2] = 1
    cond = np.array([0, 0, 0, 1], dtype=np.float32)
    miou, pc, sl = U.getSeg_metrics(yt, yp, cond)
    a = rng.rand(B, 2) * 100
    p = rng.rand(B, 2) * 2 - 1
    d, dv = U.getPoint_metric(a, p, cond, (H, W), True)
    d2, dv2 = U.getPoint_metric(a, a + 1.5, cond, (H, W), False)
    logits = rng.randn(B, 3, H, W).astype(np.float32)
    logits[0, :, 0, 0] = 1.0  # exact tie -> first index wins
    pred = U.get_predictions(torch.from_numpy(logits)).numpy()
    save("metrics", yt=yt.astype(np.uint8), yp=yp.astype(np.uint8), cond=cond, miou=np.array(miou), perclass=pc,
         scorelist=sl, pts_true=a, pts_pred=p, pdist=np.array(d), pdist_v=dv, pdist2=np.array(d2), pdist2_v=dv2,
         logits=logits, pred=pred.astype(np.uint8),
         norm=U.normPts(torch.from_numpy(a.astype(np.float32)), (H, W)).numpy(),
         unnorm=U.unnormPts(p.astype(np.float32), (H, W)))


def gold_evaluate(bd):
    """evaluate.py:112-166 on REAL frames of videos/example1.avi (two eyes of two frames), seeded weights:
    z-scored frame -> edge -> logits -> argmax -> un-normalised ellipses -> fitted ellipses."""
    import io as _io
    from PIL import Image
    U, HF = REF["utils"], REF["hf"]
    data = open("/root/reference/videos/example1.avi", "rb").read()
    frames, pos = [], 0
    while len(frames) < 120:
        a = data.find(b"\xff\xd8\xff", pos)
        b = data.find(b"\xff\xd9", a)
        pos = b + 2
        frames.append(np.asarray(Image.open(_io.BytesIO(data[a:b + 2])).convert("L")))
    eyes = np.stack([frames[j][:, 320 * i: 320 * (i + 1)] for j in (0, 119) for i in (0, 1)])   # [4,240,320] uint8
    m = ref_esf(load_setting("baseline_edge")).eval()
    H, W = 240, 320
    Hm = np.array([[W / 2, 0, W / 2], [0, H / 2, H / 2], [0, 0, 1]])
    masks, inits, fits, ops, gaps = [], [], [], [], []
    for e in eyes:
        img = e.astype(np.float64)
        img = (img - img.mean()) / img.std()                      # evaluate.py:102
        x = torch.from_numpy(img).unsqueeze(0).to(torch.float32).unsqueeze(0)
        with torch.no_grad():
            edge = bd(torch.cat((x, x, x), 1))[-1]
            lab = torch.zeros((1, H, W)); lab[..., 0, 2] = 1; lab[..., 2, 2] = 2
            out = quiet(m, x, edge, lab.long(), torch.zeros(1, 2), torch.zeros(1, 2, 5), torch.zeros(1, H, W),
                        torch.zeros(1, 3, H, W), torch.zeros(1, 4), 0, 0)
        seg = U.get_predictions(out[0]).squeeze()
        srt = out[0].sort(dim=1, descending=True)[0]
        gaps.append(int(((srt[:, 0] - srt[:, 1]) < 2e-3).sum()))
        elp = out[1].squeeze().numpy()
        ini_p = HF.my_ellipse(elp[5:10]).transform(Hm)[0][:-1]
        ini_i = HF.my_ellipse(elp[0:5]).transform(Hm)[0][:-1]
        fit_i = U.search_proper_parameter_iou_for_our_data(seg == 1, ini_i.copy())
        fit_p = U.search_proper_parameter_iou_for_our_data(seg == 2, ini_p.copy())
        masks.append(np.packbits(seg.numpy().astype(np.uint8) == 1)); masks.append(np.packbits(seg.numpy().astype(np.uint8) == 2))
        inits.append(np.stack([ini_i, ini_p])); fits.append(np.stack([fit_i, fit_p])); ops.append(npy(out[0][0, :, ::4, ::4]))
    save("evaluate_real_frames", eyes=eyes, masks=np.stack(masks), inits=np.stack(inits), fits=np.stack(fits),
         op_sub=np.stack(ops), gap_lt_2e3=np.array(gaps))


def gold_ritnet_v1():
    """The comparator model models/RITnet_v1.py (modelSummary.py:18-26 'ritnet_v1') on the golden batch: eval outputs, training
    loss, BatchNorm running statistics, per-parameter gradient norms and a few full gradients."""
    with contextlib.redirect_stdout(io.StringIO()):
        from models import RITnet_v1 as R1
    m = quiet(R1.DenseNet2D)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=0, kind="esf"))
    b = synth.make_batch(2, seed=1234)
    args = batch_args(b, torch.zeros_like(b["img"]))
    m.eval()
    with torch.no_grad():
        op, elPred, latent, loss, elOut = quiet(m, *args)
    srt = op.sort(dim=1, descending=True)[0]
    arrs = dict(B=2, img_sha=sha(b["img"]), elOut=npy(elOut), elPred=npy(elPred), latent=npy(latent), loss=npy(loss),
                op=npy(op[:, :, ::4, ::4]), op_sum=npy(op.double().sum((2, 3))), op_absmax=np.array(op.abs().max().item()),
                mask=np.packbits(npy(op.max(1)[1]).astype(np.uint8) == 1), mask2=np.packbits(npy(op.max(1)[1]).astype(np.uint8) == 2),
                gap_lt_2e3=np.array(int(((srt[:, 0] - srt[:, 1]) < 2e-3).sum())))
    m.train()
    m.zero_grad()
    op, elPred, latent, loss, elOut = quiet(m, *args)
    loss.sum().backward()
    arrs.update(t_loss=npy(loss), t_elOut=npy(elOut), t_latent=npy(latent), t_op_sub=npy(op[:, :, ::4, ::4]),
                t_bn1_rm=npy(m.enc.down_block1.bn.running_mean), t_bn1_rv=npy(m.enc.down_block1.bn.running_var),
                t_bn5_rm=npy(m.enc.down_block5.bn.running_mean), t_bn5_rv=npy(m.enc.down_block5.bn.running_var))
    names, gl2 = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            gl2.append(p.grad.double().norm().item())
    arrs.update(grad_names=np.array(names), grad_l2=np.array(gl2))
    for k in ("elReg.l2.weight", "dec.final.weight", "enc.down_block1.conv1.weight", "enc.down_block3.conv31.weight", "dec.up_block4.conv11.bias"):
        arrs["grad::" + k] = npy(dict(m.named_parameters())[k].grad)
    save("ritnet_v1_b2", **arrs)
    import json
    keys = os.path.join(HERE, "state_keys.json")
    d = json.load(open(keys))
    d["ritnet_v1"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(keys, "w") as f:
        json.dump(d, f, indent=0)


def gold_deepvog():
    """The comparator model models/deepvog_pytorch.py (modelSummary.py:26 'deepvog') in evaluation mode on the golden batch, with
    non-trivial BatchNorm statistics; one frame has its mask marked absent (cond[:,1] = 1)."""
    with contextlib.redirect_stdout(io.StringIO()):
        from models import deepvog_pytorch as DV
    m = quiet(DV.DeepVOG_pytorch)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=1, kind="esf"))
    arrs = {}
    for tag, B, absent in (("b2", 2, ()), ("b3", 3, (1,))):
        b = synth.make_batch(B, seed=1234)
        for i in absent:
            b["cond"][i, 1] = 1.0
        m.eval()
        with torch.no_grad():
            out, elPred, emb, loss, _ = quiet(m, *batch_args(b, torch.zeros_like(b["img"])))
        srt = out.sort(dim=1, descending=True)[0]
        arrs.update({tag + "_img_sha": sha(b["img"]), tag + "_loss": npy(loss), tag + "_pred_c": npy(elPred[:, :2]),
                     tag + "_pred_c2": npy(elPred[:, 5:7]), tag + "_emb": npy(emb), tag + "_op": npy(out[:, :, ::4, ::4]),
                     tag + "_op_sum": npy(out.double().sum((2, 3))), tag + "_op_absmax": np.array(out.abs().max().item()),
                     tag + "_mask": np.packbits(npy(out.max(1)[1]).astype(np.uint8) == 1),
                     tag + "_gap_lt_2e3": np.array(int(((srt[:, 0] - srt[:, 1]) < 2e-3).sum()))})
    # training mode (batch statistics) on the three-frame case: loss, running statistics of the first and a deep BatchNorm, every
    # parameter's gradient norm and a few full gradients (up_block5.conv2 / bn2 are built but unused: no gradient)
    b = synth.make_batch(3, seed=1234)
    b["cond"][1, 1] = 1.0
    m.train()
    m.zero_grad()
    out, elPred, emb, loss, _ = quiet(m, *batch_args(b, torch.zeros_like(b["img"])))
    loss.sum().backward()
    arrs.update(t_loss=npy(loss), t_op_sub=npy(out[:, :, ::4, ::4]), t_op_absmax=np.array(out.abs().max().item()),
                t_bn1_rm=npy(m.down_block1.bn1.running_mean), t_bn1_rv=npy(m.down_block1.bn1.running_var),
                t_bnu_rm=npy(m.up_block2.bn2.running_mean), t_bnu_rv=npy(m.up_block2.bn2.running_var))
    names, gl2 = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            gl2.append(p.grad.double().norm().item())
    arrs.update(grad_names=np.array(names), grad_l2=np.array(gl2))
    for k in ("down_block1.conv1.weight", "down_block2.conv2.weight", "up_block1.bn2.bias", "conv1.weight", "down_block4.bn2.weight"):
        arrs["grad::" + k] = npy(dict(m.named_parameters())[k].grad)
    save("deepvog_b2", **arrs)
    import json
    keys = os.path.join(HERE, "state_keys.json")
    d = json.load(open(keys))
    d["deepvog"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(keys, "w") as f:
        json.dump(d, f, indent=0)


def gold_augment():
    """data_augment.augment (:12-130) for the branches that do not call into OpenCV (0 flip, 3 exposure, 4 noise, 7 none) and the
    random branch selection itself: the outputs of the reference for seeded np.random states.  Branch 2 goes through cv2.LUT,
    which the shim does not have: only its table (data_augment.py:47) is recorded."""
    with contextlib.redirect_stdout(io.StringIO()):
        import data_augment as DA
    arrs = {}
    cases = [(0, 7, 100), (0, 9, 101), (3, 7, 102), (3, 8, 103), (4, 7, 104), (4, 9, 105), (7, 8, 106)]
    # randomly selected branch (choice=None): np.random seeds whose first draw (data_augment.py:23) lands on a NumPy branch
    want = {0: 1, 3: 2, 4: 2, 7: 1}
    for s in range(200):
        c = int(np.random.RandomState(s).randint(0, 8))
        if want.get(c, 0) > 0:
            want[c] -= 1
            cases.append((-1 - c, 7 + s % 3, s))
    for n, (choice, seed, npseed) in enumerate(cases):
        base, mask, pc, el = synth.augment_case(seed)
        np.random.seed(npseed)
        ob, om, opc, (op_, oi) = DA.augment(base.copy(), mask.copy(), pc.copy(), el.copy(), choice=choice if choice >= 0 else None)
        arrs["c%d_img_sha" % n] = np.array(sha(ob))
        arrs["c%d_img_rows" % n] = ob[::16]
        arrs["c%d_mask_sha" % n] = np.array(sha(om.astype(np.int64)))
        arrs["c%d_pc" % n] = np.asarray(opc, np.float64)
        arrs["c%d_el" % n] = np.stack([op_, oi]).astype(np.float64)
    arrs["cases"] = np.array(cases)
    for g in (0.6, 0.8, 1.2, 1.4):
        arrs["gamma_table_%d" % int(g * 10)] = 255.0 * (np.linspace(0, 1, 256) ** g)
    save("augment", **arrs)


def gold_keys():
    """Checkpoint key schema (name -> shape) of every reference module on the path."""
    import json
    out = {"bdcn": {k: list(v.shape) for k, v in ref_bdcn().state_dict().items()}}
    for cfg in ("baseline", "baseline_edge", "baseline_adain", "baseline_adain_edge", "baseline_input_concat"):
        m = ref_esf(load_setting(cfg))
        out["v2:" + cfg] = {k: list(v.shape) for k, v in m.state_dict().items()}
    m = ref_esf(load_setting("baseline_edge"), disentangle=True, nsets=4)
    out["v2:baseline_edge:disentangle4"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    m = ref_esf(load_setting("baseline_edge"), variant="concat")
    out["concat:baseline_edge"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote state_keys.json")


if __name__ == "__main__":
    what = sys.argv[1:] or ["bdcn", "esf", "adain", "dp", "prep", "loss", "fit", "metrics", "keys", "evaluate", "ritnet_v1", "augment", "deepvog"]
    bd = None
    if "bdcn" in what:
        bd = gold_bdcn()
    if "esf" in what:
        gold_esf(bd or ref_bdcn())
    if "adain" in what:
        gold_esf_adain_train(bd or ref_bdcn())
    if "dp" in what:
        gold_dp(bd or ref_bdcn())
    if "prep" in what:
        gold_prep()
    if "loss" in what:
        gold_losses()
    if "fit" in what:
        gold_fit()
    if "metrics" in what:
        gold_metrics()
    if "keys" in what:
        gold_keys()
    if "evaluate" in what:
        gold_evaluate(bd or ref_bdcn())
    if "ritnet_v1" in what:
        gold_ritnet_v1()
    if "augment" in what:
        gold_augment()
    if "deepvog" in what:
        gold_deepvog()

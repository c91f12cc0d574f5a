"""Synthetic TEyeD-shaped batches and seeded weights.

The reference ships neither datasets nor checkpoints, so every test, golden vector and
bench run uses (i) a batch tuple with the contract of ``CurriculumLib.py:94-166`` rendered
from random ellipses and (ii) weights regenerated from a seed on both sides (reference
modules in the container, this package's modules on the GPU box).  Host-side numpy/scipy
only; nothing here is on the measured path.
"""
import math
import re
import zlib

import numpy as np
import torch

H_DEF, W_DEF = 240, 320


# --------------------------------------------------------------------------------------
# seeded weights
# --------------------------------------------------------------------------------------
def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))
    return g


def seeded_state_dict(template, seed=0, kind="esf", gain=1.0):
    """Build a name->tensor dict with the keys/shapes of ``template`` (a state_dict).

    Every tensor depends only on (key, shape, seed), so the reference modules and this
    package's modules receive bit-identical parameters without any weight file.
    ``kind='bdcn'`` uses fan-in scaled conv weights so that activations stay O(1) through
    the 13-layer trunk and the fused edge map spans the sigmoid range (an untrained BDCN
    with its N(0,0.01) init returns 0.5 everywhere).
    """
    out = {}
    for key, ref in template.items():
        shape = tuple(ref.shape)
        g = _gen(key, seed)
        leaf = key.split(".")[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=ref.dtype)
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif "upsample" in key:  # BDCN ConvTranspose2d: bilinear kernel + small perturbation
            k = shape[-1]
            f = (k + 1) // 2
            c = f - 0.5 if k % 2 == 0 else f - 1
            r = 1 - (torch.arange(k, dtype=torch.float64) - c).abs() / f
            t = (r[:, None] * r[None, :]).to(torch.float32).reshape(shape)
            t = t * (1 + 0.05 * torch.randn(shape, generator=g))
        elif len(shape) == 4:  # conv weight [O, I, kh, kw]
            o, i, kh, kw = shape
            if kind == "bdcn":
                if "fuse" in key:
                    t = 0.1 + 0.05 * torch.randn(shape, generator=g)
                else:
                    std = gain * math.sqrt(2.0 / (i * kh * kw))
                    if "score_dsn" in key:
                        std *= 0.15  # keeps the side outputs inside the sigmoid's useful range
                    t = std * torch.randn(shape, generator=g)
            else:
                # fan-in scaling: the reference's own fan-out init (RITnet_v2.py:356-369) lets the
                # un-trained logits grow to O(1e4), where the 1e-3 absolute tolerance of the north
                # star is below fp32 round-off; a trained net has O(10) logits, which this mimics
                std = gain * math.sqrt(2.0 / (i * kh * kw))
                t = std * torch.randn(shape, generator=g)
        elif len(shape) == 2:  # linear weight [out, in]
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif len(shape) == 1:
            bn = ".bn." in key or re.search(r"\.bn\d+\.", key) is not None
            if bn and leaf == "weight":
                t = 0.5 + torch.rand(shape, generator=g)
            elif bn and leaf == "bias":
                t = 0.1 * torch.randn(shape, generator=g)
            else:  # conv / linear bias
                t = 0.05 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape)
        out[key] = t.to(ref.dtype)
    return out


# --------------------------------------------------------------------------------------
# synthetic eye frames
# --------------------------------------------------------------------------------------
def _ellipse_mask(H, W, cx, cy, a, b, th):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    X = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
    Y = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
    return (X / a) ** 2 + (Y / b) ** 2 <= 1.0


def _signed_dist(posmask):
    """helperfunctions.py:356-371 (one_hot2dist) with scipy's EDT."""
    from scipy.ndimage import distance_transform_edt as edt
    h, w = posmask.shape
    mx = math.sqrt((h - 1) ** 2 + (w - 1) ** 2)
    if not posmask.any():
        return np.zeros((h, w))
    neg = ~posmask
    return (edt(neg) * neg - (edt(posmask) - 1) * posmask) / mx


def _norm_ellipse(p, H, W):
    """Pixel ellipse (cx,cy,a,b,theta) -> [-1,1] coordinates via the conic form (the map is
    anisotropic, x by 2/W and y by 2/H), with the a<=b convention of
    helperfunctions.py:509-513 and theta in (-pi/2, pi/2]."""
    cx, cy, a, b, th = p
    c, s = math.cos(th), math.sin(th)
    R = np.array([[c, -s], [s, c]])
    Q = R @ np.diag([1 / a ** 2, 1 / b ** 2]) @ R.T  # (x-c)^T Q (x-c) = 1
    S = np.diag([W / 2.0, H / 2.0])                  # x_px - c_px = S (x_n - c_n)
    Qn = S @ Q @ S
    ev, evec = np.linalg.eigh(Qn)                    # ascending eigenvalues -> descending axes
    ax = 1 / np.sqrt(ev)                             # ax[0] >= ax[1]
    # a<=b: first axis is the minor one (eigvec of the larger eigenvalue)
    v = evec[:, 1]
    th_n = math.atan2(v[1], v[0])
    if th_n > math.pi / 2:
        th_n -= math.pi
    if th_n <= -math.pi / 2:
        th_n += math.pi
    return np.array([2 * cx / W - 1, 2 * cy / H - 1, ax[1], ax[0], th_n])


def make_batch(B, H=H_DEF, W=W_DEF, seed=1234, mask_absent_every=8):
    """Returns the dict of CPU tensors every caller feeds the path."""
    rng = np.random.RandomState(seed)
    img = np.zeros((B, 1, H, W), np.float32)
    label = np.zeros((B, H, W), np.int64)
    spat = np.zeros((B, H, W), np.float32)
    dist = np.zeros((B, 3, H, W), np.float32)
    pc = np.zeros((B, 2), np.float32)
    eln = np.zeros((B, 2, 5), np.float32)
    cond = np.zeros((B, 4), np.float32)
    ID = rng.randint(0, 4, size=(B,)).astype(np.int64)
    sc = min(H / 240.0, W / 320.0)
    for i in range(B):
        cx = W * rng.uniform(0.3, 0.7)
        cy = H * rng.uniform(0.3, 0.7)
        ia, ib = sc * rng.uniform(55, 75), sc * rng.uniform(55, 75)
        ith = rng.uniform(-1.2, 1.2)
        pr = sc * rng.uniform(18, 30)
        pa, pb = pr * rng.uniform(0.85, 1.0), pr
        pth = rng.uniform(-1.2, 1.2)
        pcx, pcy = cx + sc * rng.uniform(-6, 6), cy + sc * rng.uniform(-6, 6)
        iris = _ellipse_mask(H, W, cx, cy, ia, ib, ith)
        pup = _ellipse_mask(H, W, pcx, pcy, pa, pb, pth)
        lab = np.zeros((H, W), np.int64)
        lab[iris] = 1
        lab[pup] = 2
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        im = 0.75 + 0.15 * (xx / W) - 0.1 * (yy / H)
        im[iris] = 0.45
        im[pup] = 0.08
        im = im + rng.normal(0, 0.05, size=(H, W)).astype(np.float32)
        for _ in range(rng.randint(1, 3)):
            gx = int(pcx + sc * rng.uniform(-25, 25))
            gy = int(pcy + sc * rng.uniform(-25, 25))
            im[max(gy - 3, 0):gy + 3, max(gx - 3, 0):gx + 3] = 1.0
        im = (im - im.mean()) / im.std()
        img[i, 0] = im
        label[i] = lab
        # boundary(label) dilated 3x3 -> weights {1, 21}  (CurriculumLib.py:128-129)
        bnd = np.zeros((H, W), bool)
        bnd[:-1, :] |= lab[:-1, :] != lab[1:, :]
        bnd[1:, :] |= lab[:-1, :] != lab[1:, :]
        bnd[:, :-1] |= lab[:, :-1] != lab[:, 1:]
        bnd[:, 1:] |= lab[:, :-1] != lab[:, 1:]
        dil = bnd.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                sh = np.zeros_like(bnd)
                ys = slice(max(dy, 0), H + min(dy, 0))
                yd = slice(max(-dy, 0), H + min(-dy, 0))
                xs = slice(max(dx, 0), W + min(dx, 0))
                xd = slice(max(-dx, 0), W + min(-dx, 0))
                sh[yd, xd] = bnd[ys, xs]
                dil |= sh
        spat[i] = 1 + 20 * dil
        for c in range(3):
            dist[i, c] = _signed_dist(lab == c)
        pc[i] = (pcx, pcy)
        eln[i, 0] = _norm_ellipse((cx, cy, ia, ib, ith), H, W)
        eln[i, 1] = _norm_ellipse((pcx, pcy, pa, pb, pth), H, W)
        if mask_absent_every and (i % mask_absent_every) == mask_absent_every - 1:
            cond[i, 1:4] = 1.0  # mask + ellipses absent; only the pupil centre is annotated
            eln[i] = -1.0
    return dict(
        img=torch.from_numpy(img), label=torch.from_numpy(label), spatWts=torch.from_numpy(spat),
        distMap=torch.from_numpy(dist), pupil_center=torch.from_numpy(pc),
        elNorm=torch.from_numpy(eln), cond=torch.from_numpy(cond), ID=torch.from_numpy(ID),
        alpha=0.5)


def augment_case(seed=7, H=H_DEF, W=W_DEF):
    """One raw sample as the reference's Dataset hands it to data_augment.augment (CurriculumLib.py:114): uint8 frame, integer mask
    with the ORIGINAL labels 0..3, pupil centre and (pupil, iris) ellipse parameters in pixels / radians.  Every third seed marks the
    pupil as absent (all -1)."""
    rng = np.random.RandomState(seed)
    b = make_batch(1, H, W, seed=1000 + seed)
    im = b["img"][0, 0].numpy()
    base = np.clip((im - im.min()) / (im.max() - im.min()) * 255.0 + rng.uniform(-20, 20), 0, 255).astype(np.uint8)
    mask = b["label"][0].numpy().astype(np.int64)
    mask[mask > 0] += 1                                   # iris 2, pupil 3 as stored; sclera band below
    mask[(mask == 0) & (rng.rand(H, W) < 0.2)] = 1
    pupil_c = np.array([W * rng.uniform(0.3, 0.7), H * rng.uniform(0.3, 0.7)])
    el = np.stack([np.array([pupil_c[0], pupil_c[1], rng.uniform(15, 30), rng.uniform(12, 25), rng.uniform(-1.5, 1.5)]),
                   np.array([W * rng.uniform(0.3, 0.7), H * rng.uniform(0.3, 0.7), rng.uniform(50, 75), rng.uniform(50, 70), rng.uniform(-1.5, 1.5)])])
    if seed % 3 == 0:
        pupil_c[:] = -1
        el[0, :] = -1
    return base, mask, pupil_c, el

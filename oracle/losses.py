"""Oracle for the loss terms of the path (test infrastructure; see oracle/__init__.py).

Batched, sync-free restatements of loss.py / RITnet_v2.get_allLoss.  They follow the
reference's arithmetic exactly (including F8 of SURVEY.md: the "edge-weighted" CE is
``mean(spatWts) * CE_mean``), only the per-sample Python loops are replaced by masks.
"""
import torch
import torch.nn.functional as F


def meshgrid_xy(H, W, dtype=torch.float32):
    """utils.py:27-60 create_meshgrid(normalized): x = linspace(-1,1,W), y = linspace(-1,1,H)."""
    xs = torch.linspace(-1, 1, W, dtype=dtype)
    ys = torch.linspace(-1, 1, H, dtype=dtype)
    return xs[None, :].expand(H, W).reshape(-1), ys[:, None].expand(H, W).reshape(-1)


def norm_pts(pts, H, W):
    """utils.py:627-634 normPts: 2*x/W - 1, 2*y/H - 1."""
    out = pts.clone().reshape(-1, 2)
    out[:, 0] = 2 * (out[:, 0] / W) - 1
    out[:, 1] = 2 * (out[:, 1] / H) - 1
    return out.reshape(pts.shape)


def seg2pt(chan, gt, temperature=4):
    """loss.py:16-46 get_seg2ptLoss: soft-argmax centre of mass and its L1 to ``gt``.
    chan [B,H,W]; gt [B,2]; returns (loss [B,2], pts [B,2])."""
    B, H, W = chan.shape
    wt = F.softmax(chan.reshape(B, -1) * temperature, dim=1)
    xl, yl = meshgrid_xy(H, W, chan.dtype)
    pts = torch.stack([(wt * xl).sum(-1), (wt * yl).sum(-1)], dim=1)
    return (pts - gt).abs(), pts


def surface_loss(op_i, dist_i):
    """loss.py:86-92 SurfaceLoss for one sample: mean_c mean_hw softmax(op)*dist."""
    p = torch.softmax(op_i, dim=0)
    return (p.flatten(1) * dist_i.flatten(1)).mean(1).mean(0)


def gdice_loss(op_i, tgt_i):
    """loss.py:94-121 GDiceLoss for one sample (class weights 1/clamp(n_c^2,1e-5), 0 if absent)."""
    C = op_i.shape[0]
    p = torch.softmax(op_i, dim=0).flatten(1)
    onehot = (torch.arange(C)[:, None] == tgt_i.reshape(1, -1)).to(p.dtype)
    n = onehot.sum(1)
    w = 1.0 / (n ** 2).clamp(1e-5)
    w = torch.where(n > 0, w, torch.zeros_like(w))
    A = (w * (p * onehot).sum(1)).sum()
    Bq = (w * (p + onehot).sum(1)).sum()
    return 1 - (2.0 * A / Bq).clamp(1e-5)


def wce_loss(op_i, tgt_i, sw_i):
    """loss.py:123-137 wCE for one sample: mean(spatWts) * CE_mean(ignore the one absent class)."""
    C = op_i.shape[0]
    present = [(tgt_i == c).any().item() for c in range(C)]
    absent = [c for c in range(C) if not present[c]]
    if len(absent) > 1:
        raise ValueError("wCE supports at most one absent class (loss.py:132 rmIdx.item())")
    kw = dict(ignore_index=absent[0]) if absent else {}
    ce = F.cross_entropy(op_i.reshape(1, C, -1), tgt_i.reshape(1, -1), **kw)
    return (sw_i.reshape(1, -1) * ce).mean()


def seg_loss(op, target, spatWts, distMap, mask_present, alpha):
    """loss.py:48-69 get_segLoss: sum over valid samples / number of valid samples."""
    terms = []
    for i in range(op.shape[0]):
        if mask_present[i] == 1:
            terms.append(alpha * surface_loss(op[i], distMap[i])
                         + (1 - alpha) * gdice_loss(op[i], target[i])
                         + wce_loss(op[i], target[i], spatWts[i]))
    if not terms:
        return 0.0
    return torch.stack(terms).sum() / mask_present.to(torch.float32).sum()


def pt_loss(vec, tgt, valid):
    """loss.py:71-84 get_ptLoss: per-sample mean-L1, summed over valid / count(valid)."""
    terms = [(vec[i] - tgt[i]).abs().mean() for i in range(vec.shape[0]) if valid[i] == 1]
    if not terms:
        return 0.0
    return torch.stack(terms).sum() / valid.to(torch.float32).sum()


def conf_loss(pred_ds, gt, flag=True):
    """loss.py:139-157 conf_Loss: L1(softmax, uniform) when flag else CE."""
    if flag:
        Bn, C = pred_ds.shape
        return (F.softmax(pred_ds, dim=1) - 1.0 / C).abs().mean()
    return F.cross_entropy(pred_ds, gt)


def all_loss(op, elOut, target, pupil_center, elNorm, spatWts, distMap, cond, alpha):
    """models/RITnet_v2.py:372-432 get_allLoss.  Returns (total, pred_c_seg [B,2,2], terms)."""
    B, C, H, W = op.shape
    mask_present = (1 - cond[:, 1]).to(torch.float32)
    pc_n = norm_pts(pupil_center, H, W)
    l_pup, c_pup = seg2pt(op[:, 2], pc_n, 4)
    if mask_present.sum() > 0:
        l_iri, c_iri = seg2pt(-op[:, 0], elNorm[:, 0, :2], 4)
        m2 = torch.stack([mask_present, mask_present], dim=1)
        l_iri = (l_iri * m2).sum() / m2.sum()
    else:
        l_iri = 0.0
        c_iri = elOut[:, 5:7].clone()
    l_pup = l_pup.mean()
    pred_c = torch.stack([c_iri, c_pup], dim=1)
    l_seg2pt = 0.5 * l_pup + 0.5 * l_iri
    l_seg = seg_loss(op, target, spatWts, distMap, mask_present, alpha)
    l_pt = pt_loss(elOut[:, 5:7], pc_n, 1 - mask_present)
    l_ell = pt_loss(elOut, elNorm.reshape(-1, 10), mask_present)
    total = l_seg2pt + 20 * l_seg + 10 * (l_pt + l_ell)
    terms = dict(l_seg2pt=l_seg2pt, l_seg=l_seg, l_pt=l_pt, l_ellipse=l_ell)
    return total, pred_c, terms

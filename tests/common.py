"""Shared helpers of the test-suite: fixtures loading, seeded modules, settings."""
import hashlib
import os

import numpy as np
import torch
import yaml

import egne_amd  # noqa: F401  (registers the package alias)
from egne_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(os.path.dirname(egne_amd.__file__), "configs")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def setting(name):
    """``name`` is a config file stem, optionally followed by ``+key=int`` overrides."""
    name, *ov = name.split("+")
    with open(os.path.join(CFG, name + ".yaml")) as f:
        st = yaml.safe_load(f)
    for kv in ov:
        k, v = kv.split("=")
        st[k] = int(v)
    return st


def sha(t):
    a = t.detach().cpu().contiguous().numpy() if torch.is_tensor(t) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()


def bdcn_module(seed=0):
    from egne_amd.bdcn_new import BDCN
    m = BDCN()
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=seed, kind="bdcn"))
    return m.eval()


def esf_module(cfg, variant="v2", seed=0, disentangle=False, nsets=4):
    if variant == "v2":
        from egne_amd.models.RITnet_v2 import DenseNet2D
    else:
        from egne_amd.models.RITnet_concat import DenseNet2D
    m = DenseNet2D(dict(setting(cfg)))
    if disentangle:
        m.disentangle = True
        m.setDatasetInfo(nsets)
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=seed, kind="esf"))
    return m


def batch_args(b, edge):
    return (b["img"], edge, b["label"], b["pupil_center"], b["elNorm"], b["spatWts"], b["distMap"], b["cond"],
            b["ID"], b["alpha"])


# the golden ESF cases: name -> (cfg, variant, batch kwargs)
ESF_CASES = {
    "esf_edge_b2": ("baseline_edge", "v2", dict(B=2, seed=1234)),
    "esf_baseline_b2": ("baseline", "v2", dict(B=2, seed=1234)),
    "esf_input_concat_b2": ("baseline_input_concat", "v2", dict(B=2, seed=1234)),
    "esf_only_edge_b2": ("baseline_only_edge", "v2", dict(B=2, seed=1234)),
    "esf_concat_b2": ("baseline_edge", "concat", dict(B=2, seed=1234)),
    "esf_adain_edge_b2": ("baseline_adain_edge", "v2", dict(B=2, seed=1234)),
    "esf_adain_b2": ("baseline_adain", "v2", dict(B=2, seed=1234)),
    "esf_adain_b2_train": ("baseline_adain", "v2", dict(B=2, seed=1234)),
    "esf_adain_edge_detach_b2": ("baseline_adain_edge+seg_detach=1", "v2", dict(B=2, seed=1234)),
    "esf_edge_b2_absent1": ("baseline_edge", "v2", dict(B=2, seed=4321, mask_absent_every=2)),
    "esf_edge_b2_absent_all": ("baseline_edge", "v2", dict(B=2, seed=99, mask_absent_every=1)),
}


def eval_b1_batch():
    """Arguments exactly as evaluate.py:112-131 builds them."""
    b1 = synth.make_batch(1, seed=555)
    H, W = b1["img"].shape[-2:]
    lab = torch.zeros((1, H, W))
    lab[..., 0, 2] = 1
    lab[..., 2, 2] = 2
    b1.update(label=lab.long(), pupil_center=torch.zeros(1, 2), elNorm=torch.zeros(1, 2, 5),
              spatWts=torch.zeros(1, H, W), distMap=torch.zeros(1, 3, H, W), cond=torch.zeros(1, 4),
              ID=torch.zeros(1, dtype=torch.long), alpha=0)
    return b1


def mask_mismatch(mask, g, what, budget_key="gap_lt_2e3"):
    """Argmax-mask identity against a fixture, PIXEL by pixel (BASELINE.json: "argmax masks bit-identical"): ``mask`` [B,H,W] class
    indices, fixture ``mask`` / ``mask2`` = np.packbits of (reference argmax == 1) / (== 2).  Returns the number of differing
    pixels and asserts it stays within the fixture's count of near-tie pixels (two largest reference logits within 2e-3) -- and is
    ZERO where the fixture has no near tie.  Every call appends its counts to gpurun_out/mask_mismatch.jsonl (copied to profiles/)."""
    import json
    m = np.asarray(mask).astype(np.uint8)
    r1 = np.unpackbits(g["mask"])[:m.size].reshape(m.shape).astype(bool)
    r2 = np.unpackbits(g["mask2"])[:m.size].reshape(m.shape).astype(bool)
    ref = r1.astype(np.uint8) + 2 * r2.astype(np.uint8)
    ndiff = int(np.count_nonzero(m != ref))
    budget = int(np.asarray(g[budget_key]).sum())
    rec = {"case": what, "pixels": int(m.size), "mismatch_pixels": ndiff, "near_tie_pixels_lt_2e-3": budget}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "mask_mismatch.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    print("mask identity %s: %d of %d pixels differ (near-tie pixels in the fixture: %d)" % (what, ndiff, m.size, budget))
    assert ndiff <= budget, "%s: argmax mask differs in %d pixels, the fixture has only %d near-tie pixels" % (what, ndiff, budget)
    return ndiff

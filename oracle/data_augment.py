"""CPU restatement of the NumPy branches of the reference's data_augment.augment (TEST INFRASTRUCTURE ONLY - never imported by the
product).  Pinned by tests/golden/augment.npz, produced by the reference's own function (make_golden.py, target "augment"), for the
branches 0 (flip), 3 (exposure), 4 (noise) and 7 (none) with explicit and with randomly selected branch.  Branch 2 (gamma) calls
cv2.LUT in the reference: only its table (data_augment.py:47) is pinned, the look-up dst = table[src] is OpenCV's documented
behaviour and unpinned.  Branches 1, 5 and 6 (blur, lines, rotation) are OpenCV throughout and not restated.
"""
import numpy as np


def _absent(v):
    return bool(np.all(np.asarray(v) == -1))


def augment(base, mask, pupil_c, elParam, choice=None):
    """data_augment.py:12-130; draws from the global np.random stream in the reference's order."""
    H, W = base.shape
    pc = np.array(pupil_c, dtype=np.float64)
    pup, iri = np.array(elParam[0], dtype=np.float64), np.array(elParam[1], dtype=np.float64)
    k = int(np.random.randint(0, 8)) if choice is None else int(choice)          # :23
    img, lab = base, mask
    if k == 0:                                                                    # :25-36
        img, lab = base[:, ::-1], mask[:, ::-1]
        if not _absent(pupil_c):
            pc[0] = W - pc[0]
        for e, src in ((pup, elParam[0]), (iri, elParam[1])):
            if not _absent(src):
                e[0], e[4] = W - src[0], -src[4]
    elif k == 2:                                                                  # :44-49
        g = (0.6, 0.8, 1.2, 1.4)[np.random.randint(0, 4)]
        img = (255.0 * (np.linspace(0, 1, 256) ** g))[base]
    elif k == 3:                                                                  # :51-56
        img = np.clip(base.astype(np.float64) + (50 * np.random.rand(1) - 25), 0, 255).astype(np.uint8)
    elif k == 4:                                                                  # :58-65
        std = 14 * np.random.rand() + 2
        img = np.clip(base + np.random.normal(0.0, std, base.shape), 0, 255)
    elif k in (1, 5, 6):
        raise NotImplementedError("OpenCV branch %d is not restated" % k)
    return np.ascontiguousarray(img).astype(np.uint8), np.ascontiguousarray(lab).astype(np.int64), pc, (pup, iri)

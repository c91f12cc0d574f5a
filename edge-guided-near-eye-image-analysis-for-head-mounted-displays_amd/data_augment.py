"""data_augment.py of the reference (augment(), :12-130), batched on the device.

The reference augments ONE sample at a time on the host inside its Dataset (CurriculumLib.py:114-120).  Here a whole batch of
uint8 frames that already sits in HBM is augmented by one launch (csrc/dataprep.hip, egne_augment); the random draws stay on the
host and consume ``np.random`` in exactly the reference's order, so the same seed selects the same augmentation with the same
parameters for every frame.

    choice  reference branch                         here
    0       flip left-right (:25-36)                 device, pinned against the reference (tests/golden/augment.npz)
    1       cv2.GaussianBlur (:38-42)                NOT BUILT (OpenCV is not in the image; no fixture)
    2       gamma through cv2.LUT (:44-49)           device; the 256-entry table is the reference's expression evaluated on the host,
                                                     cv2.LUT itself (a table look-up) is unpinned
    3       exposure +/- 25 (:51-56)                 device, pinned
    4       Gaussian noise (:58-65)                  device; pinned with the host-drawn noise field (``host_noise=True``); by default the
                                                     field is drawn on the device (same distribution, different stream)
    5       cv2.line glints (:67-79)                 NOT BUILT
    6       cv2.warpAffine rotation (:100-119)       NOT BUILT
    >= 7    no change (:121-124)                     device (copy)

``augment`` keeps the reference's per-sample signature for NumPy callers; ``augment_batch`` is the path a device-side loader uses.
"""
import numpy as np
import torch

from . import _lib
from .engine import require_cuda

CV2_CHOICES = (1, 5, 6)
GAMMAS = (0.6, 0.8, 1.2, 1.4)


def gamma_table(gamma):
    """data_augment.py:47 followed by the final astype(np.uint8) of :126 (cv2.LUT returns table[pixel])."""
    return (255.0 * (np.linspace(0, 1, 256) ** gamma)).astype(np.uint8)


def draw(B, shape, choices=None, host_noise=False, on_cv2="raise"):
    """The random draws of ``B`` consecutive augment() calls, in the reference's order per call: the branch index
    (np.random.randint(0, 8), :23), then the branch's own draws.  Returns (choice int32 [B], param float64 [B], lut uint8 [B,256],
    noise float64 [B,H,W] or None).  ``on_cv2``: "raise" or "skip" (a frame that drew a cv2 branch is left unchanged)."""
    H, W = shape
    choice = np.zeros(B, np.int32)
    param = np.zeros(B, np.float64)
    lut = np.tile(np.arange(256, dtype=np.uint8), (B, 1))
    noise = None
    for b in range(B):
        c = int(np.random.randint(0, 8)) if choices is None else int(choices[b])
        if c in CV2_CHOICES:
            if on_cv2 != "skip":
                raise NotImplementedError("augment: branch %d needs OpenCV (blur / lines / rotate are not built)" % c)
            c = 7
        if c == 2:
            lut[b] = gamma_table(GAMMAS[np.random.randint(0, 4)])
        elif c == 3:
            param[b] = (50 * np.random.rand(1) - 25).item()
        elif c == 4:
            std = 14 * np.random.rand() + 2
            if host_noise:
                if noise is None:
                    noise = np.zeros((B, H, W), np.float64)
                noise[b] = np.random.normal(0.0, std, (H, W))      # already scaled: the launch multiplies by 1
                param[b] = 1.0
            else:
                param[b] = std
        choice[b] = min(c, 7)
    return choice, param, lut, noise


def augment_batch(img, label, pupil_c, elParam, choices=None, host_noise=False, on_cv2="raise"):
    """img uint8 [B,H,W] and label int64 [B,H,W] on the device; pupil_c [B,2] and elParam [B,2,5] (pixels, radians) on any device.
    Returns (img, label, pupil_c, elParam, choice) with the geometry of flipped frames mirrored as data_augment.py:29-36 does
    (entries equal to -1 everywhere mark an absent centre / ellipse and stay untouched)."""
    require_cuda(img, "img")
    require_cuda(label, "label")
    if img.dtype != torch.uint8 or img.dim() != 3 or label.dtype != torch.int64 or label.shape != img.shape:
        raise ValueError("augment_batch: img must be uint8 [B,H,W] and label int64 of the same shape")
    img, label = img.contiguous(), label.contiguous()
    B, H, W = img.shape
    dev = img.device
    choice, param, lut, noise = draw(B, (H, W), choices, host_noise, on_cv2)
    if (choice == 4).any() and noise is None:
        noise_d = torch.randn((B, H, W), dtype=torch.float64, device=dev)
    else:
        noise_d = torch.from_numpy(noise).to(dev) if noise is not None else None
    ch_d, p_d, lut_d = (torch.from_numpy(a).to(dev) for a in (choice, param, lut))
    oimg, olab = torch.empty_like(img), torch.empty_like(label)
    _lib.check(_lib.lib().egne_augment(img.data_ptr(), label.data_ptr(), ch_d.data_ptr(), p_d.data_ptr(), lut_d.data_ptr(),
                                       noise_d.data_ptr() if noise_d is not None else None, oimg.data_ptr(), olab.data_ptr(),
                                       B, H, W, _lib.stream_ptr()), "augment")
    pc = torch.as_tensor(pupil_c).clone()
    el = torch.as_tensor(elParam).clone()
    flip = torch.from_numpy(choice == 0).to(pc.device)
    if bool(flip.any()):
        ok_c = flip & ~(pc == -1).all(dim=1)
        pc[:, 0] = torch.where(ok_c, W - pc[:, 0], pc[:, 0])
        for k in range(2):
            ok = flip & ~(el[:, k] == -1).all(dim=1)
            el[:, k, 0] = torch.where(ok, W - el[:, k, 0], el[:, k, 0])
            el[:, k, 4] = torch.where(ok, -el[:, k, 4], el[:, k, 4])
    return oimg, olab, pc, el, choice


def augment(base, mask, pupil_c, elParam, choice=None):
    """The reference's signature (data_augment.py:12): one NumPy frame in, (uint8 image, int mask, centre, (pupil, iris)) out,
    computed on cuda:0 through ``augment_batch`` with the reference's random stream (host-drawn noise)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("augment: no GPU (this path has no CPU fallback)")
    img = torch.from_numpy(np.ascontiguousarray(base, dtype=np.uint8))[None].to(dev)
    lab = torch.from_numpy(np.ascontiguousarray(mask).astype(np.int64))[None].to(dev)
    pc = torch.from_numpy(np.asarray(pupil_c, dtype=np.float64).copy())[None]
    el = torch.from_numpy(np.stack([np.asarray(elParam[0], dtype=np.float64), np.asarray(elParam[1], dtype=np.float64)]))[None]
    oi, ol, pc, el, _ = augment_batch(img, lab, pc, el, None if choice is None else [choice], host_noise=True)
    return oi[0].cpu().numpy(), ol[0].cpu().numpy(), pc[0].numpy(), (el[0, 0].numpy(), el[0, 1].numpy())

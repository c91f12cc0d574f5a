"""Device-side batch preparation: what the reference's Dataset does per sample on the host
(CurriculumLib.py:127-139) for the tensors the hot path consumes, as batched HIP kernels.

    dist_maps(label)  ->  distMap   float32 [B,3,H,W]   helperfunctions.one_hot2dist per class (:356-371), bit-identical
    zscore(img)       ->  img       float32 [B,1,H,W]   (img - img.mean()) / img.std() per image (CurriculumLib.py:139)

    spatial_weights(label) -> spatWts float32 [B,H,W]  1 + 20 * dilate(Canny(label, 0, 1) / 255, (3, 3)) (CurriculumLib.py:128-129);
                                                          PARITY UNPINNED - OpenCV is not available in the build container and the
                                                          reference holds no fixture, so this follows OpenCV's published algorithm
                                                          (restated in oracle/dataprep.py, which the kernel matches bit for bit)
"""
import torch

from . import _lib
from .engine import require_cuda


def dist_maps(label, ncls=3):
    require_cuda(label, "label")
    if label.dtype != torch.int64 or label.dim() != 3:
        raise ValueError("label must be an int64 [B,H,W] tensor")
    label = label.contiguous()
    B, H, W = label.shape
    L = _lib.lib()
    out = torch.empty((B, ncls, H, W), dtype=torch.float32, device=label.device)
    ws = torch.empty(int(L.egne_dist_maps_workspace_bytes(B, H, W, ncls)), dtype=torch.uint8, device=label.device)
    _lib.check(L.egne_dist_maps(label.data_ptr(), B, H, W, ncls, out.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "dist_maps")
    return out


def zscore(img):
    require_cuda(img, "img")
    x = img.to(torch.float32).contiguous()
    B = x.shape[0]
    out = torch.empty_like(x)
    _lib.check(_lib.lib().egne_zscore(x.data_ptr(), out.data_ptr(), B, x.numel() // B, _lib.stream_ptr()), "zscore")
    return out


def spatial_weights(label):
    """CurriculumLib.py:128-129 on the device (parity unpinned, see the module docstring)."""
    require_cuda(label, "label")
    if label.dtype != torch.int64 or label.dim() != 3:
        raise ValueError("label must be an int64 [B,H,W] tensor")
    label = label.contiguous()
    B, H, W = label.shape
    out = torch.empty((B, H, W), dtype=torch.float32, device=label.device)
    _lib.check(_lib.lib().egne_spatial_weights(label.data_ptr(), B, H, W, out.data_ptr(), _lib.stream_ptr()), "spatial_weights")
    return out

#!/usr/bin/env python3
"""Video / frame inference entry point -- the reference's evaluate.py (edge -> seg -> ellipse fit).

The per-image path (evaluate.py:105-166) runs on the HIP kernels; the two ellipses of a frame are fitted
by one device launch.  OpenCV is not available in this image: MJPEG .avi files (the format of
videos/example1.avi) are decoded with PIL, results go to <video>_ellipses.npy instead of an overlay video.

    python evaluate.py --path2data videos [--max_frames 20]
    python evaluate.py --synthetic 4
"""
import argparse
import glob
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egne_amd  # noqa: E402,F401
from egne_amd import _entry  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from egne_amd.utils import fit_ellipses_from_pred  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser()                               # evaluate.py:33-65
    p.add_argument('--vid_ext', type=str, default='avi')
    p.add_argument('--path2data', type=str, default='videos')
    p.add_argument('--align_width', type=int, default=1)
    p.add_argument('--method', type=str, default='baseline')
    p.add_argument('--loadfile', type=str, default='logs/baseline_edge_16.pkl')        # evaluate.py:357
    p.add_argument('--setting', type=str, default='configs/baseline_edge.yaml')        # evaluate.py:359
    p.add_argument('--max_frames', type=int, default=0)
    p.add_argument('--synthetic', type=int, default=0)
    a = p.parse_args(argv)
    a.prec = torch.float32
    return a


def preprocess_frame(img, op_shape, align_width=True):
    """evaluate.py:69-104: fit to 240x320 by width, pad/crop rows, z-score.  Resizing needs cv2's Lanczos;
    frames that already have the target width (the example video: 320 per eye) pass straight through."""
    if not align_width:
        sys.exit('Height alignment not implemented! Exiting ...')
    scale_shift = (1, 0)
    if op_shape[1] != img.shape[1]:
        raise RuntimeError('frame width %d != %d: resizing needs OpenCV (INTER_LANCZOS4), not available here' % (img.shape[1], op_shape[1]))
    if op_shape[0] > img.shape[0]:
        pad = op_shape[0] - img.shape[0]
        img = np.pad(img, ((pad // 2, pad - pad // 2), (0, 0)))
        scale_shift = (1, pad)
    elif op_shape[0] < img.shape[0]:
        cut = img.shape[0] - op_shape[0]
        img = img[cut // 2: cut // 2 + op_shape[0]]
        scale_shift = (1, -cut)
    img = img.astype(np.float64)
    img = (img - img.mean()) / img.std()
    return torch.from_numpy(img).unsqueeze(0).to(torch.float32), scale_shift


def evaluate_ellseg_on_image(frames, model, edge_model, args=None):
    """evaluate.py:112-166 for a batch of frames [N,1,H,W] (the reference loops one frame at a time).
    Returns edge maps [N,H,W], class maps [N,H,W], pupil ellipses [N,5], iris ellipses [N,5] (pixels)."""
    from egne_amd.utils import calc_edge
    assert frames.dim() == 4, 'Frame must be [N,1,H,W]'
    dev = frames.device
    N, _, H, W = frames.shape
    ns = argparse.Namespace(prec=torch.float32, edge_thres=0)
    with torch.no_grad():
        edge = calc_edge(ns, frames, edge_model, dev)
        labels = torch.zeros((N, H, W), device=dev)
        labels[..., 0, 2] = 1                                  # evaluate.py:118-120: make all 3 classes present
        labels[..., 2, 2] = 2
        z = lambda *s: torch.zeros(s, device=dev)              # noqa: E731
        out = model(frames, edge, labels.long(), z(N, 2), z(N, 2, 5), z(N, H, W), z(N, 3, H, W), z(N, 4),
                    torch.zeros(N, dtype=torch.long, device=dev), 0)
        fit = fit_ellipses_from_pred(model.predictions(), out[1])     # [N,2,5] on the device: (iris, pupil)
        mask = model.predictions()
    fit = fit.cpu().numpy()
    return edge[:, 0].cpu().numpy(), mask.cpu().numpy(), fit[:, 1], fit[:, 0]


def rescale_to_original(seg_map, pupil_ellipse, iris_ellipse, scale_shift, orig_shape):
    """evaluate.py:169-192 (nearest-neighbour maps; ellipse centres shifted back by the row padding)."""
    pupil_ellipse, iris_ellipse = pupil_ellipse.copy(), iris_ellipse.copy()
    for e in (pupil_ellipse, iris_ellipse):
        e[1] = e[1] - np.floor(scale_shift[1] // 2)
        e[:-1] = e[:-1] * (1 / scale_shift[0])
    if scale_shift[1] > 0:
        seg_map = seg_map[scale_shift[1] // 2: seg_map.shape[0] - (scale_shift[1] - scale_shift[1] // 2)]
    elif scale_shift[1] < 0:
        seg_map = np.pad(seg_map, ((-scale_shift[1] // 2, -scale_shift[1] - (-scale_shift[1] // 2)), (0, 0)))
    return seg_map, pupil_ellipse, iris_ellipse


def mjpeg_frames(path):
    """Yield grey frames of an MJPEG .avi by scanning for JPEG SOI/EOI markers (no OpenCV)."""
    from PIL import Image
    data = open(path, 'rb').read()
    pos = 0
    while True:
        a = data.find(b'\xff\xd8\xff', pos)
        if a < 0:
            return
        b = data.find(b'\xff\xd9', a)
        if b < 0:
            return
        pos = b + 2
        try:
            yield np.asarray(Image.open(io.BytesIO(data[a:b + 2])).convert('L'))
        except Exception:
            continue


def evaluate_ellseg_per_video(path_vid, args, model, edge_model, device):
    """evaluate.py:195-308: two 320-wide eyes per 640x240 frame."""
    out, batch, meta = {}, [], []
    for j, fr in enumerate(mjpeg_frames(path_vid)):
        if args.max_frames and j >= args.max_frames:
            break
        for i in range(2):
            eye = fr[:, 320 * i: 320 * (i + 1)]
            t, ss = preprocess_frame(eye, (240, 320), args.align_width)
            batch.append(t); meta.append((j, i, ss, eye.shape))
        if len(batch) >= 32:
            _flush(batch, meta, out, model, edge_model, device)
    if batch:
        _flush(batch, meta, out, model, edge_model, device)
    np.save(os.path.splitext(path_vid)[0] + '_ellipses_' + args.method + '.npy', out, allow_pickle=True)
    return out


def _flush(batch, meta, out, model, edge_model, device):
    x = torch.stack(batch).to(device)
    _, seg, pup, iri = evaluate_ellseg_on_image(x, model, edge_model)
    for k, (j, i, ss, shp) in enumerate(meta):
        _, p, q = rescale_to_original(seg[k], pup[k], iri[k], ss, shp)
        out[(j, i)] = (q, p)                                    # (iris, pupil) as evaluate.py:272
    batch.clear(); meta.clear()


def main(argv=None):
    args = parse_args(argv)
    device = torch.device('cuda')
    setting = _entry.load_setting(args.setting)
    if args.synthetic:
        edge_net, model = _entry.seeded_networks(setting)
    else:
        for need in (args.loadfile, 'gen_00000016.pt'):        # the reference dies in torch.load (evaluate.py:360-371)
            if not os.path.exists(need):
                sys.exit('evaluate.py: weights file %r not found (use --synthetic N for seeded random weights)' % need)
        from egne_amd.bdcn_new import BDCN
        from egne_amd.modelSummary import get_model
        edge_net = BDCN()
        edge_net.load_state_dict(torch.load('gen_00000016.pt', map_location='cpu')['a'])
        model = get_model('ritnet_v2', setting)
        model.load_state_dict(torch.load(args.loadfile, map_location='cpu')['state_dict'], strict=True)
    edge_net, model = edge_net.to(device).eval(), model.to(device).eval()
    if args.synthetic:
        from egne_amd import synth
        x = synth.make_batch(args.synthetic, seed=5)['img'].to(device)
        edge, seg, pup, iri = evaluate_ellseg_on_image(x, model, edge_net)
        print('pupil ellipses:\n', pup, '\niris ellipses:\n', iri)
        return pup, iri
    res = None
    for v in sorted(glob.glob(os.path.join(args.path2data, '*.' + args.vid_ext))):
        print('evaluate {}...'.format(os.path.basename(v)))
        res = evaluate_ellseg_per_video(v, args, model, edge_net, device)
    return res


if __name__ == '__main__':
    main()

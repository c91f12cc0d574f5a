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
    p.add_argument('--model', type=str, default='ritnet_v2')                            # evaluate.py:37,362-367: 'ritnet_v2' | 'deepvog'
    p.add_argument('--loadfile', type=str, default='logs/baseline_edge_16.pkl')        # evaluate.py:357
    p.add_argument('--setting', type=str, default='configs/baseline_edge.yaml')        # evaluate.py:359
    p.add_argument('--max_frames', type=int, default=0)
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--low_latency', type=int, default=0)          # 1: the two eyes of a frame per call, results before the next frame
    a = p.parse_args(argv)
    a.prec = torch.float32
    return a


def resize_lanczos4(img, dsize):
    """cv2.resize(img, dsize=(width, height), interpolation=cv2.INTER_LANCZOS4) restated in NumPy (evaluate.py:76): separable
    8-tap Lanczos (a = 4) at source coordinate (dst + 0.5) * scale - 0.5, replicated border, uint8 in -> rounded uint8 out.
    OpenCV is not installed here, so this is checked against properties only (parity unpinned): OpenCV quantises the tap
    weights to 1/32-pixel tables in fixed point, which this float restatement does not."""
    W2, H2 = int(dsize[0]), int(dsize[1])
    src = np.asarray(img)
    out = src.astype(np.float64)
    for axis, n2 in ((0, H2), (1, W2)):
        n1 = out.shape[axis]
        if n1 == n2:
            continue
        pos = (np.arange(n2) + 0.5) * (n1 / n2) - 0.5
        base = np.floor(pos).astype(np.int64)
        frac = pos - base
        taps = np.arange(-3, 5)                                        # 8 taps around the source position
        x = frac[:, None] - taps[None, :]
        with np.errstate(invalid="ignore", divide="ignore"):
            wts = np.where(np.abs(x) < 1e-12, 1.0, np.sin(np.pi * x) * np.sin(np.pi * x / 4) / (np.pi * np.pi * x * x / 4))
        wts = np.where(np.abs(x) < 4, wts, 0.0)
        wts /= wts.sum(1, keepdims=True)
        idx = np.clip(base[:, None] + taps[None, :], 0, n1 - 1)        # BORDER_REPLICATE
        g = np.take(out, idx.reshape(-1), axis=axis)
        shp = list(out.shape)
        shp[axis:axis + 1] = [n2, 8]
        g = g.reshape(shp)
        wshape = [1] * g.ndim
        wshape[axis], wshape[axis + 1] = n2, 8
        out = (g * wts.reshape(wshape)).sum(axis + 1)
    if src.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(src.dtype)


def resize_nearest(img, dsize):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src index = floor(dst * scale) (evaluate.py:190-191)."""
    W2, H2 = int(dsize[0]), int(dsize[1])
    a = np.asarray(img)
    ys = np.minimum((np.arange(H2) * (a.shape[0] / H2)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum((np.arange(W2) * (a.shape[1] / W2)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


def preprocess_frame(img, op_shape, align_width=True):
    """evaluate.py:69-104: scale to the target WIDTH (Lanczos), pad or centre-crop the rows, z-score.  Returns the
    [1,H,W] float32 tensor and scale_shift = (scale, rows added (+) or removed (-)).  (The reference's crop branch
    indexes with floats and raises; the centre crop it intends is what runs here.)"""
    if not align_width:
        sys.exit('Height alignment not implemented! Exiting ...')
    scale_shift = (1, 0)
    if op_shape[1] != img.shape[1]:
        sc = op_shape[1] / img.shape[1]
        img = resize_lanczos4(img, (int(img.shape[1] * sc), int(img.shape[0] * sc)))
        scale_shift = (sc, 0)
    if op_shape[0] > img.shape[0]:
        pad = op_shape[0] - img.shape[0]
        img = np.pad(img, ((pad // 2, pad - pad // 2), (0, 0)))
        scale_shift = (scale_shift[0], pad)
    elif op_shape[0] < img.shape[0]:
        cut = img.shape[0] - op_shape[0]
        img = img[cut // 2: cut // 2 + op_shape[0]]
        scale_shift = (scale_shift[0], -cut)
    img = img.astype(np.float64)
    img = (img - img.mean()) / img.std()
    return torch.from_numpy(img).unsqueeze(0).to(torch.float32), scale_shift


def evaluate_ellseg_on_image(frames, model, edge_model, args=None):
    """evaluate.py:112-166 for a batch of frames [N,1,H,W] (the reference loops one frame at a time).
    Returns edge maps [N,H,W], class maps [N,H,W], pupil ellipses [N,5], iris ellipses [N,5] (pixels)."""
    from egne_amd.utils import calc_edge
    assert frames.dim() == 4, 'Frame must be [N,1,H,W]'
    ns = argparse.Namespace(prec=torch.float32, edge_thres=0)
    for attempt in (0, 1, 2):
        with torch.no_grad():
            edge = calc_edge(ns, frames, edge_model, frames.device)
        # a frame beyond the head-room of the calibrated f16 pre-scales (engine.Plan.overflowed): a plan re-calibrates on its next call, so
        # the frames simply run again.  The edge network's word is read BEFORE its map is fed on (a NaN map must not reach the model plan's
        # own calibration pass)
        if _overflowed(edge_model):
            continue
        res = _to_host(_seg_and_fit(frames, model)(edge))
        if not _overflowed(model):
            return res
    raise RuntimeError("non-finite activations after re-calibration: the input frames themselves are not finite")


def _overflowed(net):
    return bool(net.overflowed()) if hasattr(net, "overflowed") else False


def _seg_and_fit(frames, model, wfit=None):
    """Second stage of a batch (evaluate.py:117-166): ESF-Net on the frames and their edge maps, argmax mask, both ellipses
    fitted on the device.  Returns a callable of the edge maps (egne_amd.pipeline.TwoStagePipeline runs it on its second stream).
    ``wfit`` (egne_amd.pipeline.WindowedFit): the searches go to its stream and are released in the next batch's launch window; the
    third result is then a handle (``_to_host`` waits for it)."""
    dev = frames.device
    N, _, H, W = frames.shape

    def run(edge):
        with torch.no_grad():
            labels = torch.zeros((N, H, W), device=dev)
            labels[..., 0, 2] = 1                                  # evaluate.py:118-120: make all 3 classes present
            labels[..., 2, 2] = 2
            z = lambda *s: torch.zeros(s, device=dev)              # noqa: E731
            out = model(frames, edge, labels.long(), z(N, 2), z(N, 2, 5), z(N, H, W), z(N, 3, H, W), z(N, 4),
                        torch.zeros(N, dtype=torch.long, device=dev), 0)
            if wfit is not None:
                return edge[:, 0].clone(), model.predictions().clone(), wfit.submit(model.predictions(), out[1])
            fit = fit_ellipses_from_pred(model.predictions(), out[1])     # [N,2,5] on the device: (iris, pupil)
            return edge[:, 0].clone(), model.predictions().clone(), fit
    return run


def graphed_runner(warmup_frames, model, edge_model):
    """edge -> seg -> fit of a fixed small batch as one hipGraph replay (egne_amd.pipeline.GraphedFrames): the per-eye loop of
    evaluate.py:235-249 at one or two frames per call.  ``runner(frames)`` returns the same device tensors as the eager path (edge maps
    [N,H,W], class maps [N,H,W], fitted ellipses [N,2,5]); results are bit-identical to it."""
    from egne_amd.pipeline import GraphedFrames
    from egne_amd.utils import calc_edge
    ns = argparse.Namespace(prec=torch.float32, edge_thres=0)

    def stage(x):
        return _seg_and_fit(x, model)(calc_edge(ns, x, edge_model, x.device))
    return GraphedFrames(stage, warmup_frames)


def _to_host(res):
    edge, mask, fit = res
    if not torch.is_tensor(fit):       # WindowedFit.Handle: the searches may still be waiting for their window
        fit.synchronize()
        fit = fit.result
    fit = fit.cpu().numpy()
    return edge.cpu().numpy(), mask.cpu().numpy(), fit[:, 1], fit[:, 0]


def rescale_to_original(seg_map, pupil_ellipse, iris_ellipse, scale_shift, orig_shape, edge_map=None):
    """evaluate.py:169-192: ellipses back to the source frame (row shift, then 1/scale), class map (and edge map) un-padded /
    re-padded and resized to the source shape with nearest neighbour.  Returns (seg_map, pupil, iris[, edge_map])."""
    pupil_ellipse, iris_ellipse = np.array(pupil_ellipse, dtype=np.float64), np.array(iris_ellipse, dtype=np.float64)
    for e in (pupil_ellipse, iris_ellipse):
        e[1] = e[1] - np.floor(scale_shift[1] // 2)
        e[:-1] = e[:-1] * (1 / scale_shift[0])

    def fix(m):
        if m is None:
            return None
        if scale_shift[1] > 0:
            m = m[scale_shift[1] // 2: m.shape[0] - (scale_shift[1] - scale_shift[1] // 2)]
        elif scale_shift[1] < 0:
            m = np.pad(m, ((-scale_shift[1] // 2, -scale_shift[1] - (-scale_shift[1] // 2)), (0, 0)))
        return resize_nearest(m, (orig_shape[1], orig_shape[0])) if tuple(m.shape[:2]) != tuple(orig_shape[:2]) else m
    seg_map, edge_map = fix(seg_map), fix(edge_map)
    return (seg_map, pupil_ellipse, iris_ellipse) if edge_map is None else (seg_map, pupil_ellipse, iris_ellipse, edge_map)


def _draw_ellipse(img, el, colour):
    """Outline of the ellipse (cx, cy, a, b, theta) as cv2.ellipse(..., thickness 1) would put it (helperfunctions.py:606-609;
    integer centre / axes as there), without anti-aliasing: 720 boundary samples rounded to pixels."""
    if np.all(np.asarray(el) == -1) or not np.all(np.isfinite(el)):
        return
    cx, cy, a, b = (int(v) for v in el[:4])
    t = np.linspace(0, 2 * np.pi, 720, endpoint=False)
    ang = float(el[4])
    x = cx + a * np.cos(t) * np.cos(ang) - b * np.sin(t) * np.sin(ang)
    y = cy + a * np.cos(t) * np.sin(ang) + b * np.sin(t) * np.cos(ang)
    xi, yi = np.rint(x).astype(np.int64), np.rint(y).astype(np.int64)
    ok = (xi >= 0) & (xi < img.shape[1]) & (yi >= 0) & (yi < img.shape[0])
    img[yi[ok], xi[ok]] = colour


def plot_segmap_ellpreds(image, seg_map, pupil_ellipse, iris_ellipse):
    """helperfunctions.py:521-622: grey frame -> BGR, iris pixels (120,183,53), pupil pixels (36,231,253), iris ellipse in
    (255,0,0) and pupil ellipse in (0,0,255)."""
    out = np.stack([image] * 3, axis=2).astype(np.uint8)
    out[seg_map == 1] = np.array([120, 183, 53], np.uint8)
    out[seg_map == 2] = np.array([36, 231, 253], np.uint8)
    _draw_ellipse(out, iris_ellipse, np.array([255, 0, 0], np.uint8))
    _draw_ellipse(out, pupil_ellipse, np.array([0, 0, 255], np.uint8))
    return out


class MJPEGWriter:
    """Minimal AVI (RIFF) writer with Motion-JPEG frames encoded by PIL -- stands in for cv2.VideoWriter (evaluate.py:219-221;
    the reference writes mp4v, for which there is no encoder in this image).  Frames are BGR uint8 arrays as OpenCV's are."""

    def __init__(self, path, fps, size):
        self.path, self.fps, self.size, self.frames = path, max(int(fps), 1), (int(size[0]), int(size[1])), []

    def write(self, frame_bgr):
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(frame_bgr[..., ::-1])).save(buf, format='JPEG', quality=90)
        self.frames.append(buf.getvalue())

    def release(self):
        import struct
        W, H, n = self.size[0], self.size[1], len(self.frames)
        chunks, idx, off = [], [], 4
        for f in self.frames:
            pad = len(f) & 1
            chunks.append(b'00dc' + struct.pack('<I', len(f)) + f + b'\0' * pad)
            idx.append(b'00dc' + struct.pack('<III', 0x10, off, len(f)))
            off += 8 + len(f) + pad
        movi = b'movi' + b''.join(chunks)
        biggest = max((len(f) for f in self.frames), default=0)
        avih = struct.pack('<IIIIIIIIIIIIII', 1000000 // self.fps, biggest * self.fps, 0, 0x10, n, 0, 1, biggest, W, H, 0, 0, 0, 0)
        strh = b'vids' + b'MJPG' + struct.pack('<IHHIIIIIIIIhhhh', 0, 0, 0, 0, 1, self.fps, 0, n, biggest, 0xffffffff, 0, 0, 0, W, H)
        strf = struct.pack('<IiiHHIIiiII', 40, W, H, 1, 24, 0x47504a4d, W * H * 3, 0, 0, 0, 0)

        def ck(tag, data):
            return tag + struct.pack('<I', len(data)) + data + (b'\0' if len(data) & 1 else b'')

        def lst(tag, data):
            return b'LIST' + struct.pack('<I', len(data) + 4) + tag + data
        hdrl = lst(b'hdrl', ck(b'avih', avih) + lst(b'strl', ck(b'strh', strh) + ck(b'strf', strf)))
        body = b'AVI ' + hdrl + b'LIST' + struct.pack('<I', len(movi)) + movi + ck(b'idx1', b''.join(idx))
        with open(self.path, 'wb') as f:
            f.write(b'RIFF' + struct.pack('<I', len(body)) + body)


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
    """evaluate.py:195-308: every frame of the video holds two eyes side by side (320 columns each): per eye preprocess ->
    edge -> seg -> fitted ellipses -> back to the source geometry; writes <name>_result_<method>.avi (overlay: class colours,
    both ellipses, frame number) and <name>_edge_<method>.avi (255 - 255*edge), both Motion-JPEG, and the ellipse dictionary
    <name>_pred2_<method>.npy {frame: (iris, pupil)} -- eyes are batched 32 at a time instead of one by one."""
    from egne_amd.pipeline import TwoStagePipeline, WindowedFit
    stem = os.path.splitext(path_vid)[0]
    out, pending = {}, []
    vid_out = edge_out = None
    # the edge network of batch i+1 runs next to ESF-Net + fit of batch i (two HIP streams), and the host draws / encodes batch
    # i-1 meanwhile: results come back one batch late
    pipe = TwoStagePipeline(argparse.Namespace(prec=torch.float32, edge_thres=0), edge_model, torch.device(device))
    queued = []                                  # frames of the batches whose results are still on the device
    # the ellipse searches of batch i are released where ESF-Net of batch i+1 reaches its low-resolution levels (WindowedFit): a
    # batch is drawn once the NEXT batch's second stage has been queued, i.e. two batches behind the decoder
    wfit = WindowedFit(torch.device(device))
    ready = []

    redo = [False]
    live = bool(getattr(args, 'low_latency', 0))    # head-mounted-display use: a frame's ellipses before the next frame arrives
    runner = [None]

    def flush():
        if not pending:
            return
        eyes = [e for fr in pending for e in fr[2]]
        x = torch.stack([e[0] for e in eyes]).to(device)
        if live:
            # one hipGraph replay per frame pair (egne_amd.pipeline.GraphedFrames, captured on the first frame): no batching, no
            # pipelining across frames -- 4.3-4.7 ms per call on MI355X instead of a batch of 32 every ~20 ms
            if runner[0] is None or tuple(runner[0].x.shape) != tuple(x.shape):
                runner[0] = graphed_runner(x, model, edge_model)
            res = [t.clone() for t in runner[0](x)]
            if _overflowed(model) | _overflowed(edge_model):
                # beyond the head-room of the scales baked into the captured launches: capture again (its eager warm-up runs re-calibrate)
                runner[0] = graphed_runner(x, model, edge_model)
                res = [t.clone() for t in runner[0](x)]
                if _overflowed(model) | _overflowed(edge_model):
                    raise RuntimeError("non-finite activations after re-calibration: the input frames themselves are not finite")
            done = torch.cuda.Event()
            done.record()
            frames_now = list(pending)
            pending.clear()
            draw(frames_now, (tuple(res), done))
            return
        queued.append(list(pending))
        pending.clear()
        r = pipe.submit(x, _seg_and_fit(x, model, wfit))
        if r is not None:
            ready.append((queued.pop(0), r))
        while len(ready) > 1:
            draw(*ready.pop(0))

    def drain():
        r = pipe.flush()
        if r is not None and queued:
            ready.append((queued.pop(0), r))
        while ready:
            draw(*ready.pop(0))

    def draw(frames_of_batch, r):
        nonlocal vid_out, edge_out
        res, done = r
        done.synchronize()
        edge, seg, pup, iri = _to_host(res)
        if redo[0] or _overflowed(model) | _overflowed(edge_model):
            # invalid results (engine.Plan.overflowed): this batch again, back to back, on the re-calibrated plans -- and the batch
            # queued behind it too, whose edge maps were computed with the old scales
            redo[0] = not redo[0]
            torch.cuda.synchronize()    # the edge network of the NEXT batch is in flight on the pipeline's stream and owns the same plan buffers
            x = torch.stack([e[0] for _, _, fe in frames_of_batch for e in fe]).to(device)
            edge, seg, pup, iri = evaluate_ellseg_on_image(x, model, edge_model)
        k = 0
        for j, frame_bgr, fe in frames_of_batch:
            overlay, edge_frame = frame_bgr.copy(), frame_bgr.copy()
            for i, (_, ss, grey) in enumerate(fe):
                em = 255.0 - 255.0 * edge[k]                                         # evaluate.py:263-264
                sm, p, q, em = rescale_to_original(seg[k], pup[k], iri[k], ss, grey.shape, edge_map=em)
                out[j] = (q, p)                                                      # evaluate.py:269 (the second eye overwrites the first)
                out[(j, i)] = (q, p)
                overlay[:, 320 * i: 320 * (i + 1)] = plot_segmap_ellpreds(grey, sm, p, q)
                edge_frame[:, 320 * i: 320 * (i + 1)] = np.clip(em, 0, 255).astype(np.uint8)[..., None]
                k += 1
            _put_frame_number(overlay, j)
            if vid_out is None:
                Hh, Ww = frame_bgr.shape[:2]
                vid_out = MJPEGWriter(stem + '_result_' + args.method + '.avi', 30, (Ww, Hh))
                edge_out = MJPEGWriter(stem + '_edge_' + args.method + '.avi', 30, (Ww, Hh))
            vid_out.write(overlay)
            edge_out.write(edge_frame)

    for j, fr in enumerate(mjpeg_frames(path_vid)):
        if args.max_frames and j >= args.max_frames:
            break
        frame_bgr = np.stack([fr] * 3, axis=2)
        eyes = []
        for i in range(2):
            grey = fr[:, 320 * i: 320 * (i + 1)]
            t, ss = preprocess_frame(grey, (240, 320), args.align_width)
            eyes.append((t, ss, grey))
        pending.append((j, frame_bgr, eyes))
        if len(pending) >= (1 if live else 16):
            flush()
    flush()
    drain()
    assert not queued
    for w in (vid_out, edge_out):
        if w is not None:
            w.release()
    np.save(stem + '_pred2_' + args.method + '.npy', out, allow_pickle=True)
    return out


def _put_frame_number(img, j):
    """cv2.putText(frame, str(j), (10, 30), FONT_HERSHEY_PLAIN, 2.0, (0, 0, 255), 2) -- PIL's default font instead of Hershey."""
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.ascontiguousarray(img[..., ::-1]))
    ImageDraw.Draw(im).text((10, 12), str(j), fill=(255, 0, 0))
    img[...] = np.asarray(im)[..., ::-1]


def main(argv=None):
    args = parse_args(argv)
    device = torch.device('cuda')
    setting = _entry.load_setting(args.setting)
    if args.model not in ('ritnet_v2', 'deepvog'):
        sys.exit('evaluate.py: illegal model %r (evaluate.py:362-367 knows ritnet_v2 and deepvog)' % args.model)
    if args.synthetic:
        edge_net, model = _entry.seeded_networks(setting)
        if args.model == 'deepvog':
            from egne_amd import synth
            from egne_amd.modelSummary import get_model
            model = get_model('deepvog', None)
            model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=1, kind='esf'))
    else:
        for need in (args.loadfile, 'gen_00000016.pt'):        # the reference dies in torch.load (evaluate.py:360-371)
            if not os.path.exists(need):
                sys.exit('evaluate.py: weights file %r not found (use --synthetic N for seeded random weights)' % need)
        from egne_amd.bdcn_new import BDCN
        from egne_amd.modelSummary import get_model
        edge_net = BDCN()
        edge_net.load_state_dict(torch.load('gen_00000016.pt', map_location='cpu')['a'])
        model = get_model(args.model, setting)
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

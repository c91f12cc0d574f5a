"""Two-stage pipeline across batches for the inference path (test.py / evaluate.py loop: calc_edge -> model, utils.py:645-656 +
models/RITnet_v2.py:270-340): the frozen edge network of batch i runs on one HIP stream while ESF-Net (+ loss / argmax / fit) of
batch i-1 runs on another.

ESF-Net's lower levels (30x40 and 15x20 maps, the regression module: ~100 short launches) leave CUs idle that the edge
network's kernels fill, and the gaps between dependent launches of one network are covered by the other: +3.5 - 5.9 % frames/s
at B=64 on MI355X (scratch/overlap.py, bench.py).  Every batch still takes the whole path and the results are bit-identical to
the sequential loop (tests/test_gpu_nets.py::test_pipelined_inference_is_bit_identical): the two networks own disjoint launch
plans and buffers; the only shared tensors are the two edge-map slots, handed over with events.
"""
import torch

from .utils import calc_edge


class TwoStagePipeline:
    """``submit(frames, second_stage)`` queues the edge network of this batch and then runs ``second_stage(edge)`` of the PREVIOUS
    batch; it returns that previous batch's ``(result, done_event)`` (``None`` on the first call).  ``flush()`` runs what is
    still pending, joins both streams into the current one and returns the last ``(result, done_event)``.  ``result`` holds
    device tensors produced on the second stream: wait for ``done_event`` (or call ``flush``) before reading them elsewhere."""

    def __init__(self, args, edge_model, device):
        self.args, self.edge_model, self.device = args, edge_model, device
        import os
        prio = -1 if os.environ.get("EGNE_FIT_PRIO") == "1" else 0
        self.sa, self.sb = torch.cuda.Stream(device=device, priority=prio), torch.cuda.Stream(device=device, priority=prio)
        self.slots, self.ready, self.freed = [None, None], [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
        self.pending = []           # (slot, second_stage) of batches whose edge maps are queued
        self.keep = []              # (done event, closure, frames): kept alive until the second stream has finished with them
        self.n = 0

    def _second(self):
        slot, fn, frames = self.pending.pop(0)
        with torch.cuda.stream(self.sb):
            self.sb.wait_event(self.ready[slot])
            res = fn(self.slots[slot])
            self.freed[slot].record(self.sb)
            done = torch.cuda.Event()
            done.record(self.sb)
        self.keep = [k for k in self.keep if not k[0].query()] + [(done, fn, frames)]
        return res, done

    def submit(self, frames, second_stage):
        cur = torch.cuda.current_stream(self.device)
        here = torch.cuda.Event()       # both streams see what the caller's stream has queued so far (the frames' upload)
        here.record(cur)
        self.sa.wait_event(here)
        self.sb.wait_event(here)
        slot = self.n & 1
        with torch.no_grad(), torch.cuda.stream(self.sa):
            if self.n >= 2:
                self.sa.wait_event(self.freed[slot])        # the second stage has read batch n-2's edge maps out of this slot
            e = calc_edge(self.args, frames, self.edge_model, self.device)
            if self.slots[slot] is None or self.slots[slot].shape != e.shape or self.slots[slot].dtype != e.dtype:
                self.slots[slot] = torch.empty_like(e)
            self.slots[slot].copy_(e)
            self.ready[slot].record(self.sa)
        self.n += 1
        out = self._second() if self.pending else None
        self.pending.append((slot, second_stage, frames))
        return out

    def flush(self):
        out = None
        while self.pending:
            out = self._second()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.sa)
        cur.wait_stream(self.sb)
        return out


class WindowedFit:
    """The ellipse searches of a batch (utils.fit_ellipses_from_pred, evaluate.py:135-166) on a stream of their own, RELEASED where a
    later launch plan reaches its window (engine.WINDOW_NAME, default the 30x40 level of ESF-Net's encoder).

    A batch of searches is 2B sequential chains on a few dozen CUs for ~1.9 ms.  Queued right behind the network that produced the
    class maps it runs next to whatever the two network streams launch then -- mostly PERSISTENT kernels: 256 workgroups with a
    static 1/256 share of the tiles each, one per CU; the workgroup whose CU a search holds waits for another one to finish and the
    launch takes up to twice as long.  ESF-Net's low-resolution levels (and the edge network's deep trunk) are ordinary grids that
    hand a workgroup to whichever CU is free: next to them a search costs its share of the CUs and nothing more.  Measured on
    MI355X, B = 64, edge + seg + fit pipelined (scratch/ab_fit.sh): 1998-2001 frames/s released at once, 2009-2021 in a window.

    ``submit(mask, elPred, then=None)`` is called on the stream that produced the two tensors; it returns a handle: ``result`` (device
    tensor [F,2,5], valid once the handle is done), ``synchronize()`` (host waits; queues the searches at once if their window has not
    opened yet -- the last batches of a run), ``wait(stream)``.  ``then(result)`` runs on the search stream right behind the searches
    (e.g. the copy to pinned host memory)."""

    class Handle:
        def __init__(self):
            self.result, self.done, self._launch = None, None, None

        def _ensure(self):
            if self.done is None:
                from . import engine
                if self._launch in engine.WINDOW_HOOKS:
                    engine.WINDOW_HOOKS.remove(self._launch)
                self._launch(None)

        def synchronize(self):
            self._ensure()
            self.done.synchronize()

        def wait(self, stream=None):
            self._ensure()
            (stream or torch.cuda.current_stream()).wait_event(self.done)

    def __init__(self, device, windowed=True):
        from . import engine
        self.side = torch.cuda.Stream(device=device)
        self.windowed = bool(windowed) and engine.WINDOW_NAME != "none"
        self.device_index = self.side.device.index
        self._mine = []                 # hooks of this object still waiting for a window

    def close(self):
        """Drop the searches whose window never opened (an exception in the caller's loop): their hooks would otherwise keep the
        class maps alive and be launched by an unrelated plan later.  Handles already handed out still work (``synchronize`` queues
        the searches itself)."""
        from . import engine
        for h in self._mine:
            if h in engine.WINDOW_HOOKS:
                engine.WINDOW_HOOKS.remove(h)
        self._mine = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, mask, elPred, then=None):
        from . import engine
        from .utils import fit_ellipses_from_pred
        ready = torch.cuda.Event()
        ready.record()
        h = WindowedFit.Handle()
        side = self.side

        def launch(window):
            side.wait_event(ready)
            if window is not None:
                side.wait_event(window)
            with torch.cuda.stream(side), torch.no_grad():
                mask.record_stream(side)
                elPred.record_stream(side)
                h.result = fit_ellipses_from_pred(mask, elPred)
                if then is not None:
                    then(h.result)
                h.done = torch.cuda.Event()
                h.done.record(side)
        h._launch = launch
        launch.device_index = self.device_index        # engine.Plan._open_window releases it only from a plan of the same GPU
        if self.windowed:
            self._mine = [q for q in self._mine if q in engine.WINDOW_HOOKS] + [launch]
            engine.WINDOW_HOOKS.append(launch)
        else:
            launch(None)
        return h


class GraphedFrames:
    """Edge map + segmentation + ellipse fit of a FIXED small batch of frames as ONE hipGraph replay.

    The head-mounted-display case of the reference feeds the two eyes of a video frame one at a time (evaluate.py:235-249): at one
    or two frames the path is ~120 short dependent launches and the Python launch loop needs ~2.5 ms of host time per call to queue
    them -- as long as the GPU needs to run them.  A replay costs the host one call (the host thread is free for decoding / drawing
    frames, evaluate.py:195-308); the GPU time of a call is what it was (measured: 3.52 vs 3.56 ms at one frame, 5.33 vs 5.37 ms
    at two, scratch/latency2.py -- the launches were queued ahead of the GPU already).  ``stage(frames)`` -- any function of a static input tensor that only queues work on the current stream
    (egne_amd.evaluate builds edge -> ESF-Net -> argmax -> fit) -- is run a few times eagerly on ``warmup`` (this is also where the
    split-f16 kernels calibrate their power-of-two activation scales, Plan.run) and then captured; ``__call__`` copies the new
    frames into the static input and replays.  The returned tensors are the graph's own output buffers: they are overwritten by the
    next call (clone what must outlive it).

    The captured launches carry the calibrated scales and the packed weights by value / by address: capture again (``capture()``)
    after the weights change.  The scales have 32x of headroom over the warm-up frames, as in the eager path."""

    def __init__(self, stage, warmup, n_warm=3):
        self.stage = stage
        self.x = warmup.clone()
        self.n_warm = n_warm
        self.graph = None
        self.capture()

    def capture(self):
        dev = self.x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.n_warm):
                self.stage(self.x)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = self.stage(self.x)

    def __call__(self, frames):
        if frames.shape != self.x.shape:
            raise ValueError("GraphedFrames: captured for frames of shape %s, got %s" % (tuple(self.x.shape), tuple(frames.shape)))
        self.x.copy_(frames, non_blocking=True)
        self.graph.replay()
        return self.out

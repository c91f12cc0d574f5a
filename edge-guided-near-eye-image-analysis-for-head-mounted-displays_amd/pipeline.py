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

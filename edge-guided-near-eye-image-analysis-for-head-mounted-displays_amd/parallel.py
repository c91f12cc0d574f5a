"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

Replaces the reference's single-process ``torch.nn.DataParallel`` (train.py:205, test.py:296) with the
semantics it has:
  * every rank runs the frozen BDCN + ESF-Net on ITS shard with local BatchNorm statistics and local
    loss normalisation (DataParallel replicas do the same and the caller takes ``loss.mean()``,
    train.py:285);
  * gradients are averaged over ranks -- here ONE all-reduce of the flat gradient arena
    (13.45 MB fp32 for baseline_edge: latency-bound on the xGMI mesh, far below a step's compute), issued
    after backward; the dataset-identity head (``dsIdentify_lin``) is outside the optimiser in the
    reference (train.py:146) but is averaged too, which is harmless;
  * parameters and BatchNorm buffers start identical (broadcast from rank 0); checkpoints come from rank 0.
Inference shards frames with no collective at all.
"""
import os

import torch
import torch.distributed as dist


def forced():
    """EGNE_FORCE_DIST=1: build the process group and run every collective even with ONE rank -- the way to take the RCCL path
    (all-reduce of the gradient arena on its HIP stream, broadcasts, the device branch of sum_over_ranks) on a one-GPU box."""
    return os.environ.get("EGNE_FORCE_DIST", "0") not in ("", "0")


def active():
    """True when collectives are to be issued: more than one rank, or a forced one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not forced():
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # nccl = RCCL over xGMI, one rank per GPU.  EGNE_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) --
        # what the two-rank test of tests/test_gpu_scripts.py uses on a one-GPU box
        backend = backend or os.environ.get("EGNE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)
    return dist.get_rank(), dist.get_world_size()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_state(model, src=0):
    """Same parameters and BatchNorm buffers on every rank before the first step."""
    if not active():
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src)


def broadcast_buffers(model, src=0):
    """BatchNorm running statistics of rank ``src`` on every rank.  nn.DataParallel keeps replica 0's buffers
    and drops the others'; ranks here update their own, so they are re-aligned before each validation pass
    (and therefore before every checkpoint)."""
    if not active():
        return
    with torch.no_grad():
        for t in model.buffers():
            dist.broadcast(t.data, src=src)


def sum_over_ranks(values, device=None):
    """Element-wise sum of a short list of python floats over all ranks (validation means)."""
    if not active():
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def gather_lists(values):
    """Concatenation over ranks (rank order) of a python list of per-batch numbers; every rank gets the result."""
    if world_size() == 1:
        return list(values)
    out = [None] * world_size()
    dist.all_gather_object(out, list(values))
    return [v for part in out for v in part]


class _StridedShard(torch.utils.data.Sampler):
    """Validation shard of one rank: items rank, rank + world, ... in order -- every item exactly once over the ranks, nothing
    dropped or repeated (a DistributedSampler either drops the remainder or pads by repeating items)."""

    def __init__(self, n_items, rank, world):
        self.idx = list(range(rank, n_items, world))

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)


def samplers(train_set, valid_set, rank, world, seed=0):
    """(train sampler, validation sampler) that give every rank a disjoint shard.  Training: one permutation per epoch
    shared by all ranks (``set_epoch``), drop_last so that every rank takes the same number of optimiser steps (a rank with
    one step more would hang in the gradient all-reduce).  Validation: a strided shard that drops NOTHING -- the validation
    loader must keep its partial last batch too (train.py), the caller weights per-rank sums by sample counts
    (sum_over_ranks), and a rank left without any validation item is an error.  (None, None) for a single process."""
    if world == 1:
        return None, None
    from torch.utils.data.distributed import DistributedSampler
    if len(valid_set) < world:
        raise RuntimeError("validation set of %d items cannot be sharded over %d ranks: a rank would validate nothing"
                           % (len(valid_set), world))
    return (DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True, seed=seed, drop_last=True),
            _StridedShard(len(valid_set), rank, world))


def grad_arena(model):
    """The flat fp32 buffer all ``p.grad`` are views of (models with ``_ensure_grad_arena``), or a freshly
    flattened copy for foreign modules."""
    if hasattr(model, "_ensure_grad_arena"):
        return model._ensure_grad_arena(), True
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    return torch.cat([g.reshape(-1) for g in grads]), False


class GradOverlap:
    """The gradient all-reduce in TWO buckets, the first one issued from inside the backward pass ("bucketed,
    reverse-topological, overlapped with backward"; replaces the implicit reduce of nn.DataParallel, train.py:205,285).

    The flat arena holds the parameters in module order: encoder first, then decoder, style encoder / MLP, regression head,
    dataset-identity head.  Backward runs the other way round, so when its first ENCODER launch is queued every gradient behind the
    encoder's block of the arena is final: the backward plan calls ``tail_ready`` there (engine.Plan.tail_hook, on its second
    stream, behind the decoder's weight gradients) and the tail's all-reduce -- 52 % of the bytes for baseline_edge, 75 % with the
    AdaIN modules -- runs on RCCL's stream next to the encoder's backward (the longer half of the pass).  ``finish`` (from
    ``allreduce_grads``) reduces the encoder's block, waits for the tail and divides by the world size.  Element for element the
    same sums as the one-bucket form (one or two ranks: the same bits; a ring over more ranks may add them in another order)."""

    def __init__(self, model):
        self.model, self.work, self.enabled = model, None, os.environ.get("EGNE_OVERLAP_ALLREDUCE", "1") != "0"

    def split(self):
        """Elements of the arena's leading block whose gradients are NOT final at the hook: the parameters named ``enc.*`` when
        they open the parameter list (0 otherwise: nothing is issued early)."""
        n, seen_other = 0, False
        for name, p in self.model.named_parameters():
            if name.startswith("enc."):
                if seen_other:
                    return None
                n += p.numel()
            else:
                seen_other = True
        return n if seen_other else None

    def tail_ready(self):
        """Called by the backward plan.  Contract: exactly ONE backward pass per ``allreduce_grads``.  A second pass while the tail
        bucket of the first is still out (gradient accumulation, a step abandoned after an exception) would write into the arena
        slice the collective is reducing and ``finish`` would then mix summed and local gradients -- so it fails loudly; ``abandon``
        waits the collective out when a step is given up."""
        if not (self.enabled and active()):
            return
        if self.work is not None:
            raise RuntimeError("GradOverlap: a second backward pass started while the early all-reduce bucket of the previous one is still "
                               "pending -- call parallel.allreduce_grads(model) after every backward (or model.grad_comm.abandon() "
                               "to give a step up; EGNE_OVERLAP_ALLREDUCE=0 for gradient accumulation)")
        s = self.split()
        if not s:
            return
        flat, is_view = grad_arena(self.model)
        if not is_view or s >= flat.numel():
            return
        self.work = dist.all_reduce(flat[s:], op=dist.ReduceOp.SUM, async_op=True)
        self._s = s

    @property
    def pending(self):
        return self.work is not None

    def abandon(self):
        """Give the current step up (an exception between backward and the reduce): wait for the early bucket, forget it.  The arena
        then holds a partly reduced gradient -- the next backward overwrites it."""
        if self.work is not None:
            self.work.wait()
            self.work = None

    def finish(self):
        flat, _ = grad_arena(self.model)
        w0 = dist.all_reduce(flat[:self._s], op=dist.ReduceOp.SUM, async_op=True)
        self.work.wait()
        w0.wait()
        self.work = None
        flat.div_(world_size())


def overlap_grads(model):
    """Opt a model with a flat gradient arena into the two-bucket all-reduce (GradOverlap): its backward pass then issues the
    tail bucket itself and ``allreduce_grads`` finishes the job.  Harmless without a process group."""
    if hasattr(model, "_ensure_grad_arena") and getattr(model, "grad_comm", None) is None:
        model.grad_comm = GradOverlap(model)
    return model


def allreduce_grads(model, async_op=False):
    """Average gradients over ranks with one collective on the flat arena (in place) -- or, for a model opted into ``overlap_grads``
    whose backward pass already issued the tail bucket, with the rest of the two-bucket exchange."""
    n = world_size()
    if not active():
        return None
    gc = getattr(model, "grad_comm", None)
    if gc is not None and gc.pending:
        gc.finish()
        return (_Done(), lambda: None) if async_op else None
    flat, is_view = grad_arena(model)
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def finish():
        flat.div_(n)
        if not is_view:
            o = 0
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.copy_(flat[o:o + p.numel()].view_as(p.grad))
                    o += p.numel()
    if async_op:
        return work, finish
    finish()
    return None


class _Done:
    def wait(self):
        return True


def mean_loss(loss):
    """Logging only: the DataParallel caller's ``loss.mean()`` over replicas (train.py:285)."""
    n = world_size()
    if not active():
        return loss
    t = loss.detach().clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / n


def shard(n_items, rank=None, world=None):
    """Contiguous shard [lo, hi) of ``n_items`` frames for this rank (drop_last semantics of the loaders)."""
    rank = dist.get_rank() if rank is None and world_size() > 1 else (rank or 0)
    world = world or world_size()
    per = n_items // world
    return rank * per, (rank + 1) * per

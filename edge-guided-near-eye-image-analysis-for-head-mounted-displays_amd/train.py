#!/usr/bin/env python3
"""Training entry point -- the reference's train.py loop (:246-488) on the HIP path, data-parallel over
one process per GPU (torchrun) instead of nn.DataParallel.

    python train.py --synthetic 32 --batchsize 8 --epochs 2 --setting configs/baseline_edge.yaml
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --curObj ... --setting ...
"""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egne_amd  # noqa: E402,F401
from egne_amd import _entry, parallel  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from egne_amd.args import parse_args  # noqa: E402
from egne_amd.utils import calc_edge, getSeg_metrics  # noqa: E402


class EarlyStopping:
    """pytorchtools.py:11-67 ('max' mode as train.py:197-203 uses it): stop after `patience` epochs
    without an improvement of more than `delta`; saves the best model to `path`."""

    def __init__(self, patience=10, delta=0.001, path=None):
        self.patience, self.delta, self.path = patience, delta, path
        self.best, self.counter, self.early_stop = None, 0, False

    def __call__(self, metric, ckpt):
        if self.best is None or metric > self.best + self.delta:
            self.best, self.counter = metric, 0
            if self.path:
                torch.save(ckpt, self.path)
        else:
            self.counter += 1
            self.early_stop = self.counter >= self.patience


def lossandaccuracy(args, loader, model, edge_model, alpha, device):
    """utils.py:658-760 (the shipped version dies in np.stack([])): validation loss + mIoU.
    Under torchrun every rank validates ITS shard of the loader; the means are combined over ranks so that all
    ranks feed the same numbers to the LR scheduler / early stopping."""
    model.eval()
    lsum = nsamp = isum = icnt = 0.0
    train_products = getattr(edge_model, "f16_products", 0)
    edge_model.f16_products = 0          # validation: the 22-bit split products, as test.py
    try:
        for bt, batch in enumerate(loader):
            if args.test_normal and bt > 20:
                break
            img, labels, sw, dm, pc, ic, eln, cond, imInfo = batch
            for attempt in (0, 1, 2):
                with torch.no_grad():
                    edge = calc_edge(args, img.to(device), edge_model, device)
                # a batch beyond the head-room of the calibrated f16 pre-scales (engine.Plan.overflowed): a plan re-calibrates on its next
                # call, so the batch runs once more.  The edge network's word is read BEFORE its map is fed on: a NaN map would reach the
                # model plan's own calibration pass
                if bool(edge_model.overflowed()):
                    continue
                with torch.no_grad():
                    out = model(img.to(device), edge, labels.to(device).long(), pc.to(device), eln.to(device), sw.to(device),
                                dm.to(device), cond.to(device).float(), imInfo[:, 2].to(device), alpha)
                if not bool(model.overflowed()):
                    break
            else:
                # (test.py / evaluate.py raise here too: a NaN validation loss would steer the LR scheduler, early stopping and the
                # checkpoint choice)
                raise RuntimeError("validation batch %d: non-finite activations after re-calibration: the input frames themselves are not finite" % bt)
            # batches are weighted by their sample count: under torchrun the shards (and their last batches) differ in size
            n = float(img.shape[0])
            model.raise_on_loss_flags(model.loss_flags())       # two absent classes: loss.py:132 raises in the reference
            lsum += out[3].mean().item() * n
            nsamp += n
            iou = getSeg_metrics(labels.numpy(), model.predictions().cpu().numpy(), cond.numpy().astype(np.float32)[:, 1])[0]
            if iou == iou:
                isum += float(iou) * n
                icnt += n
    finally:
        model.train()
        edge_model.f16_products = train_products
    if nsamp == 0:
        raise RuntimeError("rank %d validated no batch (validation set too small for %d ranks?)" % (parallel.rank(), parallel.world_size()))
    sums = parallel.sum_over_ranks([lsum, nsamp, isum, icnt], device)
    return sums[0] / sums[1], (sums[2] / sums[3] if sums[3] else float('nan'))


def main(argv=None):
    args = parse_args(argv)
    rank, world = parallel.init()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(0)                                      # train.py:34-36
    setting = _entry.load_setting(args.setting)
    startEp = 0
    if args.synthetic:
        trainObj = _entry.SyntheticEyes(args.synthetic, seed=1234)   # same set on every rank; the sampler below shards it
        validObj = _entry.SyntheticEyes(max(args.batchsize, 4), seed=99)
        edge_net, model = _entry.seeded_networks(setting, args.model, bool(args.disentangle))
    else:
        from egne_amd.bdcn_new import BDCN
        from egne_amd.modelSummary import get_model
        with open(os.path.join(args.path2data, 'baseline', 'cond_' + args.curObj + '.pkl'), 'rb') as f:
            trainObj, validObj, _ = pickle.load(f)            # train.py:86-92
        edge_net = BDCN()
        edge_net.load_state_dict(torch.load('gen_00000016.pt', map_location='cpu')['a'])   # train.py:124-129
        model = get_model(args.model, setting)
        if args.disentangle:                                   # train.py:182-186
            model.disentangle = True
            model.setDatasetInfo(int(torch.unique(trainObj.imList[:, 2]).numel()))
    logdir = os.path.join('logs', args.model, args.expname)
    if args.resume:                                            # train.py:149-160: priority 1) checkpoint.pt 2) --loadfile
        found = [f for f in (os.path.join(logdir, 'checkpoint.pt'), args.loadfile) if f and os.path.exists(f)]
        if not found:
            sys.exit('--resume: neither %s nor --loadfile %r exists' % (os.path.join(logdir, 'checkpoint.pt'), args.loadfile))
        netDict = torch.load(found[0], map_location='cpu')
        model.load_state_dict(netDict['state_dict'], strict=False)      # the dataset-identity head is never checkpointed
        startEp = netDict['epoch'] + 1 if 'epoch' in netDict else 0
    model.selfCorr = bool(args.selfCorr)
    edge_net, model = edge_net.to(device).eval(), model.to(device).to(args.prec).train()
    if args.prec in (torch.float16, torch.bfloat16) and os.environ.get("EGNE_EDGE_PRODUCTS", "1") == "1" and hasattr(edge_net, "f16_products"):
        # --prec 16: the model rounds the edge map to bf16 (8-bit significand) on entry, so the frozen edge network in front of it runs
        # on plain f16 operands (11 bits, fp32 accumulation: one MFMA per product instead of three; egne_conv_desc.f16_products).
        # EGNE_EDGE_PRODUCTS=3 keeps the 22-bit split; validation / test.py / evaluate.py always use it
        edge_net.f16_products = 1
    parallel.broadcast_state(model)
    parallel.overlap_grads(model)       # DP: the decoder-side half of the gradient all-reduce goes out from inside the backward pass (EGNE_OVERLAP_ALLREDUCE=0: one collective after it)
    params = [p for n, p in model.named_parameters() if 'dsIdentify' not in n]     # train.py:146-148
    # train.py:148 (Adam, default betas / eps); on the GPU torch's fused multi-tensor form: a handful of launches per step instead of ~75
    optimizer = torch.optim.Adam(params, lr=args.lr, fused=bool(params) and all(p.is_cuda for p in params))
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'max', patience=5, factor=0.1)   # train.py:192
    os.makedirs(os.path.join(logdir, 'weights'), exist_ok=True)
    stopper = EarlyStopping(patience=10, delta=0.001, path=os.path.join(logdir, 'checkpoint.pt') if rank == 0 else None)
    # DataParallel scatters every batch over the GPUs (train.py:205); here every rank draws batches of --batchsize from
    # ITS disjoint shard of one shared per-epoch permutation, so the global batch is world * batchsize
    tsamp, vsamp = parallel.samplers(trainObj, validObj, rank, world)
    trainloader = DataLoader(trainObj, batch_size=args.batchsize, shuffle=tsamp is None, sampler=tsamp, num_workers=args.workers,
                             drop_last=True)
    # validation keeps its partial last batch (nothing is dropped, parallel.samplers); one process: as train.py:110-121
    validloader = DataLoader(validObj, batch_size=args.batchsize, shuffle=False, sampler=vsamp, num_workers=args.workers,
                             drop_last=vsamp is None and len(validObj) >= args.batchsize)
    # one 4-byte read of the edge plan's sticky overflow word per step, where the step's stream has the edge map (EGNE_TRAIN_OVF_CHECK=0: off)
    check_edge = os.environ.get("EGNE_TRAIN_OVF_CHECK", "1") != "0" and hasattr(edge_net, "overflowed")
    redo_edge = [False]
    pipe = None
    if getattr(args, "pipeline", 0):
        from egne_amd.pipeline import TwoStagePipeline
        pipe = TwoStagePipeline(args, edge_net, device)
    for epoch in range(startEp, args.epochs):
        alpha = epoch / args.epochs                                                 # helperfunctions.linVal, train.py:248
        if tsamp is not None:
            tsamp.set_epoch(epoch)
        t_edge = t_net = 0.0
        t0 = time.time()
        for bt, batch in enumerate(trainloader):
            if (args.overfit and bt >= args.overfit) or (args.test_normal and bt > 20):
                break
            img, labels, sw, dm, pc, ic, eln, cond, imInfo = batch

            def rest(edge, img=img, labels=labels, sw=sw, dm=dm, pc=pc, eln=eln, cond=cond, imInfo=imInfo):
                ov = check_edge and bool(edge_net.overflowed())          # (reads AND clears the plan's sticky word)
                if check_edge and (redo_edge[0] or ov):
                    # the frozen edge network left the f16 range of its calibrated pre-scales (engine.Plan.overflowed): this edge map holds
                    # NaNs, and a step on it would put them into the BatchNorm statistics, the Adam state and the weights.  The plan is
                    # marked for re-calibration, so the map is computed again, here, behind everything in flight.  Under the pipeline the
                    # edge network of the NEXT batch is already queued on the other stream, in the same plan buffers and with the old scales,
                    # and its flag went with the word this step cleared: whenever a step cleared the word, the next step computes its map
                    # again as well (two attempts each: the plan may first have to re-calibrate on this batch's frames)
                    cleared = ov
                    torch.cuda.synchronize()
                    for attempt in (0, 1):
                        edge = calc_edge(args, img.to(device), edge_net, device).clone()
                        if not edge_net.overflowed():
                            break
                        cleared = True
                    else:
                        raise RuntimeError("non-finite edge maps after re-calibration: the input frames themselves are not finite")
                    redo_edge[0] = pipe is not None and cleared
                    if pipe is not None:
                        pipe.sa.wait_stream(torch.cuda.current_stream())
                optimizer.zero_grad()
                if args.device_prep:    # bit-identical to the Dataset's one_hot2dist maps, 54k frames/s instead of 123 per host core
                    from egne_amd import dataprep
                    dm = dataprep.dist_maps(labels.to(device).long())
                    if args.device_prep >= 2:   # the boundary weights too (CurriculumLib.py:128-129; parity unpinned: no OpenCV to check against)
                        sw = dataprep.spatial_weights(labels.to(device).long())
                out = model(img.to(device), edge, labels.to(device).long(), pc.to(device), eln.to(device), sw.to(device),
                            dm.to(device), cond.to(device).float(), imInfo[:, 2].to(device), alpha)
                loss = out[3].mean()                                                   # train.py:285
                loss.backward()
                parallel.allreduce_grads(model)
                optimizer.step()
                return loss.detach(), model.loss_flags()

            if pipe is not None:
                # --pipeline 1: the frozen edge network of this batch runs on a second stream next to the previous batch's
                # forward / backward / optimiser step (egne_amd.pipeline); the logged loss is the previous batch's
                r = pipe.submit(img.to(device), rest)
                if r is None:
                    continue
                torch.cuda.current_stream().wait_event(r[1])    # the step ran on the pipeline's stream: its loss is final behind this event
                loss, flags = r[0]
            else:
                torch.cuda.synchronize(); ta = time.time()
                edge = calc_edge(args, img.to(device), edge_net, device)               # frozen, no_grad (train.py:266)
                torch.cuda.synchronize(); tb = time.time()
                loss, flags = rest(edge)
                torch.cuda.synchronize(); tc = time.time()
                t_edge += tb - ta; t_net += tc - tb
            if bt % 30 == 0:
                model.raise_on_loss_flags(flags)       # (checked where the loop synchronises anyway: loss.py:132 raises in the reference)
            if rank == 0 and bt % 30 == 0:
                # (per-stage timers need the stages back to back: --pipeline 0)
                timers = '' if pipe is not None else ' edge {:.3f}s net {:.3f}s'.format(t_edge, t_net)
                print('Epoch:{} [{}/{}], Loss: {:.3f}{}'.format(epoch, bt, len(trainloader), parallel.mean_loss(loss.detach()).item(), timers))
            elif world > 1 and bt % 30 == 0:
                parallel.mean_loss(loss.detach())
        if pipe is not None:
            pipe.flush()                       # the last batch of the epoch
        parallel.broadcast_buffers(model)      # validate with rank 0's BatchNorm statistics (DataParallel keeps replica 0's)
        vloss, viou = lossandaccuracy(args, validloader, model, edge_net, alpha, device)
        if rank == 0:
            print('Epoch {} done in {:.1f}s: valid loss {:.4f} mIoU {:.4f}'.format(epoch, time.time() - t0, vloss, viou))
            ckpt = _entry.checkpoint_dict(model, epoch)
            torch.save(ckpt, os.path.join(logdir, 'weights', '{}_{}.pkl'.format(args.model, epoch)))   # train.py:486-488
            stopper(viou if viou == viou else -vloss, ckpt)
        scheduler.step(viou if viou == viou else -vloss)
        stop = torch.tensor([1.0 if stopper.early_stop else 0.0], device=device)
        if world > 1:
            torch.distributed.broadcast(stop, src=0)
        if stop.item() > 0:
            break
    return model


if __name__ == '__main__':
    main()

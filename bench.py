#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json): eye-frames/s at 320x240.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch that is already resident in HBM.  ONE run
measures every leg BASELINE.json's metric / north_star name and prints ONE JSON line:

  value / ms_per_step  inference, BASELINE.json configs[1]: frozen BDCN edge extractor -> ESF-Net (baseline_edge,
                       chz=32) -> loss head + argmax mask, batch 64 per GPU (W warm-up steps, exactly K timed steps);
  with_fit             the same step followed by evaluate.py's ellipse-fit stage (2 fits per frame, seeds computed on the
                       device, fitted ellipses copied to the host every step): the north-star "edge+seg+fit";
  exact_fp32           the same step with the split-f16 products switched off (every conv on v_mfma_f32_32x32x2_f32);
  train                fwd+bwd: frozen edge net forward, ESF-Net forward + backward, gradient all-reduce (N>1), Adam
                       step, batch 256 per GPU (BASELINE.json configs[2] shape; fp32 storage, fp32 accumulation; 3x3 forward
                       convolutions, their data and weight gradients on split-f16 products, 1x1 convolutions exact fp32).

For N>1 the driver launches one process per GPU (torch.distributed.run, RANK/LOCAL_RANK/WORLD_SIZE in the env);
`python bench.py --gpus N` without that environment spawns the N rank processes itself (before anything touches the
GPU in the parent).  Inference shards frames across ranks with no data-path collective (replicas);
training averages gradients with one RCCL all-reduce per step.  Scaling is weak: every rank keeps its own batch.

The JSON line also carries
  roofline      the dominant kernel family of the inference step (split-f16 convolutions: 3 x v_mfma_f32_32x32x16_f16
                per product): algorithmic conv FLOPs of one step / summed duration of its launches, measured with HIP
                events on the launch stream inside the timed region, against 2500 / 3 TFLOP/s;
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's PyTorch path) timed on the host cores on a bounded
                sample (B=2, rank 0 at N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"; the split kernel issues 3 MFMAs per product
ROUND = "r02"
FP32_FAM = ("conv_igemm", "conv3x3_halo", "conv3x3_smallcin", "conv_wgrad")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per inference step (default 64)")
    ap.add_argument("--train-batch", type=int, default=256, help="frames per GPU per training step (BASELINE.json configs[2]: 256)")
    ap.add_argument("--train-steps", type=int, default=4)
    ap.add_argument("--train-storage", choices=("fp32", "bf16", "both"), default="both",
                    help="activation / activation-gradient storage of the training leg: bf16 (what BASELINE.json configs[2..4] name; "
                         "fp32 accumulation, fp32 master weights), fp32 (the parity anchor), or both legs (default)")
    ap.add_argument("--mode", choices=("all", "infer", "train", "prep"), default="all",
                    help="all: every leg in one JSON line (default); infer / train: that leg only; "
                         "prep: device-side batch preparation (distance maps + z-score, SURVEY.md 8f N1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="inference legs: edge network and ESF-Net of a batch back to back on one stream")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--config", default="baseline_edge", help="configs/<name>.yaml (baseline_adain_edge = BASELINE.json configs[3])")
    ap.add_argument("--chz", type=int, default=32, help="ESF-Net base width (64 = BASELINE.json configs[4]'s wider model)")
    ap.add_argument("--fit", action="store_true", help="--mode infer: the headline step includes the ellipse-fit stage")
    ap.add_argument("--layers", action="store_true", help="print a per-launch time / TFLOP/s table to stderr")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start the N rank processes (nothing has touched the GPU here)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


def cpu_baseline(setting, bd_sd, net_sd, B, iters):
    """The oracle (CPU port of the reference path: edge + seg + loss, eval, no_grad) on host cores."""
    import torch
    import egne_amd  # noqa: F401
    from egne_amd import synth
    from oracle import bdcn as obdcn, esfnet as oesf
    # the box exposes 256 logical cores but over-subscribing torch's intra-op pool is pathologically slow
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(min(ncores, 32))
    b = synth.make_batch(B, seed=1234)
    times = []
    with torch.no_grad():
        for i in range(iters + 1):
            t0 = time.perf_counter()
            e = obdcn.calc_edge(bd_sd, b["img"])
            oesf.esf_forward(net_sd, setting, b["img"], e, b["label"], b["pupil_center"], b["elNorm"], b["spatWts"],
                             b["distMap"], b["cond"], b["ID"], b["alpha"])
            if i > 0:  # first iteration is warm-up
                times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 4), "unit": "eye-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "B=%d edge+seg+loss fp32 eval, %d timed iterations (median), torch CPU %d threads of %s logical cores"
                      % (B, iters, torch.get_num_threads(), os.cpu_count())}


def bench_prep(a):
    """--mode prep: distance maps (exact EDT x 3 classes) + z-score + boundary weights of B frames per step, inputs resident in HBM."""
    import torch
    import egne_amd  # noqa: F401
    from egne_amd import _lib, dataprep, synth
    _lib.lib()
    torch.cuda.set_device(0)
    B = a.batch or 64
    base = synth.make_batch(min(B, 8), seed=1234)
    rep = (B + 7) // 8
    lab = torch.cat([base["label"]] * rep)[:B].cuda()
    img = torch.cat([base["img"]] * rep)[:B].cuda()

    def step():
        return dataprep.dist_maps(lab), dataprep.zscore(img), dataprep.spatial_weights(lab)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    H, W = lab.shape[1:]
    nbytes = B * H * W * (8 + 3 * 4 + 4 + 4 + 8 + 4)  # label read, 3 maps written, image read + written, label read again + weights written
    ach = nbytes * a.steps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    res = {"metric": "eye-frames/sec (320x240) device-side batch preparation: 3 signed distance maps (exact EDT) + z-score + boundary weights",
           "value": round(B * a.steps / dt, 1), "unit": "eye-frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32 squared distances, f64 sqrt / statistics, f32 out", "data": "synthetic",
           "config": {"workload": "SURVEY.md 8f N1: CurriculumLib.py:128-139 for a batch of %d label maps / frames (the boundary weights are parity-unpinned)" % B,
                      "frames_per_gpu_per_step": B},
           "roofline": {"bound": "hbm", "kernel": "edt_rows_k (brute-force row minimum from LDS: 320 candidates per pixel; latency / "
                                                   "issue bound, far below the HBM roof by design - it is 40x faster than the network)",
                        "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}}
    if not a.no_cpu_baseline:
        from oracle import dataprep as oprep
        n = 4
        t0 = time.perf_counter()
        oprep.dist_maps(lab[:n].cpu().numpy())
        oprep.zscore(img[:n].cpu().numpy())
        oprep.spatial_weights(lab[:n].cpu().numpy())
        res["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 2), "unit": "eye-frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames, scipy.ndimage.distance_transform_edt x 6 per frame + numpy z-score + numpy Canny restatement, one core" % n}
    print(json.dumps(res), flush=True)


class Bench:
    """Networks, synthetic batch and the timed legs of one rank."""

    def __init__(self, a):
        import torch
        import yaml
        import egne_amd  # noqa: F401
        from egne_amd import _lib, synth
        from egne_amd.bdcn_new import BDCN
        from egne_amd.models.RITnet_v2 import DenseNet2D
        self.a, self.torch = a, torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, self.world))
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world)
        _lib.lib()  # fail loudly if the HIP extension is missing
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        with open(os.path.join(os.path.dirname(egne_amd.__file__), "configs", a.config + ".yaml")) as f:
            self.setting = yaml.safe_load(f)
        bd = BDCN()
        bd.load_state_dict(synth.seeded_state_dict(bd.state_dict(), kind="bdcn"))
        net = DenseNet2D(dict(self.setting), chz=a.chz)
        net.load_state_dict(synth.seeded_state_dict(net.state_dict(), kind="esf"))
        self.bd_sd = {k: v.clone() for k, v in bd.state_dict().items()}
        self.net_sd = {k: v.clone() for k, v in net.state_dict().items()}
        self.bd, self.net = bd.to(self.dev).eval(), net.to(self.dev).eval()
        self.args = types.SimpleNamespace(prec=torch.float32, edge_thres=0)
        self._batches = {}

    def batch(self, B):
        """Synthetic TEyeD-shaped batch: render 8 distinct frames on the host, tile to B."""
        torch = self.torch
        if B not in self._batches:
            from egne_amd import synth
            base = synth.make_batch(min(B, 8), seed=1234 + self.rank)
            rep = (B + base["img"].shape[0] - 1) // base["img"].shape[0]
            self._batches = {B: {k: (torch.cat([v] * rep)[:B].to(self.dev) if torch.is_tensor(v) else v) for k, v in base.items()}}
        return self._batches[B]

    def barrier(self):
        if self.world > 1:
            self.torch.distributed.barrier()
        self.torch.cuda.synchronize()

    def free_plans(self):
        self.bd._plans.clear()
        self.net._plans.clear()
        self.net._last_plan = None
        self._batches = {}
        self.torch.cuda.empty_cache()

    # ------------------------------------------------------------------------------------------------------------
    def timed(self, step, steps, warmup, events=True):
        """W untimed steps, then exactly K steps between two barrier + synchronize brackets; MAX over ranks."""
        torch = self.torch
        from egne_amd import engine as _engine
        flush = getattr(step, "flush", lambda: None)
        for _ in range(warmup):
            step()
        flush()                     # (a pipelined step: nothing in flight when the timed region starts)
        ev = []
        self.bd._events = self.net._events = (ev if events and not os.environ.get("EGNE_BENCH_NO_EVENTS") else None)
        # HIP events around every conv launch (the roofline families); all ~600 launches of a step only with --layers:
        # an event pair costs ~2 us of GPU time, 2.5 % of the step when every launch carries one
        _engine.EVENT_KINDS = None if self.a.layers else {"conv_f16x3", "conv_bf16", "conv_igemm", "conv3x3_halo", "conv3x3_smallcin", "conv_wgrad"}
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        last = flush()              # ... and drained inside it: exactly K batches take the whole path
        out = last if last is not None else out
        self.barrier()
        dt = time.perf_counter() - t0
        self.bd._events = self.net._events = None
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = tt.item()
        return dt, ev, out

    def families(self, events, steps, dt):
        fam, per_layer = {}, {}
        for kind, flops, e0, e1, lname in events:
            d = fam.setdefault(kind, [0.0, 0.0, 0])
            sec = e0.elapsed_time(e1) * 1e-3
            d[0] += sec
            d[1] += flops
            d[2] += 1
            pl_ = per_layer.setdefault(lname, [0.0, flops, kind])
            pl_[0] += sec / steps
        if self.a.layers and self.rank == 0:
            from egne_amd import engine as _eng
            for lname, (sec, fl, kind) in per_layer.items():
                nb = _eng.LAYER_BYTES.get(lname, 0.0)
                print("%-26s %-18s %9.1f us %8.2f GFLOP %7.1f TFLOP/s %7.2f GB %5.2f TB/s" %
                      (lname, kind, sec * 1e6, fl / 1e9, fl / sec / 1e12 if sec > 0 else 0, nb / 1e9, nb / sec / 1e12 if sec > 0 else 0), file=sys.stderr)
        return fam

    def rooflines(self, fam, steps, B, dt):
        conv_t, conv_f, conv_n = [sum(fam.get(k, [0.0, 0.0, 0])[i] for k in FP32_FAM) for i in range(3)]
        sub = {k.split(":")[1]: v for k, v in fam.items() if k.startswith("conv_f16x3:")}
        sp_t, sp_f, sp_n = [sum(v[i] for v in sub.values()) for i in range(3)]
        achieved = conv_f / conv_t / 1e12 if conv_t > 0 else 0.0
        r_fp32 = {"bound": "mfma", "kernel": "exact-fp32 conv family on v_mfma_f32_32x32x2_f32: conv_igemm_kernel, conv3x3_halo_kernel, "
                  "conv3x3_c4_kernel (+ conv_wgrad_kernel, conv3x3_wgrad_halo_kernel in training)",
                  "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                  "launches_per_step": conv_n // max(steps, 1), "avg_launch_ms": round(1e3 * conv_t / max(conv_n, 1), 4),
                  "algorithmic_gflop_per_frame": round(conv_f / steps / B / 1e9, 2), "time_share": round(conv_t / dt, 4)}
        sp_ach = sp_f / sp_t / 1e12 if sp_t > 0 else 0.0
        r_split = {"bound": "mfma", "kernel": "split-f16 conv family (fp32 tensors, 3 x v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 per product, fp32 "
                   "accumulate): conv_f16x3_big_kernel, conv3x3_rw_kernel, conv3x3_rs_kernel, fused_1x1_3x3_kernel, msblock_dil_kernel, "
                   "conv3x3_halo_f16_kernel, conv_f16x3_kernel, conv1x1 kernels; inference plans of BDCN and ESF-Net, 3x3 forward convolutions "
                   "data and weight gradients of training plans",
                   "achieved": round(sp_ach, 2), "peak": round(PEAK_F16_MFMA_TFLOPS / 3, 1),
                   "unit": "TFLOP/s (algorithmic, fp32-equivalent; peak = 2500 dense f16 MFMA / 3 MFMAs per product)",
                   "frac": round(sp_ach / (PEAK_F16_MFMA_TFLOPS / 3), 4), "traffic": None,
                   "launches_per_step": sp_n // max(steps, 1), "avg_launch_ms": round(1e3 * sp_t / max(sp_n, 1), 4),
                   "algorithmic_gflop_per_frame": round(sp_f / steps / B / 1e9, 2), "time_share": round(sp_t / dt, 4),
                   "by_kernel": {k: {"tflops": round(v[1] / v[0] / 1e12, 1) if v[0] > 0 else 0.0, "time_share": round(v[0] / dt, 4),
                                     "launches_per_step": v[2] // max(steps, 1)} for k, v in sorted(sub.items())}}
        return r_split, r_fp32, sp_t, conv_t

    # ------------------------------------------------------------------------------------------------------------
    def infer_step(self, B, fit, pipeline=None):
        torch = self.torch
        from egne_amd.utils import calc_edge, fit_ellipses_from_pred
        t, bd, net, args, dev = self.batch(B), self.bd, self.net, self.args, self.dev
        # evaluate.py:135-166 per batch: seeds from elPred, both ellipses of every frame fitted on the device, results copied to
        # the host.  The fit of batch i runs on a second HIP stream next to the network of batch i+1 (a sequential search on
        # 2B workgroups leaves most of the chip idle); the host takes delivery of batch i-1's ellipses before it queues batch i+1.
        host = [torch.empty((B, 2, 5), dtype=torch.float64).pin_memory() for _ in range(2)] if fit else None
        side = torch.cuda.Stream(device=dev) if fit else None
        state = {"n": 0, "done": [None, None]}

        # Two-stage pipeline across batches (unless --no-pipeline): the frozen edge network of batch i runs on stream A while
        # ESF-Net (+ loss / argmax) of batch i-1 runs on stream B.  ESF-Net's small levels (30x40 and 15x20 maps, the regression
        # module, ~100 short launches) leave CUs idle that the edge network's kernels fill: +3.5 % (scratch/overlap.py).  Every
        # batch still takes the whole path; the pipeline is empty when the timed region starts and is drained inside it.
        from egne_amd.pipeline import TwoStagePipeline
        use_pipe = (not self.a.no_pipeline) if pipeline is None else pipeline
        pipe = TwoStagePipeline(args, bd, dev) if use_pipe else None

        def esf(edge):
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"],
                      t["ID"], t["alpha"])
            if fit:
                k = state["n"] & 1
                mask, elp = net.predictions(), out[1]
                ready_f = torch.cuda.Event()
                ready_f.record()
                if state["done"][k] is not None:
                    state["done"][k].synchronize()          # the ellipses of two batches ago have landed in host[k]
                side.wait_event(ready_f)
                with torch.cuda.stream(side):
                    mask.record_stream(side)
                    elp.record_stream(side)
                    host[k].copy_(fit_ellipses_from_pred(mask, elp), non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(side)
                state["done"][k] = done
                state["n"] += 1
            return out

        def step():
            with torch.no_grad():
                if pipe is None:
                    return esf(calc_edge(args, t["img"], bd, dev))
                r = pipe.submit(t["img"], esf)
                return r[0] if r is not None else None

        def flush():
            if pipe is None:
                return None
            with torch.no_grad():
                r = pipe.flush()
            return r[0] if r is not None else None
        step.flush = flush
        return step

    def leg_infer(self, steps, warmup, fit=False, events=True, pipeline=None):
        B = self.a.batch or 64
        dt, ev, out = self.timed(self.infer_step(B, fit, pipeline), steps, warmup, events)
        assert self.torch.isfinite(out[3]).all()
        return B, dt, ev

    def leg_train(self, steps, warmup, events=True, pipeline=None, storage="fp32"):
        torch = self.torch
        from egne_amd import parallel
        from egne_amd.utils import calc_edge
        B = self.a.train_batch
        t, bd, net, args, dev = self.batch(B), self.bd, self.net, self.args, self.dev
        net.to(torch.bfloat16 if storage == "bf16" else torch.float32)       # storage of the training plan's activations (models/RITnet_v2.py: DenseNet2D.to)
        net.train()
        parallel.broadcast_state(net)
        opt = torch.optim.Adam([p for n, p in net.named_parameters() if "dsIdentify" not in n], lr=5e-4)

        def rest(edge):   # train.py:284-287: forward, loss.backward(), (DP) gradient all-reduce, Adam
            opt.zero_grad()          # (train.py:284; PyTorch's default drops the gradient views, the model re-attaches its flat arena with one fill)
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"],
                      t["ID"], t["alpha"])
            out[3].backward()
            parallel.allreduce_grads(net)
            opt.step()
            return [o.detach() for o in out]

        # the frozen edge network of the NEXT batch (train.py:266, no gradient) runs on a second stream next to this batch's
        # forward / backward / optimiser step (egne_amd.pipeline; --no-pipeline: back to back)
        from egne_amd.pipeline import TwoStagePipeline
        use_pipe = (not self.a.no_pipeline) if pipeline is None else pipeline
        pipe = TwoStagePipeline(args, bd, dev) if use_pipe else None

        def step():
            if pipe is None:
                return rest(calc_edge(args, t["img"], bd, dev))
            r = pipe.submit(t["img"], rest)
            return r[0] if r is not None else None

        def flush():
            r = pipe.flush() if pipe is not None else None
            return r[0] if r is not None else None
        step.flush = flush
        dt, ev, out = self.timed(step, steps, warmup, events)
        assert torch.isfinite(out[3]).all()
        net.eval()
        net.to(torch.float32)
        return B, dt, ev


def _debug_prio():
    """diagnostic: EGNE_CONS_PRIO=1 raises the issue priority of the consumer waves of the role-split kernels"""
    if os.environ.get('EGNE_CONS_PRIO'):
        import ctypes as C
        from egne_amd import _lib
        L = _lib.lib()
        v = int(os.environ['EGNE_CONS_PRIO'])
        L.egne_fused_debug.argtypes = [C.c_int, C.c_void_p]; L.egne_msdil_debug.argtypes = [C.c_int, C.c_void_p]
        L.egne_fused_debug(128 * v, None); L.egne_msdil_debug(128 * v, None); L.egne_rs_debug_prio(v)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a)
    if a.mode == "prep":
        return bench_prep(a)
    import torch
    from egne_amd import engine as _engine
    bn = Bench(a)
    _debug_prio()
    world, rank = bn.world, bn.rank
    res = {"metric": "eye-frames/sec (320x240): inference edge+seg (value), +fit, exact fp32, and train fwd+bwd at %d MI355X" % world,
           "value": None, "unit": "eye-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 storage; inference products split into f16 hi/lo pairs (3 f16 MFMAs per product, 22-bit significand), "
                    "f32 accumulate; training: the same products for 3x3 forward convolutions, data and weight gradients, exact f32 MFMA for 1x1", "data": "synthetic"}
    arith = ("fp32 tensors everywhere; inference: split-f16 MFMA products (22-bit significand) with fp32 accumulation where eligible, "
             "exact fp32 elsewhere; training: split-f16 products for the 3x3 forward convolutions, their data and weight gradients (pre-scales "
             "measured on the device every step), exact fp32 MFMA for 1x1 convolutions and everything else")

    if a.mode in ("all", "infer"):
        fit_ = a.fit and a.mode == "infer"
        if a.no_pipeline:
            B, dt, ev = bn.leg_infer(a.steps, a.warmup, fit=fit_)
            dt_k = dt
        else:
            # `value`: the pipelined loop.  Kernel durations for `roofline`: a second timed region of the same run with the two
            # stages of a batch back to back on ONE stream -- under the pipeline an event pair on one stream also spans the other
            # stream's kernels (their sum was 1.75x the step), which says nothing about the kernel between them.
            B, dt, _ = bn.leg_infer(a.steps, a.warmup, fit=fit_, events=False)
            _, dt_k, ev = bn.leg_infer(a.steps, 1, fit=fit_, pipeline=False)
        fam = bn.families(ev, a.steps, dt_k)
        r_split, r_fp32, sp_t, conv_t = bn.rooflines(fam, a.steps, B, dt_k)
        for r in (r_split, r_fp32):
            r["region_ms_per_step"] = round(1e3 * dt_k / a.steps, 3)       # time_share x this = summed kernel time per step
            r["measured_in"] = ("the timed region itself" if a.no_pipeline else
                                "second timed region of this run, stages back to back on one stream: %.3f ms per step "
                                "(time_share refers to it)" % (1e3 * dt_k / a.steps))
        try:   # HBM traffic per launch from the committed PMC passes of this round (bench.py cannot run rocprofv3 on itself)
            with open(os.path.join(ROOT, "profiles", ROUND + "_pmc_traffic.json")) as f:
                tr = json.load(f)["families"]
            if B == 64:
                r_split["traffic"] = tr["split_f16"]["hbm_bytes_per_launch"]
                r_fp32["traffic"] = tr["fp32_conv"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        timed_t = sum(v[0] for v in fam.values())
        res.update({
            "value": round(B * a.steps * world / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 3),
            "config": {"workload": "BASELINE.json configs[1]: %s.yaml (chz=%d) inference, batch=%d/GPU, fp32, 240x320 synthetic IR "
                                   "frames, seeded random-init weights" % (a.config, a.chz, B),
                       "frames_per_gpu_per_step": B, "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                       "ellipse_fit_stage": bool(a.fit and a.mode == "infer"), "arithmetic": arith,
                       "parallelism": "replicas x%d (frames sharded, no collective)" % world,
                       "pipeline": ("none: edge network and ESF-Net of a batch back to back on one stream" if a.no_pipeline else
                                    "two stages across batches: the frozen edge network of batch i on one HIP stream while ESF-Net + loss "
                                    "/ argmax of batch i-1 runs on another; empty when the timed region starts, drained inside it "
                                    "(`roofline` / kernel_time_share: see roofline.measured_in)")},
            "roofline": r_split if sp_t >= conv_t else r_fp32, "roofline_secondary": r_fp32 if sp_t >= conv_t else r_split,
            "algorithmic_gflop_per_frame_total": round((r_split["algorithmic_gflop_per_frame"] + r_fp32["algorithmic_gflop_per_frame"]), 2),
            "kernel_time_share": dict({k.split(":")[0]: 0.0 for k in fam}),
        })
        share = {}
        for k, v in fam.items():
            share[k.split(":")[0]] = share.get(k.split(":")[0], 0.0) + v[0] / dt_k
        if not a.layers:
            share["untimed (elementwise, reductions, layout, loss, host gaps)"] = 1.0 - timed_t / dt_k
        res["kernel_time_share"] = {k: round(v, 4) for k, v in sorted(share.items())}

    if a.mode == "all":
        # ---- edge + seg + fit (north-star target) ----
        B, dt, _ = bn.leg_infer(a.steps, 2, fit=True, events=False)
        res["with_fit"] = {"value": round(B * a.steps * world / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 3), "steps": a.steps,
                           "what": "the inference step + evaluate.py's fit stage: seeds from elPred on the device, 2 ellipse searches per "
                                   "frame in one launch on a second HIP stream (overlaps the next batch's network), fitted ellipses "
                                   "copied to pinned host memory every step; all fits complete inside the timed region"}
        # ---- exact fp32 (split products off) ----
        old = (_engine.F16X3_ENABLED, _engine.ESF_SPLIT)
        _engine.F16X3_ENABLED = _engine.ESF_SPLIT = False
        bn.free_plans()
        ks = max(2, a.steps // 2)
        B, dt, ev = bn.leg_infer(ks, 2, pipeline=False)       # (reference leg: stages back to back, kernel durations undisturbed)
        fam = bn.families(ev, ks, dt) if not a.layers else {}
        _, r32, _, _ = bn.rooflines(fam, ks, B, dt)
        res["exact_fp32"] = {"value": round(B * ks * world / dt, 2), "ms_per_step": round(1e3 * dt / ks, 3), "steps": ks,
                             "what": "same step with EGNE_F16X3=0 EGNE_ESF_SPLIT=0: every convolution on v_mfma_f32_32x32x2_f32",
                             "roofline": {k: r32[k] for k in ("bound", "achieved", "peak", "unit", "frac", "time_share")}}
        _engine.F16X3_ENABLED, _engine.ESF_SPLIT = old
        bn.free_plans()

    if a.mode in ("all", "train"):
        steps = a.train_steps if a.mode == "all" else a.steps
        warm = 2 if a.mode == "all" else a.warmup
        torch.cuda.reset_peak_memory_stats()
        sto = "bf16" if a.train_storage in ("bf16", "both") else "fp32"
        if a.no_pipeline:
            B, dt, ev = bn.leg_train(steps, warm, storage=sto)
            dt_k = dt
        else:       # value from the pipelined loop, kernel durations from a second region with the stages back to back (as for inference)
            B, dt, _ = bn.leg_train(steps, warm, events=False, storage=sto)
            _, dt_k, ev = bn.leg_train(steps, 1, pipeline=False, storage=sto)
        fam = bn.families(ev, steps, dt_k)
        rsp, r32, sp_t, conv_t = bn.rooflines(fam, steps, B, dt_k)
        for r in (rsp, r32):
            r["region_ms_per_step"] = round(1e3 * dt_k / steps, 3)
            r["measured_in"] = ("the timed region itself" if a.no_pipeline else
                                "second timed region of this run, stages back to back on one stream: %.3f ms per step "
                                "(time_share refers to it)" % (1e3 * dt_k / steps))
        rdom, rsec = (rsp, r32) if sp_t >= conv_t else (r32, rsp)        # the family with the larger share of the step first
        tr = {"value": round(B * steps * world / dt, 2), "unit": "eye-frames/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
              "warmup": warm, "frames_per_gpu_per_step": B, "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
              "what": "BASELINE.json configs[2] shape: %s.yaml (chz=%d) train step = frozen BDCN forward + ESF-Net forward + backward + "
                      "gradient all-reduce + Adam, batch=%d/GPU, fp32 storage and accumulation (3x3 forward convs, data and weight gradients on "
                      "split-f16 products, 1x1 exact fp32; EGNE_TRAIN_SPLIT=0 for all-fp32)" % (a.config, a.chz, B),
              "parallelism": "dp%d (one flat RCCL all-reduce of the gradient arena per step)" % world,
              "pipeline": ("none" if a.no_pipeline else "the frozen edge network of batch i+1 on a second HIP stream next to forward / backward / "
                           "Adam of batch i; empty when the timed region starts, drained inside it"),
              "roofline": {k: rdom[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_ms",
                                                "algorithmic_gflop_per_frame", "time_share", "region_ms_per_step", "measured_in")},
              "roofline_secondary": {k: rsec[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step",
                                                          "avg_launch_ms", "algorithmic_gflop_per_frame", "time_share", "region_ms_per_step", "measured_in")}}
        if a.mode == "train":
            res.update({"metric": "eye-frames/sec (320x240) train step: edge fwd + ESF-Net fwd+bwd + grad all-reduce + Adam",
                        "value": tr["value"], "ms_per_step": tr["ms_per_step"], "roofline": rdom, "roofline_secondary": rsec,
                        "config": {"workload": tr["what"], "frames_per_gpu_per_step": B, "peak_hbm_gb": tr["peak_hbm_gb"],
                                   "parallelism": tr["parallelism"], "arithmetic": arith}})
        else:
            res["train"] = tr
        bn.free_plans()

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and a.mode != "train":
            res["cpu_baseline"] = cpu_baseline(bn.setting, bn.bd_sd, bn.net_sd, a.cpu_batch, a.cpu_iters)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

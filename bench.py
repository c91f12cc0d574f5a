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
                       step, batch 256 per GPU (BASELINE.json configs[2] shape) with bf16 STORAGE of activations and activation
                       gradients (fp32 accumulation, fp32 master weights: what configs[2..4] name); `train.fp32_storage` is the
                       same step with fp32 storage (the plan the gradient fixtures pin to the reference).  --config
                       baseline_adain_edge / --chz 64 with --mode train select the configs[3] / configs[4] shapes.

For N>1 the driver launches one process per GPU (torch.distributed.run, RANK/LOCAL_RANK/WORLD_SIZE in the env);
`python bench.py --gpus N` without that environment spawns the N rank processes itself (before anything touches the
GPU in the parent).  Inference shards frames across ranks with no data-path collective (replicas);
training averages gradients with one RCCL all-reduce per step.  Scaling is weak: every rank keeps its own batch.

The JSON line also carries
  roofline      the dominant kernel family of the inference step (split-f16 convolutions: 3 x v_mfma_f32_32x32x16_f16
                per product): algorithmic conv FLOPs of one step / summed duration of its launches, measured with HIP
                events on the launch stream inside the timed region, against 2500 / 3 TFLOP/s;
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's PyTorch path) timed on the host cores per BASELINE.md
                section 3: BDCN forward, ESF forward, ESF forward+backward+Adam and the fit on their own, B=2 and B=8, bounded
                samples (rank 0 at N=1 only); `parity_sample` compares the GPU path with the oracle on that B=2 sample;
  ranks         ranks seen, every rank's own frames/s (min / max) and the whole-job value per GPU (= `value` at --gpus 1);
                training legs add `allreduce_ms_per_step` (HIP events around the RCCL call).
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (as egne_amd/__init__.py: before the first HIP call of the process)

PEAK_HBM_GBS = 8000.0
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"; the split kernel issues 3 MFMAs per product
ROUND = "r06"
FP32_FAM = ("conv_igemm", "conv3x3_halo", "conv3x3_smallcin", "conv_wgrad", "conv3x3_narrow")     # exact-fp32 kernels (the last one on the vector ALU)


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
    ap.add_argument("--edge-products", choices=("auto", "1", "3"), default="auto",
                    help="training legs: products per multiply of the frozen edge network (egne_conv_desc.f16_products).  auto = what train.py does: "
                         "1 (plain f16 operands, fp32 accumulate) next to a bf16-storage plan, which rounds the edge map to bf16 on entry; 3 (the "
                         "22-bit split) next to an fp32-storage plan.  The bf16 leg also reports its rate with 3 (`edge_split3_value`)")
    ap.add_argument("--mode", choices=("all", "infer", "train", "prep"), default="all",
                    help="all: every leg in one JSON line (default); infer / train: that leg only; "
                         "prep: device-side batch preparation (distance maps + z-score, SURVEY.md 8f N1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="inference legs: edge network and ESF-Net of a batch back to back on one stream")
    ap.add_argument("--cpu-budget", type=float, default=7.0, help="seconds per timed part of the CPU baseline (4 parts x 2 batch sizes)")
    ap.add_argument("--config", default="baseline_edge", help="configs/<name>.yaml (baseline_adain_edge = BASELINE.json configs[3])")
    ap.add_argument("--chz", type=int, default=32, help="ESF-Net base width (64 = BASELINE.json configs[4]'s wider model)")
    ap.add_argument("--fit", action="store_true", help="--mode infer: the headline step includes the ellipse-fit stage")
    ap.add_argument("--layers", action="store_true", help="print a per-launch time / TFLOP/s table to stderr")
    ap.add_argument("--force-dist", action="store_true",
                    help="build the RCCL process group even with ONE rank (EGNE_FORCE_DIST=1): the training legs then run the gradient all-reduce, "
                         "the parameter broadcast and the timing all-gather through RCCL and report allreduce_ms_per_step > 0")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline (0: min(affinity cores, 16), see --cpu-thread-sweep)")
    ap.add_argument("--cpu-thread-sweep", action="store_true",
                    help="time the CPU baseline's B=2 edge + seg part at 16 / 32 / 64 / 128 / all logical cores and print the table (one-off: which "
                         "thread count is the fair baseline)")
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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _sources_sha16():
    """Fingerprint of what decides kernel times and HBM traffic: the HIP sources, the C-ABI header and the plan builders.  The PMC
    traffic files under profiles/ carry the fingerprint of the tree they were measured on; the bench line says whether it still
    matches (`roofline.traffic_sources_match`): a copied traffic figure from an older build is labelled as such."""
    import glob
    import hashlib
    pkg = os.path.join(ROOT, "edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "egne_hip.h")] + [os.path.join(pkg, f) for f in ("engine.py", "esf_engine.py", "bdcn_new.py")])
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _timeit(fn, budget_s, warm=2, lo=5, hi=5):
    """``warm`` untimed calls, then between ``lo`` and ``hi`` timed ones -- as many as fit ``budget_s`` judging by the last warm-up.
    Returns (median, min, timed calls, warm-ups) in seconds."""
    last = None
    for _ in range(warm):
        t0 = time.perf_counter()
        fn()
        last = time.perf_counter() - t0
    n = max(lo, min(hi, int(budget_s / max(last, 1e-9))))
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0], n, warm


def cpu_thread_sweep(setting, bd_sd, net_sd, budget_s=6.0):
    """One-off: edge + seg of the B = 2 sample (the `cpu_baseline.value` part) at several torch thread counts on this box."""
    import torch
    import egne_amd  # noqa: F401
    from egne_amd import synth
    from oracle import bdcn as obdcn, esfnet as oesf
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    b = synth.make_batch(2, seed=1234)
    rows = []
    # (more threads than the box's CPU share only over-subscribe it: measured 6.1 frames/s at 8, 6.0 at 16, 4.0 at 32, 1.7 at 64, 0.8 at 128 threads and
    #  no result in seven minutes at 256 on a 16-core share of an EPYC 9575F -- the sweep stops at 64)
    for n in [t for t in (4, 8, 16, 32, 64) if t <= ncores]:
        torch.set_num_threads(n)
        keep = {}

        def f():
            with torch.no_grad():
                keep["e"] = obdcn.calc_edge(bd_sd, b["img"])
                oesf.esf_forward(net_sd, setting, b["img"], keep["e"], b["label"], b["pupil_center"], b["elNorm"], b["spatWts"], b["distMap"],
                                 b["cond"], b["ID"], b["alpha"])
        med, mn, k, w = _timeit(f, budget_s, warm=1, lo=2, hi=4)
        rows.append({"threads": n, "median_s": round(med, 3), "min_s": round(mn, 3), "frames_per_s": round(2 / med, 3), "timed": k})
        print("cpu thread sweep: %4d threads  %.3f s  %.2f frames/s" % (n, med, 2 / med), file=sys.stderr, flush=True)
    return {"what": "oracle edge + seg + loss, fp32, eval, B=2, torch CPU at several intra-op thread counts (one-off sweep: which count is the fair baseline)",
            "affinity_cores": ncores, "os_cpu_count": os.cpu_count(), "cpu_model": _cpu_model(), "rows": rows,
            "best": max(rows, key=lambda r: r["frames_per_s"])}


def cpu_baseline(setting, bd_sd, net_sd, batches=(2, 8), budget_s=7.0, parity=None, threads=0):
    """the oracle (CPU restatement of the reference path, pinned by the reference-generated fixtures) timed on
    this box's host cores, on bounded samples: BDCN forward, ESF-Net forward (+ loss), ESF-Net forward + backward + Adam, and the
    ellipse fit, each on its own; B = 2 (BASELINE.json configs[0]) and B = 8; 2 warm-ups and up to 5 timed iterations per part (as
    many as fit ~7 s per part: the whole leg stays near a minute), median and minimum.  ``value`` = frames/s of edge + seg at B = 2
    (the sum of the two medians), what round 1 / 2 reported.  ``parity``: callable(batch, edge, logits) -> dict, run on the B = 2
    sample so that the line also says how the GPU path compares with the oracle on it."""
    import numpy as np
    import torch
    import egne_amd  # noqa: F401
    from egne_amd import synth
    from oracle import bdcn as obdcn, esfnet as oesf, fit as ofit
    # the box exposes 256 logical cores but a one-GPU lease has a 16-core share: the thread sweep (profiles/r04_cpu_thread_sweep.json:
    # 6.1 frames/s at 8 threads, 6.0 at 16, 4.0 at 32, 1.7 at 64, 0.8 at 128) puts the fair baseline at <= 16 threads
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(threads if threads > 0 else min(ncores, 16))
    out = {"kind": "port", "unit": "eye-frames/s", "cores": torch.get_num_threads(), "os_cpu_count": os.cpu_count(),
           "affinity_cores": ncores, "cpu_model": _cpu_model(), "torch_threads": torch.get_num_threads(),
           "protocol": "BASELINE.md section 3: 2 warm-ups (1 for the training part), 5 timed iterations per part, median and min",
           "parts": {}}
    t_all = time.perf_counter()
    for B in batches:
        b = synth.make_batch(B, seed=1234)
        keep = {}

        def f_bdcn():
            with torch.no_grad():
                keep["e"] = obdcn.calc_edge(bd_sd, b["img"])

        def f_esf():
            with torch.no_grad():
                keep["o"] = oesf.esf_forward(net_sd, setting, b["img"], keep["e"], b["label"], b["pupil_center"], b["elNorm"], b["spatWts"],
                                             b["distMap"], b["cond"], b["ID"], b["alpha"])
        sd = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else v.clone()) for k, v in net_sd.items()}
        opt = torch.optim.Adam([v for k, v in sd.items() if v.requires_grad and "dsIdentify" not in k], lr=5e-4)

        def f_train():          # train.py:284-287 on the oracle: zero_grad, forward (training mode), backward, Adam
            opt.zero_grad()
            oesf.esf_forward(sd, setting, b["img"], keep["e"], b["label"], b["pupil_center"], b["elNorm"], b["spatWts"],
                             b["distMap"], b["cond"], b["ID"], b["alpha"], training=True)[3].sum().backward()
            opt.step()
        rec = {}
        for name, fn in (("bdcn_forward", f_bdcn), ("esf_forward_loss", f_esf), ("esf_forward_backward_adam", f_train)):
            med, mn, n, w = _timeit(fn, budget_s, warm=2 if name != "esf_forward_backward_adam" else 1)
            rec[name] = {"median_s": round(med, 4), "min_s": round(mn, 4), "frames_per_s": round(B / med, 3), "timed": n, "warmups": w}
        if B == batches[0]:
            # evaluate.py:148-151: two ellipse searches per frame from the network's own mask and ellipse head
            mask = keep["o"][0].argmax(1).numpy()
            H, W = mask.shape[1:]
            Hm = np.array([[W / 2, 0, W / 2], [0, H / 2, H / 2], [0, 0, 1]])
            elp = keep["o"][1].numpy()

            def f_fit():
                for c, sl in ((1, slice(0, 5)), (2, slice(5, 10))):
                    ofit.fit_ellipse(mask[0] == c, list(ofit.transform(elp[0, sl].astype(np.float64), Hm)))
            med, mn, n, w = _timeit(f_fit, budget_s, warm=1)
            rec["ellipse_fit_one_frame"] = {"median_s": round(med, 4), "min_s": round(mn, 4), "frames_per_s": round(1.0 / med, 3), "timed": n,
                                            "warmups": w, "what": "2 ellipse searches (iris, pupil) of ONE frame, one core (sequential numpy search)"}
            if parity is not None:
                out["parity_sample"] = parity(b, keep["e"], keep["o"])
        rec["edge_seg_frames_per_s"] = round(B / (rec["bdcn_forward"]["median_s"] + rec["esf_forward_loss"]["median_s"]), 3)
        out["parts"]["B=%d" % B] = rec
    first = out["parts"]["B=%d" % batches[0]]
    out["value"] = first["edge_seg_frames_per_s"]
    out["sample"] = ("edge + seg + loss, fp32, eval, B=%d: sum of the median BDCN and ESF-Net forward times; parts (BDCN forward, ESF forward, "
                     "ESF forward+backward+Adam, fit) at B=%s in `parts`; torch CPU %d threads of %s logical cores (%s); whole leg %.0f s"
                     % (batches[0], "/".join(str(x) for x in batches), torch.get_num_threads(), os.cpu_count(), out["cpu_model"],
                        time.perf_counter() - t_all))
    return out


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
        self.dist = self.world > 1 or a.force_dist           # collectives are issued (more than one rank, or a forced one-rank RCCL group)
        if self.dist:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if a.force_dist:
                os.environ["EGNE_FORCE_DIST"] = "1"
            torch.cuda.set_device(local)
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

    def parity_sample(self, b, edge_ref, ref):
        """The GPU path on the CPU baseline's B = 2 sample against what the oracle just computed there (the oracle as the checker,
        inside the cpu_baseline leg only): edge map and logit errors, and the argmax masks pixel by pixel -- mismatches are
        counted, and so are the pixels whose two largest reference logits lie within 2e-3 (the only ones allowed to differ)."""
        torch = self.torch
        from egne_amd.utils import calc_edge
        dev = self.dev
        self.net.load_state_dict(self.net_sd)            # (the training legs have stepped the weights and the BatchNorm statistics)
        self.net.eval()
        with torch.no_grad():
            edge = calc_edge(self.args, b["img"].to(dev), self.bd, dev)
            out = self.net(b["img"].to(dev), edge, b["label"].to(dev), b["pupil_center"].to(dev), b["elNorm"].to(dev), b["spatWts"].to(dev),
                           b["distMap"].to(dev), b["cond"].to(dev), b["ID"].to(dev), b["alpha"])
            mask = self.net.predictions().cpu()
        top2 = ref[0].topk(2, dim=1).values
        near = (top2[:, 0] - top2[:, 1]) < 2e-3
        diff = mask != ref[0].argmax(1)
        return {"frames": int(b["img"].shape[0]), "edge_max_abs_err": float((edge.cpu() - edge_ref).abs().max()),
                "logits_max_abs_err": float((out[0].cpu() - ref[0]).abs().max()), "loss_rel_err": float(abs(out[3].item() - ref[3].item()) / abs(ref[3].item())),
                "mask_mismatch_pixels": int(diff.sum()), "mask_mismatch_pixels_outside_near_ties": int((diff & ~near).sum()),
                "near_tie_pixels_lt_2e-3": int(near.sum()), "pixels": int(mask.numel())}

    def barrier(self):
        if self.dist:
            self.torch.distributed.barrier()
        self.torch.cuda.synchronize()

    def leg_latency(self, calls=100):
        """edge + seg + fit of ONE call of 1 / 2 frames, synchronised after every call (a caller that needs this frame's ellipses
        before the next frame arrives): eager launch loop and hipGraph replay (egne_amd.pipeline.GraphedFrames); results of both
        are checked to be identical.  Returns (ms of the faster form at two frames, details)."""
        torch = self.torch
        from egne_amd.evaluate import _seg_and_fit, graphed_runner
        from egne_amd.utils import calc_edge
        self.net.eval()
        out = {"what": "wall ms per call of edge map + ESF-Net + argmax mask + both ellipse fits, torch.cuda.synchronize() after every call, frames "
                       "resident in HBM, median of %d calls; `no_fit`: the same without the fit stage; `graph`: the call as one hipGraph replay" % calls}

        def med(fn):
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return round(1e3 * sorted(ts)[len(ts) // 2], 3)
        for B in (1, 2):
            x = self.batch(B)["img"]

            def eager(fit=True):
                with torch.no_grad():
                    e = calc_edge(self.args, x, self.bd, self.dev)
                    if fit:
                        return _seg_and_fit(x, self.net)(e)
                    z = lambda *s: torch.zeros(s, device=self.dev)          # noqa: E731
                    lab = torch.zeros((B, H_, W_), dtype=torch.long, device=self.dev)
                    self.net(x, e, lab, z(B, 2), z(B, 2, 5), z(B, H_, W_), z(B, 3, H_, W_), z(B, 4), torch.zeros(B, dtype=torch.long, device=self.dev), 0)
                    return self.net.predictions()
            H_, W_ = x.shape[2:]
            for _ in range(5):
                want = eager()
            torch.cuda.synchronize()
            run = graphed_runner(x, self.net, self.bd)
            got = run(x)
            torch.cuda.synchronize()
            same = all(torch.equal(p, q) for p, q in zip(want, got))
            eager(False)
            out["b%d" % B] = {"eager": med(eager), "graph": med(lambda: run(x)), "no_fit": med(lambda: eager(False)), "graph_equals_eager": bool(same)}
        self.free_plans()
        return min(out["b2"]["eager"], out["b2"]["graph"]), out

    def free_plans(self):
        self.bd._plans.clear()
        self.net._plans.clear()
        self.net._last_plan = None
        self._batches = {}
        import gc
        gc.collect()                # a training plan and its backward plan refer to each other: only the cycle collector frees their buffers
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
        _engine.EVENT_KINDS = None if self.a.layers else {"conv_f16x3", "conv_bf16", "conv_igemm", "conv3x3_halo", "conv3x3_smallcin", "conv_wgrad", "conv3x3_narrow"}
        self.barrier()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = step()
        last = flush()              # ... and drained inside it: exactly K batches take the whole path
        out = last if last is not None else out
        self.barrier()
        dt = time.perf_counter() - t0
        self.bd._events = self.net._events = None
        self.rank_dts = [dt]
        if self.dist:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            every = [torch.zeros_like(tt) for _ in range(self.world)]
            torch.distributed.all_gather(every, tt)             # every rank's own clock around its K steps (both barriers inside)
            self.rank_dts = [float(x.item()) for x in every]
            dt = max(self.rank_dts)                              # the job's time: the slowest rank
        return dt, ev, out

    def rank_report(self, B, steps):
        """What the multi-GPU scaling run needs to be read without guessing: how many ranks took part, each rank's own rate, and
        the whole-job value per GPU (at --gpus 1 this IS `value`)."""
        rates = [B * steps / t for t in self.rank_dts]
        return {"ranks_seen": int(self.torch.distributed.get_world_size()) if self.dist else self.world,
                "process_group": ("nccl (RCCL), %d rank(s)%s" % (self.world, ", forced" if self.a.force_dist else "")) if self.dist else "none (one process)",
                "per_rank_frames_per_s_min": round(min(rates), 2), "per_rank_frames_per_s_max": round(max(rates), 2),
                "value_per_gpu": round(B * steps / max(self.rank_dts), 2)}

    def families(self, events, steps, dt):
        from egne_amd.engine import LAYER_BYTES as _LB
        fam, per_layer = {}, {}
        for kind, flops, e0, e1, lname in events:
            d = fam.setdefault(kind, [0.0, 0.0, 0, 0.0])
            sec = e0.elapsed_time(e1) * 1e-3
            d[0] += sec
            d[1] += flops
            d[2] += 1
            d[3] += _LB.get(lname, 0.0)          # algorithmic bytes of the launch (engine.LAYER_BYTES: inputs read once, output written once)
            pl_ = per_layer.setdefault(lname, [0.0, flops, kind])
            pl_[0] += sec / steps
        if self.a.layers and self.rank == 0:
            from egne_amd import engine as _eng
            for lname, (sec, fl, kind) in per_layer.items():
                nb = _eng.LAYER_BYTES.get(lname, 0.0)
                print("%-26s %-18s %9.1f us %8.2f GFLOP %7.1f TFLOP/s %7.2f GB %5.2f TB/s" %
                      (lname, kind, sec * 1e6, fl / 1e9, fl / sec / 1e12 if sec > 0 else 0, nb / 1e9, nb / sec / 1e12 if sec > 0 else 0), file=sys.stderr)
        return fam

    def rooflines(self, fam, steps, B, dt, products=3):
        """``products``: f16 MFMAs per multiply in the split family of this leg -- 3 (hi hi + hi lo + lo hi) everywhere except the
        frozen edge network next to a bf16-storage training plan (``edge_products`` = 1: plain f16 operands, ONE MFMA per product; that
        leg's split family is the edge network alone).  The family's MFMA roof is 2500 / products (round-5 verdict: the bf16 leg had
        been priced against 2500 / 3)."""
        conv_t, conv_f, conv_n = [sum(fam.get(k, [0.0, 0.0, 0])[i] for k in FP32_FAM) for i in range(3)]
        conv_b = sum(fam.get(k, [0.0, 0.0, 0, 0.0])[3] for k in FP32_FAM)
        sub = {k.split(":")[1]: v for k, v in fam.items() if k.startswith("conv_f16x3:")}
        sp_t, sp_f, sp_n = [sum(v[i] for v in sub.values()) for i in range(3)]
        sp_b = sum(v[3] for v in sub.values())
        achieved = conv_f / conv_t / 1e12 if conv_t > 0 else 0.0
        r_fp32 = {"bound": "mfma", "kernel": "exact-fp32 conv family on v_mfma_f32_32x32x2_f32: conv_igemm_kernel, conv3x3_halo_kernel, "
                  "conv3x3_c4_kernel (+ conv_wgrad_kernel, conv3x3_wgrad_halo_kernel in training)",
                  "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                  "launches_per_step": conv_n // max(steps, 1), "avg_launch_ms": round(1e3 * conv_t / max(conv_n, 1), 4),
                  "algorithmic_gflop_per_frame": round(conv_f / steps / B / 1e9, 2), "time_share": round(conv_t / dt, 4),
                  "algorithmic_bytes_per_step": int(conv_b / max(steps, 1))}
        sp_ach = sp_f / sp_t / 1e12 if sp_t > 0 else 0.0
        r_split = {"bound": "mfma", "kernel": "split-f16 conv family (fp32 tensors, %d x v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 per product, fp32 " % products +
                   "accumulate): conv_f16x3_big_kernel (plain f16: conv_f16_big1_kernel), conv3x3_rw_kernel, conv3x3_rs_kernel, fused_1x1_3x3_kernel, msblock_dil_kernel (plain f16: msdil1_kernel), "
                   "conv3x3_halo_f16_kernel, conv_f16x3_kernel, conv1x1 kernels; inference plans of BDCN and ESF-Net, 3x3 forward convolutions "
                   "data and weight gradients of training plans",
                   "achieved": round(sp_ach, 2), "peak": round(PEAK_F16_MFMA_TFLOPS / products, 1), "mfmas_per_product": products,
                   "unit": "TFLOP/s (algorithmic, fp32-equivalent; peak = 2500 dense f16 MFMA / %d MFMA%s per product%s)"
                           % (products, "s" if products > 1 else "", "" if products == 3 else
                              ": plain f16 operands in the deep trunk, resident-weights, halo, flat and dilated-group kernels, conv1_1 / conv1_2 / pool1 stored as f16; conv1_1 keeps three"),
                   "frac": round(sp_ach / (PEAK_F16_MFMA_TFLOPS / products), 4), "traffic": None,
                   "launches_per_step": sp_n // max(steps, 1), "avg_launch_ms": round(1e3 * sp_t / max(sp_n, 1), 4),
                   "algorithmic_gflop_per_frame": round(sp_f / steps / B / 1e9, 2), "time_share": round(sp_t / dt, 4),
                   "algorithmic_bytes_per_step": int(sp_b / max(steps, 1)),
                   "by_kernel": {k: {"tflops": round(v[1] / v[0] / 1e12, 1) if v[0] > 0 else 0.0, "time_share": round(v[0] / dt, 4),
                                     "launches_per_step": v[2] // max(steps, 1)} for k, v in sorted(sub.items())}}
        bf = {k.split(":")[1]: v for k, v in fam.items() if k.startswith("conv_bf16:")}
        self.r_bf16 = None
        if bf:
            bt, bfl, bn_, bby = [sum(v[i] for v in bf.values()) for i in range(4)]
            gbs = bby / bt / 1e9 if bt > 0 else 0.0
            self.r_bf16 = {"bound": "hbm", "kernel": "bf16-storage conv family of the training plan (bf16 tensors, one v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 "
                           "per product, fp32 accumulate): conv3x3_bf16_kernel (3x3 forward + data gradients), conv1x1_bf16_kernel (1x1 over raw "
                           "slices, forward + data gradients), wgrad3x3_bf16_kernel, wgrad1x1_bf16_kernel",
                           "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s (algorithmic bytes: every input slice read once, the output "
                           "written once, the accumulated slice of a data gradient read once more; bf16)",
                           "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "traffic_source": "not measured in this run",
                           "launches_per_step": bn_ // max(steps, 1), "avg_launch_ms": round(1e3 * bt / max(bn_, 1), 4),
                           "algorithmic_gb_per_frame": round(bby / steps / B / 1e9, 4), "algorithmic_gflop_per_frame": round(bfl / steps / B / 1e9, 2),
                           "tflops": round(bfl / bt / 1e12, 1) if bt > 0 else 0.0, "time_share": round(bt / dt, 4),
                           "by_kernel": {k: {"gb_per_s": round(v[3] / v[0] / 1e9, 1) if v[0] > 0 else 0.0, "tflops": round(v[1] / v[0] / 1e12, 1) if v[0] > 0 else 0.0,
                                             "time_share": round(v[0] / dt, 4), "launches_per_step": v[2] // max(steps, 1)} for k, v in sorted(bf.items())}}
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
        # (EGNE_FIT_PRIO=1: the network's two streams at high priority, the fit stream at normal priority -- measured)
        state = {"n": 0, "done": [None, None]}

        # Two-stage pipeline across batches (unless --no-pipeline): the frozen edge network of batch i runs on stream A while
        # ESF-Net (+ loss / argmax) of batch i-1 runs on stream B.  ESF-Net's small levels (30x40 and 15x20 maps, the regression
        # module, ~100 short launches) leave CUs idle that the edge network's kernels fill: +3.5 % (scratch/overlap.py).  Every
        # batch still takes the whole path; the pipeline is empty when the timed region starts and is drained inside it.
        from egne_amd.pipeline import TwoStagePipeline
        use_pipe = (not self.a.no_pipeline) if pipeline is None else pipeline
        pipe = TwoStagePipeline(args, bd, dev) if use_pipe else None
        from egne_amd.pipeline import WindowedFit
        wfit = WindowedFit(dev, windowed=use_pipe) if fit else None

        def esf(edge):
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"],
                      t["ID"], t["alpha"])
            if fit:
                k = state["n"] & 1
                mask, elp = net.predictions(), out[1]
                if state["done"][k] is not None:
                    state["done"][k].synchronize()          # the ellipses of two batches ago have landed in host[k]
                # (egne_amd.pipeline.WindowedFit: the searches are released where the NEXT batch's ESF-Net reaches its low-resolution
                #  levels, next to launches that hand one workgroup to each free CU -- not next to persistent ones)
                state["done"][k] = wfit.submit(mask, elp, then=lambda res, k=k: host[k].copy_(res, non_blocking=True))
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
            for d in state["done"]:          # every search of the timed region is finished inside it
                if d is not None:
                    d.synchronize()
            return r[0] if r is not None else None
        step.flush = flush
        return step

    def leg_infer(self, steps, warmup, fit=False, events=True, pipeline=None):
        B = self.a.batch or 64
        dt, ev, out = self.timed(self.infer_step(B, fit, pipeline), steps, warmup, events)
        assert steps == 0 or self.torch.isfinite(out[3]).all()
        return B, dt, ev

    def leg_train(self, steps, warmup, events=True, pipeline=None, storage="fp32", edge_products=None):
        torch = self.torch
        if edge_products is None:
            edge_products = {"auto": 1 if storage == "bf16" else 3, "1": 1, "3": 3}[self.a.edge_products]
        self.bd.f16_products = 1 if edge_products == 1 else 0        # (egne_amd/train.py: --prec 16 sets it on its edge network)
        self.edge_products = edge_products
        from egne_amd import parallel
        from egne_amd.utils import calc_edge
        B = self.a.train_batch
        t, bd, net, args, dev = self.batch(B), self.bd, self.net, self.args, self.dev
        net.to(torch.bfloat16 if storage == "bf16" else torch.float32)       # storage of the training plan's activations (models/RITnet_v2.py: DenseNet2D.to)
        net.train()
        parallel.broadcast_state(net)
        parallel.overlap_grads(net)      # (as egne_amd/train.py: two-bucket all-reduce, the first bucket issued inside the backward pass)
        opt = torch.optim.Adam([p for n, p in net.named_parameters() if "dsIdentify" not in n], lr=5e-4, fused=True)     # (as egne_amd/train.py)

        def rest(edge):   # train.py:284-287: forward, loss.backward(), (DP) gradient all-reduce, Adam
            opt.zero_grad()          # (train.py:284; PyTorch's default drops the gradient views, the model re-attaches its flat arena with one fill)
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"],
                      t["ID"], t["alpha"])
            out[3].backward()
            if self.dist:            # one flat RCCL all-reduce of the gradient arena (13.45 MB for baseline_edge), timed with HIP events on the stream it runs on
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                parallel.allreduce_grads(net)
                e1.record()
                self.ar_events.append((e0, e1))
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
        self.ar_events = []
        dt, ev, out = self.timed(step, steps, warmup, events)
        assert steps == 0 or torch.isfinite(out[3]).all()
        ar = [a.elapsed_time(b) for a, b in self.ar_events[-steps:]] if self.ar_events else []
        self.allreduce_ms = round(sum(ar) / len(ar), 4) if ar else 0.0
        self.grad_bytes = int(net._grad_flat.numel() * 4) if getattr(net, "_grad_flat", None) is not None else 0
        net.eval()
        net.to(torch.float32)
        self.bd.f16_products = 0
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
           "dtype": "f32 tensors; inference products split into f16 hi/lo pairs (3 f16 MFMAs per product, 22-bit significand), f32 accumulate "
                    "(`exact_fp32`: every product on the f32 MFMA); `train`: bf16 storage, f32 accumulate, f32 master weights (and an f32-storage leg)",
           "data": "synthetic"}
    arith = ("fp32 tensors everywhere; inference: split-f16 MFMA products (22-bit significand) with fp32 accumulation where eligible, "
             "exact fp32 elsewhere; training: split-f16 products for the 3x3 forward convolutions, their data and weight gradients (pre-scales "
             "measured on the device every step), exact fp32 MFMA for 1x1 convolutions and everything else")

    if a.steps == 0:
        # warm-up only (plan building, weight packing, calibration, W untimed steps, nothing else): the run a profiler trace of
        # `--steps K` is compared with, so that per-step kernel counts and times of the STEADY state come out by subtraction
        if a.mode == "train":
            bn.leg_train(0, a.warmup, events=False, pipeline=not a.no_pipeline, storage="bf16" if a.train_storage in ("bf16", "both") else "fp32")
        else:
            bn.leg_infer(0, a.warmup, fit=a.fit, events=False, pipeline=not a.no_pipeline)
        if rank == 0:
            print(json.dumps({"warmup_only": True, "warmup": a.warmup, "mode": a.mode}), flush=True)
        return
    if a.mode in ("all", "infer"):
        fit_ = a.fit and a.mode == "infer"
        if a.no_pipeline:
            B, dt, ev = bn.leg_infer(a.steps, a.warmup, fit=fit_)
            dt_k = dt
            ranks_inf = bn.rank_report(B, a.steps)
        else:
            # `value`: the pipelined loop.  Kernel durations for `roofline`: a second timed region of the same run with the two
            # stages of a batch back to back on ONE stream -- under the pipeline an event pair on one stream also spans the other
            # stream's kernels (their sum was 1.75x the step), which says nothing about the kernel between them.
            B, dt, _ = bn.leg_infer(a.steps, a.warmup, fit=fit_, events=False)
            ranks_inf = bn.rank_report(B, a.steps)
            _, dt_k, ev = bn.leg_infer(a.steps, 1, fit=fit_, pipeline=False)
        fam = bn.families(ev, a.steps, dt_k)
        r_split, r_fp32, sp_t, conv_t = bn.rooflines(fam, a.steps, B, dt_k)
        for r in (r_split, r_fp32):
            r["region_ms_per_step"] = round(1e3 * dt_k / a.steps, 3)       # time_share x this = summed kernel time per step
            r["measured_in"] = ("the timed region itself" if a.no_pipeline else
                                "second timed region of this run, stages back to back on one stream: %.3f ms per step "
                                "(time_share refers to it)" % (1e3 * dt_k / a.steps))
        # HBM traffic per launch: bench.py cannot run rocprofv3 on itself, so the figure is COPIED from the committed PMC passes of the
        # latest round that has them (profiles/rNN_pmc_traffic.json, same command, B = 64) and labelled as such
        for r in (r_split, r_fp32):
            r["traffic_source"] = "not measured in this run"
        for rnd in (ROUND, "r05", "r04", "r02"):
            try:
                with open(os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")) as f:
                    trj = json.load(f)
                    tr = trj["families"]
                for r in (r_split, r_fp32):
                    r["traffic_sources_match"] = trj.get("sources_sha16") == _sources_sha16()
                if B == 64:
                    r_split["traffic"] = tr["split_f16"]["hbm_bytes_per_launch"]
                    r_fp32["traffic"] = tr["fp32_conv"]["hbm_bytes_per_launch"]
                    for r, key in ((r_split, "split_f16"), (r_fp32, "fp32_conv")):
                        r["traffic_source"] = ("copied from profiles/%s_pmc_traffic.json (rocprofv3 --pmc passes over `bench.py --mode infer --no-pipeline`, "
                                               "FETCH_SIZE x2 + WRITE_SIZE per launch), not measured in this run" % rnd)
                        # per STEP, next to the algorithmic bytes of the same launches (every input slice read once, the output written once):
                        # a ratio well above 1 would mean wasted re-reads
                        r["traffic_bytes_per_step"] = int(1e9 * (tr[key]["hbm_read_gb_per_step"] + tr[key]["hbm_write_gb_per_step"]))
                        if r.get("algorithmic_bytes_per_step"):
                            r["traffic_over_algorithmic"] = round(r["traffic_bytes_per_step"] / r["algorithmic_bytes_per_step"], 3)
                break
            except Exception:
                continue
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
            "ranks": ranks_inf,
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

    if a.mode == "all" and world == 1:
        # ---- one or two frames per call: the head-mounted-display case (evaluate.py:235-249 feeds one eye at a time) ----
        res["latency_b2_ms"], res["latency"] = bn.leg_latency()

    def train_leg(storage):
        steps = a.train_steps if a.mode == "all" else a.steps
        warm = 2 if a.mode == "all" else a.warmup
        bn.free_plans()
        torch.cuda.reset_peak_memory_stats()
        if a.no_pipeline:
            B, dt, ev = bn.leg_train(steps, warm, storage=storage)
            dt_k = dt
            ranks = bn.rank_report(B, steps)
        else:       # value from the pipelined loop, kernel durations from a second region with the stages back to back (as for inference)
            B, dt, _ = bn.leg_train(steps, warm, events=False, storage=storage)
            ranks = bn.rank_report(B, steps)
            _, dt_k, ev = bn.leg_train(steps, 1, pipeline=False, storage=storage)
        fam = bn.families(ev, steps, dt_k)
        rsp, r32, sp_t, conv_t = bn.rooflines(fam, steps, B, dt_k, products=1 if (storage == "bf16" and bn.edge_products == 1) else 3)
        rbf = bn.r_bf16
        meas = ("the timed region itself" if a.no_pipeline else
                "second timed region of this run, stages back to back on one stream: %.3f ms per step (time_share refers to it)" % (1e3 * dt_k / steps))
        for r in (rsp, r32) + ((rbf,) if rbf else ()):
            r["region_ms_per_step"] = round(1e3 * dt_k / steps, 3)
            r["measured_in"] = meas
            r.setdefault("traffic_source", "not measured in this run")
        if rbf and storage == "bf16" and B == 256 and a.config == "baseline_edge" and a.chz == 32:
            try:        # HBM bytes per launch of the bf16 family from the committed PMC passes of the same command (not measured in this run)
                rnd_t = next(r for r in (ROUND, "r05") if os.path.exists(os.path.join(ROOT, "profiles", r + "_pmc_traffic_train_b256.json")))
                with open(os.path.join(ROOT, "profiles", rnd_t + "_pmc_traffic_train_b256.json")) as f:
                    trj = json.load(f)
                    rbf["traffic"] = trj["families"]["bf16_conv"]["hbm_bytes_per_launch"]
                    rbf["traffic_sources_match"] = trj.get("sources_sha16") == _sources_sha16()
                    rbf["step_hbm_gb"] = round(sum(v["hbm_read_gb_per_step"] + v["hbm_write_gb_per_step"] for v in trj["families"].values()), 1)
                rbf["traffic_source"] = ("copied from profiles/%s_pmc_traffic_train_b256.json (rocprofv3 --pmc passes over `bench.py --mode train --train-batch 256 "
                                         "--train-storage bf16 --no-pipeline`, FETCH_SIZE x2 + WRITE_SIZE per launch), not measured in this run" % rnd_t)
            except Exception:
                pass
        cands = sorted([r for r in (rbf, rsp, r32) if r and r["time_share"] > 0], key=lambda r: -r["time_share"])
        rdom, rsec = cands[0], (cands[1] if len(cands) > 1 else None)        # the family with the larger share of the step first
        which = {"baseline_edge": "configs[2]", "baseline_adain_edge": "configs[3] (per-GPU shard: 256 of the global 1024)"}.get(a.config, a.config)
        if a.chz == 64:
            which = "configs[4] (the 64-channel model, per-GPU shard: 256 of the global 2048)"
        if storage == "bf16":
            what = ("BASELINE.json %s shape: %s.yaml (chz=%d) train step = frozen BDCN forward + ESF-Net forward + backward + gradient all-reduce + "
                    "Adam, batch=%d/GPU, bf16 STORAGE of activations and activation gradients in HBM, fp32 accumulation, fp32 master weights / "
                    "gradient arena / optimiser (3x3 and raw-slice 1x1 convolutions, their data and weight gradients on bf16 MFMAs with weights "
                    "rounded to bf16; everything else fp32 arithmetic on bf16 tensors)" % (which, a.config, a.chz, B))
            dty = ("bf16 storage, f32 accumulate, f32 master weights (frozen edge network: f32 tensors, %s)"
                   % ("plain f16 operands / f32 accumulate in the deep trunk layers -- its output is rounded to bf16 on entry; `edge_split3_value`: "
                      "the same step with the 22-bit split products" if bn.edge_products == 1 else "split-f16 products"))
        else:
            what = ("BASELINE.json %s shape with FP32 storage (the parity anchor): %s.yaml (chz=%d) train step = frozen BDCN forward + ESF-Net "
                    "forward + backward + gradient all-reduce + Adam, batch=%d/GPU, fp32 storage and accumulation (3x3 forward convs, data and "
                    "weight gradients on split-f16 products, 1x1 exact fp32; EGNE_TRAIN_SPLIT=0 for all-fp32)" % (which, a.config, a.chz, B))
            dty = "f32 storage; 3x3 products split into f16 hi/lo pairs (22-bit significand), f32 accumulate; 1x1 exact f32 MFMA"
        keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_sources_match", "step_hbm_gb", "launches_per_step", "avg_launch_ms",
                "algorithmic_gflop_per_frame", "time_share", "region_ms_per_step", "measured_in", "by_kernel", "algorithmic_gb_per_frame", "tflops",
                "mfmas_per_product", "algorithmic_bytes_per_step")
        tr = {"value": round(B * steps * world / dt, 2), "unit": "eye-frames/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
              "warmup": warm, "frames_per_gpu_per_step": B, "dtype": dty, "storage": storage,
              "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "what": what,
              "parallelism": "dp%d (one flat RCCL all-reduce of the %.2f MB gradient arena per step)" % (world, bn.grad_bytes / 1e6),
              "allreduce_ms_per_step": bn.allreduce_ms, "ranks": ranks,
              "pipeline": ("none" if a.no_pipeline else "the frozen edge network of batch i+1 on a second HIP stream next to forward / backward / "
                           "Adam of batch i; empty when the timed region starts, drained inside it"),
              "roofline": {k: rdom[k] for k in keys if k in rdom}}
        if rsec is not None:
            tr["roofline_secondary"] = {k: rsec[k] for k in keys if k in rsec}
        tr["edge_products"] = bn.edge_products
        if storage == "bf16" and bn.edge_products == 1 and a.edge_products == "auto":
            # the same step with the frozen edge network on the 22-bit split products (what the fp32-storage leg and every inference leg use)
            s3 = max(2, steps // 2)
            B3, dt3, _ = bn.leg_train(s3, 1, events=False, storage=storage, edge_products=3)
            tr["edge_split3_value"] = round(B3 * s3 * world / dt3, 2)
            tr["edge_split3_ms_per_step"] = round(1e3 * dt3 / s3, 3)
        bn.free_plans()
        return tr, rdom, rsec

    if a.mode in ("all", "train"):
        first = "bf16" if a.train_storage in ("bf16", "both") else "fp32"
        tr, rdom, rsec = train_leg(first)
        if a.train_storage == "both":
            tr["fp32_storage"], _, _ = train_leg("fp32")        # the parity anchor (the gradient fixtures pin THIS plan to the reference's fp32 gradients)
            tr["fp32_storage"].pop("roofline_secondary", None)
        if a.mode == "train":
            res.update({"metric": "eye-frames/sec (320x240) train step: edge fwd + ESF-Net fwd+bwd + grad all-reduce + Adam",
                        "value": tr["value"], "ms_per_step": tr["ms_per_step"], "dtype": tr["dtype"], "roofline": rdom,
                        "config": {"workload": tr["what"], "frames_per_gpu_per_step": tr["frames_per_gpu_per_step"], "peak_hbm_gb": tr["peak_hbm_gb"],
                                   "parallelism": tr["parallelism"], "storage": tr["storage"]},
                        "allreduce_ms_per_step": tr["allreduce_ms_per_step"], "ranks": tr["ranks"]})
            if rsec is not None:
                res["roofline_secondary"] = rsec
            if "fp32_storage" in tr:
                res["fp32_storage"] = tr["fp32_storage"]
            for k in ("edge_products", "edge_split3_value", "edge_split3_ms_per_step"):
                if k in tr:
                    res[k] = tr[k]
        else:
            res["train"] = tr

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and a.mode != "train":
            res["cpu_baseline"] = cpu_baseline(bn.setting, bn.bd_sd, bn.net_sd, budget_s=a.cpu_budget, parity=bn.parity_sample, threads=a.cpu_threads)
            if a.cpu_thread_sweep:
                res["cpu_thread_sweep"] = cpu_thread_sweep(bn.setting, bn.bd_sd, bn.net_sd)
        # the other legs' headline numbers as top-level scalars AND inside `config` (a record that keeps only the standard keys still has them)
        extra = {}
        if "with_fit" in res:
            extra.update(with_fit_value=res["with_fit"]["value"], with_fit_ms_per_step=res["with_fit"]["ms_per_step"])
        if "exact_fp32" in res:
            extra.update(exact_fp32_value=res["exact_fp32"]["value"])
        if "train" in res:
            extra.update(train_value=res["train"]["value"], train_ms_per_step=res["train"]["ms_per_step"],
                         train_allreduce_ms_per_step=res["train"]["allreduce_ms_per_step"])
            if "edge_split3_value" in res["train"]:
                extra.update(train_edge_split3_value=res["train"]["edge_split3_value"])
            if "fp32_storage" in res["train"]:
                extra.update(train_fp32_storage_value=res["train"]["fp32_storage"]["value"])
        if "latency_b2_ms" in res:
            extra.update(latency_b2_ms_edge_seg_fit=res["latency_b2_ms"])
        res.update(extra)
        if isinstance(res.get("config"), dict):
            res["config"].update(extra)
        print(json.dumps(res), flush=True)
    if bn.dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json): eye-frames/s at 320x240.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch that is already resident in HBM:
frozen BDCN edge extractor -> ESF-Net (baseline_edge, chz=32) -> loss head + argmax mask
(BASELINE.json configs[1]: inference, batch 64 per GPU, fp32).  For N>1 the driver launches one
process per GPU (torch.distributed.run); frames shard across ranks with no data-path collective
(inference replicas), so scaling is weak: every rank processes its own 64 frames.

The JSON line also carries
  roofline     -- the dominant kernel family (implicit-GEMM conv on fp32 MFMA): algorithmic conv FLOPs
                  of one step / the summed duration of its conv launches, measured with HIP events
                  on the launch stream inside the timed region, against the 157.3 TFLOP/s fp32 MFMA peak;
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's PyTorch path) timed on the host
                  cores on a bounded sample (B=2, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"; the split kernel issues 3 MFMAs per product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 64 infer / 32 train)")
    ap.add_argument("--mode", choices=("infer", "train", "prep"), default="infer",
                    help="infer: BASELINE.json configs[1] (headline); train: fwd+bwd+all-reduce+Adam step (fp32); "
                         "prep: device-side batch preparation (distance maps + z-score, SURVEY.md 8f N1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--config", default="baseline_edge", help="configs/<name>.yaml (baseline_adain_edge = BASELINE.json configs[3])")
    ap.add_argument("--chz", type=int, default=32, help="ESF-Net base width (64 = BASELINE.json configs[4]'s wider model)")
    ap.add_argument("--fit", action="store_true", help="inference: also run the ellipse-fit stage of evaluate.py (2 fits per frame)")
    ap.add_argument("--layers", action="store_true", help="print a per-launch time / TFLOP/s table to stderr")
    return ap.parse_args()


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
    """--mode prep: distance maps (exact EDT x 3 classes) + z-score of B frames per step, inputs resident in HBM."""
    import numpy as np
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
        return dataprep.dist_maps(lab), dataprep.zscore(img)
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
    nbytes = B * H * W * (8 + 3 * 4 + 4 + 4)          # label read, 3 maps written, image read + written
    ach = nbytes * a.steps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    res = {"metric": "eye-frames/sec (320x240) device-side batch preparation: 3 signed distance maps (exact EDT) + z-score",
           "value": round(B * a.steps / dt, 1), "unit": "eye-frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32 squared distances, f64 sqrt / statistics, f32 out", "data": "synthetic",
           "config": {"workload": "SURVEY.md 8f N1: CurriculumLib.py:131-139 for a batch of %d label maps / frames" % B,
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
        res["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 2), "unit": "eye-frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames, scipy.ndimage.distance_transform_edt x 6 per frame + numpy z-score, one core" % n}
    print(json.dumps(res), flush=True)


def main():
    a = parse()
    if a.mode == "prep":
        return bench_prep(a)
    import torch
    import yaml
    import egne_amd  # noqa: F401
    from egne_amd import _lib, synth
    from egne_amd.bdcn_new import BDCN
    from egne_amd.models.RITnet_v2 import DenseNet2D
    from egne_amd.utils import calc_edge

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    _lib.lib()  # fail loudly if the HIP extension is missing
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    with open(os.path.join(os.path.dirname(egne_amd.__file__), "configs", a.config + ".yaml")) as f:
        setting = yaml.safe_load(f)
    bd = BDCN()
    bd.load_state_dict(synth.seeded_state_dict(bd.state_dict(), kind="bdcn"))
    net = DenseNet2D(dict(setting), chz=a.chz)
    net.load_state_dict(synth.seeded_state_dict(net.state_dict(), kind="esf"))
    bd_sd = {k: v.clone() for k, v in bd.state_dict().items()}
    net_sd = {k: v.clone() for k, v in net.state_dict().items()}
    bd, net = bd.to(dev).eval(), net.to(dev).eval()
    train = a.mode == "train"
    if train:
        from egne_amd import parallel
        net.train()
        parallel.broadcast_state(net)
        opt = torch.optim.Adam([p for n, p in net.named_parameters() if "dsIdentify" not in n], lr=5e-4)

    B = a.batch or (32 if train else 64)
    # synthetic TEyeD-shaped batch: render 8 distinct frames on the host, tile to B
    base = synth.make_batch(min(B, 8), seed=1234 + rank)
    rep = (B + base["img"].shape[0] - 1) // base["img"].shape[0]
    t = {k: (torch.cat([v] * rep)[:B].to(dev) if torch.is_tensor(v) else v) for k, v in base.items()}
    args = types.SimpleNamespace(prec=torch.float32, edge_thres=0)

    def step():
        if train:   # train.py:262-287: frozen edge net, forward, loss.backward(), (DP) gradient all-reduce, Adam
            edge = calc_edge(args, t["img"], bd, dev)
            opt.zero_grad(set_to_none=False)
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"],
                      t["ID"], t["alpha"])
            out[3].backward()
            parallel.allreduce_grads(net)
            opt.step()
            return [o.detach() for o in out]
        with torch.no_grad():
            edge = calc_edge(args, t["img"], bd, dev)
            out = net(t["img"], edge, t["label"], t["pupil_center"], t["elNorm"], t["spatWts"], t["distMap"], t["cond"],
                      t["ID"], t["alpha"])
            if a.fit:   # evaluate.py:135-151: un-normalise the predicted ellipses (host, float64) and fit both on the device
                import numpy as np
                from egne_amd import ellipse
                from egne_amd.utils import fit_ellipses
                ep = out[1].cpu().numpy().astype(np.float64)
                Hm = np.array([[160.0, 0, 160.0], [0, 120.0, 120.0], [0, 0, 1]])
                init = np.stack([ellipse.transform(ep[i, k:k + 5], Hm) for i in range(B) for k in (0, 5)])
                fit_ellipses(net.predictions(), [i for i in range(B) for _ in (0, 1)], [1, 2] * B, init)
            return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    events = []
    bd._events = net._events = (None if os.environ.get("EGNE_BENCH_NO_EVENTS") else events)
    # HIP events around every conv launch (the roofline families); all ~600 launches of a step only with --layers:
    # an event pair costs ~2 us of GPU time, 2.5 % of the step when every launch carries one
    from egne_amd import engine as _engine
    _engine.EVENT_KINDS = None if a.layers else {"conv_f16x3", "conv_igemm", "conv3x3_halo", "conv3x3_smallcin", "conv_wgrad"}
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    bd._events = net._events = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out[3]).all()

    # per-kernel-family time from the HIP events recorded on the launch stream during the timed steps
    fam = {}
    per_layer = {}
    for kind, flops, e0, e1, lname in events:
        d = fam.setdefault(kind, [0.0, 0.0, 0])
        d[0] += e0.elapsed_time(e1) * 1e-3
        d[1] += flops
        d[2] += 1
        pl_ = per_layer.setdefault(lname, [0.0, flops, kind])
        pl_[0] += e0.elapsed_time(e1) * 1e-3 / a.steps
    FP32_FAM = ("conv_igemm", "conv3x3_halo", "conv3x3_smallcin", "conv_wgrad")
    conv_t, conv_f, conv_n = [sum(fam.get(k, [0.0, 0.0, 0])[i] for k in FP32_FAM) for i in range(3)]
    # the split-f16 family is reported as a whole and per kernel (kind "conv_f16x3:<kernel>")
    sub = {k.split(":")[1]: v for k, v in fam.items() if k.startswith("conv_f16x3:")}
    for k in [k for k in fam if k.startswith("conv_f16x3:")]:
        v = fam.pop(k)
        d = fam.setdefault("conv_f16x3", [0.0, 0.0, 0])
        d[0] += v[0]; d[1] += v[1]; d[2] += v[2]
    sp_t, sp_f, sp_n = fam.get("conv_f16x3", [0.0, 0.0, 0])
    if a.layers and rank == 0:
        for lname, (sec, fl, kind) in per_layer.items():
            print("%-26s %-18s %9.1f us %8.2f GFLOP %7.1f TFLOP/s" % (lname, kind, sec * 1e6, fl / 1e9, fl / sec / 1e12 if sec > 0 else 0),
                  file=sys.stderr)
    frames = B * a.steps * world
    res = None
    if rank == 0:
        achieved = conv_f / conv_t / 1e12 if conv_t > 0 else 0.0
        res = {
            "metric": ("eye-frames/sec (320x240) train step: edge fwd + ESF-Net fwd+bwd + grad all-reduce + Adam" if train else
                       "eye-frames/sec (320x240) inference edge+seg (BDCN -> ESF-Net -> loss/argmax)"),
            "value": round(frames / dt, 2), "unit": "eye-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%s.yaml (chz=%d) TRAIN step (BASELINE.json configs[2] shape in fp32, batch=%d/GPU), 240x320 "
                                    "synthetic TEyeD-shaped batch, seeded random-init weights" % (a.config, a.chz, B)) if train
                       else ("BASELINE.json configs[1]: %s.yaml (chz=%d) inference, batch=%d/GPU, fp32, "
                             "240x320 synthetic IR frames, seeded random-init weights" % (a.config, a.chz, B)),
                       "frames_per_gpu_per_step": B, "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                       "ellipse_fit_stage": bool(a.fit),
                       "arithmetic": "fp32 tensors everywhere; training: exact fp32 MFMA; inference: split-f16 MFMA products "
                                     "(22-bit significand) with fp32 accumulation where eligible, exact fp32 elsewhere",
                       "parallelism": ("dp%d (one flat RCCL all-reduce of 13.45 MB per step)" % world) if train
                       else "replicas x%d (frames sharded, no collective)" % world},
            "roofline": None, "roofline_secondary": None,
            "algorithmic_gflop_per_frame_total": round((conv_f + sp_f) / a.steps / B / 1e9, 2),
            # share of the timed region (wall clock); without --layers only the conv families carry HIP events
            "kernel_time_share": dict({k: round(v[0] / dt, 4) for k, v in sorted(fam.items())},
                                      **({} if a.layers else {"untimed (elementwise, reductions, layout, loss, host gaps)":
                                                              round(1.0 - sum(x[0] for x in fam.values()) / dt, 4)})),
        }
        tot_t = max(dt, 1e-9)
        r_fp32 = {"bound": "mfma", "kernel": "exact-fp32 implicit-GEMM conv family on v_mfma_f32_32x32x2_f32: conv_igemm_kernel, "
                  "conv3x3_halo_kernel, conv3x3_c4_kernel (+ conv_wgrad_kernel, conv3x3_wgrad_halo_kernel in train mode); all training convs, fused-affine / wide 1x1 convs in inference",
                  "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                  "launches_per_step": conv_n // max(a.steps, 1), "avg_launch_ms": round(1e3 * conv_t / max(conv_n, 1), 4),
                  "algorithmic_gflop_per_frame": round(conv_f / a.steps / B / 1e9, 2), "time_share": round(conv_t / tot_t, 4)}
        sp_ach = sp_f / sp_t / 1e12 if sp_t > 0 else 0.0
        r_split = {"bound": "mfma", "kernel": "split-f16 conv family (fp32 tensors, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate): "
                   "conv_f16x3_big_kernel (deep 256-wide trunk tiles), conv3x3_halo_f16_kernel, conv_f16x3_kernel, conv1x1_f16x3_kernel; inference plans of BDCN and ESF-Net",
                   "achieved": round(sp_ach, 2), "peak": round(PEAK_F16_MFMA_TFLOPS / 3, 1),
                   "unit": "TFLOP/s (algorithmic, fp32-equivalent; peak = 2500 dense f16 MFMA / 3 MFMAs per product)",
                   "frac": round(sp_ach / (PEAK_F16_MFMA_TFLOPS / 3), 4), "traffic": None,
                   "launches_per_step": sp_n // max(a.steps, 1), "avg_launch_ms": round(1e3 * sp_t / max(sp_n, 1), 4),
                   "algorithmic_gflop_per_frame": round(sp_f / a.steps / B / 1e9, 2), "time_share": round(sp_t / tot_t, 4),
                   # per kernel: big = conv_f16x3_big_kernel (MFMA-issue bound), halo / lattice = conv3x3_halo_f16_kernel (narrow
                   # full-resolution layers and the lattice launches are HBM-bound), flat = conv_f16x3_kernel, stream1x1 /
                   # gemm1x1 = the 1x1 kernels (HBM-bound), first = conv3x3_c4_f16_kernel (HBM write stream)
                   "by_kernel": {k: {"tflops": round(v[1] / v[0] / 1e12, 1) if v[0] > 0 else 0.0, "time_share": round(v[0] / tot_t, 4),
                                     "launches_per_step": v[2] // max(a.steps, 1)} for k, v in sorted(sub.items())}}
        # HBM traffic per launch from the committed PMC run (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE);
        # inference plans only -- bench.py cannot run the PMC passes itself
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                tr = json.load(f)["families"]
            if not train and B == 64:
                r_split["traffic"] = tr["split_f16"]["hbm_bytes_per_launch"]
                r_fp32["traffic"] = tr["fp32_conv"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        res["roofline"], res["roofline_secondary"] = (r_split, r_fp32) if sp_t > conv_t else (r_fp32, r_split)
        if world == 1 and not a.no_cpu_baseline and not train:
            res["cpu_baseline"] = cpu_baseline(setting, bd_sd, net_sd, a.cpu_batch, a.cpu_iters)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

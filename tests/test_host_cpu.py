"""CPU (-m "not gpu"): host logic, key schemas, C-ABI surface, DP collectives over gloo."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from common import GOLD, ROOT, bdcn_module, esf_module, gold


def test_state_dict_schema_matches_reference():
    """Checkpoint key names / shapes must round-trip unchanged."""
    import yaml
    import egne_amd
    from egne_amd.models.RITnet_v2 import DenseNet2D
    from egne_amd.models.RITnet_concat import DenseNet2D as DC
    ref = json.load(open(os.path.join(GOLD, "state_keys.json")))
    cfgd = os.path.join(os.path.dirname(egne_amd.__file__), "configs")
    got = {k: list(v.shape) for k, v in bdcn_module().state_dict().items()}
    assert got == ref["bdcn"]
    from egne_amd.modelSummary import get_model, model_dict
    assert {"ritnet_v1", "ritnet_v2", "ritnet_concat", "deepvog"} <= set(model_dict)  # modelSummary.py:18-26 registry
    assert {k: list(v.shape) for k, v in get_model("ritnet_v1", None).state_dict().items()} == ref["ritnet_v1"]
    assert {k: list(v.shape) for k, v in get_model("deepvog", None).state_dict().items()} == ref["deepvog"]
    for key, want in ref.items():
        if key in ("bdcn", "ritnet_v1", "deepvog"):
            continue
        parts = key.split(":")
        st = yaml.safe_load(open(os.path.join(cfgd, parts[1] + ".yaml")))
        m = (DenseNet2D if parts[0] == "v2" else DC)(st)
        if len(parts) > 2:
            m.setDatasetInfo(4)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == want, key


def test_param_counts():
    """ESF edge 3 363 494, adain_edge 6 353 810, baseline 2 608 965, BDCN 16 302 120."""
    n = lambda m: sum(p.numel() for p in m.parameters())  # noqa: E731
    assert n(esf_module("baseline_edge")) == 3363494
    assert n(esf_module("baseline_adain_edge")) == 6353810
    assert n(esf_module("baseline")) == 2608965
    assert n(bdcn_module()) == 16302120


def test_width_generalisation_builds():
    """chz != 32 crashes in the reference; here the widths follow section 8a-note."""
    from common import setting
    from egne_amd.models.RITnet_v2 import DenseNet2D
    from egne_amd.esf_engine import dec_sizes
    assert dec_sizes(32, 1.2, True, "v2") == dict(ip=[306, 180, 100, 62], op=[180, 100, 62, 32], skip=[243, 172, 102, 64])
    assert dec_sizes(32, 1.2, False, "v2")["ip"] == [153, 115, 76, 38]
    assert dec_sizes(32, 1.2, True, "concat") == dict(ip=[306, 115, 76, 38], op=[115, 76, 38, 32], skip=[486, 344, 204, 128])
    for chz in (16, 64):
        m = DenseNet2D(dict(setting("baseline_edge")), chz=chz)
        assert m.elReg.c1.weight.shape[1] == 2 * int(1.2 * chz * 4)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports exactly what include/egne_hip.h declares."""
    from egne_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "egne_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(egne_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "header declares %s but the library does not export it" % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert L.egne_version() >= 100
    # struct layouts agree between ctypes and the C compiler
    assert L.egne_sizeof(0) == __import__("ctypes").sizeof(_lib.ConvDesc)
    assert L.egne_sizeof(1) == __import__("ctypes").sizeof(_lib.LossDesc)
    assert L.egne_sizeof(2) == __import__("ctypes").sizeof(_lib.BdcnTailDesc)
    assert L.egne_sizeof(3) == __import__("ctypes").sizeof(_lib.Dst)
    assert L.egne_sizeof(4) == __import__("ctypes").sizeof(_lib.ConvQuery)
    assert L.egne_sizeof(5) == __import__("ctypes").sizeof(_lib.ConvChoice)


def test_no_cpu_fallback_and_loud_failure():
    from egne_amd import synth
    m = esf_module("baseline_edge")
    b = synth.make_batch(1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(b["img"], b["img"], b["label"], b["pupil_center"], b["elNorm"], b["spatWts"], b["distMap"], b["cond"], b["ID"], 0.5)
    with pytest.raises(RuntimeError):
        bdcn_module()(torch.zeros(1, 3, 32, 32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_metrics_match_reference():
    """utils.getSeg_metrics / getPoint_metric / get_predictions / normPts / unnormPts (host side)."""
    from egne_amd import utils as U
    g = gold("metrics")
    miou, pc, sl = U.getSeg_metrics(g["yt"].astype(np.int64), g["yp"].astype(np.int64), g["cond"])
    np.testing.assert_allclose(miou, g["miou"], rtol=1e-12)
    np.testing.assert_allclose(pc, g["perclass"], rtol=1e-12)
    np.testing.assert_allclose(sl, g["scorelist"], rtol=1e-12, equal_nan=True)
    H, W = g["yt"].shape[1:]
    d, dv = U.getPoint_metric(g["pts_true"], g["pts_pred"], g["cond"], (H, W), True)
    np.testing.assert_allclose(d, g["pdist"], rtol=1e-6)
    np.testing.assert_allclose(dv, g["pdist_v"], rtol=1e-6, atol=1e-9)
    d2, _ = U.getPoint_metric(g["pts_true"], g["pts_true"] + 1.5, g["cond"], (H, W), False)
    np.testing.assert_allclose(d2, g["pdist2"], rtol=1e-6)
    pred = U.get_predictions(torch.from_numpy(g["logits"])).numpy()
    assert np.array_equal(pred.astype(np.uint8), g["pred"])
    np.testing.assert_allclose(U.normPts(torch.from_numpy(g["pts_true"].astype(np.float32)), (H, W)).numpy(), g["norm"])
    np.testing.assert_allclose(U.unnormPts(g["pts_pred"].astype(np.float32), (H, W)), g["unnorm"])
    mg = U.create_meshgrid(5, 7)
    assert tuple(mg.shape) == (1, 5, 7, 2) and mg[0, 0, 0, 0] == -1 and mg[0, 4, 6, 1] == 1


def test_seeded_weights_are_deterministic():
    from egne_amd import synth
    m = esf_module("baseline_edge")
    a = synth.seeded_state_dict(m.state_dict(), seed=3)
    b = synth.seeded_state_dict(m.state_dict(), seed=3)
    c = synth.seeded_state_dict(m.state_dict(), seed=4)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["enc.head.conv1.weight"], c["enc.head.conv1.weight"])


def test_grad_arena_views():
    """p.grad are views of one flat buffer; zero_grad(set_to_none=True) is survived."""
    m = esf_module("baseline_edge")
    flat = m._ensure_grad_arena()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    p = m.enc.head.conv1.weight
    p.grad.fill_(2.0)
    assert flat[:p.numel()].eq(2.0).all()
    for q in m.parameters():
        q.grad = None
    flat2 = m._ensure_grad_arena()
    assert flat2.data_ptr() == flat.data_ptr() and p.grad.data_ptr() == flat.data_ptr() and flat.eq(0).all()


_DP_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch, torch.distributed as dist
from common import esf_module
from egne_amd import parallel
rank, world = parallel.init('gloo')
assert world == 2
m = esf_module('baseline_edge', seed=rank)            # ranks start different on purpose
parallel.broadcast_state(m)
ref = esf_module('baseline_edge', seed=0)
assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref.state_dict().values()))
flat = m._ensure_grad_arena()
g = torch.Generator().manual_seed(100 + rank)
flat.copy_(torch.randn(flat.numel(), generator=g))
mine = flat.clone()
parallel.allreduce_grads(m)
# expected: mean of both ranks' gradients, visible through every p.grad view
g0 = torch.randn(flat.numel(), generator=torch.Generator().manual_seed(100))
g1 = torch.randn(flat.numel(), generator=torch.Generator().manual_seed(101))
assert torch.allclose(flat, (g0 + g1) / 2, atol=1e-6)
w = m.dec.final.conv2.weight
off = sum(p.numel() for p in list(m.parameters())[:[id(q) for q in m.parameters()].index(id(w))])
assert torch.allclose(w.grad.reshape(-1), ((g0 + g1) / 2)[off:off + w.numel()], atol=1e-6)
# async form + loss averaging + sharding
work, fin = parallel.allreduce_grads(m, async_op=True); work.wait(); fin()
l = parallel.mean_loss(torch.tensor([float(rank + 1)]))
assert abs(l.item() - 1.5) < 1e-6
lo, hi = parallel.shard(10)
assert (lo, hi) == (rank * 5, rank * 5 + 5)
# one SGD step from identical weights + averaged grads keeps the replicas identical
opt = torch.optim.SGD(m.parameters(), lr=0.1); opt.step()
chk = torch.stack([p.detach().double().sum() for p in m.parameters()]).sum().reshape(1)
both = [torch.zeros_like(chk) for _ in range(2)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1])
# data sharding of train.py: disjoint shards of ONE per-epoch permutation, reshuffled by set_epoch, same length on every rank
from egne_amd import _entry
ds = _entry.SyntheticEyes(10, seed=1)
ts, vs = parallel.samplers(ds, ds, rank, world)
for ep in range(2):
    ts.set_epoch(ep)
    idx = torch.tensor(list(ts))
    allidx = [torch.zeros_like(idx) for _ in range(2)]
    dist.all_gather(allidx, idx)
    assert len(idx) == 5 and len(set(allidx[0].tolist()) | set(allidx[1].tolist())) == 10, allidx
    if ep == 0: first = idx.clone()
assert not torch.equal(first, idx)
assert sorted(list(vs)) == list(range(rank, 10, 2))
# validation shards drop nothing and repeat nothing, also when the set does not divide by the world size
_, vs11 = parallel.samplers(ds, _entry.SyntheticEyes(11, seed=1), rank, world)
cnt = torch.tensor([len(vs11)]); both_n = [torch.zeros_like(cnt) for _ in range(2)]
dist.all_gather(both_n, cnt)
assert list(vs11) == list(range(rank, 11, 2)) and int(both_n[0] + both_n[1]) == 11
try:
    parallel.samplers(ds, _entry.SyntheticEyes(1, seed=1), rank, world)
    raise SystemExit("a validation set smaller than the world must be refused")
except RuntimeError:
    pass
s = parallel.sum_over_ranks([1.0 + rank, 10.0])
assert s == [3.0, 20.0]
m.enc.head.bn.running_mean.fill_(float(rank + 7))
parallel.broadcast_buffers(m)
assert m.enc.head.bn.running_mean.eq(7.0).all()
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
"""


def test_data_parallel_collectives_gloo_world2(tmp_path):
    """N>1 path on CPU: two processes over gloo -- broadcast, flat-arena all-reduce(avg), shard, step."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "ok" in o, "rank %d failed:\n%s" % (r, o[-3000:])


_DP4_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch, torch.distributed as dist
from common import esf_module
from egne_amd import parallel
rank, world = parallel.init('gloo')
assert world == 4
m = esf_module('baseline_adain_edge', seed=rank)
parallel.broadcast_state(m)
flat = m._ensure_grad_arena()
gens = [torch.randn(flat.numel(), generator=torch.Generator().manual_seed(300 + r)) for r in range(4)]
want = (gens[0] + gens[1] + gens[2] + gens[3]) / 4
# one bucket
flat.copy_(gens[rank]); parallel.allreduce_grads(m)
assert torch.allclose(flat, want, atol=1e-6)
# two buckets: the tail (everything behind the encoder's block) goes out first, as the backward plan's hook does
parallel.overlap_grads(m)
gc = m.grad_comm
s = gc.split()
names = [n for n, _ in m.named_parameters()]
nenc = sum(p.numel() for n, p in m.named_parameters() if n.startswith('enc.'))
assert s == nenc and 0 < s < flat.numel() and names[0].startswith('enc.'), (s, nenc)
flat.copy_(gens[rank])
gc.tail_ready()
assert gc.pending
# (the encoder's gradients are still being written while the tail is in flight)
flat[:s].add_(1.0); flat[:s].sub_(1.0)
parallel.allreduce_grads(m)
assert not gc.pending and torch.allclose(flat, want, atol=1e-6)
# a backward pass that never reached the hook (a foreign model, a plan without encoder launches): the one-bucket path
flat.copy_(gens[rank]); parallel.allreduce_grads(m)
assert torch.allclose(flat, want, atol=1e-6)
# shards, samplers and scalar sums at four ranks
lo, hi = parallel.shard(1024)
assert (lo, hi) == (rank * 256, rank * 256 + 256)
assert parallel.sum_over_ranks([float(rank), 1.0]) == [6.0, 4.0]
from egne_amd import _entry
ts, vs = parallel.samplers(_entry.SyntheticEyes(10, seed=1), _entry.SyntheticEyes(11, seed=1), rank, world)
idx = torch.tensor(list(ts)); allidx = [torch.zeros_like(idx) for _ in range(4)]
dist.all_gather(allidx, idx)
assert len(idx) == 2 and len(set(sum((a.tolist() for a in allidx), []))) == 8
assert list(vs) == list(range(rank, 11, 4))
opt = torch.optim.SGD(m.parameters(), lr=0.1); opt.step()
chk = torch.stack([p.detach().double().sum() for p in m.parameters()]).sum().reshape(1)
every = [torch.zeros_like(chk) for _ in range(4)]
dist.all_gather(every, chk)
assert all(torch.equal(every[0], e) for e in every)
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
"""


def test_data_parallel_collectives_gloo_world4(tmp_path):
    """Four ranks over gloo (BASELINE.json configs[3]: DP over 4 GPUs): the flat all-reduce, its two-bucket form whose tail bucket is
    issued first (parallel.GradOverlap: what the backward plan does when it reaches the encoder), shards, samplers."""
    script = tmp_path / "dp4_worker.py"
    script.write_text(_DP4_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="4", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(4)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "ok" in o, "rank %d failed:\n%s" % (r, o[-3000:])


_DP8_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch, torch.distributed as dist
torch.set_num_threads(1)
from common import esf_module, setting, batch_args
from egne_amd import parallel, synth, _entry
from egne_amd.models.RITnet_v2 import DenseNet2D
from oracle import esfnet as oesf
rank, world = parallel.init('gloo')
assert world == 8
# -- (1) one data-parallel step of train.py's loop with REAL gradients: every rank owns one frame of a global batch of 8 (its
#    shard of one shared permutation), computes loss and gradients on it with its own BatchNorm statistics and its own loss
#    normalisation (oracle autograd: the HIP path needs a GPU), the arena is averaged over the ranks, Adam steps
cfg = 'baseline_edge'
m = esf_module(cfg, seed=rank, disentangle=True)          # ranks start different on purpose
parallel.broadcast_state(m)
parallel.overlap_grads(m)
params = [p for n, p in m.named_parameters() if 'dsIdentify' not in n]
opt = torch.optim.Adam(params, lr=5e-4)
ts, vs = parallel.samplers(_entry.SyntheticEyes(16, seed=1), _entry.SyntheticEyes(9, seed=2), rank, world)
ts.set_epoch(0)
mine = list(ts)
assert len(mine) == 2
idx = torch.tensor(mine); allidx = [torch.zeros_like(idx) for _ in range(8)]
dist.all_gather(allidx, idx)
assert len(set(sum((a.tolist() for a in allidx), []))) == 16           # disjoint shards of one permutation
assert list(vs) == list(range(rank, 9, 8))                               # rank 0 validates two frames, the others one: nothing dropped
H, W = 240, 320                                                          # (the regression head's Linear layer fixes the frame size)
b = synth.make_batch(1, H=H, W=W, seed=1000 + mine[0], mask_absent_every=8)
sd = {k: (v.detach().clone().requires_grad_(v.dtype.is_floating_point and k in dict(m.named_parameters())) if torch.is_tensor(v) else v)
      for k, v in m.state_dict().items()}
edge = torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(rank))
out = oesf.esf_forward(sd, setting(cfg), *batch_args(b, edge), training=True, disentangle=True)
out[3].mean().backward()
flat = m._ensure_grad_arena()
flat.zero_()
for n, p in m.named_parameters():
    if sd[n].grad is not None:
        p.grad.copy_(sd[n].grad)
local = flat.clone()
assert local.abs().sum() > 0 and torch.isfinite(local).all()
every = [torch.zeros_like(local) for _ in range(8)]
dist.all_gather(every, local)
want = torch.stack(every).double().sum(0).div(8).float()
# the backward plan's hook issues the tail bucket (everything behind the encoder's block), allreduce_grads the rest
gc = m.grad_comm
gc.tail_ready()
assert gc.pending
try:
    gc.tail_ready()                                   # a second backward pass before the reduce must fail loudly (round-5 advisor finding)
    raise SystemExit('a second tail_ready with the first bucket pending must raise')
except RuntimeError:
    pass
parallel.allreduce_grads(m)
assert not gc.pending
err = (flat - want).abs().max().item() / want.abs().max().item()
assert err < 1e-6, err
opt.step()
chk = torch.stack([p.detach().double().sum() for p in m.parameters()] + [p.detach().double().abs().sum() for p in m.parameters()]).reshape(-1)
allchk = [torch.zeros_like(chk) for _ in range(8)]
dist.all_gather(allchk, chk)
assert all(torch.equal(allchk[0], c) for c in allchk), 'replicas diverged after one data-parallel Adam step'
ref0 = esf_module(cfg, seed=0, disentangle=True)
moved = sum(float((p - q).abs().sum()) for (n, p), (_, q) in zip(m.named_parameters(), ref0.named_parameters()) if 'dsIdentify' not in n)
assert moved > 0
# -- (2) the two-bucket split on the arenas of configs[3] (AdaIN modules) and configs[4] (64-channel model)
for cfg2, chz in (('baseline_adain_edge', 32), ('baseline_edge', 64)):
    m2 = DenseNet2D(dict(setting(cfg2)), chz=chz)
    parallel.overlap_grads(m2)
    f2 = m2._ensure_grad_arena()
    s = m2.grad_comm.split()
    nenc = sum(p.numel() for n, p in m2.named_parameters() if n.startswith('enc.'))
    assert s == nenc and 0 < s < f2.numel(), (cfg2, chz, s, nenc, f2.numel())
    g = torch.Generator().manual_seed(7)
    base = torch.randn(f2.numel(), generator=g)
    f2.copy_(base * (rank + 1))
    m2.grad_comm.tail_ready()
    assert m2.grad_comm.pending
    parallel.allreduce_grads(m2)
    assert torch.allclose(f2, base * 4.5, rtol=1e-5, atol=1e-6), (cfg2, chz)
    if rank == 0:
        print('arena', cfg2, chz, f2.numel() * 4 // 1024, 'KiB, early bucket %%.0f %%%%' %% (100.0 * (f2.numel() - s) / f2.numel()))
lo, hi = parallel.shard(2048)
assert (lo, hi) == (rank * 256, rank * 256 + 256)                        # configs[4]: global batch 2048 = 8 x 256
assert parallel.sum_over_ranks([float(rank), 1.0]) == [28.0, 8.0]
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
"""


def test_data_parallel_step_gloo_world8(tmp_path):
    """Eight ranks over gloo (BASELINE.json configs[4]: DP over 8 GPUs; replaces nn.DataParallel, train.py:205,285): a data-parallel
    Adam step with real per-rank gradients (one frame per rank, local BatchNorm / loss normalisation, oracle autograd standing in for
    the HIP backward, which needs a GPU), identical replicas afterwards; the two-bucket all-reduce on the AdaIN and 64-channel
    arenas; shards of the global batch of 2048."""
    script = tmp_path / "dp8_worker.py"
    script.write_text(_DP8_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29651", WORLD_SIZE="8", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(8)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "ok" in o, "rank %d failed:\n%s" % (r, o[-3000:])


def test_bench_refuses_eight_gpus_in_a_world_of_four():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--mode", "train"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode != 0 and b"--gpus 8 but WORLD_SIZE=4" in p.stdout


def test_ellipse_transform_matches_reference():
    from egne_amd import ellipse
    g = gold("ellipse_transform")
    Hm = np.array([[160.0, 0, 160.0], [0, 120.0, 120.0], [0, 0, 1]])
    for p, ref in zip(g["params"], g["out"]):
        np.testing.assert_allclose(ellipse.transform(p, Hm), ref, rtol=1e-12, atol=1e-12)


def test_entry_args_and_checkpoint_format():
    from egne_amd import _entry
    from egne_amd.args import parse_args, parse_precision
    a = parse_args(["--synthetic", "4", "--setting", "configs/baseline_edge.yaml"])
    assert a.lr == 5e-4 and a.batchsize == 12 and a.epochs == 40 and a.disentangle == 1 and a.prec == torch.float32
    assert parse_precision(64) == torch.float32 and parse_precision(16) == torch.float16
    with pytest.raises(SystemExit):
        parse_args([])
    st = _entry.load_setting("configs/baseline_edge.yaml")
    assert st["add_edge"] == 1 and st["feature_channels"] == 153
    m = esf_module("baseline_edge", disentangle=True)
    ck = _entry.checkpoint_dict(m, 3)
    assert ck["epoch"] == 3 and not any("dsIdentify" in k for k in ck["state_dict"]) and "enc.head.conv1.weight" in ck["state_dict"]
    ds = _entry.SyntheticEyes(2)
    s = ds[1]
    assert len(s) == 9 and tuple(s[0].shape) == (1, 240, 320) and s[7].dtype == torch.bool and tuple(s[8].shape) == (3,)


def test_bench_refuses_a_world_that_does_not_match_gpus():
    """bench.py --gpus N must either run N ranks or fail loudly (round 1 silently ran one rank)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode != 0 and b"--gpus 2 but WORLD_SIZE=1" in p.stdout


def test_evaluate_front_and_back_end(tmp_path):
    """evaluate.py's host stages that OpenCV did in the reference (resize, overlay, video I/O): properties only -- OpenCV is
    not installed, so nothing here is pinned against it."""
    from egne_amd import evaluate as E
    rng = np.random.RandomState(0)
    flat = np.full((100, 200), 77, np.uint8)
    assert np.unique(E.resize_lanczos4(flat, (320, 160))).tolist() == [77]          # weights sum to one
    ramp = np.tile(np.linspace(0, 255, 640)[None, :], (240, 1)).astype(np.uint8)
    half = E.resize_lanczos4(ramp, (320, 120))
    assert half.shape == (120, 320) and half.dtype == np.uint8
    assert np.abs(half[60].astype(float) - np.linspace(0, 255, 320)).max() < 2.0    # a linear ramp stays linear
    img = rng.randint(0, 256, (240, 320)).astype(np.uint8)
    assert np.array_equal(E.resize_lanczos4(img, (320, 240)), img)                  # same size: untouched
    assert np.array_equal(E.resize_nearest(img, (640, 480))[::2, ::2], img)
    # 400 wide -> scale 0.8 -> 192 rows -> 24 rows of padding above and below; z-scored
    t, ss = E.preprocess_frame(rng.randint(0, 256, (240, 400)).astype(np.uint8), (240, 320))
    assert tuple(t.shape) == (1, 240, 320) and ss == (0.8, 48) and abs(float(t.mean())) < 1e-6 and abs(float(t.std()) - 1) < 1e-3
    t2, ss2 = E.preprocess_frame(rng.randint(0, 256, (300, 320)).astype(np.uint8), (240, 320))
    assert tuple(t2.shape) == (1, 240, 320) and ss2 == (1, -60)
    # back to the source geometry: rows un-padded, ellipse shifted by pad / 2 and scaled by 1 / 0.8
    seg = np.zeros((240, 320), np.int64)
    seg[24:216] = 1
    s2, p, q, e2 = E.rescale_to_original(seg, np.array([160., 120., 20., 10., 0.3]), np.array([160., 120., 60., 50., 0.1]), ss, (240, 400),
                                         edge_map=np.ones((240, 320)))
    assert s2.shape == (240, 400) and e2.shape == (240, 400) and (s2 == 1).all()
    np.testing.assert_allclose(p, [200.0, 120.0, 25.0, 12.5, 0.3])
    # overlay colours and ellipse outlines
    segm = np.zeros((240, 320), np.int64)
    segm[100:140, 100:200] = 1
    segm[110:130, 140:160] = 2
    ov = E.plot_segmap_ellpreds(img, segm, np.array([150., 120., 12., 9., 0.]), np.array([150., 120., 55., 35., 0.2]))
    assert ov.shape == (240, 320, 3) and tuple(ov[105, 105]) == (120, 183, 53) and tuple(ov[125, 145]) == (36, 231, 253)
    assert tuple(ov[120, 162]) == (0, 0, 255) and (ov == np.array([255, 0, 0], np.uint8)).all(2).sum() > 100
    # Motion-JPEG AVI round trip through the package's own reader
    w = E.MJPEGWriter(str(tmp_path / "t.avi"), 30, (320, 240))
    smooth = np.tile(np.linspace(20, 230, 320)[None, :], (240, 1)).astype(np.uint8)
    for k in range(3):
        w.write(np.stack([np.roll(smooth, 10 * k, 1)] * 3, axis=2))
    w.release()
    back = list(E.mjpeg_frames(str(tmp_path / "t.avi")))
    assert len(back) == 3 and back[0].shape == (240, 320)
    assert np.abs(back[1].astype(int) - np.roll(smooth, 10, 1)).mean() < 3


def test_first_writer_bookkeeping_of_gradient_slices():
    """engine.Plan.first_touch (host logic of the backward plan, train.py:285-286): while the backward plan is built every access to
    a gradient twin goes through gp / gbuf, which record channel ranges in execution order; a data gradient may STORE (instead of
    accumulate) only into channels nobody asked for before it.  Also the device-side pre-scale words: one per call, published
    words found again by (buffer, slice, samples)."""
    from egne_amd.engine import Piece, Plan
    pl = Plan(torch.device("cpu"), train=True)
    a, b = pl.buf(2, 4, 4, 96), pl.buf(2, 4, 4, 32)
    x, x1, x22 = Piece(a, 0, 32), Piece(a, 32, 32), Piece(a, 64, 30)
    assert not pl.first_touch(a, 0, 32)            # outside build_backward nothing is a first touch
    pl._touching, pl._touched = True, {}
    assert pl.first_touch(a, 32, 32) and pl.first_touch(a, 0, 96)
    g1 = pl.gp(x1)
    assert g1.buf is pl.gbuf(a, False) and (g1.off, g1.Cp) == (32, 32)
    assert not pl.first_touch(a, 32, 32) and not pl.first_touch(a, 0, 64) and not pl.first_touch(a, 56, 16)
    assert pl.first_touch(a, 0, 32) and pl.first_touch(a, 64, 32)      # neighbours on both sides are still untouched
    pl.gp(x22)
    assert not pl.first_touch(a, 64, 8) and pl.first_touch(a, 0, 32)
    assert pl.first_touch(b, 0, 32)
    pl.gbuf(b)                                     # a whole-buffer access (loss / softmax backward) touches every channel
    assert not pl.first_touch(b, 8, 8)
    pl._touching = False
    # device-side pre-scale words
    w0 = pl._publish_absmax(x, 2)
    w1 = pl._new_slot()
    assert w1 == w0 + 4 and pl.ndyn == 2 and pl.dynbuf.dtype == torch.int32
    assert pl._absmax_of[(id(a), 0, 32, 0, 2)][0] == w0


def test_augment_draws_consume_the_random_stream_like_the_reference():
    """egne_amd.data_augment.draw (host side of the batched device augmentation): for every NumPy branch it leaves np.random in the
    state the reference's augment() leaves it in (restated in oracle/data_augment.py, itself pinned by tests/golden/augment.npz), so a
    seeded loader selects the same augmentations; the gamma tables are the reference's (data_augment.py:47) after its uint8 cast; the
    OpenCV branches are refused or skipped on request."""
    import numpy as np
    from egne_amd import data_augment as DA, synth
    from oracle import data_augment as oaug
    g = np.load(os.path.join(GOLD, "augment.npz"))
    for gm in (0.6, 0.8, 1.2, 1.4):
        assert np.array_equal(DA.gamma_table(gm), g["gamma_table_%d" % int(gm * 10)].astype(np.uint8))
    base, mask, pc, el = synth.augment_case(8)
    for choice in (0, 2, 3, 4, 7):
        for seed in (1, 2):
            np.random.seed(seed)
            oaug.augment(base, mask, pc, el, choice)
            want = np.random.get_state()[1].copy(), np.random.get_state()[2]
            np.random.seed(seed)
            ch, param, lut, noise = DA.draw(1, base.shape, [choice], host_noise=True)
            got = np.random.get_state()[1], np.random.get_state()[2]
            assert np.array_equal(want[0], got[0]) and want[1] == got[1], "branch %d leaves another random state" % choice
            assert ch[0] == choice and (noise is not None) == (choice == 4)
    # branch drawn at random: same first draw, same follow-up draws
    for seed in range(12):
        first = int(np.random.RandomState(seed).randint(0, 8))
        np.random.seed(seed)
        if first in DA.CV2_CHOICES:
            with pytest.raises(NotImplementedError):
                DA.draw(1, base.shape)
            np.random.seed(seed)
            ch, _, _, _ = DA.draw(1, base.shape, on_cv2="skip")
            assert ch[0] == 7
            continue
        oaug.augment(base, mask, pc, el)
        want = np.random.get_state()
        np.random.seed(seed)
        ch, _, _, _ = DA.draw(1, base.shape, host_noise=True)
        got = np.random.get_state()
        assert ch[0] == min(first, 7) and np.array_equal(want[1], got[1]) and want[2] == got[2]

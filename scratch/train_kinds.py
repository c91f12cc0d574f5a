import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, collections
from common import esf_module, batch_args
from egne_amd import synth, engine
print("TRAIN_SPLIT", engine.TRAIN_SPLIT)
m = esf_module("baseline_edge").cuda().train()
b = synth.make_batch(2, seed=1)
args = [a.cuda() if torch.is_tensor(a) else a for a in batch_args(b, b["img"])]
out = m(*args)
pl = m._last_plan
c = collections.Counter(k for k, _ in pl.meta)
print(c)
print([(n, k) for (_, _, n), (k, _) in zip(pl.calls, pl.meta) if "enc.b0" in n][:12])

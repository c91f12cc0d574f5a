"""Shared plumbing of the entry scripts: package import when run as a file, settings, synthetic data,
checkpoint helpers (formats of train.py:445-447,486-488 and pytorchtools.py:60-67)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import egne_amd  # noqa: E402,F401

import torch  # noqa: E402
import yaml  # noqa: E402


def load_setting(path):
    """yaml.safe_load (the reference's bare yaml.load(stream) raises on PyYAML >= 6)."""
    if not os.path.exists(path):
        alt = os.path.join(os.path.dirname(os.path.abspath(__file__)), path)
        path = alt if os.path.exists(alt) else path
    with open(path) as f:
        return yaml.safe_load(f)


class SyntheticEyes(torch.utils.data.Dataset):
    """Stands in for the pickled CurriculumLib.DataLoader_riteyes objects (absent):
    returns the same 9-tuple per sample (CurriculumLib.py:166)."""

    def __init__(self, n, seed=1234, dataset_id=0):
        from egne_amd import synth
        self.b = synth.make_batch(n, seed=seed)
        self.n, self.ds = n, dataset_id
        self.imList = torch.zeros(n, 3, dtype=torch.long)
        self.imList[:, 0] = torch.arange(n)
        self.imList[:, 2] = self.b["ID"]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        b = self.b
        iris_c = torch.stack([(b["elNorm"][i, 0, 0] + 1) * 160, (b["elNorm"][i, 0, 1] + 1) * 120])
        return (b["img"][i], b["label"][i], b["spatWts"][i], b["distMap"][i], b["pupil_center"][i], iris_c,
                b["elNorm"][i], b["cond"][i].bool(), self.imList[i])


def seeded_networks(setting, model_name="ritnet_v2", disentangle=False, nsets=4):
    """Random-init (seeded) BDCN + ESF-Net for --synthetic runs."""
    from egne_amd import synth
    from egne_amd.bdcn_new import BDCN
    from egne_amd.modelSummary import get_model
    bd = BDCN()
    bd.load_state_dict(synth.seeded_state_dict(bd.state_dict(), kind="bdcn"))
    net = get_model(model_name, dict(setting))
    if disentangle:
        net.disentangle = True
        net.setDatasetInfo(nsets)
    net.load_state_dict(synth.seeded_state_dict(net.state_dict(), kind="esf"))
    return bd, net


def checkpoint_dict(model, epoch):
    """train.py:445-447: state_dict without the dataset-identity head + epoch."""
    sd = {k: v for k, v in model.state_dict().items() if "dsIdentify_lin" not in k}
    return {"state_dict": sd, "epoch": epoch}

"""-m gpu: the entry points (test.py / train.py / evaluate.py mirrors) end to end on synthetic frames."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_test_py_synthetic(capsys):
    from egne_amd import test as T
    miou, pd, idist, loss = T.main(["--synthetic", "4", "--batchsize", "2", "--setting", "configs/baseline_edge.yaml"])
    assert np.isfinite(miou) and 0 <= miou <= 1 and np.isfinite(loss)
    assert "mIoU" in capsys.readouterr().out


def test_train_py_synthetic_loss_decreases(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from egne_amd import train as TR
    m = TR.main(["--synthetic", "4", "--batchsize", "2", "--epochs", "2", "--setting", "configs/baseline_edge.yaml",
                 "--disentangle", "1", "--expname", "t"])
    ck = torch.load(os.path.join("logs", "ritnet_v2", "t", "weights", "ritnet_v2_1.pkl"), map_location="cpu")
    assert set(ck) == {"state_dict", "epoch"} and ck["epoch"] == 1
    assert not any("dsIdentify" in k for k in ck["state_dict"])


def test_evaluate_py_synthetic():
    from egne_amd import evaluate as E
    pup, iri = E.main(["--synthetic", "2"])
    assert pup.shape == (2, 5) and iri.shape == (2, 5) and np.isfinite(pup).all() and np.isfinite(iri).all()

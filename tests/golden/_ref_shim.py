"""Import the reference (read-only at /root/reference) inside THIS container only.

Used by make_golden.py to produce the committed fixtures.  It never runs on the GPU box
(/root/reference does not exist there) and nothing in the product imports it.
Shim per SURVEY.md section 8c: stub cv2/skimage/torchvision (touched only at import time),
restore the NumPy aliases the reference still uses, make Tensor.cuda() the identity for the
CPU run, and do not write bytecode into the read-only tree.
"""
import os
import sys
import types

REF = "/root/reference"


def install():
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    import numpy as np
    import torch

    for name in ("cv2", "skimage", "skimage.draw", "torchvision", "torchvision.utils"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].draw = sys.modules["skimage.draw"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.utils"].make_grid = lambda *a, **k: None
    for alias, typ in (("int", int), ("bool", bool), ("float", float)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")


def reference_modules():
    install()
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        import bdcn_new
        import utils as ref_utils
        import loss as ref_loss
        import helperfunctions as ref_hf
        from models import RITnet_v2, RITnet_concat
    return dict(bdcn_new=bdcn_new, utils=ref_utils, loss=ref_loss, hf=ref_hf,
                RITnet_v2=RITnet_v2, RITnet_concat=RITnet_concat)

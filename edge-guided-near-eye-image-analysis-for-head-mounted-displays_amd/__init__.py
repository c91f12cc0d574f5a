"""MI355X-native hot path of edge-guided near-eye segmentation (BDCN edge extractor -> ESF-Net).

Layout: ``csrc/`` hand-written HIP kernels + the C-ABI (``libegne_hip.so``), ``_lib.py`` the
ctypes binding, and host-side mirrors of the reference's Python surface (``bdcn_new``,
``utils``, ``loss``, ``models.RITnet_v2`` / ``models.RITnet_concat``, ``modelSummary``,
``args``, ``test`` / ``train`` / ``evaluate``).  There is no CPU fallback: every compute
entry point raises if the HIP library is missing.
"""
import os as _os

# The inference / training loops run the frozen edge network, ESF-Net and the ellipse fit on three HIP streams next to torch's
# own (pipeline.py, utils.fit_ellipses_from_pred callers).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4), and two streams that share a queue run their kernels one after the other: the 3 ms fit launch then delays
# the next batch's network by 3 ms although it needs a hundredth of the chip (measured: B=64 step with fit 36.6 ms with 4 queues,
# 34.4 ms with 8).  Read by the runtime when it initialises, i.e. at the first HIP call of the process: set before that, and only
# if the user has not chosen a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"

"""MI355X-native hot path of edge-guided near-eye segmentation (BDCN edge extractor -> ESF-Net).

Layout: ``csrc/`` hand-written HIP kernels + the C-ABI (``libegne_hip.so``), ``_lib.py`` the
ctypes binding, and host-side mirrors of the reference's Python surface (``bdcn_new``,
``utils``, ``loss``, ``models.RITnet_v2`` / ``models.RITnet_concat``, ``modelSummary``,
``args``, ``test`` / ``train`` / ``evaluate``).  There is no CPU fallback: every compute
entry point raises if the HIP library is missing.
"""
__version__ = "0.1.0"

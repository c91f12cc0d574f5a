"""Skip-concat variant of ESF-Net -- drop-in for the reference's ``models/RITnet_concat.py``.

Identical to ``RITnet_v2.DenseNet2D`` except that the edge encoder's skips are also fed to every
up block (``up_block.forward(prev, e_prev, x)``, RITnet_concat.py:79-88,240-247) and the decoder
widths are ip [306,115,76,38] / op [115,76,38,32] (:164-169); no AdaIN, no input_concat/only_edge.
"""
from .RITnet_v2 import DenseNet2D as _DenseNet2D_v2
from .RITnet_v2 import getSizes  # noqa: F401


class DenseNet2D(_DenseNet2D_v2):
    variant = "concat"

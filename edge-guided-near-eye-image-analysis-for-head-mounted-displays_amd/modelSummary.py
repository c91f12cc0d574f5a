"""Model registry (reference: modelSummary.py:18-26, which instantiates classes that do not exist and
calls DenseNet2D() without its required ``setting``).  Here the dictionary holds factories.  'ritnet_v1' is the comparator of
models/RITnet_v1.py and 'deepvog' the one of models/deepvog_pytorch.py (evaluation and training; fp32 storage only, bf16 storage refused); neither takes a ``setting``: the edge options
do not apply to them."""
from .models.RITnet_concat import DenseNet2D as DN_concat
from .models.RITnet_v1 import DenseNet2D as DN_v1
from .models.RITnet_v2 import DenseNet2D as DN_v2
from .models.deepvog_pytorch import DeepVOG_pytorch

model_dict = {'ritnet_v2': DN_v2, 'ritnet_concat': DN_concat, 'ritnet_v1': lambda setting=None, **kw: DN_v1(**kw),
              'deepvog': lambda setting=None, **kw: DeepVOG_pytorch(**kw)}


def get_model(name, setting, **kw):
    if name not in model_dict:
        raise KeyError('unknown model %r (the HIP path builds: %s)' % (name, ', '.join(sorted(model_dict))))
    return model_dict[name](setting, **kw)

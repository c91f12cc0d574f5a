"""Model registry (reference: modelSummary.py:18-26, which instantiates classes that do not exist and
calls DenseNet2D() without its required ``setting``).  Here the dictionary holds factories."""
from .models.RITnet_concat import DenseNet2D as DN_concat
from .models.RITnet_v2 import DenseNet2D as DN_v2

model_dict = {'ritnet_v2': DN_v2, 'ritnet_concat': DN_concat}


def get_model(name, setting, **kw):
    if name not in model_dict:
        raise KeyError('unknown model %r (the HIP path builds: %s)' % (name, ', '.join(sorted(model_dict))))
    return model_dict[name](setting, **kw)

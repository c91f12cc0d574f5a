"""Import alias for the package directory.

The package lives in ``edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd/``
(not a valid Python identifier).  ``import egne_amd`` registers that directory as the
package ``egne_amd`` so tests, bench.py and the entry scripts can import it normally.
"""
import importlib.util
import os
import sys

PKG_DIRNAME = "edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd"
_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), PKG_DIRNAME)
_spec = importlib.util.spec_from_file_location(
    "egne_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["egne_amd"] = _mod
_spec.loader.exec_module(_mod)

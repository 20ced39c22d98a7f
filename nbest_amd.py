"""Import alias: the product package lives in ``n-best-asr-transformer_amd/`` (a directory name
Python cannot import directly); ``import nbest_amd`` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "n-best-asr-transformer_amd")
_spec = importlib.util.spec_from_file_location(
    "nbest_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nbest_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import alias: the package lives in ./light-vllm_amd/ (a hyphen is not importable).

`import light_vllm_amd` executes this file, which loads ./light-vllm_amd/__init__.py
as the package `light_vllm_amd` (sub-modules resolve inside that directory) and
replaces itself in sys.modules with it.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "light-vllm_amd")
_spec = importlib.util.spec_from_file_location(
    "light_vllm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["light_vllm_amd"] = _mod
_spec.loader.exec_module(_mod)

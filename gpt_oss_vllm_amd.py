"""Importable alias of the `gpt-oss-vllm_amd/` package (a hyphen cannot appear in an import statement).

`import gpt_oss_vllm_amd` returns the real package object; its submodules are registered under both
names so that each is loaded once.
"""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("gpt-oss-vllm_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("gpt-oss-vllm_amd."):
        sys.modules[__name__ + _name[len("gpt-oss-vllm_amd"):]] = _mod
sys.modules[__name__] = _pkg

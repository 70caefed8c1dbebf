"""MI355X-native NF4 MoE decode path (gfx950): Python host side of libmoe355.so.

The directory name carries a hyphen, so import it as `gpt_oss_vllm_amd` (the alias module at the
repository root) or with importlib.import_module("gpt-oss-vllm_amd").
"""
from . import lib  # noqa: F401

__all__ = ["lib"]

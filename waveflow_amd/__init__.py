"""waveflow_amd -- MI355X (gfx950) implementation of waveflow's flow-density hot path.

The arithmetic is hand-written HIP behind a C ABI (include/waveflow_hip.h, libwaveflow_hip.so);
this package re-creates the reference's model_factory / wavefunctions / flows call surface on it.
There is no CPU fallback: without the built library and a gfx950 device the hot path raises.
"""
from . import flows, model_factory, wavefunctions  # noqa: F401
from .core import DeviceModel, DeviceParams, build_tables, flatten_params, tree_leaves  # noqa: F401

__all__ = ["flows", "model_factory", "wavefunctions", "DeviceModel", "DeviceParams", "build_tables", "flatten_params", "tree_leaves"]

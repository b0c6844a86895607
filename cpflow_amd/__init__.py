"""cpflow_amd — MI355X-native engine for cpflow's CP-template synthesis hot path.

Public names follow the reference package (cpflow/__init__.py:5-10):
    Results, Decomposition, Synthesize, StaticOptions, AdaptiveOptions, Ansatz
plus the layer helpers of cpflow/topology.py.  All numerics run in hand-written HIP
kernels reached through the C ABI of include/cpflow_amd.h; there is no CPU fallback.
"""
__version__ = '0.1.0'

from .ansatz import Ansatz  # noqa: F401
from .options import AdaptiveOptions, BasicOptions, RegularizationOptions, StaticOptions  # noqa: F401
from .synthesis import (Decomposition, HSTLoss, RelativePhaseLoss, Results, StatePrepLoss,  # noqa: F401
                        Synthesize)
from .topology import chain_layer, connected_layer, fill_layers, num_qubits_from_layer, ring_layer  # noqa: F401

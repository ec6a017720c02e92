"""coivo_amd -- MI355X-native implementation of ColVO's DCDP+LCC training hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
every hot op is hand-written HIP for gfx950 behind the C-ABI in include/colvo.h.
"""
__version__ = "0.1.0"

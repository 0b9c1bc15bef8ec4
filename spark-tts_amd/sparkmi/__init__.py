"""sparkmi -- MI355X-native Spark-TTS inference hot path.

Host side (Python on PyTorch-ROCm) of ``libsparkmi.so``: hand-written HIP kernels for gfx950
behind a flat C ABI (``include/sparkmi.h``).  Importing this package does not touch the GPU;
constructing any engine without the built library raises (there is no CPU fallback).
"""
__version__ = "0.1.0"

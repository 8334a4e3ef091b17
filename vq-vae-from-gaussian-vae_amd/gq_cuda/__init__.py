"""Drop-in for the reference's ``gq_cuda`` extension package
(gq_cuda_extension/gq_cuda/__init__.py:3): importing it registers the torch op
``extension_cpp::gq`` (schema of csrc/gq_cuda.cpp:29-31) with a HIP
implementation from libgqhip.so, and exposes ``gq_cuda.ops.gq_cuda``."""
from . import _C, ops  # noqa: F401

__all__ = ["_C", "ops"]

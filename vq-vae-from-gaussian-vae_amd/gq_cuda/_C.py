"""Stand-in for the reference's ``gq_cuda._C`` (gq_cuda_extension/gq_cuda/csrc/gq_cuda.cpp:8-25): there an EMPTY extension module
whose only purpose is that importing it loads the shared object and runs its ``TORCH_LIBRARY`` static initialisers.  Here the op is
registered by ``gq_cuda.ops`` on top of libgqhip.so, so this module is empty too -- it exists so that ``import gq_cuda._C`` and
``from gq_cuda import _C`` (gq_cuda/__init__.py:3) keep working."""

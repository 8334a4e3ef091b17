"""``gq_cuda.ops.gq_cuda(a, b, c, out, d, e, f, g) -> None`` -- same signature and
argument meaning as gq_cuda_extension/gq_cuda/ops.py:7-8: a=mu [b,dim], b=std
[b,dim], c=codebook [n,dim], out [b,n] mutated in place, d=dim, e=b, f=n, g=beta.
Dispatch key "CUDA" (PyTorch-ROCm's key for HIP devices), launched on the current
stream, asynchronous, like gq_cuda.cu:114-116.  Error behaviour mirrors the
TORCH_CHECKs at gq_cuda.cu:91-101 (RuntimeError on shape/dtype/device)."""
import torch
from torch import Tensor

from pit_hip import _lib

__all__ = ["gq_cuda"]

_LIB = torch.library.Library("extension_cpp", "DEF")
_LIB.define("gq(Tensor a, Tensor b, Tensor c, Tensor(a!) out, int d, int e, int f, float g) -> ()")


def _gq_hip(a: Tensor, b: Tensor, c: Tensor, out: Tensor, d: int, e: int, f: int, g: float) -> None:
    if a.shape != b.shape:
        raise RuntimeError("gq: mu and std sizes differ")
    for t, name in ((a, "mu"), (b, "std"), (c, "noise"), (out, "result")):
        if t.dtype != torch.float32:
            raise RuntimeError(f"gq: {name} must be float32")
        if not t.is_cuda:
            raise RuntimeError(f"gq: {name} must be on a HIP device")
    if not out.is_contiguous():
        raise RuntimeError("gq: result must be contiguous")
    if a.shape != (e, d) or c.shape != (f, d) or out.shape != (e, f):
        raise RuntimeError("gq: d/e/f do not match tensor shapes")
    _lib.gq_scores(a, b, c, out, g)


_LIB.impl("gq", _gq_hip, "CUDA")


def gq_cuda(a: Tensor, b: Tensor, c: Tensor, out: Tensor, d: int, e: int, f: int, g: float) -> None:
    torch.ops.extension_cpp.gq.default(a, b, c, out, d, e, f, g)

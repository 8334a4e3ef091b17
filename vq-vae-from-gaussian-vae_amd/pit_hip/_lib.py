"""ctypes binding of libgqhip.so (the C ABI declared in include/gqhip.h).

torch is used only for device memory and the current HIP stream: every call
hands raw ``data_ptr()`` values and ``torch.cuda.current_stream().cuda_stream``
to the library.  There is NO CPU fallback here: if the library is missing or a
tensor is not on a HIP device the call raises.
"""
from __future__ import annotations

import ctypes
import threading
import math
import os
import subprocess
from typing import Optional, Tuple

import torch

_CSRC = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc"))
LIB_PATH = os.environ.get("GQHIP_LIB", os.path.join(_CSRC, "libgqhip.so"))  # GQHIP_LIB: diagnostic builds

ABI_VERSION = 8
GNSTAT_WORDS = 8     # int64 words per (image, group) statistics record (gqhip.h: gqhip_gnstat_t)
GQHIP_LAYOUT = {"bchw": 0, "blc": 1}
GQHIP_GROUP_STRIDED = 0
GQHIP_GROUP_CONTIGUOUS = 1

_lib: Optional[ctypes.CDLL] = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_SIGNATURES = {
    # name: (restype, argtypes)
    "gqhip_abi_version": (ctypes.c_int, []),
    "gqhip_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "gqhip_last_hip_error": (ctypes.c_int, []),
    "gqhip_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gqhip_cb_cache_bytes": (_i64, [_i64, _i64]),
    "gqhip_grid_search_applies": (ctypes.c_int, [_i64, _i64]),
    "gqhip_cb_cache_degenerate": (ctypes.c_int, [_vp, _i64, _i64]),
    "gqhip_debug_grid": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_i64)]),
    "gqhip_set_filter": (ctypes.c_int, [ctypes.c_int]),
    "gqhip_get_filter": (ctypes.c_int, []),
    "gqhip_debug_plan": (ctypes.c_int, [_i64, _i64, _i64, ctypes.POINTER(_i64)]),
    "gq_scores_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_double, _vp]),
    "gq_argmax_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_double, _vp, _i64, _vp, _i64, _vp]),
    "gq_quantize_z_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, _vp, _i64, _vp, _i64, _vp]),
    "gq_quantize_z_gauss_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                               ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                               ctypes.c_double, ctypes.c_int, _vp, _i64, _vp, _i64, _vp]),
    "vq_quantize_z_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_int, _vp, _i64, _vp, _i64, _vp]),
    "gq_dequant_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int,
                                      ctypes.c_int, _vp]),
    "vq_argmin_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "lfq_pack_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "lfq_unpack_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "fsq_quantize_f32": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), _i64, _vp, _vp, _i64, _vp]),
    "fsq_dequant_f32": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), _i64, _vp, _i64, _vp]),
    "gn_silu_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_double, ctypes.c_int,
                                   ctypes.c_int, _vp, _vp]),
    "add_bias_stats_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "gn_apply_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_double, ctypes.c_int, _vp, _vp]),
    "wino_in_nhwc_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "wino4_in_nhwc_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "wino4_out_nhwc_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, _vp]),
    "wino_out_res_nhwc_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int,
                                              ctypes.c_float, _vp]),
    "wino_in_nhwc_f16x3": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_int, ctypes.c_float, _vp]),
    "wino_in_nhwc_f16x2": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_int, ctypes.c_float, _vp]),
    "wino_gemm_f16x2": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "upconv2x_f16x3": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64,
                                      _vp]),
    "attn_split_qkv_f16x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_float, ctypes.c_float, _vp]),
    "attn_softmax_split_f16x3": (ctypes.c_int, [_vp, _vp, _i64, _i64, ctypes.c_float, _vp]),
    "wino_in_gn_nhwc_f16x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_double,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp]),
    "wino_in_gn_nhwc_f16x2": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_double,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp]),
    "conv3x3_gn_f16x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_double, ctypes.c_int, ctypes.c_float, _vp,
                                         _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, ctypes.c_float, _vp]),
    "conv1x1_f16x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                      _vp]),
    "conv3x3s2_f16x3": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                        _i64, _vp]),
    "conv3x3_gn_small_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp, _i64, _i64,
                                             _i64, _i64, _i64, _vp]),
    "conv3x3_cin_small_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "conv3x3_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp, _i64, _i64, _i64,
                                   _i64, _i64, _vp]),
    "gqhip_checksum_tensors": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "gn_stats_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "wino_in_gn_nhwc_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_double,
                                            ctypes.c_int, _vp]),
    "wino4_in_gn_nhwc_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_double,
                                             ctypes.c_int, _vp]),
    "wino_out_nhwc_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, _vp]),
    "f16_scales_from_gn_stats": (ctypes.c_int, [_vp, _i64, ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "upsample2x_nhwc_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "add_bias_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, _vp]),
    "gqhip_stats_prezeroed": (ctypes.c_int, [ctypes.c_int]),
    "gq_index_histogram": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "gq_indices_to_u16": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "gq_indices_from_u16": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "gq_step_record_workspace_bytes": (_i64, [_i64, _i64]),
    "gq_step_record_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "gqhip_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "gqhip_profile_reserve": (ctypes.c_int, [ctypes.c_int]),
    "gqhip_profile_collect": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "gqhip_debug_enable": (ctypes.c_int, [ctypes.c_int]),
    "gqhip_debug_counters": (ctypes.c_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class GqHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libgqhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args + ["libgqhip.so"])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the library, failing loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GqHipError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C vq-vae-from-gaussian-vae_amd/csrc`). There is no CPU fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.gqhip_abi_version() != ABI_VERSION:
            raise GqHipError("libgqhip.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        L = lib()
        raise GqHipError(f"{what}: {L.gqhip_status_string(rc).decode()} (hipError {L.gqhip_last_hip_error()})")


def _dev(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GqHipError(f"{name} must be a tensor on a HIP device (got {getattr(t, 'device', type(t))}); "
                         "the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise GqHipError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Workspace:
    """Caller-owned scratch, grown on demand and reused across calls, plus the persistent CODEBOOK CACHE of the shapes that keep
    one (gqhip.h: gqhip_cb_cache_bytes -- dims 4 / 8: the spatial index of the pruned search; the library validates it against a
    content hash of the codebook on every call and rebuilds it in-stream when it is stale, so nothing here has to track the
    codebook).  One workspace serves one stream at a time (its header carries the per-call counters).  While a HIP graph is being
    captured the buffers are never replaced: a captured graph has the pointers baked in, so growing them then -- or after the
    capture, by an eager call with a bigger shape -- would leave the graph writing into freed memory; size them with a warm-up
    call first (``reserve``), and a capture that would need to grow raises instead."""

    def __init__(self) -> None:
        self.buf: Optional[torch.Tensor] = None
        self.cache_buf: Optional[torch.Tensor] = None
        self._cache_key = None
        self._cache_calls = 0     # calls that were handed the current cache buffer
        self.no_search = False    # the dim-4 index of this codebook was found degenerate (gqhip_cb_cache_degenerate): dense path from then on
        self._pinned = False      # True once a graph capture has used this buffer: it must never be replaced

    def cache(self, n: int, dim: int, device) -> Tuple[Optional[int], int]:
        """(pointer, bytes) of the codebook cache for (n, dim), or (None, 0) for shapes that keep none."""
        need = lib().gqhip_cb_cache_bytes(n, dim)
        if need <= 0 or self.no_search:
            return None, 0
        key = (n, dim, device)
        if self.cache_buf is None or self._cache_key != key or self.cache_buf.numel() < need:
            if torch.cuda.is_current_stream_capturing() or self._pinned:
                raise GqHipError("the codebook cache would have to be (re)allocated during / after a HIP graph capture that uses "
                                 "this Workspace: run a warm-up call of the same codebook shape before capturing")
            self.cache_buf = torch.zeros(need, dtype=torch.uint8, device=device)    # zeros: no stamp -> built by the first call
            self._cache_key = key
            self._cache_calls = 0
        # ONE look at what the index builder found, at the call after the one that built it (a 4-KiB synchronous copy, once per
        # cache buffer): a clustered / collapsed codebook puts most codes into one sub-leaf, and the search then hands every row to
        # the block-per-row finish kernel -- milliseconds per call where filter + re-rank takes ~100 us.  Such a book goes
        # back to the dense path for good (always exact; a codebook that changes every step should not pass a cache at all: vq.py).
        self._cache_calls += 1
        if self._cache_calls == 2 and lib().gqhip_grid_search_applies(n, dim) and not torch.cuda.is_current_stream_capturing():
            if lib().gqhip_cb_cache_degenerate(self.cache_buf.data_ptr(), n, dim) == 1:
                self.no_search = True
                return None, 0
        return self.cache_buf.data_ptr(), self.cache_buf.numel()

    def reserve(self, rows: int, n: int, dim: int, device) -> None:
        """Size the scratch and the codebook cache for a shape ahead of a graph capture."""
        self.get(rows, n, dim, device)
        self.cache(n, dim, device)

    def get(self, rows: int, n: int, dim: int, device) -> Tuple[int, int]:
        need = lib().gqhip_workspace_bytes(rows, n, dim)
        if need < 0:
            raise GqHipError(f"unsupported shape rows={rows} n={n} dim={dim}")
        capturing = torch.cuda.is_current_stream_capturing()
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            if capturing or self._pinned:
                raise GqHipError("workspace would have to grow during / after a HIP graph capture that uses it: "
                                 "run a warm-up call of the largest shape before capturing, or use a separate Workspace")
            self.buf = torch.empty(need, dtype=torch.uint8, device=device)
        if capturing:
            self._pinned = True
        return self.buf.data_ptr(), self.buf.numel()


def gq_scores(mu, sd, cb, out, beta: float = 1.0) -> None:
    """Compat op (reference gq_cuda.ops.gq_cuda): fills out[rows, n] in place."""
    mu, sd, cb = (_dev(t, torch.float32, n) for t, n in ((mu, "mu"), (sd, "std"), (cb, "codebook")))
    if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()):
        raise GqHipError("out must be a contiguous float32 HIP tensor")
    rows, dim = mu.shape
    n = cb.shape[0]
    if sd.shape != mu.shape or cb.shape[1] != dim or tuple(out.shape) != (rows, n):
        raise GqHipError("shape mismatch in gq_scores")
    with torch.cuda.device(mu.device):
        _check(lib().gq_scores_f32(mu.data_ptr(), sd.data_ptr(), cb.data_ptr(), out.data_ptr(), dim, rows, n,
                                   float(beta), _stream()), "gq_scores_f32")


def gq_argmax(mu, sd, cb, beta: float = 1.0, logsd=None, ws: Optional[Workspace] = None, want_zhat: bool = True):
    """Fused score+argmax+gather on (mu, sd[, log sd]) rows -> (idx int64 [rows], zhat [rows, dim])."""
    mu, sd, cb = (_dev(t, torch.float32, n) for t, n in ((mu, "mu"), (sd, "std"), (cb, "codebook")))
    logsd = None if logsd is None else _dev(logsd, torch.float32, "logsd")
    rows, dim = mu.shape
    n = cb.shape[0]
    if sd.shape != mu.shape or cb.shape[1] != dim:
        raise GqHipError("shape mismatch in gq_argmax")
    ws = ws or Workspace()
    idx = torch.empty(rows, dtype=torch.int64, device=mu.device)
    zhat = torch.empty(rows, dim, dtype=torch.float32, device=mu.device) if want_zhat else None
    with torch.cuda.device(mu.device):
        wptr, wbytes = ws.get(rows, n, dim, mu.device)
        cptr, cbytes = ws.cache(n, dim, mu.device)
        _check(lib().gq_argmax_f32(mu.data_ptr(), sd.data_ptr(), _ptr(logsd), cb.data_ptr(), idx.data_ptr(),
                                   _ptr(zhat), dim, rows, n, float(beta), wptr, wbytes, cptr, cbytes, _stream()), "gq_argmax_f32")
    return idx, zhat


def gq_quantize_z(z, cb, dim: int, layout: str, grouping: int, lv_range=(-30.0, 20.0), beta: float = 1.0,
                  ws: Optional[Workspace] = None, return_operands: bool = False, noise=None):
    """Module-level fused quantiser on the encoder output z (see gqhip.h).  ``noise`` (same shape as the returned
    zhat): the first launch also writes ``zhat_noquant = mu + noise * sd`` (gaussian.py:121), returned as a third value."""
    z, cb = _dev(z, torch.float32, "z"), _dev(cb, torch.float32, "codebook")
    n = cb.shape[0]
    if layout == "bchw":
        B, c2, L = z.shape[0], z.shape[1], int(z[0, 0].numel())
    else:
        B, L, c2 = z.shape
    c = c2 // 2
    K = c // dim
    rows = B * L * K
    ws = ws or Workspace()
    dev = z.device
    if layout == "bchw":
        idx = torch.empty((B, K) + tuple(z.shape[2:]), dtype=torch.int64, device=dev)
        zhat = torch.empty((B, c) + tuple(z.shape[2:]), dtype=torch.float32, device=dev)
    else:
        idx = torch.empty((B, L, K), dtype=torch.int64, device=dev)
        zhat = torch.empty((B, L, c), dtype=torch.float32, device=dev)
    noquant = None
    if noise is not None:
        noise = _dev(noise, torch.float32, "noise")
        if tuple(noise.shape) != tuple(zhat.shape):
            raise GqHipError(f"noise must have the shape of zhat {tuple(zhat.shape)}, got {tuple(noise.shape)}")
        noquant = torch.empty_like(zhat)
    mu_o = sd_o = None
    if return_operands:
        mu_o = torch.empty(rows, dim, dtype=torch.float32, device=dev)
        sd_o = torch.empty(rows, dim, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        wptr, wbytes = ws.get(max(rows, 1), n, dim, dev)
        cptr, cbytes = ws.cache(n, dim, dev)
        _check(lib().gq_quantize_z_f32(z.data_ptr(), _ptr(noise), cb.data_ptr(), idx.data_ptr(), zhat.data_ptr(),
                                       _ptr(noquant), _ptr(mu_o), _ptr(sd_o), B, L, c, dim, n, GQHIP_LAYOUT[layout],
                                       grouping, float(lv_range[0]), float(lv_range[1]), float(beta),
                                       wptr, wbytes, cptr, cbytes, _stream()), "gq_quantize_z_f32")
    out = (idx, zhat)
    if return_operands:
        out = out + (mu_o, sd_o)
    if noise is not None:
        out = out + (noquant,)
    return out


def gq_quantize_z_gauss(z, cb, dim: int, layout: str, grouping: int, noise, lam_state, log2n: int, tolerance: float,
                        lam_factor: float, lam_range, lam_max_decreases: bool, use_ste: bool = True, lv_range=(-30.0, 20.0),
                        beta: float = 1.0, ws: Optional[Workspace] = None, want_std: bool = True):
    """GaussianQuantRegularizer2's eval forward in one call (gqhip.h: gq_quantize_z_gauss_f32).  ``lam_state``: float64 [3] device
    tensor {lam, lam_min, lam_max}, advanced in place.  Returns (idx, zhat, zhat_noquant, std or None, scalars) with ``scalars`` a
    fresh 64-byte device buffer: float32 view [0:4] = kl_loss, bits-mean, bits-min, bits-max; float64 view of bytes 32..56 = the
    lambdas after the update."""
    z, cb, noise = _dev(z, torch.float32, "z"), _dev(cb, torch.float32, "codebook"), _dev(noise, torch.float32, "noise")
    if not (lam_state.is_cuda and lam_state.dtype == torch.float64 and lam_state.numel() == 3 and lam_state.is_contiguous()):
        raise GqHipError("lam_state must be a contiguous float64 [3] tensor on the HIP device")
    n = cb.shape[0]
    if layout == "bchw":
        B, c2, L = z.shape[0], z.shape[1], int(z[0, 0].numel())
        tail = tuple(z.shape[2:])
    else:
        B, L, c2 = z.shape
    c = c2 // 2
    K = c // dim
    rows = B * L * K
    ws = ws or Workspace()
    dev = z.device
    if layout == "bchw":
        idx = torch.empty((B, K) + tail, dtype=torch.int64, device=dev)
        zhat = torch.empty((B, c) + tail, dtype=torch.float32, device=dev)
    else:
        idx = torch.empty((B, L, K), dtype=torch.int64, device=dev)
        zhat = torch.empty((B, L, c), dtype=torch.float32, device=dev)
    if tuple(noise.shape) != tuple(zhat.shape):
        raise GqHipError(f"noise must have the shape of zhat {tuple(zhat.shape)}, got {tuple(noise.shape)}")
    noquant = torch.empty_like(zhat)
    pure = torch.empty_like(zhat) if use_ste else None
    std = torch.empty_like(zhat) if want_std else None
    scalars = torch.empty(64, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        wptr, wbytes = ws.get(max(rows, 1), n, dim, dev)
        cptr, cbytes = ws.cache(n, dim, dev)
        _check(lib().gq_quantize_z_gauss_f32(z.data_ptr(), noise.data_ptr(), cb.data_ptr(), idx.data_ptr(), zhat.data_ptr(),
                                             _ptr(pure), noquant.data_ptr(), _ptr(std), scalars.data_ptr(), lam_state.data_ptr(), B, L, c, dim,
                                             n, GQHIP_LAYOUT[layout], grouping, float(lv_range[0]), float(lv_range[1]),
                                             float(beta), 1 if use_ste else 0, float(log2n), float(tolerance), float(lam_factor),
                                             float(lam_range[0]), float(lam_range[1]), 1 if lam_max_decreases else 0,
                                             wptr, wbytes, cptr, cbytes, _stream()), "gq_quantize_z_gauss_f32")
    return idx, zhat, (pure if use_ste else zhat), noquant, std, scalars


def vq_quantize_z(z, emb, dim: int, layout: str, beta: float, legacy: bool, ws: Optional[Workspace] = None):
    """VQQuantizer's eval forward in one call (gqhip.h: vq_quantize_z_f32): z [B, c, ...] ("bchw") or [B, L, c] ("blc") ->
    (idx [B, K, ...] / [B, L, K], z_q in the layout of z, loss float32 [2] = {codebook_loss, mean((e - z)^2)})."""
    z, emb = _dev(z, torch.float32, "z"), _dev(emb, torch.float32, "embedding")
    n = emb.shape[0]
    if layout == "bchw":
        B, c, L = z.shape[0], z.shape[1], int(z[0, 0].numel())
    else:
        B, L, c = z.shape
    if c % dim or emb.shape[1] != dim:
        raise GqHipError("shape mismatch in vq_quantize_z")
    K = c // dim
    rows = B * L * K
    ws = ws or Workspace()
    dev = z.device
    idx = torch.empty(((B, K) + tuple(z.shape[2:])) if layout == "bchw" else (B, L, K), dtype=torch.int64, device=dev)
    zq = torch.empty_like(z)
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        wptr, wbytes = ws.get(max(rows, 1), n, dim, dev)
        cptr, cbytes = ws.cache(n, dim, dev)
        _check(lib().vq_quantize_z_f32(z.data_ptr(), emb.data_ptr(), idx.data_ptr(), zq.data_ptr(), loss.data_ptr(), B, L, c, dim,
                                       n, GQHIP_LAYOUT[layout], float(beta), 1 if legacy else 0, wptr, wbytes, cptr, cbytes,
                                       _stream()), "vq_quantize_z_f32")
    return idx, zq, loss


def gq_dequant(idx, cb, dim: int, layout: str, grouping: int):
    idx, cb = _dev(idx, torch.int64, "indices"), _dev(cb, torch.float32, "codebook")
    if layout == "bchw":
        B, K, L = idx.shape[0], idx.shape[1], int(idx[0, 0].numel())
        zhat = torch.empty((B, K * dim) + tuple(idx.shape[2:]), dtype=torch.float32, device=idx.device)
    else:
        B, L, K = idx.shape
        zhat = torch.empty((B, L, K * dim), dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        _check(lib().gq_dequant_f32(idx.data_ptr(), cb.data_ptr(), zhat.data_ptr(), B, L, K, dim, cb.shape[0],
                                    GQHIP_LAYOUT[layout], grouping, _stream()), "gq_dequant_f32")
    return zhat


def vq_argmin(z, emb, ws: Optional[Workspace] = None, use_cache: bool = True):
    """``use_cache=False``: no codebook cache in this call -- the dense filter + re-rank path whatever the shape.  For codebooks
    that change every step (training): a dim-4 index would be rebuilt by one block on every call."""
    z, emb = _dev(z, torch.float32, "z"), _dev(emb, torch.float32, "embedding")
    rows, dim = z.shape
    n = emb.shape[0]
    ws = ws or Workspace()
    idx = torch.empty(rows, dtype=torch.int64, device=z.device)
    zq = torch.empty(rows, dim, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        wptr, wbytes = ws.get(max(rows, 1), n, dim, z.device)
        cptr, cbytes = ws.cache(n, dim, z.device) if use_cache else (None, 0)
        _check(lib().vq_argmin_f32(z.data_ptr(), emb.data_ptr(), idx.data_ptr(), zq.data_ptr(), dim, rows, n,
                                   wptr, wbytes, cptr, cbytes, _stream()), "vq_argmin_f32")
    return idx, zq


def lfq_pack(x):
    x = _dev(x, torch.float32, "x")
    rows, nbits = x.shape
    idx = torch.empty(rows, dtype=torch.int64, device=x.device)
    q = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().lfq_pack_f32(x.data_ptr(), idx.data_ptr(), q.data_ptr(), rows, nbits, _stream()), "lfq_pack_f32")
    return idx, q


def lfq_unpack(idx, nbits: int):
    idx = _dev(idx, torch.int64, "indices")
    rows = idx.numel()
    q = torch.empty(rows, nbits, dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        _check(lib().lfq_unpack_f32(idx.data_ptr(), q.data_ptr(), rows, nbits, _stream()), "lfq_unpack_f32")
    return q


def _levels(levels):
    arr = (ctypes.c_int32 * len(levels))(*[int(v) for v in levels])
    return arr, len(levels)


def fsq_quantize(z, levels):
    """z [rows, nlev] -> (zhat [rows, nlev] fp32, packed indices [rows] int32)."""
    z = _dev(z, torch.float32, "z")
    rows, nlev = z.shape
    arr, n = _levels(levels)
    assert n == nlev
    zhat = torch.empty_like(z)
    idx = torch.empty(rows, dtype=torch.int32, device=z.device)
    with torch.cuda.device(z.device):
        _check(lib().fsq_quantize_f32(z.data_ptr(), arr, n, zhat.data_ptr(), idx.data_ptr(), rows, _stream()),
               "fsq_quantize_f32")
    return zhat, idx


def fsq_dequant(idx, levels):
    idx = _dev(idx, torch.int32, "indices")
    arr, n = _levels(levels)
    rows = idx.numel()
    zhat = torch.empty(rows, n, dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        _check(lib().fsq_dequant_f32(idx.data_ptr(), arr, n, zhat.data_ptr(), rows, _stream()), "fsq_dequant_f32")
    return zhat


def image_layout(x: torch.Tensor):
    """0 = NCHW contiguous, 1 = NHWC (torch channels_last) dense, None = neither (caller falls back)."""
    if x.dim() != 4:
        return None
    if x.is_contiguous():
        return 0
    if x.is_contiguous(memory_format=torch.channels_last):
        return 1
    return None


class StatsArena:
    """All GroupNorm statistics records of ONE forward of a module, carved out of one int64 buffer that a single fill zeroes at
    the forward's start (gqhip.h:gqhip_stats_prezeroed) -- instead of one 32-KB memset launch in front of every kernel that leaves
    statistics behind (~60 per encode + decode step).  Grows to the forward's demand; records that do not fit come from
    torch.empty and are zeroed by the entry point as before."""

    def __init__(self) -> None:
        self.buf, self.used, self.want = None, 0, 0

    def begin(self, device) -> None:
        if self.buf is None or self.buf.device != device or self.buf.numel() < self.want:
            self.buf = torch.zeros(max(self.want, 1), dtype=torch.int64, device=device)
        elif self.used:
            self.buf[: self.used].zero_()
        self.used = self.want = 0

    def take(self, nwords: int):
        self.want += nwords
        if self.buf is None or self.used + nwords > self.buf.numel():
            return None
        t = self.buf[self.used: self.used + nwords]
        self.used += nwords
        return t


class _ArenaState(threading.local):      # per thread, like the library's flag (gqhip_stats_prezeroed is thread-local)
    def __init__(self) -> None:
        self.state = {"active": None, "flag": 0}


_ARENA_TLS = _ArenaState()


class stats_arena:
    """``with stats_arena(arena, device): ...`` -- the statistics records allocated inside come from ``arena``."""

    def __init__(self, arena: StatsArena, device) -> None:
        self.arena, self.device = arena, device

    def __enter__(self):
        self.prev = _ARENA_TLS.state["active"]
        self.arena.begin(self.device)
        _ARENA_TLS.state["active"] = self.arena
        return self.arena

    def __exit__(self, *exc):
        _ARENA_TLS.state["active"] = self.prev
        _set_prezeroed(0)


def _set_prezeroed(on: int) -> None:
    if _ARENA_TLS.state["flag"] != on:
        lib().gqhip_stats_prezeroed(on)
        _ARENA_TLS.state["flag"] = on


def _stats_records(nwords: int, device):
    """An int64 tensor for ``nwords`` words of statistics records; tells the library whether it is already zero."""
    a = _ARENA_TLS.state["active"]
    t = a.take(nwords) if a is not None else None
    if t is not None and t.device == device:
        _set_prezeroed(1)
        return t
    _set_prezeroed(0)
    return torch.empty(nwords, dtype=torch.int64, device=device)


def gn_nhwc_ok(C: int, groups: int) -> bool:
    cpg = C // groups
    return C % groups == 0 and cpg % 4 == 0 and C % 4 == 0 and 256 % (C // 4) == 0 and groups <= 64


def gn_silu(x, gamma, beta, groups: int, eps: float, silu: bool = True, pre_bias=None):
    """Fused GroupNorm(+SiLU) on an NCHW or channels_last fp32 HIP tensor (see gqhip.h:gn_silu_f32)."""
    layout = image_layout(x)
    if not x.is_cuda or x.dtype != torch.float32 or layout is None:
        raise GqHipError("gn_silu needs a dense fp32 NCHW / channels_last HIP tensor")
    B, C = x.shape[0], x.shape[1]
    HW = x.shape[2] * x.shape[3]
    # statistics scratch: from the forward's arena (one per module, thread and stream, never under a graph capture:
    # modules/unet.py:_with_stats_arena) or a fresh stream-ordered tensor from the caching allocator
    ws = _stats_records(GNSTAT_WORDS * B * groups, x.device)
    y = torch.empty_like(x)  # preserves the memory format
    with torch.cuda.device(x.device):
        _check(lib().gn_silu_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(pre_bias), y.data_ptr(), B, C, HW,
                                 groups, float(eps), 1 if silu else 0, layout, ws.data_ptr(), _stream()), "gn_silu_f32")
    return y


def add_bias(a, b, bias=None):
    """y = a + b (+ bias[c]) on NCHW / channels_last fp32 HIP tensors of the same layout."""
    layout = image_layout(a)
    if layout is None or image_layout(b) != layout or not a.is_cuda or a.dtype != torch.float32:
        raise GqHipError("add_bias needs two dense fp32 HIP tensors of the same layout")
    B, C = a.shape[0], a.shape[1]
    HW = a.shape[2] * a.shape[3]
    y = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _check(lib().add_bias_f32(a.data_ptr(), b.data_ptr(), _ptr(bias), y.data_ptr(), B, C, HW, layout, _stream()),
               "add_bias_f32")
    return y


def add_bias_stats(a, b, bias, groups: int):
    """channels_last only: (a + b (+ bias[c]), GroupNorm statistics of that sum [2 * B * groups] fp64)."""
    if image_layout(a) != 1 or image_layout(b) != 1 or not a.is_cuda or a.dtype != torch.float32 or not gn_nhwc_ok(a.shape[1], groups):
        raise GqHipError("add_bias_stats needs two dense channels_last fp32 HIP tensors with a GroupNorm-compatible C")
    B, C = a.shape[0], a.shape[1]
    HW = a.shape[2] * a.shape[3]
    y = torch.empty_like(a)
    stats = _stats_records(GNSTAT_WORDS * B * groups, a.device)
    with torch.cuda.device(a.device):
        _check(lib().add_bias_stats_f32(a.data_ptr(), b.data_ptr(), _ptr(bias), y.data_ptr(), B, C, HW, groups,
                                        stats.data_ptr(), _stream()), "add_bias_stats_f32")
    return y, stats


def gn_apply(x, gamma, beta, groups: int, eps: float, silu: bool, stats):
    """channels_last only: GroupNorm(+SiLU) of x from statistics computed by add_bias_stats."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32:
        raise GqHipError("gn_apply needs a dense channels_last fp32 HIP tensor")
    B, C = x.shape[0], x.shape[1]
    HW = x.shape[2] * x.shape[3]
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().gn_apply_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), B, C, HW, groups,
                                  float(eps), 1 if silu else 0, stats.data_ptr(), _stream()), "gn_apply_f32")
    return y


def gn_stats_values(stats):
    """[2 * n_bg] fp64 (sum, sum of squares per (image, group)) from the fixed-point statistics records the kernels
    accumulate (gqhip.h: gqhip_gnstat_t; the readers' evaluation order).  For tests and diagnostics."""
    r = stats.view(-1, GNSTAT_WORDS).double()
    s = (r[:, 2] * 2.0 ** 24 + r[:, 1] * 2.0 ** -16) + r[:, 0] * 2.0 ** -56
    ss = (r[:, 5] * 2.0 ** 24 + r[:, 4] * 2.0 ** -16) + r[:, 3] * 2.0 ** -56
    bad = r[:, 6] != 0
    s = torch.where(bad, torch.full_like(s, float("nan")), s)
    ss = torch.where(bad, torch.full_like(ss, float("nan")), ss)
    return torch.stack([s, ss], 1).flatten()


def gn_stats(x, groups: int, pre_bias=None):
    """channels_last only: GroupNorm statistics of x (+ pre_bias[c]) as [2 * B * groups] fp64 (sum, sum of squares)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or not gn_nhwc_ok(x.shape[1], groups):
        raise GqHipError("gn_stats needs a dense channels_last fp32 HIP tensor with a GroupNorm-compatible C")
    B, C = x.shape[0], x.shape[1]
    stats = _stats_records(GNSTAT_WORDS * B * groups, x.device)
    with torch.cuda.device(x.device):
        _check(lib().gn_stats_f32(x.data_ptr(), _ptr(pre_bias), B, C, x.shape[2] * x.shape[3], groups, stats.data_ptr(),
                                  _stream()), "gn_stats_f32")
    return stats


def wino_conv3x3(x, U, gn=None, residual=None, bias=None, stats_groups: int = 0, f16=None):
    """3x3 stride-1 padding-1 convolution of a channels_last fp32 HIP tensor by Winograd F(2x2, 3x3):
    U [16, Cin, Cout] = G g G^T (see unet._wino_weights).  Returns [B, Cout, H, W] channels_last, no bias.
    ``gn`` = (gamma, beta, groups, eps, silu, stats, pre_bias): the convolution's input is SiLU(GroupNorm(x + pre_bias)),
    applied inside the input transform (the normalised tensor is never materialised).
    ``stats_groups`` > 0: the output transform also adds bias[c] (+ ``residual``) and returns
    (y, GroupNorm statistics of y) -- the tail of a ResnetBlock, or conv1 + the statistics norm2 needs, in one pass.
    ``f16`` = (U3, u_scale, x_bound): run the 16 / 36 GEMMs as ONE fp16 batched GEMM with fp32 accumulation over a K axis
    that carries the three products of two-term fp16 splits (see gqhip.h:wino_in_nhwc_f16x3): U3 [T, 3 Cin, Cout] fp16 =
    [U_h; U_l; U_h] of U * u_scale, x_bound >= max|x| (guarantees the scaled transform stays inside fp16's range)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 4 or x.shape[2] % 2 or x.shape[3] % 2:
        raise GqHipError("wino_conv3x3 needs a dense channels_last fp32 HIP tensor, C % 4 == 0, even H and W")
    B, C, H, W = x.shape
    cout = U.shape[2]
    f4 = U.shape[0] == 36                                     # F(4x4,3x3): 6x6 tiles, 36 GEMMs
    if f4 and (H % 4 or W % 4):
        raise GqHipError("F(4x4,3x3) needs H, W multiples of 4")
    t = 4 if f4 else 2
    tiles = B * (H // t) * (W // t)
    L = lib()
    mscale = 1.0
    with torch.cuda.device(x.device):
        if f16 is not None:
            U3, u_scale, x_bound = f16[:3]
            # |B^T d B| <= amp * max|d| (amp = squared max abs row sum of B^T: 100 for F(4x4,3x3), 4 for F(2x2,3x3))
            amp = 100.0 if f4 else 4.0
            v_scale = 2.0 ** math.floor(math.log2(32768.0 / (amp * max(float(x_bound), 1e-30))))
            v_scale = min(v_scale, 2.0 ** 14)
            use_own = len(f16) > 3 and f16[3] is not None and tiles % 256 == 0 and own_gemm_fits(U.shape[0], tiles, cout, C)
            if use_own:
                # [h | l] operand (4 bytes per element) + our own GEMM kernel forming the three products (wino_gemm_f16x2:
                # weights in operand order)
                V = torch.empty((U.shape[0], tiles, 2 * C), dtype=torch.float16, device=x.device)
                if gn is not None:
                    gamma, beta, groups, eps, silu, stats, pre_bias = gn
                    _check(L.wino_in_gn_nhwc_f16x2(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(pre_bias),
                                                   stats.data_ptr(), V.data_ptr(), B, H, W, C, groups, float(eps),
                                                   1 if silu else 0, t, float(v_scale), _stream()), "wino_in_gn_nhwc_f16x2")
                else:
                    _check(L.wino_in_nhwc_f16x2(x.data_ptr(), V.data_ptr(), B, H, W, C, t, float(v_scale), _stream()),
                           "wino_in_nhwc_f16x2")
                M = torch.empty((U.shape[0], tiles, cout), dtype=torch.float32, device=x.device)
                _check(L.wino_gemm_f16x2(V.data_ptr(), f16[3].data_ptr(), M.data_ptr(), U.shape[0], tiles, C, cout,
                                         _stream()), "wino_gemm_f16x2")
                V = None
            else:
                V = torch.empty((U.shape[0], tiles, 3 * C), dtype=torch.float16, device=x.device)
            if V is None:
                pass
            elif gn is not None:     # x_bound then bounds SiLU(GroupNorm(x)), the tensor the transform actually sees
                gamma, beta, groups, eps, silu, stats, pre_bias = gn
                _check(L.wino_in_gn_nhwc_f16x3(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(pre_bias),
                                               stats.data_ptr(), V.data_ptr(), B, H, W, C, groups, float(eps),
                                               1 if silu else 0, t, float(v_scale), _stream()), "wino_in_gn_nhwc_f16x3")
            else:
                _check(L.wino_in_nhwc_f16x3(x.data_ptr(), V.data_ptr(), B, H, W, C, t, float(v_scale), _stream()),
                       "wino_in_nhwc_f16x3")
            if V is not None:
                M = torch.bmm(V, U3, out_dtype=torch.float32)     # ONE fp16 GEMM per tile position, fp32 accumulate
            mscale = 1.0 / (v_scale * u_scale)
        else:
            V = torch.empty((U.shape[0], tiles, C), dtype=x.dtype, device=x.device)
            if gn is not None:
                gamma, beta, groups, eps, silu, stats, pre_bias = gn
                _check((L.wino4_in_gn_nhwc_f32 if f4 else L.wino_in_gn_nhwc_f32)(
                    x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(pre_bias), stats.data_ptr(), V.data_ptr(), B, H, W, C,
                    groups, float(eps), 1 if silu else 0, _stream()), "wino_in_gn_nhwc_f32")
            else:
                _check((L.wino4_in_nhwc_f32 if f4 else L.wino_in_nhwc_f32)(x.data_ptr(), V.data_ptr(), B, H, W, C, _stream()),
                       "wino_in_nhwc_f32")
            M = torch.bmm(V, U)                               # 16 / 36 GEMMs [tiles, Cin] x [Cin, Cout] (hipBLASLt)
        y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        if stats_groups:
            if (residual is not None and (image_layout(residual) != 1 or tuple(residual.shape) != tuple(y.shape))) \
                    or not gn_nhwc_ok(cout, stats_groups):
                raise GqHipError("fused Winograd tail needs a channels_last residual of the output shape and a GroupNorm-compatible C")
            stats = _stats_records(GNSTAT_WORDS * B * stats_groups, x.device)
            _check(L.wino_out_res_nhwc_f32(M.data_ptr(), _ptr(residual), _ptr(bias), y.data_ptr(), stats.data_ptr(),
                                           B, H, W, cout, stats_groups, t, float(mscale), _stream()), "wino_out_res_nhwc_f32")
            return y, stats
        _check((L.wino4_out_nhwc_f32 if f4 else L.wino_out_nhwc_f32)(M.data_ptr(), y.data_ptr(), B, H, W, cout,
                                                                     float(mscale), _stream()), "wino_out_nhwc_f32")
    return y


_BMM_OUT_DTYPE = None


def bmm_out_dtype_ok(device) -> bool:
    """torch.bmm / torch.mm(..., out_dtype=torch.float32) on fp16 operands (fp32 accumulation AND fp32 result: what the fp16 x 3
    library-GEMM routes need) exists in recent PyTorch-ROCm only.  Probed once, on the device, with a 16 x 16 product; the conv
    stack takes its fp32-GEMM routes when it is missing (pit_hip/modules/unet.py) instead of raising a TypeError mid-forward."""
    global _BMM_OUT_DTYPE
    if _BMM_OUT_DTYPE is None:
        try:
            a = torch.ones(1, 16, 16, dtype=torch.float16, device=device)
            _BMM_OUT_DTYPE = bool(torch.bmm(a, a, out_dtype=torch.float32).dtype == torch.float32)
        except (TypeError, RuntimeError):
            _BMM_OUT_DTYPE = False
    return _BMM_OUT_DTYPE


ATTN_L_OK = (64, 256, 1024, 2304, 4096)   # token counts attn_softmax_split_f16x3 is instantiated for


def attention_f16x3(qkv, q_bound: float, v_bound: float):
    """softmax(q k^T C^-1/2) v for the fused projection qkv [B, L, 3C] fp32 (dense), single head: two library fp16 GEMMs with
    fp32 accumulation over K axes of two-term fp16 splits (gqhip.h:attn_split_qkv_f16x3) -- fp32-grade results at several
    times the rate of the fp32 GEMMs (a split-bf16 emulation on gfx950).  ``q_bound`` >= max(|q|, |k|), ``v_bound`` >= max|v|
    (rigorous bounds from the caller: GroupNorm bound x weight row sums).  Returns (O_raw [B, L, C] fp32, post_scale) with
    attention output = O_raw * post_scale (a power of two the consumer folds into its own scale)."""
    if not (qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3 and qkv.is_contiguous() and qkv.shape[2] % 12 == 0):
        raise GqHipError("attention_f16x3 needs a dense fp32 HIP tensor [B, L, 3C], C % 4 == 0")
    B, Ltok, C3 = qkv.shape
    C = C3 // 3
    if Ltok not in ATTN_L_OK:
        raise GqHipError("attention_f16x3: token count %d not in %s" % (Ltok, ATTN_L_OK))
    pow2 = lambda bound: min(2.0 ** math.floor(math.log2(32768.0 / max(float(bound), 1e-30))), 2.0 ** 14)
    # q k^T sums C products of scaled operands in fp32: keep sq^2 C q_bound^2 well inside fp32 (it always is: <= 2^30 C)
    sq, sv = pow2(q_bound), pow2(v_bound)
    L_ = lib()
    with torch.cuda.device(qkv.device):
        Q3 = torch.empty((B, Ltok, C3), dtype=torch.float16, device=qkv.device)
        K3 = torch.empty((B, Ltok, C3), dtype=torch.float16, device=qkv.device)
        V3 = torch.empty((B, 3 * Ltok, C), dtype=torch.float16, device=qkv.device)
        _check(L_.attn_split_qkv_f16x3(qkv.data_ptr(), Q3.data_ptr(), K3.data_ptr(), V3.data_ptr(), B, Ltok, C, sq, sv,
                                       _stream()), "attn_split_qkv_f16x3")
        S = torch.bmm(Q3, K3.transpose(1, 2), out_dtype=torch.float32)
        P3 = torch.empty((B, Ltok, 3 * Ltok), dtype=torch.float16, device=qkv.device)
        _check(L_.attn_softmax_split_f16x3(S.data_ptr(), P3.data_ptr(), B * Ltok, Ltok, float(C) ** -0.5 / (sq * sq), _stream()),
               "attn_softmax_split_f16x3")
        O = torch.bmm(P3, V3, out_dtype=torch.float32)
    return O, 1.0 / (16384.0 * sv)


def own_gemm_fits(positions: int, tiles: int, cout: int, cin: int = 256) -> bool:
    """Where wino_gemm_f16x2 replaces the library's K-concatenated GEMM: wherever its grid (256 x 256 tiles at Cout % 256 == 0,
    else 256 x 128) fills whole rounds of the chip reasonably (36 x 1024 tiles x 512 channels = 288 blocks = 1.125 rounds does
    not).  Alone the kernel is 1.05-1.21x the library at 256 input channels and at 16 x 4096 tiles, 0.91-0.95x at the two largest
    512-channel shapes (tools/wino_gemm2_bench.py) -- but its [h | l] operand also takes a third off what the input transform
    writes, so the step as a whole is faster with it everywhere (33.9 -> 33.4 ms)."""
    if cout % 256 == 0 and cin % 64 == 0:
        rounds = positions * (tiles // 256) * (cout // 256) / 256.0      # 8-wave blocks, one per CU
    else:
        rounds = positions * (tiles // 256) * (cout // 128) / 512.0      # 4-wave blocks, two per CU
    return rounds >= 1.0 and rounds / math.ceil(rounds) >= 0.85


def wino_weights_operand_order(h, l):
    """(U_h, U_l) [T, Cin, Cout] fp16 -> Wf [T, Cin/16, Cout/32, 2, 64, 8]: the B operands of v_mfma_f32_32x32x16_f16 as
    wino_gemm_f16x2 loads them (lane (c, hh) of column tile nt: k = 16 chunk + 8 hh .. + 7 of column 32 nt + c)."""
    T, cin, cout = h.shape
    planes = torch.stack([h, l], 0).reshape(2, T, cin // 16, 2, 8, cout // 32, 32)   # [pl, t, kc, hh, e, nt, c]
    return planes.permute(1, 2, 5, 0, 3, 6, 4).reshape(T, cin // 16, cout // 32, 2, 64, 8).contiguous()


def conv3_weights_f16(weight):
    """Operand-order fp16 x 3 weights of a [Cout, Cin, 3, 3] kernel for conv3x3_direct: (Wf [Cin/16, 9, Cout/32, 2, 64, 8]
    fp16, u_scale) -- see gqhip.h:conv3x3_gn_f16x3."""
    cout, cin, kk = weight.shape[0], weight.shape[1], weight.shape[2]
    if cout not in ((128, 256, 512, 1536) if kk == 1 else (128, 256)) or cin % 16 or tuple(weight.shape[2:]) not in ((3, 3), (1, 1)):
        raise GqHipError("conv3_weights_f16 needs a [128 | 256, Cin % 16 == 0, 3, 3] or [128 | 256 | 512 | 1536, Cin % 16 == 0, 1, 1] kernel")
    w = weight.detach().float()
    amax = float(w.abs().max())
    u_scale = 2.0 ** math.floor(math.log2(16384.0 / max(amax, 1e-30))) if amax > 0 else 1.0
    ws = w * u_scale
    hi = ws.half()
    lo = (ws - hi.float()).half()
    planes = torch.stack([hi, lo], 0)                                    # [plane, n, k, ky, kx]
    # n = 32 tile + c, k = 16 chunk + 8 h + e  ->  [chunk, ky, kx, tile, plane, h, c, e]
    p7 = planes.reshape(2, cout // 32, 32, cin // 16, 2, 8, kk, kk).permute(3, 6, 7, 1, 0, 4, 2, 5)
    return p7.reshape(cin // 16, kk * kk, cout // 32, 2, 64, 8).contiguous(), u_scale


def conv3x3_direct(x, wf, u_scale: float, x_bound: float, gn, residual=None, bias=None, stats_groups: int = 0):
    """3x3 stride-1 padding-1 convolution Cin -> Cout (128 or 256) of SiLU(GroupNorm(x)) for a channels_last fp32 HIP tensor x, as
    a direct (implicit GEMM) fp16 x 3 convolution with the normalisation applied while the patch is staged
    (gqhip.h:conv3x3_gn_f16x3); ``wf, u_scale`` from conv3_weights_f16, ``x_bound`` >= max|activated tensor|, ``gn`` as in
    wino_conv3x3.  Returns y, or (y, statistics of y) when ``stats_groups`` > 0 (+ bias, + residual in either case)."""
    if (image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 32 or x.shape[1] > 512 or x.shape[2] % 8
            or x.shape[3] % 32 or gn is None):
        raise GqHipError("conv3x3_direct needs a dense channels_last fp32 HIP tensor, C % 32 == 0, C <= 512, H % 8 == 0, W % 32 == 0, "
                         "and the GroupNorm that feeds the convolution")
    B, C, H, W = x.shape
    cout = wf.shape[2] * 32
    if wf.shape[0] * 16 != C:
        raise GqHipError("conv3x3_direct: weights are for %d input channels, x has %d" % (wf.shape[0] * 16, C))
    if residual is not None and (image_layout(residual) != 1 or tuple(residual.shape) != (B, cout, H, W)):
        raise GqHipError("conv3x3_direct: residual must be channels_last [B, Cout, H, W]")
    v_scale = min(2.0 ** math.floor(math.log2(32768.0 / max(float(x_bound), 1e-30))), 2.0 ** 14)
    L = lib()
    with torch.cuda.device(x.device):
        y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ostats = _stats_records(GNSTAT_WORDS * B * stats_groups, x.device) if stats_groups else None
        mscale = 1.0 / (v_scale * u_scale)
        gamma, beta, groups, eps, silu, stats, pre_bias = gn
        _check(L.conv3x3_gn_f16x3(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(pre_bias), stats.data_ptr(),
                                  groups, float(eps), 1 if silu else 0, float(v_scale), wf.data_ptr(), _ptr(bias),
                                  _ptr(residual), y.data_ptr(), _ptr(ostats), B, H, W, C, cout, max(stats_groups, 1),
                                  mscale, _stream()), "conv3x3_gn_f16x3")
    return (y, ostats) if stats_groups else y


def conv1x1_direct(x, wf, u_scale: float, scale, residual=None, bias=None, stats_groups: int = 0, pre_bias=None,
                   post_scale: float = 1.0):
    """1x1 convolution Cin -> Cout (128 | 256) of a channels_last fp32 HIP tensor as an fp16 x 3 GEMM over its pixels with the
    split of x inside the kernel (gqhip.h:conv1x1_f16x3).  ``wf, u_scale`` from conv3_weights_f16 of the [Cout, Cin, 1, 1]
    kernel; ``pre_bias``: per-channel bias still pending on x (added before the split); ``scale``: a float bound >=
    max|x + pre_bias|, or the device float[2] of f16_scales(statistics of x + pre_bias, 1.0, u_scale)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 32 or (x.shape[2] * x.shape[3]) % 256:
        raise GqHipError("conv1x1_direct needs a dense channels_last fp32 HIP tensor, C % 32 == 0, H * W % 256 == 0")
    B, C, H, W = x.shape
    cout = wf.shape[2] * 32
    if wf.shape[0] * 16 != C or wf.shape[1] != 1:
        raise GqHipError("conv1x1_direct: weights do not match (need conv3_weights_f16 of a [Cout, %d, 1, 1] kernel)" % C)
    if residual is not None and (image_layout(residual) != 1 or tuple(residual.shape) != (B, cout, H, W)):
        raise GqHipError("conv1x1_direct: residual must be channels_last [B, Cout, H, W]")
    if torch.is_tensor(scale):
        if post_scale != 1.0:
            raise GqHipError("conv1x1_direct: post_scale needs a host-side scale bound")
        sdev, v_scale, mscale = scale.data_ptr(), 0.0, 0.0
    else:
        v_scale = min(2.0 ** math.floor(math.log2(32768.0 / max(float(scale), 1e-30))), 2.0 ** 14)
        # post_scale: x is post_scale^-1 times the tensor meant (a power of two its producer left pending)
        sdev, mscale = None, float(post_scale) / (v_scale * u_scale)
    with torch.cuda.device(x.device):
        y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ostats = _stats_records(GNSTAT_WORDS * B * stats_groups, x.device) if stats_groups else None
        _check(lib().conv1x1_f16x3(x.data_ptr(), _ptr(pre_bias), wf.data_ptr(), sdev, float(v_scale), float(mscale), _ptr(bias), _ptr(residual),
                                   y.data_ptr(), _ptr(ostats), B, H * W, C, cout, max(stats_groups, 1), _stream()),
               "conv1x1_f16x3")
    return (y, ostats) if stats_groups else y


def conv3s2_weights_f16(weight):
    """Operand-order fp16 x 3 weights of a [Cout, Cin, 3, 3] kernel for conv3x3s2_direct (stride 2): k-steps in the order
    (phase, chunk, tap of the phase) -- gqhip.h:conv3x3s2_f16x3.  Returns (Wf [9 Cin/16, Cout/32, 2, 64, 8], u_scale)."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cout not in (128, 256, 512) or cin % 16 or tuple(weight.shape[2:]) != (3, 3):
        raise GqHipError("conv3s2_weights_f16 needs a [128 | 256 | 512, Cin % 16 == 0, 3, 3] kernel")
    w = weight.detach().float()
    amax = float(w.abs().max())
    u_scale = 2.0 ** math.floor(math.log2(16384.0 / max(amax, 1e-30))) if amax > 0 else 1.0
    ws = w * u_scale
    hi = ws.half()
    lo = (ws - hi.float()).half()
    planes = torch.stack([hi, lo], 0)                                    # [plane, n, k, ky, kx]
    std = planes.reshape(2, cout // 32, 32, cin // 16, 2, 8, 3, 3).permute(3, 6, 7, 1, 0, 4, 2, 5)   # [chunk, ky, kx, tile, plane, h, c, e]
    steps = []
    for a, b in ((0, 0), (0, 1), (1, 0), (1, 1)):
        taps = [(dy, dx) for dy in range(2 if a == 0 else 1) for dx in range(2 if b == 0 else 1)]
        for chunk in range(cin // 16):
            for dy, dx in taps:
                steps.append(std[chunk, 2 * dy + a, 2 * dx + b])
    return torch.stack(steps, 0).reshape(9 * (cin // 16), cout // 32, 2, 64, 8).contiguous(), u_scale


def conv3x3s2_direct(x, wf, u_scale: float, scale, bias=None, stats_groups: int = 0):
    """The reference's Downsample convolution -- F.pad(x, (0, 1, 0, 1)) then 3x3 stride 2 -- of a channels_last fp32 HIP tensor
    as an fp16 x 3 convolution on the four phase images (gqhip.h:conv3x3s2_f16x3).  ``wf, u_scale`` from conv3s2_weights_f16;
    ``scale``: a float bound >= max|x| or the device float[2] of f16_scales.  Returns y [B, Cout, H/2, W/2] (+ bias), or
    (y, statistics of y)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 16 or x.shape[2] % 16 or x.shape[3] % 64:
        raise GqHipError("conv3x3s2_direct needs a dense channels_last fp32 HIP tensor, C % 16 == 0, H % 16 == 0, W % 64 == 0")
    B, C, H, W = x.shape
    cout = wf.shape[1] * 32
    if wf.shape[0] != 9 * (C // 16):
        raise GqHipError("conv3x3s2_direct: weights do not match the input channels")
    if torch.is_tensor(scale):
        sdev, v_scale, mscale = scale.data_ptr(), 0.0, 0.0
    else:
        v_scale = min(2.0 ** math.floor(math.log2(32768.0 / max(float(scale), 1e-30))), 2.0 ** 14)
        sdev, mscale = None, 1.0 / (v_scale * u_scale)
    with torch.cuda.device(x.device):
        y = torch.empty((B, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ostats = _stats_records(GNSTAT_WORDS * B * stats_groups, x.device) if stats_groups else None
        _check(lib().conv3x3s2_f16x3(x.data_ptr(), wf.data_ptr(), sdev, float(v_scale), float(mscale), _ptr(bias), y.data_ptr(),
                                     _ptr(ostats), B, H, W, C, cout, max(stats_groups, 1), _stream()), "conv3x3s2_f16x3")
    return (y, ostats) if stats_groups else y


def upconv_weights_f16(phase_matrix, cin: int, cout: int):
    """Operand-order fp16 x 3 weights of upconv2x_direct from the sub-pixel phase matrix [4 Cin, 4 Cout] (row (2u + v) Cin + ci,
    column (2a + b) Cout + co: Upsample._phase_weights): (Wf [4, 4 Cin/16, Cout/32, 2, 64, 8] fp16 -- [phase][chunk * 4 + tap]
    [column tile][plane][lane (hh, c)][e] = plane[tap, 16 chunk + 8 hh + e, phase, 32 tile + c] --, u_scale)."""
    if cout not in (128, 256, 512) or cin % 16 or tuple(phase_matrix.shape) != (4 * cin, 4 * cout):
        raise GqHipError("upconv_weights_f16 needs the [4 Cin, 4 Cout] phase matrix, Cin % 16 == 0, Cout in (128, 256, 512)")
    w = phase_matrix.detach().float()
    amax = float(w.abs().max())
    u_scale = 2.0 ** math.floor(math.log2(16384.0 / max(amax, 1e-30))) if amax > 0 else 1.0
    ws = w * u_scale
    hi = ws.half()
    lo = (ws - hi.float()).half()
    planes = torch.stack([hi, lo], 0).reshape(2, 4, cin // 16, 2, 8, 4, cout // 32, 32)   # [pl, tap, chunk, hh, e, phase, nt, c]
    wf = planes.permute(5, 2, 1, 6, 0, 3, 7, 4).reshape(4, 4 * (cin // 16), cout // 32, 2, 64, 8)
    return wf.contiguous(), u_scale


def upconv2x_direct(x, wf, u_scale: float, scale, bias=None, stats_groups: int = 0):
    """The reference's Upsample (nearest x2 + conv 3x3) of a channels_last fp32 HIP tensor as the direct sub-pixel fp16 x 3
    convolution (gqhip.h:upconv2x_f16x3).  ``wf, u_scale`` from upconv_weights_f16; ``scale``: a float bound >= max|x| or the
    device float[2] of f16_scales.  Returns y [B, Cout, 2H, 2W] (+ bias), or (y, statistics of y)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 16 or x.shape[2] % 8 or x.shape[3] % 32:
        raise GqHipError("upconv2x_direct needs a dense channels_last fp32 HIP tensor, C % 16 == 0, H % 8 == 0, W % 32 == 0")
    B, C, H, W = x.shape
    cout = wf.shape[2] * 32
    if wf.shape[0] != 4 or wf.shape[1] != 4 * (C // 16):
        raise GqHipError("upconv2x_direct: weights do not match the input channels")
    if torch.is_tensor(scale):
        sdev, v_scale, mscale = scale.data_ptr(), 0.0, 0.0
    else:
        v_scale = min(2.0 ** math.floor(math.log2(32768.0 / max(float(scale), 1e-30))), 2.0 ** 14)
        sdev, mscale = None, 1.0 / (v_scale * u_scale)
    with torch.cuda.device(x.device):
        y = torch.empty((B, cout, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ostats = _stats_records(GNSTAT_WORDS * B * stats_groups, x.device) if stats_groups else None
        _check(lib().upconv2x_f16x3(x.data_ptr(), wf.data_ptr(), sdev, float(v_scale), float(mscale), _ptr(bias), y.data_ptr(),
                                    _ptr(ostats), B, H, W, C, cout, max(stats_groups, 1), _stream()), "upconv2x_f16x3")
    return (y, ostats) if stats_groups else y


def conv3x3_gn_small(x, w_ohwi, bias, gn):
    """conv3x3(SiLU(GroupNorm(x))) into 1..4 channels (gqhip.h:conv3x3_gn_small_f32): x channels_last fp32 [B, Cin, H, W],
    w_ohwi [Cout, 3, 3, Cin] fp32 contiguous, ``gn`` = (gamma, beta, groups, eps, silu, stats, pre_bias)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 32 or x.shape[2] % 16 or x.shape[3] % 16:
        raise GqHipError("conv3x3_gn_small needs a dense channels_last fp32 HIP tensor, C % 32 == 0, H % 16 == 0, W % 16 == 0")
    B, C, H, W = x.shape
    cout = w_ohwi.shape[0]
    if tuple(w_ohwi.shape) != (cout, 3, 3, C) or not w_ohwi.is_contiguous() or w_ohwi.dtype != torch.float32:
        raise GqHipError("conv3x3_gn_small: weights must be fp32 [Cout, 3, 3, Cin] contiguous")
    gamma, beta, groups, eps, silu, stats, pre_bias = gn
    y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _check(lib().conv3x3_gn_small_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(pre_bias), stats.data_ptr(),
                                          groups, float(eps), 1 if silu else 0, w_ohwi.data_ptr(), _ptr(bias), y.data_ptr(),
                                          B, H, W, C, cout, _stream()), "conv3x3_gn_small_f32")
    return y


def checksum_table(tensors):
    """Device table for checksum_tensors: one {pointer, 32-bit words} entry per tensor (dense fp32 / int32 HIP tensors)."""
    dev = tensors[0].device
    rows = []
    for t in tensors:
        if not (t.is_cuda and t.element_size() == 4 and t.data_ptr() % 16 == 0 and t.device == dev):
            raise GqHipError("checksum_table needs 16-byte aligned 4-byte-element tensors on one HIP device")
        rows += [t.data_ptr(), t.numel()]
    return torch.tensor(rows, dtype=torch.int64).to(dev)


def checksum_tensors(table, out):
    """out[t] (int64, device) = content hash of tensor t of ``table`` (gqhip.h:gqhip_checksum_tensors); asynchronous."""
    with torch.cuda.device(table.device):
        _check(lib().gqhip_checksum_tensors(table.data_ptr(), table.numel() // 2, out.data_ptr(), _stream()), "gqhip_checksum_tensors")
    return out


def conv_f32_ok(cin: int, cout: int, H: int, W: int, gn: bool) -> bool:
    """Shapes conv3x3_f32 tiles (gqhip.h)."""
    return cout % 4 == 0 and (cin % 64 == 0 and (not gn or cin <= 1024) or (not gn and cin in (8, 16, 32)))


def conv_f32_weights(weight):
    """[Cout, Cin, 3, 3] fp32 -> the operand order of conv3x3_f32: [ceil(Cout/32)][9][Cin/8][64][4], element
    (t, tap, g, 32 h + j, m) = w[32 t + j][8 g + 4 h + m][tap // 3][tap % 3], zero rows beyond Cout."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cin % 8 or tuple(weight.shape[2:]) != (3, 3):
        raise GqHipError("conv_f32_weights needs a [Cout, Cin % 8 == 0, 3, 3] weight")
    nt = (cout + 31) // 32
    w = weight.detach().float().reshape(cout, cin, 9)
    if nt * 32 != cout:
        w = torch.cat([w, w.new_zeros(nt * 32 - cout, cin, 9)], 0)
    return w.reshape(nt, 32, cin // 8, 2, 4, 9).permute(0, 5, 2, 3, 1, 4).contiguous()


def conv3x3_f32(x, wk, cout: int, bias=None, gn=None):
    """3x3 convolution (stride 1, padding 1) of a channels_last fp32 HIP tensor on the fp32 matrix cores with a fixed summation
    order (gqhip.h:conv3x3_f32: bit-reproducible).  ``wk`` from conv_f32_weights; ``gn`` = (gamma, beta, groups, eps, silu, stats,
    pre_bias): the input is act(GroupNorm(x + pre_bias)), applied while the patch is staged."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32:
        raise GqHipError("conv3x3_f32 needs a dense channels_last fp32 HIP tensor")
    B, C, H, W = x.shape
    if not conv_f32_ok(C, cout, H, W, gn is not None) or wk.numel() != ((cout + 31) // 32) * 9 * (C // 8) * 256:
        raise GqHipError(f"conv3x3_f32: shape {tuple(x.shape)} -> {cout} channels is not tiled by the kernel")
    y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    gamma = beta = stats = pre_bias = None
    groups, eps, silu = 1, 0.0, False
    if gn is not None:
        gamma, beta, groups, eps, silu, stats, pre_bias = gn
    with torch.cuda.device(x.device):
        _check(lib().conv3x3_f32(x.data_ptr(), _ptr(gamma), _ptr(beta), _ptr(pre_bias), _ptr(stats), groups, float(eps),
                                 1 if silu else 0, wk.data_ptr(), _ptr(bias), y.data_ptr(), B, H, W, C, cout, _stream()),
               "conv3x3_f32")
    return y


def conv_cin_small_ok(cin: int, cout: int, H: int, W: int) -> bool:
    """Shapes conv3x3_cin_small_f32 tiles (gqhip.h): the encoder's conv_in."""
    return 1 <= cin <= 4 and cout == 128 and H % 8 == 0 and W % 32 == 0 and H >= 8 and W >= 32


def conv_cin_small_weights(weight):
    """[128, Cin <= 4, 3, 3] fp32 -> [9 Cin, 128]: row tap * Cin + ci, column co."""
    cout, cin = weight.shape[0], weight.shape[1]
    return weight.detach().float().permute(2, 3, 1, 0).reshape(9 * cin, cout).contiguous()


def conv3x3_cin_small(x, wk, bias=None, stats_groups: int = 0):
    """The encoder's conv_in on libgqhip (gqhip.h:conv3x3_cin_small_f32): 3x3 / stride 1 / pad 1, Cin <= 4 -> 128 channels of a
    channels_last fp32 HIP image, fp32 FMAs in a fixed order, bias added; ``stats_groups`` = 32: returns (y, statistics of y)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32:
        raise GqHipError("conv3x3_cin_small needs a dense channels_last fp32 HIP tensor")
    B, C, H, W = x.shape
    if not conv_cin_small_ok(C, wk.shape[1], H, W) or wk.shape[0] != 9 * C or stats_groups not in (0, 32):
        raise GqHipError(f"conv3x3_cin_small: shape {tuple(x.shape)} is not tiled by the kernel")
    with torch.cuda.device(x.device):
        y = torch.empty((B, 128, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ostats = _stats_records(GNSTAT_WORDS * B * 32, x.device) if stats_groups else None
        _check(lib().conv3x3_cin_small_f32(x.data_ptr(), wk.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(ostats), B, H, W, C, 128,
                                           32, _stream()), "conv3x3_cin_small_f32")
    return (y, ostats) if stats_groups else y


def f16_scales(stats, amp: float, u_scale: float):
    """Device float[2] = (v_scale, 1 / (v_scale * u_scale)) for an fp16 x 3 GEMM whose activation operand x has the
    GroupNorm statistics ``stats`` (n_bg records of GNSTAT_WORDS int64: gqhip_gnstat_t): v_scale = the largest power of two with
    amp * sqrt(max sum of squares) * v_scale <= 32768.  Computed on the device (no sync)."""
    if not (stats.is_cuda and stats.dtype == torch.int64 and stats.is_contiguous() and stats.numel() % GNSTAT_WORDS == 0):
        raise GqHipError("f16_scales needs the statistics tensor of add_bias_stats / wino_conv3x3")
    out = torch.empty(2, dtype=torch.float32, device=stats.device)
    with torch.cuda.device(stats.device):
        _check(lib().f16_scales_from_gn_stats(stats.data_ptr(), stats.numel() // GNSTAT_WORDS, float(amp), float(u_scale),
                                              out.data_ptr(), _stream()), "f16_scales_from_gn_stats")
    return out


def upsample2x_nhwc(x):
    """Nearest x2 upsample of a channels_last fp32 HIP tensor [B, C, H, W] -> [B, C, 2H, 2W] (channels_last)."""
    if image_layout(x) != 1 or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 4:
        raise GqHipError("upsample2x_nhwc needs a dense channels_last fp32 HIP tensor with C % 4 == 0")
    B, C, H, W = x.shape
    y = torch.empty((B, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _check(lib().upsample2x_nhwc_f32(x.data_ptr(), y.data_ptr(), B, H, W, C, _stream()), "upsample2x_nhwc_f32")
    return y


def index_histogram(idx, n: int):
    idx = _dev(idx, torch.int64, "indices")
    hist = torch.empty(n, dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        _check(lib().gq_index_histogram(idx.data_ptr(), idx.numel(), n, hist.data_ptr(), _stream()),
               "gq_index_histogram")
    return hist


def indices_to_u16(idx):
    idx = _dev(idx, torch.int64, "indices")
    out = torch.empty(idx.shape, dtype=torch.uint16, device=idx.device)
    with torch.cuda.device(idx.device):
        _check(lib().gq_indices_to_u16(idx.data_ptr(), out.data_ptr(), idx.numel(), _stream()), "gq_indices_to_u16")
    return out


def indices_from_u16(u16):
    if not (u16.is_cuda and u16.dtype == torch.uint16):
        raise GqHipError("expected a uint16 HIP tensor")
    u16 = u16.contiguous()
    out = torch.empty(u16.shape, dtype=torch.int64, device=u16.device)
    with torch.cuda.device(u16.device):
        _check(lib().gq_indices_from_u16(u16.data_ptr(), out.data_ptr(), u16.numel(), _stream()),
               "gq_indices_from_u16")
    return out


def step_record_ok(x, x_rec, idx) -> bool:
    """gq_step_record_f32 applies: fp32 HIP images of one dense layout (NCHW or channels_last), int64 indices on the same device."""
    if not (x.is_cuda and x_rec.is_cuda and idx.is_cuda and x.dtype == torch.float32 and x_rec.dtype == torch.float32
            and idx.dtype == torch.int64 and x.shape == x_rec.shape and x.dim() == 4):
        return False
    la, lb = image_layout(x), image_layout(x_rec)
    return la is not None and la == lb


def step_record(x, x_rec, idx, rec, ws_cache: dict):
    """rec[:] = [ per-image PSNR(x, x_rec; zero_mean) as fp32 bits | idx as uint16 pairs ] in ONE launch (gqhip.h:
    gq_step_record_f32; eval.py:152-154,165-169).  ``ws_cache``: the caller's dict that keeps the (zeroed, self-resetting)
    workspace between calls."""
    B = x.shape[0]
    per = x[0].numel()
    idx = idx.contiguous()      # (any memory order of the SAME logical [B, K, h, w] order is fine for the caller; contiguous = logical order)
    need = lib().gq_step_record_workspace_bytes(B, per)
    key = (x.device, B, per)
    ws = ws_cache.get(key)
    if ws is None:
        ws = ws_cache[key] = torch.zeros(max(need, 8), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().gq_step_record_f32(x.data_ptr(), x_rec.data_ptr(), idx.data_ptr(), rec.data_ptr(), B, per, idx.numel(),
                                        ws.data_ptr(), ws.numel(), _stream()), "gq_step_record_f32")
    return rec


def profile_enable(on: bool) -> None:
    lib().gqhip_profile_enable(1 if on else 0)


def profile_reserve(pairs: int) -> None:
    """Pre-create event pairs so that profiled launches create nothing (call outside the timed region)."""
    _check(lib().gqhip_profile_reserve(int(pairs)), "gqhip_profile_reserve")


def profile_collect() -> Tuple[int, float]:
    n = ctypes.c_int(0)
    ms = ctypes.c_double(0.0)
    lib().gqhip_profile_collect(ctypes.byref(n), ctypes.byref(ms))
    return n.value, ms.value


def debug_enable(on: bool) -> None:
    lib().gqhip_debug_enable(1 if on else 0)


FILTER_KINDS = {"auto": 0, "fp32": 1, "bf16": 2, "mixed": 3}


def set_filter(kind: str) -> None:
    """"auto": fp16 + fp8 MFMA filter at dim 16 (Gaussian score), split-bf16 at the other MFMA dims and for VQ (other dims use
    the exhaustive kernel either way); "fp32": always the fp32 MFMA filter; "bf16": split-bf16 wherever it applies.
    Process-wide; indices are identical whichever runs (the exact re-rank decides)."""
    if kind not in FILTER_KINDS:
        raise GqHipError(f"unknown filter {kind!r} (expected one of {sorted(FILTER_KINDS)})")
    _check(lib().gqhip_set_filter(FILTER_KINDS[kind]), "gqhip_set_filter")


def get_filter() -> str:
    k = lib().gqhip_get_filter()
    return next(name for name, v in FILTER_KINDS.items() if v == k)


def debug_plan(rows: int, n: int, dim: int) -> dict:
    out = (_i64 * 8)()
    _check(lib().gqhip_debug_plan(rows, n, dim, out), "gqhip_debug_plan")
    keys = ("rec_offset", "nsplit", "gt", "tiles_per_split", "bf16", "ef_coeff", "rt", "waves")
    return dict(zip(keys, (int(v) for v in out)))


def debug_records(ws: Workspace, rows: int, n: int, dim: int):
    """Candidate records the filter left in the workspace, decoded from their 16-byte form (csrc/gq_common.h:Rec):
    (m [sets, rows, 4] fp32: m1 and the re-rank's view m1 - gap of m2..m4; ids [sets, rows, 3] int32: GLOBAL half-group ids)."""
    pl = debug_plan(rows, n, dim)
    sets = pl["nsplit"]
    raw = ws.buf[pl["rec_offset"]: pl["rec_offset"] + sets * rows * 16].reshape(sets, rows, 16)
    m1 = raw[..., 0:4].contiguous().view(torch.float32)                                   # [sets, rows, 1]
    gaps = raw[..., 4:10].contiguous().view(torch.float16).to(torch.float64)               # [sets, rows, 3]
    m = torch.cat([m1.to(torch.float64), m1.to(torch.float64) - gaps], dim=-1).to(torch.float32)
    local = raw[..., 10:16].contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    halves = 2 if pl["bf16"] in (2, 3) else 1                                              # one set per (split, lane half)
    split = torch.arange(sets, device=raw.device, dtype=torch.int32) // halves
    base2 = 2 * ((split * pl["tiles_per_split"]) // pl["gt"])
    return m, local + base2[:, None, None]


def debug_grid(ws: Workspace) -> dict:
    """Grid search statistics of the last call on ``ws`` (counted only while ``debug_enable(True)``; synchronous)."""
    out = (_i64 * 4)()
    cptr = None if ws.cache_buf is None else ws.cache_buf.data_ptr()
    _check(lib().gqhip_debug_grid(ws.buf.data_ptr(), cptr, out), "gqhip_debug_grid")
    # out[0]: visited sub-leaves (1/4096 of the codebook each); "leaves": the same in leaves (four sub-leaves, 1/1024 of the codebook)
    return {"sub_leaves": out[0], "leaves": out[0] / 4.0, "exact_codes": out[1], "scanned_rows": out[2], "index_current": out[3]}


def debug_counters(ws: Workspace) -> Tuple[int, int]:
    fb, rr = _i64(0), _i64(0)
    _check(lib().gqhip_debug_counters(ws.buf.data_ptr(), ctypes.byref(fb), ctypes.byref(rr)), "gqhip_debug_counters")
    return fb.value, rr.value

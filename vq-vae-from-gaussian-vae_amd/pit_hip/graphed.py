"""HIP-graph replay of the inference path (serving: one ``hipGraphLaunch`` instead of ~600 kernel launches).

``GraphedAutoencoder(vae, example)`` captures ``indices = vae.quant(x)[1]; rec = vae.decode(zhat)`` for the
example's shape once (after a warm-up that sizes every workspace) and replays it for new inputs of the same
shape.  Everything on the path is capture-safe: the C-ABI entry points never synchronise or allocate, and the
PyTorch convs run with MIOpen's immediate mode.  ``zhat_noquant`` (an RNG draw in the reference's eval branch,
gaussian.py:121) is not part of the captured outputs.  Measured on MI355X (tools/latency.py): replay is 1.00x the
eager path at B = 1, 4, 16 -- the kernels, not their launches, bound this path -- so this is a convenience for
launch-constrained hosts, not a speed-up.
"""
from __future__ import annotations

import torch


class GraphedAutoencoder:
    def __init__(self, vae, example: torch.Tensor, warmup: int = 2) -> None:
        assert example.is_cuda, "HIP graphs need a HIP device"
        self.vae = vae
        self._x = example.clone()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    z, ind = vae.quant(self._x)
                    vae.decode(z)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                z, self._indices = vae.quant(self._x)
                self._rec = vae.decode(z)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor):
        """Returns (reconstruction, indices); the tensors are the graph's static outputs (clone to keep)."""
        assert x.shape == self._x.shape, "a captured graph serves one input shape"
        self._x.copy_(x)
        self._graph.replay()
        return self._rec, self._indices

"""Inference engine wiring encoder -> regularizer -> decoder.

Mirror of the reference's ``AutoencodingEngine`` API for the hot path
(pit/models/autoencoder.py:359-423): ``encode / decode / quant / dequant /
forward`` with the same arguments and the same ``state_dict`` prefixes
(``encoder.``, ``decoder.``, ``regularization.``), minus Lightning and the
training loop.  ``loss_config`` and optimizer arguments are accepted and
ignored (the reference builds LPIPS + discriminator unless ``eval_only``;
inference never touches them)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from ..util import instantiate_from_config


class AutoencodingEngine(nn.Module):
    def __init__(self, *args, input_key: str = "img", eval_only: bool = True, encoder_config: Dict,
                 decoder_config: Dict, regularizer_config: Dict, loss_config: Optional[Dict] = None,
                 ckpt_path: Optional[str] = None, ckpt_engine: Union[None, str, dict] = None,
                 clamp_range=None, latent_stats: bool = False, use_vf=None, **ignored) -> None:
        super().__init__()
        if use_vf is not None:
            raise NotImplementedError("use_vf (foundation-model alignment) is training-only and out of scope")
        self.input_key = input_key
        self.encoder: nn.Module = instantiate_from_config(encoder_config)
        self.decoder: nn.Module = instantiate_from_config(decoder_config)
        self.regularization: nn.Module = instantiate_from_config(regularizer_config)
        self.clamp_range = clamp_range
        self.latent_stats = latent_stats
        if latent_stats:
            zc = encoder_config["params"]["z_channels"]
            self.latent_mean = nn.Parameter(torch.zeros([1, zc, 1, 1]), requires_grad=False)
            self.latent_std = nn.Parameter(torch.zeros([1, zc, 1, 1]), requires_grad=False)
        ckpt = ckpt_path if ckpt_path is not None else ckpt_engine
        if ckpt_path is not None:
            assert ckpt_engine is None, "Can't set ckpt_engine and ckpt_path"
        if ckpt is not None:
            self.init_from_ckpt(ckpt)

    def init_from_ckpt(self, path, ignore_keys=()):
        # autoencoder.py:318-329; `loss.*` keys are simply unexpected under strict=False
        sd = torch.load(path, map_location="cpu")["state_dict"]
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        self.invalidate_caches()
        print("Missing keys: ", missing)
        print(f"Restored from {path}")
        return missing, unexpected

    def invalidate_caches(self) -> None:
        """Drop the conv stack's weight-derived caches (see pit_hip.modules.unet.invalidate_caches): needed only after
        weights were changed through ``param.data`` (no version bump), e.g. an EMA swap."""
        from ..modules.unet import invalidate_caches

        invalidate_caches(self)

    def get_input(self, batch: Dict) -> torch.Tensor:
        return batch[self.input_key]

    def get_last_layer(self):
        return self.decoder.get_last_layer()

    def encode(self, x: torch.Tensor, return_reg_log: bool = False, unregularized: bool = False):
        z = self.encoder(x)
        if unregularized:
            return z, dict()
        z, reg_log = self.regularization(z)
        if self.latent_stats:
            z = (z - self.latent_mean) / self.latent_std
        if return_reg_log:
            return z, reg_log
        return z

    def decode(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        if self.latent_stats:
            z = z * self.latent_std + self.latent_mean
        return self.decoder(z, **kwargs)

    def quant(self, x):
        z, reg_log = self.encode(x, return_reg_log=True)
        return z, reg_log["indices"]

    def dequant(self, incides):
        xhat = self.decode(self.regularization.dequant(incides))
        if self.clamp_range is not None:
            xhat = torch.clamp(xhat, self.clamp_range[0], self.clamp_range[1])
        return xhat

    def forward(self, x: torch.Tensor, encoder_grad: bool = True, **additional_decode_kwargs
                ) -> Tuple[torch.Tensor, torch.Tensor, dict]:
        if encoder_grad:
            z, reg_log = self.encode(x, return_reg_log=True)
        else:
            with torch.no_grad():
                z, reg_log = self.encode(x, return_reg_log=True)
        dec = self.decode(z, **additional_decode_kwargs)
        if self.clamp_range is not None:
            dec = torch.clamp(dec, self.clamp_range[0], self.clamp_range[1])
        return z, dec, reg_log

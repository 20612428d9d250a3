"""``build_model(cfg) -> nn.Module``: the reference's model factory (``/root/reference/models/model_builder.py:14-242``)
for the models on the hot path.

* ``cfg.model in {'UNet', 'UNetSmall'}`` -- MONAI BasicUNet topology (BASELINE.json configs 1-3; the reference has no
  UNet).
* ``cfg.model == 'nnFormerUNETR'`` -- ``SwinTransformerNNFormer`` encoder + ``SwinUNETRCustom`` decoder, the branch
  at ``model_builder.py:15-66``.
* ``cfg.model == 'SwinDepth'`` -- the same wiring around the ``SwinDepth`` encoder (depthwise-conv + BatchNorm MLP), the
  branch at ``model_builder.py:120-171``.
* ``cfg.model == 'SwInception'`` -- the same wiring around the ``SwInception`` encoder (Inception-head MLP: Conv3d +
  BatchNorm3d + GELU branches with 6- / 38-channel convolutions at width 48), the branch at ``model_builder.py:67-119``.
* ``cfg.model == 'SwinSegFormer'`` -- ``SwinTransformerNNFormer`` + the progressive-fusion ``SegFormerHead``
  (``model_builder.py:173-189``, ``models/segmentors/segformer_head.py``).
* ``cfg.model == 'SegFormer3D'`` -- ``MixVisionTransformer`` + ``SegFormerHeadOfficial`` (``model_builder.py:190-205``).
* ``cfg.model == 'SwinUNETR'`` -- the vendored MONAI variant of ``models/segmentors/swin_unetr_official.py`` (window 7,
  ``feature_size = cfg.hidden_dim``), the literal "Swin-UNETR 48-feat" of BASELINE.json configs[3] (the reference keeps
  the class but wires no ``build_model`` branch to it).
Every returned module obeys the engine contract ``model((vol, rel_crop_loc, affine_xyz)) -> logits`` and keeps the
reference's / MONAI's state-dict key layout.  ``cfg.compute_dtype``: 'bf16' (default) or 'f32'.
"""
from __future__ import annotations

import torch

from .unet import UNET_FEATURES, UNet

OUT_OF_SCOPE = ("GCViTUNETR", "FocalNetUNETR")


def _dtype(cfg):
    name = str(getattr(cfg, "compute_dtype", "bf16")).lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    raise ValueError(f"compute_dtype must be bf16 or f32, got {name}")


def _t3(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def build_model(cfg):
    name = cfg.model
    if name in UNET_FEATURES:
        return UNet(cfg.in_chans, cfg.output_dim, UNET_FEATURES[name], compute_dtype=_dtype(cfg))
    if name in ("nnFormerUNETR", "SwinDepth", "SwInception"):
        from .swin_unetr import SwInception, SwinDepth, SwinTransformerNNFormer, SwinUNETRCustom
        for flag in ("learned_cls_vectors", "rel_pos_bias_affine", "rel_crop_pos_emb", "abs_pos_emb", "global_token"):
            if getattr(cfg, flag, False):
                raise NotImplementedError(f"--{flag} is outside the hot-path scope of this build (SURVEY.md section 2)")
        ws = cfg.window_size if isinstance(cfg.window_size, (tuple, list)) else (cfg.window_size,) * len(cfg.depths)
        enc_cls = {"SwinDepth": SwinDepth, "SwInception": SwInception}.get(name, SwinTransformerNNFormer)   # model_builder.py:67-171
        encoder = enc_cls(pretrain_img_size=_t3(cfg.vol_size), patch_size=_t3(cfg.patch_size), in_chans=cfg.in_chans,
                          embed_dim=cfg.hidden_dim, depths=tuple(cfg.depths), num_heads=tuple(cfg.num_heads),
                          window_size=tuple(ws), qkv_bias=cfg.qkv_bias, mlp_ratio=getattr(cfg, "mlp_ratio", 4.0),
                          compute_dtype=_dtype(cfg))
        return SwinUNETRCustom(encoder, in_channels=cfg.in_chans, out_channels=cfg.output_dim,
                               img_size=_t3(cfg.vol_size), hidden_size=cfg.hidden_dim, patch_size=_t3(cfg.patch_size),
                               compute_dtype=_dtype(cfg))
    if name == "SwinSegFormer":                                 # model_builder.py:173-189
        from .segformer3d import SegFormerHead
        from .swin_unetr import SwinTransformerNNFormer
        if getattr(cfg, "abs_pos_emb", False):
            raise NotImplementedError("--abs_pos_emb is outside the hot-path scope of this build (SURVEY.md section 2)")
        if len(cfg.depths) != 4:
            raise ValueError("SwinSegFormer's head fuses five feature maps: a four-stage encoder (--depths a b c d)")
        ws = cfg.window_size if isinstance(cfg.window_size, (tuple, list)) else (cfg.window_size,) * len(cfg.depths)
        encoder = SwinTransformerNNFormer(pretrain_img_size=_t3(cfg.vol_size), patch_size=_t3(cfg.patch_size),
                                          in_chans=cfg.in_chans, embed_dim=cfg.hidden_dim, depths=tuple(cfg.depths),
                                          num_heads=tuple(cfg.num_heads), window_size=tuple(ws), qkv_bias=cfg.qkv_bias,
                                          compute_dtype=_dtype(cfg))
        return SegFormerHead(encoder, [cfg.hidden_dim * 2 ** i for i in range(len(cfg.depths) + 1)], cfg.output_dim,
                             compute_dtype=_dtype(cfg))
    if name == "SegFormer3D":                                   # model_builder.py:190-205
        from .segformer3d import MixVisionTransformer, SegFormerHeadOfficial
        enc = MixVisionTransformer(img_size=cfg.vol_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
                                   embed_dim=cfg.hidden_dim, depths=tuple(cfg.depths), num_heads=tuple(cfg.num_heads),
                                   sr_ratios=(8, 4, 2, 1), qkv_bias=cfg.qkv_bias, compute_dtype=_dtype(cfg))
        return SegFormerHeadOfficial(enc, [cfg.hidden_dim * 2 ** i for i in range(len(cfg.depths))], cfg.output_dim,
                                     compute_dtype=_dtype(cfg))
    if name == "SwinUNETR":
        from .swin_unetr_official import SwinUNETR
        return SwinUNETR(_t3(cfg.vol_size), cfg.in_chans, cfg.output_dim, depths=tuple(cfg.depths),
                         num_heads=tuple(cfg.num_heads), feature_size=cfg.hidden_dim, compute_dtype=_dtype(cfg))
    if name in OUT_OF_SCOPE:
        raise NotImplementedError(f"model '{name}' is a research variant outside this build's hot-path scope "
                                  f"(SURVEY.md section 2); available: {sorted(UNET_FEATURES)} + ['nnFormerUNETR', 'SwinDepth', 'SwInception', 'SwinSegFormer', 'SegFormer3D', 'SwinUNETR']")
    raise ValueError(f"unknown cfg.model '{name}'")

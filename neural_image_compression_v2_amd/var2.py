"""Configuration: the names of the reference's Projects/var2.py as a dataclass.  The reference sets module
globals and overrides them from ``NAME=value`` argv with exec (var2.py:90-95); here the same names are fields,
``from_argv`` parses the same syntax without exec, and the derived values (var2.py:107-123) are properties.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Sequence

import torch

from .utils import bits2dtype_torch, judge_value

_TYPES = {int: "int", float: "float", bool: "bool", str: "str"}


@dataclass
class Settings:
    IMAGE_PATH: str = "data/sancho_512.png"
    PROJECT_NAME: str = "image_compression"
    IMAGE_DTYPE: str = "image"
    COMPRESSION_METHOD: int = 1          # 1: 2D, 2: 3D flattened to 2D, 3: 3D pyramid, 4: proposed (tetrahedral G0)
    MLP_NUM_DTYPE: int = 32
    NUM_EPOCHS: int = 1000
    UNIFORM_DISTRIBUTION_RATE: float = 0.05
    IMAGE_3D_SIZE: int = 64
    IMAGE_SIZE: int = 512
    IMAGE_DIMENSION: int = 2
    MAX_MIP_LEVEL: int = 9
    IMAGE_BITS: int = 8
    OUTPUT_BITS: int = 8
    FEATURE_PYRAMID_CHANNELS: int = 12
    PE_CHANNELS: int = 6
    FP_BITS: int = 8
    HIDDEN_LAYER_CHANNELS: int = 64
    CROP_MIP_LEVEL: int = 8
    NUM_CROPS: int = 8
    INTERVAL_PRINT: int = 100
    INTERVAL_SAVE_MODEL: int = 100000
    TF_NO_MIP: bool = True
    TF_USE_TRI_PE: bool = True
    TF_SPLIT_BF16: bool = True           # not in the reference: fused training steps and decodes run their matrix products as hi + lo bf16
                                         # pairs on the bf16 matrix pipe (2D: all of them, 1.8x; 3D: the chained ones, 1.2x; gradients within
                                         # 5e-6 of the fp32 kernels).  False: fp32 MFMAs throughout
    TF_GRID_BF16: bool = False           # not in the reference: bfloat16 grid STORAGE with fp32 masters (MLP_NUM_DTYPE = 16 is the reference's own route and
                                         # maps the grids to float16, utils.py:301-313; this flag selects bfloat16 instead)
    TF_PLAIN_BF16: int = 0               # not in the reference: fused training steps in PLAIN 16-bit products (NIC_FLAG_BF16: one bf16 value per operand, fp32
                                         # accumulation; every layout, 3 or 5 Linear layers; 1.5x (2D) / 2.6x (3D) faster than the split products; outputs
                                         # within 2e-4 of fp32 arithmetic, gradients ~5e-3 - BASELINE.json's "bf16").  1 / True: on; 0 / False (default): the
                                         # reference-faithful arithmetic (split-bf16 = fp32-equivalent, or fp32 with TF_SPLIT_BF16 = False) in every
                                         # dimension; -1: on for 3D fits only (round 3's default).  A fit decodes in the arithmetic it trained in (`decode_image`)
    TF_PLAIN_FP16: bool = False          # with TF_PLAIN_BF16: the products run on fp16 operands instead (NIC_FLAG_FP16: the reference's own 16-bit type,
                                         # utils.py:301-313, `v_mfma_f32_16x16x32_f16`; dZ carried with a static power-of-two loss scale) - BASELINE config 3's "fp16"
    DECODER_LINEAR_LAYERS: int = 3       # not in the reference (depth is hard-coded at 3, image_compression.py:57-64): 5 = the "4 x 64" decoder of
                                         # the north star; served by the fused 2D step / decode with TF_SPLIT_BF16
    TF_DEVICE_SAMPLER: bool = False      # not in the reference: LOD and crop origins from the counter-based device sampler (sampler.py) and targets
                                         # from the resident RGBX pyramid instead of the host RNG calls of random_crop_dataset - no host
                                         # round-trip per step; False keeps the reference's RNG streams (random / torch.randint) draw for draw
    SAMPLER_SEED: int = 0
    TF_MIP_FILTER: str = "resize"       # not in the reference: how TF_DEVICE_SAMPLER builds its RGBX mip levels - "resize" = the reference's own transforms.Resize chain
                                         # (Pillow's BILINEAR resize of the original image, image_compression.py:432-440, bit-exact), "box" = 2 x 2 box filter level by level
    TF_TRAIN_MODEL: bool = True
    TF_SHOW_RESULT: bool = False
    TF_PRINT_LOG: bool = True
    TF_PRINT_PSNR: bool = True
    TF_WRITE_TIME: bool = True
    TF_WRITE_PSNR: bool = True

    def __post_init__(self):
        if self.TF_NO_MIP:
            self.MAX_MIP_LEVEL = 0                                   # var2.py:112-113

    @classmethod
    def from_argv(cls, argv: Sequence[str]) -> "Settings":
        kw = {}
        names = {f.name: f.type for f in fields(cls)}
        for arg in argv:
            for name, typ in names.items():
                if arg.startswith(name + "="):
                    t = typ if isinstance(typ, str) else typ.__name__
                    kw[name] = judge_value(arg, t, name)
        return cls(**kw)

    @property
    def DEVICE(self):
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    @property
    def plain_bf16(self) -> bool:
        """TF_PLAIN_BF16 resolved: explicit 0 / 1 (False / True), or by dimension when set to -1"""
        v = int(self.TF_PLAIN_BF16)
        return self.FP_DIMENSION == 3 if v < 0 else bool(v)

    @property
    def FEATURE_PYRAMID_SIZE(self):
        return self.IMAGE_SIZE // 4                                   # var2.py:107

    @property
    def FP_DIMENSION(self):
        return 2 if self.COMPRESSION_METHOD == 2 else self.IMAGE_DIMENSION

    @property
    def DECODER_INPUT_CHANNELS(self):
        C, P, D = self.FEATURE_PYRAMID_CHANNELS, self.PE_CHANNELS, self.FP_DIMENSION
        if self.COMPRESSION_METHOD == 4:
            return C * (pow(2, 2) + 1) + P * D + 1                    # var2.py:117-118
        return C * (pow(2, D) + 1) + P * D + 1

    @property
    def CROP_SIZE(self):
        return pow(2, self.CROP_MIP_LEVEL)

    @property
    def MLP_DTYPE(self):
        return bits2dtype_torch(self.MLP_NUM_DTYPE, "float")

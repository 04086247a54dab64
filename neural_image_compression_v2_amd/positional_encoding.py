"""nn.Module form of the triangular positional encoding (lookup table + gather) - the call surface of the
reference's Projects/positional_encoding.py:11-59.  The table is built and gathered by HIP kernels.

The reference's 2D module cannot run as committed (it passes 4 positional arguments to a function that needs
device and dtype, positional_encoding.py:55 vs utils.py:266); here it behaves like its functional twin
``triangular_positional_encoding_2d`` (utils.py:246-263), which is what the signature promises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import utils


class TriangularPositionalEncoding1D(nn.Module):
    """rows (1 freq, 0 offset), (2, 0), (2, .5), (4, 0), (4, .5), constant (positional_encoding.py:10-34)"""

    def __init__(self, sequence_length=8, octaves=3, include_constant=True, device=None):
        super().__init__()
        self.include_constant = include_constant
        self.octaves = octaves
        self.sequence_length = sequence_length
        dev = torch.device("cuda") if device is None else torch.device(device)
        self.register_buffer("encodings", utils.triangular_positional_encoding_1d(dev, torch.float32, sequence_length, octaves,
                                                                                  include_constant))

    def forward(self, coordinates):
        """[b, L] integer coordinates -> [b, rows, L] (positional_encoding.py:36-42)"""
        return utils.lut_gather(self.encodings, coordinates)


class TriangularPositionalEncoding2D(nn.Module):
    def __init__(self, sequence_length=8, octaves=3, include_constant=True, device=None):
        super().__init__()
        self.include_constant = include_constant
        self.octaves = octaves
        self.sequence_length = sequence_length
        self.encoding = TriangularPositionalEncoding1D(sequence_length, octaves, include_constant, device)

    def forward(self, coordinates, h, w, stride=1):
        """[b, 2] tile origins -> [b, 2*rows, h, w], x block then y block (positional_encoding.py:54-59)"""
        full_x, full_y = utils.convert_coordinate_start(coordinates, h, w, stride=stride)
        b = coordinates.shape[0]
        return torch.cat([self.encoding(full_x).view(b, -1, h, w), self.encoding(full_y).view(b, -1, h, w)], dim=1)

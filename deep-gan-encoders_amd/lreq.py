"""Equalised-learning-rate parameter holders (reference model/utils/lreq.py:39-156, implicit
lreq: weights are stored already scaled, and tagged with `lr_equalization_coef` which
LREQAdam reads, model/utils/custom_adam.py:71-72).  Forward math lives in the HIP kernels."""
import numpy as np
import torch
from torch import nn


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, gain=np.sqrt(2.0), lrmul=1.0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.std = gain / np.sqrt(in_features) * lrmul
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * (self.std / lrmul))
        setattr(self.weight, "lr_equalization_coef", self.std)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
            setattr(self.bias, "lr_equalization_coef", lrmul)
        else:
            self.register_parameter("bias", None)


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, gain=np.sqrt(2.0),
                 lrmul=1.0):
        super().__init__()
        if stride != 1 or padding != kernel_size // 2:
            raise ValueError("only stride-1 'same' convolutions are on the hot path")
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.std = gain / np.sqrt(kernel_size * kernel_size * in_channels)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size) * (self.std / lrmul))
        setattr(self.weight, "lr_equalization_coef", self.std)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
            setattr(self.bias, "lr_equalization_coef", lrmul)
        else:
            self.register_parameter("bias", None)

"""Layer-level modules of the inference StyleGAN2 (drop-in for maua/GAN/wrappers/inference/stylegan2.py:29-384:
FullyConnectedLayer, Conv2dLayer, SynthesisLayer, ToRGBLayer, SynthesisBlock) on the C-ABI operator layer (maua_amd.ops ->
maua_modconv2d / maua_bias_act / maua_upfirdn2d / maua_matmul_nt).  SynthesisNetwork (maua_amd.stylegan2) runs a whole
forward in ONE library call and is what the render path uses; these classes are the same arithmetic one layer at a time, for
callers that build or hook their own stacks.  Parameter names and shapes are the reference's, so a block's ``state_dict()``
is the ``bs.<i>.`` slice of a SynthesisNetwork's.  Activations are NCHW tensors on the HIP device (f32, or bf16 when given)."""
from math import sqrt

import numpy as np
import torch

from . import ops

_DEF_GAIN = {"linear": 1.0, "relu": sqrt(2.0), "lrelu": sqrt(2.0), "tanh": 1.0, "sigmoid": 1.0, "elu": 1.0, "selu": 1.0,
             "softplus": 1.0, "swish": sqrt(2.0)}      # ops.py:17-62 activation_funcs[...]["def_gain"]


class FullyConnectedLayer(torch.nn.Module):
    """inference/stylegan2.py:29-58"""

    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1.0, bias_init=0.0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight.detach() * self.weight_gain
        b = None if self.bias is None else self.bias.detach() * self.bias_gain
        lead = x.shape[:-1]
        # (:54-57: the "linear" branch is F.linear(x, w, b); every other activation multiplies by w.T - i.e. computes x @ w,
        #  which only works for square layers: quirk Q3, kept because the reference mapper runs through it)
        y = ops.matmul_nt(x.reshape(-1, self.in_features), w if self.activation == "linear" else w.T.contiguous())
        y = ops.bias_act(y.reshape(-1, self.out_features, 1, 1), b, act=self.activation).reshape(*lead, self.out_features)
        return y


class Conv2dLayer(torch.nn.Module):
    """inference/stylegan2.py:61-112 (down = 1: the render path never down-samples)"""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation="linear", up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, trainable=True):
        super().__init__()
        if down != 1:
            raise NotImplementedError("down-sampling is not on the render path")
        self.in_channels, self.out_channels, self.activation = in_channels, out_channels, activation
        self.up, self.down, self.conv_clamp = up, down, conv_clamp
        self.register_buffer("resample_filter", ops.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = _DEF_GAIN[activation]
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size])
        b = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(b) if b is not None else None
        else:
            self.register_buffer("weight", weight)
            if b is not None:
                self.register_buffer("bias", b)
            else:
                self.bias = None

    def forward(self, x, gain=1.0):
        w = self.weight.detach() * self.weight_gain
        if self.up == 2 and w.shape[-1] == 1:   # ops.py:201-205: a 1 x 1 kernel commutes with the up-sampling - convolve first
            x = ops.upsample2d(ops.conv2d_resample(x=x, w=w, padding=0), self.resample_filter, up=2)
        else:
            x = ops.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down, padding=self.padding)
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return ops.bias_act(x, None if self.bias is None else self.bias.detach(), act=self.activation,
                            gain=self.act_gain * gain, clamp=clamp)


class SynthesisLayer(torch.nn.Module):
    """inference/stylegan2.py:195-250: styles = affine(w) -> modulated_conv2d (+ noise_const) -> bias_act, the last two in
    one kernel launch (maua_modconv2d's fused epilogue)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation="lrelu",
                 resample_filter=[1, 3, 3, 1], conv_clamp=None):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.register_buffer("resample_filter", ops.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = _DEF_GAIN[activation]
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer("noise_const", torch.randn([resolution, resolution]))
        self.noise_adjusted = False
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode="const", gain=1.0):
        styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == "random":
            noise = torch.randn([x.shape[0], 1, x.shape[2] * self.up, x.shape[3] * self.up])
        if self.use_noise and noise_mode == "const":
            noise = self.noise_const
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return ops.modulated_conv2d(x=x, weight=self.weight.detach(), styles=styles, noise=noise, up=self.up, padding=self.padding,
                                    resample_filter=self.resample_filter, bias=self.bias.detach(), act=self.activation,
                                    gain=self.act_gain * gain, clamp=clamp)


class ToRGBLayer(torch.nn.Module):
    """inference/stylegan2.py:253-272"""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / sqrt(in_channels * (kernel_size ** 2))
        self.padding = kernel_size // 2

    def forward(self, x, w):
        styles = self.affine(w) * self.weight_gain
        return ops.modulated_conv2d(x=x, weight=self.weight.detach(), styles=styles, demodulate=False, padding=self.padding,
                                    bias=self.bias.detach(), clamp=self.conv_clamp)


class SynthesisBlock(torch.nn.Module):
    """inference/stylegan2.py:275-384 ("skip" - the reference networks - and "orig"; "resnet" needs the 1 x 1 up-sampling
    skip convolution as well, which runs through Conv2dLayer)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture="skip",
                 resample_filter=[1, 3, 3, 1], conv_clamp=256.0, use_fp16=False, **layer_kwargs):
        super().__init__()
        if architecture not in ("orig", "skip", "resnet"):
            raise ValueError(architecture)
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16 = is_last, architecture, use_fp16
        self.register_buffer("resample_filter", ops.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = 0
        self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution])) if in_channels == 0 else None
        self.conv0 = None
        if in_channels != 0:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                    **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == "skip":
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
            self.num_torgb += 1
        self.skip = None
        if in_channels != 0 and architecture == "resnet":
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter)

    def forward(self, x, img, ws, noise_mode="const"):
        w_idx = 0
        if self.in_channels == 0:
            x = self.const.detach().unsqueeze(0).expand(ws.shape[0], -1, -1, -1).contiguous()
            x = self.conv1(x, ws[:, w_idx], noise_mode, gain=1.0)
            w_idx += 1
        elif self.architecture == "resnet":
            y = self.skip(x, gain=sqrt(0.5))
            x = self.conv0(x, ws[:, w_idx], noise_mode, gain=1.0)
            x = self.conv1(x, ws[:, w_idx + 1], noise_mode, gain=sqrt(0.5))
            w_idx += 2
            x = ops.add(y, x)
        else:
            x = self.conv0(x, ws[:, w_idx], noise_mode, gain=1.0)
            x = self.conv1(x, ws[:, w_idx + 1], noise_mode, gain=1.0)
            w_idx += 2
        if img is not None:
            img = ops.upsample2d(img, self.resample_filter)
        if self.is_last or self.architecture == "skip":
            y = self.torgb(x, ws[:, w_idx]).float()
            img = ops.add(img, y) if img is not None else y
        return x, img

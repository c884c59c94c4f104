"""Guided diffusion on the HIP device (BASELINE configs[3]: "guided-diffusion 256x256, 100-step DDIM,
audio-onset-switched prompts").

Drop-in surface (same names, arguments and defaults):
  * ``create_models`` / ``GradientGuidedConditioning`` / ``GuidedDiffusion``  <- maua/diffusion/processors/guided.py:164-339
  * ``UNetModel``, ``SpacedDiffusion``, ``create_model_and_diffusion``, ``model_and_diffusion_defaults`` <- the slice of
    the un-vendored ``maua/submodules/guided_diffusion`` (unet.py, gaussian_diffusion.py, respace.py, script_util.py) that
    guided.py calls.  The submodule is EMPTY in the reference checkout (no pinned revision): the published algorithm is
    restated, **parity unpinned**; state-dict keys are upstream's, so a released checkpoint loads unchanged.
The network runs behind the C ABI (``maua_unet_*``, csrc/unet.hip); the sampler's arithmetic is ``maua_ddim_step`` /
``maua_axpby_rows``; schedules are float64 numpy on the host exactly like gaussian_diffusion.py builds them.

Conditioning speeds (guided.py:212-274): "fast" - the reference's DEFAULT (:287) - differentiates through the in-tree secondary
model (``SecondaryDiffusionImageNet2``, :68-143): forward and vector-Jacobian product run behind ``maua_secondary_*``
(csrc/secondary.hip evaluates the transposed network by hand; pinned by tests/golden/g28_secondary.npz, which the reference's own
classes generated); "hyper" (:248-249: the x0 estimate from the known noise, Jacobian 1 / alpha) is exact; "regular" (:250-252)
back-propagates through the 553 M-parameter UNet itself: ``UNetModel.forward_keep`` + ``UNetModel.vjp`` (maua_unet_forward_keep /
maua_unet_vjp - the network walked backwards in the library: every convolution's gradient is the same MFMA convolution with the
transposed kernel, GroupNorm / SiLU / resampling and attention have hand-written input-gradient kernels).  All three samplers of guided.py:302-311 exist ("ddim":
configs[3]; "p"; "plms": the fork's sampler restated from its published algorithm).

Grad modules: ``maua_amd.grad.CLIPGrads`` (round 6: maua/grad.py:96-165 - MauaCutouts, the CLIP image tower forward AND its input
gradient, spherical distance to the target embeddings - inside the library, also inside the captured guided loop) is what
configs[3]'s "text prompts" go through; the perceptor itself is an un-vendored pip dependency (published architecture restated,
**parity unpinned**; no weights in the image: benchmarks run it random-init like the UNet) and the text tower stays outside (target
embeddings are handed in).  ``MSEGuide`` (image targets) is the module that needs no perceptor.  Any other caller-supplied object with
the contract of maua/grad.py:15-25 (``scale``, ``set_targets(prompts)``, ``__call__(img, t) -> d loss / d img`` on the device) works
step by step; ColorMatch / VGG / LPIPS grads (:48-93, :167-199) are not built.
"""
import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L


# ------------------------------------------------------------------------------------------------------- network
def _channel_mult_for(image_size):
    """script_util.create_model"""
    table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
    if image_size not in table:
        raise ValueError(f"unsupported image size: {image_size}")
    return table[image_size]


class UNetModel(torch.nn.Module):
    """guided_diffusion.unet.UNetModel (constructor arguments as upstream; ``attention_resolutions`` are down-sampling
    rates).  Only the flag set guided.py:171-190 selects is implemented: use_scale_shift_norm, resblock_updown, legacy
    attention order, num_head_channels heads, no class conditioning, dropout 0.
    ``dtype``: torch.bfloat16 (default; the reference runs fp16) or torch.float32 (exact-f32 MFMA parity mode)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, dtype=torch.bfloat16, generator=None, vjp=False):
        super().__init__()
        self._vjp = bool(vjp)   # also prepare the transposed weights of the input gradient (forward_keep / vjp)
        if not (use_scale_shift_norm and resblock_updown) or use_new_attention_order or num_classes is not None or dims != 2 \
                or dropout != 0 or num_head_channels not in (32, 64):
            raise NotImplementedError("only the configuration of maua's create_models is implemented: use_scale_shift_norm, "
                                      "resblock_updown, legacy attention order, num_head_channels 32 / 64, unconditional")
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(int(a) for a in attention_resolutions)
        self.channel_mult, self.num_head_channels = tuple(channel_mult), num_head_channels
        self.dtype = dtype
        self._structure = self._build_structure()
        self._params = self._init_params(generator)
        self._net = None

    # -- structure: UNetModel.__init__'s module tree as layer descriptions -------------------------------------
    def _build_structure(self):
        mc, cm, nrb = self.model_channels, self.channel_mult, self.num_res_blocks
        ch = int(cm[0] * mc)
        inp, chans, ds = [[("conv", self.in_channels, ch)]], [ch], 1
        for level, mult in enumerate(cm):
            for _ in range(nrb):
                layers = [("res", ch, int(mult * mc))]
                ch = int(mult * mc)
                if ds in self.attention_resolutions:
                    layers.append(("attn", ch))
                inp.append(layers)
                chans.append(ch)
            if level != len(cm) - 1:
                inp.append([("res", ch, ch)])
                chans.append(ch)
                ds *= 2
        mid = [("res", ch, ch), ("attn", ch), ("res", ch, ch)]
        out = []
        for level, mult in list(enumerate(cm))[::-1]:
            for i in range(nrb + 1):
                ich = chans.pop()
                layers = [("res", ch + ich, int(mc * mult))]
                ch = int(mc * mult)
                if ds in self.attention_resolutions:
                    layers.append(("attn", ch))
                if level and i == nrb:
                    layers.append(("res", ch, ch))
                    ds //= 2
                out.append(layers)
        return dict(input=inp, middle=mid, output=out, final_ch=ch)

    def _param_shapes(self):
        emb = self.model_channels * 4
        shapes = {"time_embed.0.weight": (emb, self.model_channels), "time_embed.0.bias": (emb,),
                  "time_embed.2.weight": (emb, emb), "time_embed.2.bias": (emb,)}

        def layer(pfx, l):
            if l[0] == "conv":
                shapes[pfx + ".weight"], shapes[pfx + ".bias"] = (l[2], l[1], 3, 3), (l[2],)
            elif l[0] == "res":
                _, ci, co = l
                shapes[pfx + ".in_layers.0.weight"] = shapes[pfx + ".in_layers.0.bias"] = (ci,)
                shapes[pfx + ".in_layers.2.weight"], shapes[pfx + ".in_layers.2.bias"] = (co, ci, 3, 3), (co,)
                shapes[pfx + ".emb_layers.1.weight"], shapes[pfx + ".emb_layers.1.bias"] = (2 * co, emb), (2 * co,)
                shapes[pfx + ".out_layers.0.weight"] = shapes[pfx + ".out_layers.0.bias"] = (co,)
                shapes[pfx + ".out_layers.3.weight"], shapes[pfx + ".out_layers.3.bias"] = (co, co, 3, 3), (co,)
                if ci != co:
                    shapes[pfx + ".skip_connection.weight"], shapes[pfx + ".skip_connection.bias"] = (co, ci, 1, 1), (co,)
            else:
                c = l[1]
                shapes[pfx + ".norm.weight"] = shapes[pfx + ".norm.bias"] = (c,)
                shapes[pfx + ".qkv.weight"], shapes[pfx + ".qkv.bias"] = (3 * c, c, 1), (3 * c,)
                shapes[pfx + ".proj_out.weight"], shapes[pfx + ".proj_out.bias"] = (c, c, 1), (c,)
        s = self._structure
        for i, layers in enumerate(s["input"]):
            for j, l in enumerate(layers):
                layer(f"input_blocks.{i}.{j}", l)
        for j, l in enumerate(s["middle"]):
            layer(f"middle_block.{j}", l)
        for i, layers in enumerate(s["output"]):
            for j, l in enumerate(layers):
                layer(f"output_blocks.{i}.{j}", l)
        shapes["out.0.weight"] = shapes["out.0.bias"] = (s["final_ch"],)
        shapes["out.2.weight"], shapes["out.2.bias"] = (self.out_channels, s["final_ch"], 3, 3), (self.out_channels,)
        return shapes

    @L.host_threads(8)
    def _init_params(self, generator):
        """Random initialisation (there is no network access for the released checkpoints): fan-in scaled normal weights,
        GroupNorm scales around 1.  Upstream's ``zero_module`` layers (every ResBlock's last conv, every proj_out, the
        output conv) are drawn like the others - a zero-initialised network would make every sample a no-op."""
        g = generator or torch.Generator().manual_seed(0)
        p = {}
        for name, shape in self._param_shapes().items():
            if ".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name or name.startswith("out.0."):
                p[name] = (1 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.1 * torch.randn(shape, generator=g)
            elif name.endswith(".bias"):
                p[name] = 0.1 * torch.randn(shape, generator=g)
            else:
                fan_in = int(np.prod(shape[1:]))
                p[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        return p

    # -- parameters ------------------------------------------------------------------------------------------
    def state_dict(self, *a, **k):
        return dict(self._params)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise KeyError(f"missing {missing[:4]}..., unexpected {unexpected[:4]}...")
        for k in self._params:
            if k in sd:
                if tuple(sd[k].shape) != tuple(self._params[k].shape):
                    raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(self._params[k].shape)}")
                self._params[k] = sd[k].detach().float().cpu().contiguous()
        self._destroy()

    def convert_to_fp16(self):
        """guided.py:198-199 calls this; the compute type here is the constructor's ``dtype``."""
        return self

    def requires_grad_(self, flag=True):
        return self

    def _destroy(self):
        if self._net is not None:
            L.lib().maua_unet_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        L.require_device()
        if self._net is None:
            lib = L.lib()
            net = C.c_void_p()
            cm = (C.c_float * len(self.channel_mult))(*[float(m) for m in self.channel_mult])
            ads = (C.c_int * max(1, len(self.attention_resolutions)))(*self.attention_resolutions)
            L.check(lib.maua_unet_create(L.ctx(), self.image_size, self.in_channels, self.model_channels, self.out_channels,
                                         self.num_res_blocks, cm, len(self.channel_mult), ads,
                                         len(self.attention_resolutions), self.num_head_channels, L.dtype_id(self.dtype),
                                         C.byref(net)))
            if self._vjp:
                L.check(lib.maua_unet_set_option(net, b"vjp", 1))
            for k, v in self._params.items():
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(lib.maua_unet_load(net, k.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
            # nn.py timestep_embedding's frequency table, computed on the host with the reference's own expression
            half = self.model_channels // 2
            freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).numpy()
            L.check(lib.maua_unet_load(net, b"timestep_embedding.freqs", freqs.ctypes.data_as(C.c_void_p), C.c_size_t(half)))
            self._net = net
        else:
            L.ctx()
        return self._net

    def graph_active(self):
        """True when the in-library sampler loop replays a captured hipGraph (False: capture unavailable, eager launches)."""
        a = C.c_int()
        L.check(L.lib().maua_unet_graph_active(self._handle(), C.byref(a)))
        return bool(a.value)

    def guided_graph_active(self):
        """... the same for the guided loop (maua_ddim_guided_loop)."""
        a = C.c_int()
        L.check(L.lib().maua_unet_guided_graph_active(self._handle(), C.byref(a)))
        return bool(a.value)

    def set_option(self, key, value):
        """Library options of the network object ("route", "psum_off": see maua_unet_set_option)."""
        L.check(L.lib().maua_unet_set_option(self._handle(), key.encode(), int(value)))

    def set_route(self, route):
        """0: per-shape kernel routing (default), 1: generic 3x3 kernel only, 2: no split-K gather GEMM."""
        L.check(L.lib().maua_unet_set_option(self._handle(), b"route", int(route)))

    def forward(self, x, timesteps, y=None, out=None, keep=False):
        """x [N, C, H, W], timesteps [N] -> [N, out_channels, H, W] (float32).  ``keep``: leave what ``vjp`` reads on the device."""
        if y is not None:
            raise NotImplementedError("class conditioning is not on this path")
        x = L.dev_tensor(x, torch.float32)
        t = L.dev_tensor(torch.as_tensor(timesteps), torch.float32).reshape(-1)
        B, _, H, W = x.shape
        if t.numel() != B:
            raise ValueError("one timestep per sample")
        if out is None:
            out = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        if keep:
            self.enable_vjp()
            L.check(L.lib().maua_unet_forward_keep(self._handle(), L.ptr(x), L.ptr(t), B, H, W, L.ptr(out)))
            self._kept = (B, H, W)
        else:
            L.check(L.lib().maua_unet_forward(self._handle(), L.ptr(x), L.ptr(t), B, H, W, L.ptr(out)))
            self._kept = None
        return out

    def enable_vjp(self):
        """Prepare the network for ``forward_keep`` / ``vjp`` (the library object is rebuilt once with the transposed weights)."""
        if not self._vjp:
            self._vjp = True
            self._destroy()
        return self

    def forward_keep(self, x, timesteps, out=None):
        """``forward`` that keeps the GroupNorm inputs / statistics and the attention operands for ``vjp`` (back to back)."""
        return self.forward(x, timesteps, out=out, keep=True)

    def vjp(self, g_out):
        """(d out / d x)^T g_out of the last ``forward_keep``: what ``torch.autograd.grad(out, x, g_out)`` returns in the reference
        (guided.py:268 for speed "regular").  g_out [N, out_channels, H, W] -> [N, in_channels, H, W] float32."""
        kept = getattr(self, "_kept", None)
        if kept is None:
            raise RuntimeError("vjp: call forward_keep first (no other forward in between)")
        g = L.dev_tensor(g_out, torch.float32).contiguous()
        B, H, W = kept
        if tuple(g.shape) != (B, self.out_channels, H, W):
            raise ValueError(f"vjp: g_out must be {(B, self.out_channels, H, W)}, got {tuple(g.shape)}")
        gx = torch.empty((B, self.in_channels, H, W), dtype=torch.float32, device=g.device)
        L.check(L.lib().maua_unet_vjp(self._handle(), L.ptr(g), B, H, W, L.ptr(gx)))
        return gx


# ------------------------------------------------------------------------------------------------------ diffusion
def space_timesteps(num_timesteps, section_counts):
    """respace.space_timesteps"""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = num_timesteps // len(section_counts), num_timesteps % len(section_counts)
    start_idx, all_steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac_stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            all_steps.append(start_idx + round(cur))
            cur += frac_stride
        start_idx += size
    return set(all_steps)


class SpacedDiffusion:
    """respace.SpacedDiffusion over gaussian_diffusion.GaussianDiffusion (linear schedule, epsilon prediction, learned-range
    variance): float64 tables of the respaced process, ``timestep_map``, ``q_sample`` and ``ddim_sample``."""

    def __init__(self, use_timesteps, betas, rescale_timesteps=True):
        base = np.array(betas, dtype=np.float64)
        self.original_num_steps, self.rescale_timesteps = len(base), rescale_timesteps
        self.use_timesteps = set(use_timesteps)
        ac, last, new_betas, self.timestep_map = np.cumprod(1.0 - base, axis=0), 1.0, [], []
        for i, a in enumerate(ac):
            if i in self.use_timesteps:
                new_betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        self.betas = b = np.array(new_betas, dtype=np.float64)
        self.num_timesteps = int(b.shape[0])
        self.alphas_cumprod = np.cumprod(1.0 - b, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        # q(x_{t-1} | x_t, x_0) (p_sample)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:])) \
            if len(b) > 1 else np.log(np.maximum(self.posterior_variance, 1e-20))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(1.0 - b) / (1.0 - self.alphas_cumprod)

    # -- helpers ---------------------------------------------------------------------------------------------
    def model_timesteps(self, t):
        """_WrappedModel.__call__: respaced index -> the timestep the network sees (float when rescaled)."""
        ts = torch.tensor(self.timestep_map, device=t.device)[t.long()]
        return ts.float() * (1000.0 / self.original_num_steps) if self.rescale_timesteps else ts.float()

    @staticmethod
    def _model_output(model, x, mt, cond_fn):
        """The step's model evaluation.  A conditioning that differentiates through this very network (speed "regular") evaluates
        it again on the same (x, t) - guided.py:251 inside cond_fn, after the sampler's own call: here it runs ONCE, as a kept forward
        whose result is offered to the conditioning (same kernels, same bits as two separate evaluations)."""
        if cond_fn is not None and hasattr(cond_fn, "offer") and getattr(cond_fn, "model", None) is model \
                and getattr(cond_fn, "speed", "fast") not in ("hyper", "fast"):
            out = model.forward_keep(x, mt)
            cond_fn.offer(x, mt, out)
            return out
        return model(x, mt)

    def _f32(self, arr, t):
        """_extract_into_tensor: the table values at t as float32 (host)."""
        return torch.from_numpy(arr)[torch.as_tensor(t).long().cpu()].float()

    def step_coefficients(self, t, eta=0.0):
        """[len(t), 8] float32 coefficients of maua_ddim_step, evaluated like ddim_sample evaluates them (float32 tensors
        extracted from the float64 tables, then float32 arithmetic)."""
        t = torch.as_tensor(t).long().cpu().reshape(-1)
        ab, abp = self._f32(self.alphas_cumprod, t), self._f32(self.alphas_cumprod_prev, t)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        cf = torch.zeros((len(t), 8), dtype=torch.float32)
        cf[:, 0] = self._f32(self.sqrt_recip_alphas_cumprod, t)
        cf[:, 1] = self._f32(self.sqrt_recipm1_alphas_cumprod, t)
        cf[:, 2] = (1 - ab).sqrt()
        cf[:, 3] = torch.sqrt(abp)
        cf[:, 4] = torch.sqrt(1 - abp - sigma ** 2)
        cf[:, 5] = sigma * (t != 0).float()
        return cf

    # -- gaussian_diffusion.py ---------------------------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        x = L.dev_tensor(x_start, torch.float32)
        noise = torch.randn_like(x) if noise is None else L.dev_tensor(noise, torch.float32)
        ab = torch.stack([self._f32(self.sqrt_alphas_cumprod, t), self._f32(self.sqrt_one_minus_alphas_cumprod, t)], 1)
        ab = L.dev_tensor(ab.contiguous(), torch.float32)
        out = torch.empty_like(x)
        L.check(L.lib().maua_axpby_rows(L.ctx(x.device), L.ptr(x), L.ptr(noise), L.ptr(ab), x.shape[0],
                                        C.c_long(x[0].numel()), L.ptr(out)))
        return out

    def ddim_sample(self, model, x, t, clip_denoised=False, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0,
                    noise=None):
        """One DDIM step -> {"sample", "pred_xstart"}; ``cond_fn(x, model_timesteps)`` -> gradient (condition_score)."""
        if clip_denoised or denoised_fn is not None:
            raise NotImplementedError("clip_denoised / denoised_fn (the reference passes clip_denoised=False)")
        x = L.dev_tensor(x, torch.float32)
        mt = self.model_timesteps(torch.as_tensor(t).to(x.device))
        out = self._model_output(model, x, mt, cond_fn)
        grad = None
        if cond_fn is not None:
            grad = L.dev_tensor(cond_fn(x, mt), torch.float32)
        if eta != 0.0 and noise is None:
            noise = torch.randn_like(x)
        if noise is not None:   # a caller's tensor: onto the device, f32, contiguous, and of the sample's shape (the kernel takes a raw pointer)
            noise = L.dev_tensor(noise, torch.float32)
            if tuple(noise.shape) != tuple(x.shape):
                raise ValueError(f"ddim_sample: noise shape {tuple(noise.shape)} != sample shape {tuple(x.shape)}")
        cf = L.dev_tensor(self.step_coefficients(t, eta), torch.float32)
        sample, pred = torch.empty_like(x), torch.empty_like(x)
        B, Cc = x.shape[0], x.shape[1]
        L.check(L.lib().maua_ddim_step(L.ctx(x.device), L.ptr(x), L.ptr(out), L.ptr(grad),
                                       L.ptr(noise if eta != 0.0 else None), L.ptr(cf), B, Cc, out.shape[1],
                                       C.c_long(x[0, 0].numel()), L.ptr(sample), L.ptr(pred)))
        return {"sample": sample, "pred_xstart": pred}

    def p_sample(self, model, x, t, clip_denoised=False, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """gaussian_diffusion.py p_sample (guided.py:302-303): one ancestral step with the learned-range variance ->
        {"sample", "pred_xstart"}; ``cond_fn``: condition_mean (mean + variance * gradient)."""
        if clip_denoised or denoised_fn is not None:
            raise NotImplementedError("clip_denoised / denoised_fn (the reference passes clip_denoised=False)")
        x = L.dev_tensor(x, torch.float32)
        mt = self.model_timesteps(torch.as_tensor(t).to(x.device))
        out = self._model_output(model, x, mt, cond_fn)
        B, Cc = x.shape[0], x.shape[1]
        if out.shape[1] != 2 * Cc:
            raise NotImplementedError("p_sample needs the learned-range variance channels (learn_sigma)")
        grad = None if cond_fn is None else L.dev_tensor(cond_fn(x, mt), torch.float32)
        noise = torch.randn_like(x) if noise is None else L.dev_tensor(noise, torch.float32)
        tt = torch.as_tensor(t).long().cpu().reshape(-1)
        cf = torch.zeros((len(tt), 8), dtype=torch.float32)
        cf[:, 0] = self._f32(self.sqrt_recip_alphas_cumprod, tt)
        cf[:, 1] = self._f32(self.sqrt_recipm1_alphas_cumprod, tt)
        cf[:, 2] = self._f32(self.posterior_mean_coef1, tt)
        cf[:, 3] = self._f32(self.posterior_mean_coef2, tt)
        cf[:, 4] = self._f32(self.posterior_log_variance_clipped, tt)
        cf[:, 5] = self._f32(np.log(self.betas), tt)
        cf[:, 6] = (tt != 0).float()
        cf = L.dev_tensor(cf, torch.float32)
        sample, pred = torch.empty_like(x), torch.empty_like(x)
        L.check(L.lib().maua_p_sample_step(L.ctx(x.device), L.ptr(x), L.ptr(out), L.ptr(grad), L.ptr(noise), L.ptr(cf), B, Cc,
                                           C.c_long(x[0, 0].numel()), L.ptr(sample), L.ptr(pred)))
        return {"sample": sample, "pred_xstart": pred}

    def _plms_model_output(self, model, x, t, cond_fn):
        """plms_sample's get_model_output -> (eps, pred_xstart after condition_score, unconditioned pred_xstart)"""
        mt = self.model_timesteps(torch.as_tensor(t).to(x.device))
        out = self._model_output(model, x, mt, cond_fn)
        grad = None if cond_fn is None else L.dev_tensor(cond_fn(x, mt), torch.float32)
        cf = L.dev_tensor(self.step_coefficients(t), torch.float32)
        eps, pred, pred_orig = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        L.check(L.lib().maua_plms_eps(L.ctx(x.device), L.ptr(x), L.ptr(out), L.ptr(grad), L.ptr(cf), x.shape[0], x.shape[1],
                                      out.shape[1], C.c_long(x[0, 0].numel()), L.ptr(eps), L.ptr(pred), L.ptr(pred_orig)))
        return eps, pred, pred_orig

    def _plms_update(self, x, t, eps_list, weights, divisor, pred):
        tt = torch.as_tensor(t).long().cpu().reshape(-1)
        abp = self._f32(self.alphas_cumprod_prev, tt)
        cf = torch.zeros((len(tt), 8), dtype=torch.float32)
        cf[:, 0] = self._f32(self.sqrt_recip_alphas_cumprod, tt)
        cf[:, 1] = self._f32(self.sqrt_recipm1_alphas_cumprod, tt)
        cf[:, 2] = torch.sqrt(abp)
        cf[:, 3] = torch.sqrt(1 - abp)
        cf[:, 4] = (tt != 0).float()
        cf = L.dev_tensor(cf, torch.float32)
        ptrs = (C.c_void_p * len(eps_list))(*[e.data_ptr() for e in eps_list])
        ws = (C.c_float * len(weights))(*weights)
        sample = torch.empty_like(x)
        L.check(L.lib().maua_plms_update(L.ctx(x.device), L.ptr(x), ptrs, ws, len(eps_list), C.c_float(divisor), L.ptr(pred),
                                         L.ptr(cf), x.shape[0], C.c_long(x[0].numel()), L.ptr(sample)))
        return sample

    def plms_sample(self, model, x, t, clip_denoised=False, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    cond_fn_with_grad=False, order=2, old_out=None):
        """plms_sample of the guided-diffusion fork behind maua's submodule (guided.py:308-311; un-vendored: the published
        pseudo linear multistep sampler, Liu et al. 2022, as that fork states it): the first call of an order > 1 run takes a
        pseudo improved-Euler step (a second model evaluation at t - 1), later calls the Adams-Bashforth combination of the
        last ``order`` epsilons.  -> {"sample", "pred_xstart" (unconditioned), "old_eps"}; pass the result back as ``old_out``."""
        if clip_denoised or denoised_fn is not None or cond_fn_with_grad:
            raise NotImplementedError("clip_denoised / denoised_fn / cond_fn_with_grad")
        if not int(order) or not 1 <= order <= 4:
            raise ValueError("order is invalid (should be int from 1-4).")
        x = L.dev_tensor(x, torch.float32)
        t = torch.as_tensor(t).long()
        eps, pred, pred_orig = self._plms_model_output(model, x, t, cond_fn)
        if order > 1 and old_out is None:
            old_eps = [eps]
            # mean_pred = pred * sqrt(ac_prev) + sqrt(1 - ac_prev) * eps: the update with eps' = eps - but on `pred`, not on a
            # prediction re-derived from eps (the same value up to rounding; the reference uses out["pred_xstart"] here)
            tt = t.cpu().reshape(-1)
            abp = self._f32(self.alphas_cumprod_prev, tt)
            ab2 = L.dev_tensor(torch.stack([torch.sqrt(abp), torch.sqrt(1 - abp)], 1).contiguous(), torch.float32)
            mean_pred = torch.empty_like(x)
            L.check(L.lib().maua_axpby_rows(L.ctx(x.device), L.ptr(pred), L.ptr(eps), L.ptr(ab2), x.shape[0],
                                            C.c_long(x[0].numel()), L.ptr(mean_pred)))
            eps_2, _, _ = self._plms_model_output(model, mean_pred, t - 1, cond_fn)
            sample = self._plms_update(x, t, [eps, eps_2], [1.0, 1.0], 2.0, pred)
        else:
            old_eps = [] if old_out is None else list(old_out["old_eps"])
            old_eps.append(eps)
            cur_order = min(order, len(old_eps))
            newest_first = old_eps[::-1][:cur_order]
            weights, div = {1: ([1.0], 1.0), 2: ([3.0, -1.0], 2.0), 3: ([23.0, -16.0, 5.0], 12.0),
                            4: ([55.0, -59.0, 37.0, -9.0], 24.0)}[cur_order]
            sample = self._plms_update(x, t, newest_first, weights, div, pred)
        if len(old_eps) >= order:
            old_eps.pop(0)
        return {"sample": sample, "pred_xstart": pred_orig, "old_eps": old_eps}

    def ddim_sample_loop(self, model, x, start_step=None, n_steps=None, use_graph=True):
        """The unconditioned loop inside the library (one hipGraph per shape): t = start_step, start_step - 1, ...
        -> (x after the last step, its pred_xstart).  x is updated in place."""
        start = self.num_timesteps - 1 if start_step is None else int(start_step)
        n_steps = start + 1 if n_steps is None else int(n_steps)
        ts = torch.arange(start, start - n_steps, -1)
        ts = torch.where(ts < 0, ts + self.num_timesteps, ts)  # (negative t index the tables from the end, like numpy)
        mt = np.ascontiguousarray(self.model_timesteps(ts).numpy(), dtype=np.float32)
        cf = np.ascontiguousarray(self.step_coefficients(ts).numpy(), dtype=np.float32)
        x = L.dev_tensor(x, torch.float32)
        pred = torch.empty_like(x)
        B, _, H, W = x.shape
        L.check(L.lib().maua_ddim_sample_loop(model._handle(), L.ptr(x), B, H, W, mt.ctypes.data_as(C.c_void_p),
                                              cf.ctypes.data_as(C.c_void_p), n_steps, int(bool(use_graph)), L.ptr(pred)))
        return x, pred

    def ddim_guided_loop(self, model, conditioning, x, start_step=None, n_steps=None, use_graph=True):
        """The GUIDED loop inside the library (guided.py:302-311, 333-337 with cond_fn = ``conditioning``: speed "fast", one
        image-MSE grad module whose target is set): UNet forward, secondary forward, grad module, secondary VJP and the DDIM
        update of all steps as one hipGraph.  Same arithmetic, same kernels and the same host-evaluated coefficients as the
        step-by-step path (``ddim_sample`` + ``GradientGuidedConditioning.forward``), so the results are identical.
        -> (x after the last step, its pred_xstart); x is updated in place."""
        start = self.num_timesteps - 1 if start_step is None else int(start_step)
        n_steps = start + 1 if n_steps is None else int(n_steps)
        ts = torch.arange(start, start - n_steps, -1)
        ts = torch.where(ts < 0, ts + self.num_timesteps, ts)
        mtt = self.model_timesteps(ts)
        mt = np.ascontiguousarray(mtt.numpy(), dtype=np.float32)
        cf = np.ascontiguousarray(self.step_coefficients(ts).numpy(), dtype=np.float32)
        gc = np.ascontiguousarray(conditioning.guide_coefficients(mtt).numpy(), dtype=np.float32)
        gm = conditioning.grad_modules[0]
        x = L.dev_tensor(x, torch.float32)
        B, _, H, W = x.shape
        spec = gm.graph_spec() if hasattr(gm, "graph_spec") else None
        # the other modules of the list as library objects (VGGGrads, LPIPSGrads, ColorMatchGrads), evaluated and summed per step
        guides = [m.graph_guide(B, H, W) for m in conditioning.grad_modules[(1 if spec is not None else 0):] if hasattr(m, "graph_guide")]
        if any(g is None for g in guides):
            raise ValueError("ddim_guided_loop: a grad module has no target yet (set_targets first)")
        L.check(L.lib().maua_unet_set_guides(model._handle(), (C.c_void_p * max(len(guides), 1))(*[g.h for g in guides]), len(guides)))
        if spec is not None:
            # text-prompt guidance (CLIPGrads, maua/grad.py:96-165): the cutout rectangles of every step and cutout batch are drawn
            # here, in the order the step-by-step path draws them (per step: grad.py:149's t[[0]] = that step's model timestep)
            gm.install_targets(0, B)
            rects, mult = np.stack([gm.draw_rects(0, H, W, mtt[s:s + 1]) for s in range(n_steps)]), None
            if gm.merge_cutouts:   # (identical rectangles of a cutout batch - the whole-image cutouts of a square image - pass once)
                rects, mult = gm.merge_identical(rects)
            rects = np.ascontiguousarray(rects, dtype=np.int32)
            L.check(L.lib().maua_unet_set_clip_guide(model._handle(), spec["clip"]._handle(), rects.ctypes.data_as(C.c_void_p),
                                                     None if mult is None else mult.ctypes.data_as(C.c_void_p), n_steps,
                                                     rects.shape[2], spec["batches"], C.c_float(spec["scale"]), C.c_float(spec["clamp"])))
            tgt, tstride, mse_k = None, 0, 0.0
            # a text-guided step is ~2 500 launches (8 cutout batches x ~200 for the tower's forward + backward): 250 000 for a
            # 100-step loop.  The loop still runs inside the library in one call, but launch by launch - the launches' host cost
            # (~12 ms) is 4 % of such a step, and capturing a graph of that size faulted inside the HIP runtime (ROCm 7.0.2); short
            # loops (tests, previews) are captured
            if n_steps * spec["batches"] > 64:
                use_graph = False
        elif guides:
            L.check(L.lib().maua_unet_set_clip_guide(model._handle(), None, None, None, 0, 0, 0, C.c_float(0.0), C.c_float(0.0)))
            tgt, tstride, mse_k = None, 0, 0.0
        else:
            L.check(L.lib().maua_unet_set_clip_guide(model._handle(), None, None, None, 0, 0, 0, C.c_float(0.0), C.c_float(0.0)))
            tgt = L.dev_tensor(gm.target, torch.float32)
            mse_k = gm.factor(x[0].numel())
            if tuple(tgt.shape) == tuple(x.shape[1:]):
                tstride = 0
            elif tuple(tgt.shape) == tuple(x.shape):
                tstride = x[0].numel()
            else:
                raise ValueError(f"ddim_guided_loop: target shape {tuple(tgt.shape)} fits neither {tuple(x.shape[1:])} nor {tuple(x.shape)}")
        pred = torch.empty_like(x)
        if conditioning.speed == "fast":
            second = conditioning.model._handle()
        else:   # "regular": the gradient goes through the UNet itself (maua_ddim_guided_loop without a secondary model)
            if conditioning.model is not model:
                raise ValueError('speed="regular" differentiates through the network that is being sampled')
            model.enable_vjp()
            second = None
        L.check(L.lib().maua_ddim_guided_loop(model._handle(), second, L.ptr(x), B, H, W,
                                              mt.ctypes.data_as(C.c_void_p), cf.ctypes.data_as(C.c_void_p),
                                              gc.ctypes.data_as(C.c_void_p), n_steps, L.ptr(tgt), C.c_long(tstride),
                                              C.c_float(mse_k), int(bool(use_graph)), L.ptr(pred)))
        return x, pred


def model_and_diffusion_defaults():
    """script_util.model_and_diffusion_defaults (the keys guided.py:171-190 reads or overrides)."""
    return dict(image_size=64, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1, num_head_channels=-1,
                attention_resolutions="16,8", channel_mult="", dropout=0.0, class_cond=False, use_checkpoint=False,
                use_scale_shift_norm=True, resblock_updown=False, use_fp16=False, use_new_attention_order=False,
                learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
                predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks, channel_mult, num_heads,
                               num_head_channels, num_heads_upsample, attention_resolutions, dropout, diffusion_steps,
                               noise_schedule, timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, resblock_updown, use_fp16,
                               use_new_attention_order, dtype=torch.bfloat16, generator=None):
    """script_util.create_model_and_diffusion"""
    if class_cond or predict_xstart or use_kl or noise_schedule != "linear":
        raise NotImplementedError("unconditional epsilon model on the linear schedule (what create_models configures)")
    cm = _channel_mult_for(image_size) if channel_mult == "" else tuple(float(m) for m in str(channel_mult).split(","))
    ads = tuple(image_size // int(r) for r in str(attention_resolutions).split(","))
    model = UNetModel(image_size=image_size, in_channels=3, model_channels=num_channels,
                      out_channels=6 if learn_sigma else 3, num_res_blocks=num_res_blocks, attention_resolutions=ads,
                      dropout=dropout, channel_mult=cm, num_heads=num_heads, num_head_channels=num_head_channels,
                      num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                      resblock_updown=resblock_updown, use_fp16=use_fp16, use_new_attention_order=use_new_attention_order,
                      dtype=dtype, generator=generator)
    scale = 1000 / diffusion_steps
    betas = np.linspace(scale * 0.0001, scale * 0.02, diffusion_steps, dtype=np.float64)
    diffusion = SpacedDiffusion(space_timesteps(diffusion_steps, timestep_respacing or [diffusion_steps]), betas,
                                rescale_timesteps=rescale_timesteps)
    return model, diffusion


CHECKPOINTS = {"uncondImageNet512": ("modelzoo/512x512_diffusion_uncond_finetune_008100.pt", 512),
               "uncondImageNet256": ("modelzoo/256x256_diffusion_uncond.pt", 256)}


def create_models(checkpoint="uncondImageNet512", timestep_respacing="100", diffusion_steps=1000, use_secondary=False,
                  allow_random_init=False, dtype=torch.bfloat16, generator=None, secondary_dtype=torch.float32, secondary_exact=False,
                  **overrides):
    """guided.py:164-209.  The checkpoint file is loaded when it exists (upstream state-dict keys); there is no network
    access to download it: without the file this raises unless ``allow_random_init`` (seeded random weights of the same
    architecture - what the bench and the tests run).  ``overrides``: model_config entries (tests build small networks)."""
    import os
    path, size = CHECKPOINTS[checkpoint]
    cfg = model_and_diffusion_defaults()
    cfg.update({"attention_resolutions": "32, 16, 8", "class_cond": False, "diffusion_steps": diffusion_steps,
                "rescale_timesteps": True, "timestep_respacing": timestep_respacing, "learn_sigma": True,
                "noise_schedule": "linear", "num_channels": 256, "num_head_channels": 64, "num_res_blocks": 2,
                "resblock_updown": True, "use_fp16": True, "use_scale_shift_norm": True, "image_size": size})
    cfg.update(overrides)
    model, diffusion = create_model_and_diffusion(**cfg, dtype=dtype, generator=generator)
    if os.path.exists(path):
        model.load_state_dict(torch.load(path, map_location="cpu"))
    elif not allow_random_init:
        raise FileNotFoundError(f"{path} not found (the reference downloads it; this box has no network): place the file "
                                "there or pass allow_random_init=True")
    secondary = None
    if use_secondary:   # guided.py:198-205
        spath = "modelzoo/secondary_model_imagenet_2.pth"
        # (the reference keeps this 13.9 M-parameter model in fp32, guided.py:198-205, whatever the UNet runs in: float32 tensors here
        #  too; by default its products run as bf16 split products, ~2^-17 each - the guidance gradient stays within 1e-3 of the
        #  reference's float32 autograd at a quarter of the matrix-core cycles; secondary_exact=True: the exact-f32 matrix path;
        #  secondary_dtype=torch.bfloat16 is an opt-in, 3.5 % off in L2 norm)
        secondary = SecondaryDiffusionImageNet2(dtype=secondary_dtype, generator=generator, exact=secondary_exact)
        if os.path.exists(spath):
            secondary.load_state_dict(torch.load(spath, map_location="cpu"))
        elif not allow_random_init:
            raise FileNotFoundError(f"{spath} not found (the reference downloads it; this box has no network): place the file "
                                    "there or pass allow_random_init=True")
    return model, diffusion, secondary


def secondary_conv_keys():
    """State-dict key prefixes of SecondaryDiffusionImageNet2's 24 convolutions in execution order (what torch names the nested
    Sequential / SkipBlock modules of guided.py:77-134), index = the C ABI's convolution number."""
    keys = ["net.0.0", "net.1.0"]
    pfx = "net.2.main"
    for _ in range(4):
        keys += [pfx + ".1.0", pfx + ".2.0"]
        pfx += ".3.main"
    keys += [pfx + f".{i}.0" for i in (1, 2, 3, 4)]
    for _ in range(4):
        pfx = pfx[:-len(".3.main")]
        keys += [pfx + ".4.0", pfx + ".5.0"]
    return keys + ["net.3.0", "net.4"]


class DiffusionOutput:
    """guided.py:33-37"""

    def __init__(self, v, pred, eps):
        self.v, self.pred, self.eps = v, pred, eps


class SecondaryDiffusionImageNet2(torch.nn.Module):
    """guided.py:68-143 on the device: ``forward(input, t) -> DiffusionOutput(v, pred, eps)`` and - what the "fast" conditioning
    needs instead of autograd - ``vjp(g_v) -> (d v / d input)^T g_v`` for the input of the LAST forward.  State-dict keys are the
    reference's (``timestep_embed.weight``, ``net.<...>.weight / .bias``), so its checkpoint loads unchanged.  ``dtype``:
    torch.bfloat16 (default) or torch.float32 (the reference keeps this model in fp32).  ``exact`` (float32 only): True = every
    product on the exact-f32 matrix path (v_mfma_f32_32x32x2_f32); False = MAUA_F32_SPLIT: float32 tensors, each product as three
    bf16 split products on the bf16 matrix cores (x = hi + lo; ~2^-17 per product, f32 accumulate - finer than the TF32 (2^-11) a CUDA
    device gives the reference's fp32 convolutions by default), 4 x fewer matrix-core cycles."""

    def __init__(self, dtype=torch.bfloat16, generator=None, exact=True):
        super().__init__()
        self.dtype = dtype
        self.exact = bool(exact) or dtype != torch.float32
        self._keys = secondary_conv_keys()
        g = generator if generator is not None else torch.Generator().manual_seed(0)
        ci, co = C.c_int(), C.c_int()
        self._params = {"timestep_embed.weight": torch.randn(8, 1, generator=g)}
        for i, k in enumerate(self._keys):   # torch's Conv2d default scale (kaiming-uniform bound 1 / sqrt(fan_in)) as a normal
            L.check(L.lib().maua_secondary_conv_shape(i, C.byref(ci), C.byref(co)))
            self._params[k + ".weight"] = torch.randn(co.value, ci.value, 3, 3, generator=g) / math.sqrt(3.0 * 9 * ci.value)
            self._params[k + ".bias"] = torch.randn(co.value, generator=g) / math.sqrt(3.0 * 9 * ci.value)
        self._net = None
        self._last = None   # (B, H, W) of the last forward: what vjp differentiates

    def state_dict(self, *a, **k):
        return dict(self._params)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._params if k not in sd]
        extra = [k for k in sd if k not in self._params]
        if strict and (missing or extra):
            raise RuntimeError(f"SecondaryDiffusionImageNet2.load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        for k, v in sd.items():
            if k in self._params:
                v = torch.as_tensor(v).detach().float().cpu()
                if tuple(v.shape) != tuple(self._params[k].shape):
                    raise RuntimeError(f"{k}: shape {tuple(v.shape)} != {tuple(self._params[k].shape)}")
                self._params[k] = v.contiguous()
        self._destroy()
        self._last = None
        return torch.nn.modules.module._IncompatibleKeys(missing, extra)

    def eval(self):
        return self

    def requires_grad_(self, flag=True):
        return self

    def to(self, *a, **k):
        return self

    def _destroy(self):
        if self._net is not None:
            L.lib().maua_secondary_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        if self._net is None:
            net = C.c_void_p()
            L.check(L.lib().maua_secondary_create(L.ctx("cuda"), L.dtype_id(self.dtype) if self.exact else L.F32_SPLIT, C.byref(net)))
            lib = L.lib()

            def up(i, what, t):
                t = t.contiguous().float()
                L.check(lib.maua_secondary_load(net, i, what, C.c_void_p(t.data_ptr()), C.c_size_t(t.numel())))
            up(0, 2, self._params["timestep_embed.weight"].reshape(-1))
            for i, k in enumerate(self._keys):
                up(i, 0, self._params[k + ".weight"])
                up(i, 1, self._params[k + ".bias"])
            self._net = net
        return self._net

    def forward(self, input, t):
        x = L.dev_tensor(input, torch.float32)
        t = L.dev_tensor(torch.as_tensor(t), torch.float32).reshape(-1)
        B, _, H, W = x.shape
        if t.numel() != B:
            raise ValueError("one timestep per sample")
        v, pred, eps = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        net = self._handle()
        L.ctx(x.device)
        L.check(L.lib().maua_secondary_forward(net, L.ptr(x), L.ptr(t), B, H, W, L.ptr(v), L.ptr(pred), L.ptr(eps)))
        self._last = (B, H, W)
        return DiffusionOutput(v, pred, eps)

    def vjp(self, g_v):
        """(d v / d input)^T g_v for the input of the last ``forward``."""
        if self._last is None:
            raise RuntimeError("SecondaryDiffusionImageNet2.vjp: call forward first (the product is taken at the last forward's input)")
        g = L.dev_tensor(g_v, torch.float32)
        B, H, W = self._last
        if tuple(g.shape) != (B, 3, H, W):
            raise ValueError(f"vjp: gradient shape {tuple(g.shape)} does not match the last forward {(B, 3, H, W)}")
        out = torch.empty_like(g)
        L.check(L.lib().maua_secondary_vjp(self._handle(), L.ptr(g), B, H, W, L.ptr(out)))
        return out


class GradientGuidedConditioning(torch.nn.Module):
    """guided.py:212-274.  speed="hyper": img = (x - sigma * noise) / alpha is the clean-image estimate from the KNOWN noise; the grad
    modules return d loss / d img, and d img / d x = 1 / alpha, so cond_fn = -sum(grads) / alpha.  speed="fast" (the reference's
    default): img = pred * sigma + x * (1 - sigma) with pred = x * a_c - v(x, t_c) * s_c from the secondary model at the cosine time
    t_c = atan2(sigma, alpha) * 2 / pi (a_c, s_c = cos, sin(t_c * pi / 2)), so
        -J^T g = -[(sigma * a_c + 1 - sigma) * g - sigma * s_c * (d v / d x)^T g]
    with the last term from ``SecondaryDiffusionImageNet2.vjp`` (the reference gets the same number from torch.autograd.grad)."""

    def __init__(self, diffusion, model, grad_modules, speed="fast"):
        super().__init__()
        if speed == "fast" and not isinstance(model, SecondaryDiffusionImageNet2):
            raise ValueError('speed="fast" needs the secondary model (create_models(use_secondary=True))')
        if speed not in ("hyper", "fast"):      # guided.py:217-218: anything else is "regular", through the UNet itself
            if not hasattr(model, "forward_keep"):
                raise ValueError('speed="regular" needs the diffusion UNet (a UNetModel) as its model')
            model.enable_vjp()
        self.speed, self.model, self.grad_modules = speed, model, list(grad_modules)
        self.diffusion = diffusion
        self._offer = None
        self.timestep_map = list(diffusion.timestep_map)
        self.sqrt_alphas_cumprod = torch.from_numpy(diffusion.sqrt_alphas_cumprod).float()
        self.sqrt_one_minus_alphas_cumprod = torch.from_numpy(diffusion.sqrt_one_minus_alphas_cumprod).float()
        self.noise = None

    def set_targets(self, prompts, noise, per_sample=False):
        self.noise = noise
        for gm in self.grad_modules:
            if per_sample:
                gm.set_targets_per_sample(prompts)   # (only modules that declare it: GuidedDiffusion.run checks)
            else:
                gm.set_targets(prompts)

    def offer(self, x, t, out):
        """The sampler's own kept evaluation of the network on (x, t) (SpacedDiffusion._model_output): the next ``forward`` on the same
        sample tensor and timesteps differentiates through it instead of evaluating the network a second time."""
        self._offer = (x, torch.as_tensor(t).detach().float().cpu().clone(), out)

    def per_sample_prompts(self):
        return all(hasattr(gm, "set_targets_per_sample") for gm in self.grad_modules)

    def _sum_grads(self, img, ot):
        """guided.py:258-266: the sum of the modules' gradients, a module whose gradient holds a NaN skipped - inside the library
        (maua_grad_accumulate: no torch kernels, no host round trip per step)."""
        img_grad = torch.empty_like(img)
        lib, ctx = L.lib(), L.ctx(img.device)
        for k, gm in enumerate(self.grad_modules):
            sub = L.dev_tensor(gm(img, ot), torch.float32)
            L.check(lib.maua_grad_accumulate(ctx, L.ptr(sub), L.ptr(img_grad), C.c_long(img.numel()), int(k == 0)))
        return img_grad

    def guide_coefficients(self, t):
        """[len(t), 5] float32 = {cos_t, sigma, 1 - sigma, -(sigma a_c + 1 - sigma), sigma s_c} of the "fast" conditioning at the
        model timesteps ``t``, evaluated exactly like ``forward`` evaluates them (host float32 tensor arithmetic, :249-252 / :266-268);
        what maua_ddim_guided_loop takes per step."""
        idx = torch.tensor([self.timestep_map.index(int(v)) for v in torch.as_tensor(t).long().cpu().reshape(-1)])
        alpha, sigma = self.sqrt_alphas_cumprod[idx], self.sqrt_one_minus_alphas_cumprod[idx]
        if self.speed not in ("hyper", "fast"):   # "regular": {-, sigma, 1 - sigma, -(sigma ra + 1 - sigma), sigma rm} (forward's own expressions)
            ra = self.diffusion._f32(self.diffusion.sqrt_recip_alphas_cumprod, idx)
            rm = self.diffusion._f32(self.diffusion.sqrt_recipm1_alphas_cumprod, idx)
            return torch.stack([torch.zeros_like(sigma), sigma, 1 - sigma, -(sigma * ra + 1 - sigma), sigma * rm], 1).float().contiguous()
        cosine_t = torch.atan2(sigma, alpha) * 2 / math.pi
        a_c, s_c = torch.cos(cosine_t * math.pi / 2), torch.sin(cosine_t * math.pi / 2)
        return torch.stack([cosine_t, sigma, 1 - sigma, -(sigma * a_c + 1 - sigma), sigma * s_c], 1).float().contiguous()

    def graphable(self, shape=None):
        """True when the whole guided step is library work (speed "fast" or "regular", exactly one grad module the library has: the
        image-MSE module with a target, or CLIPGrads with one perceptor): the sampler loop then runs as one hipGraph
        (SpacedDiffusion.ddim_guided_loop)."""
        if self.speed == "hyper" or not self.grad_modules:
            return False
        mods = self.grad_modules
        if len(mods) == 1 and isinstance(mods[0], MSEGuide):
            return mods[0].target is not None
        # CLIPGrads (one perceptor, targets set) first, then modules the library holds as guide objects (VGGGrads, LPIPSGrads,
        # ColorMatchGrads: maua_unet_set_guides) - summed in the list's order, like the step-by-step path sums them
        rest = mods[1:] if hasattr(mods[0], "graph_spec") else mods
        if hasattr(mods[0], "graph_spec") and mods[0].graph_spec() is None:
            return False
        return all(hasattr(gm, "graph_guide") and not hasattr(gm, "graph_spec") for gm in rest) and \
            all(gm.graph_ready(shape) for gm in rest)

    def forward(self, x, t, kw={}):
        ot = t.clone()
        idx = torch.tensor([self.timestep_map.index(int(v)) for v in t.long().cpu()])
        alpha, sigma = self.sqrt_alphas_cumprod[idx], self.sqrt_one_minus_alphas_cumprod[idx]
        x = L.dev_tensor(x, torch.float32)
        B, row = x.shape[0], C.c_long(x[0].numel())
        lib, ctx = L.lib(), L.ctx(x.device)
        img = torch.empty_like(x)
        if self.speed == "hyper":
            ab = L.dev_tensor(torch.stack([1 / alpha, -sigma / alpha], 1).contiguous(), torch.float32)
            L.check(lib.maua_axpby_rows(ctx, L.ptr(x), L.ptr(self.noise), L.ptr(ab), B, row, L.ptr(img)))
            return -self._sum_grads(img, ot) / alpha.to(x.device).reshape(-1, 1, 1, 1)
        if self.speed != "fast":
            # "regular" (:250-252): img = pred_xstart * sigma + x * (1 - sigma), pred_xstart = ra * x - rm * eps(x, t) from
            # p_mean_variance (gaussian_diffusion.py _predict_xstart_from_eps; eps = the first half of the learn_sigma output), so
            #     -J^T g = -[(sigma * ra + 1 - sigma) * g - sigma * rm * (d eps / d x)^T g]
            ra, rm = self.diffusion._f32(self.diffusion.sqrt_recip_alphas_cumprod, idx), self.diffusion._f32(self.diffusion.sqrt_recipm1_alphas_cumprod, idx)
            nc = x.shape[1]
            offer, self._offer = self._offer, None
            mt = self.diffusion.model_timesteps(idx)
            if offer is not None and offer[0].data_ptr() == x.data_ptr() and offer[0].shape == x.shape \
                    and torch.equal(offer[1], mt.float().cpu()) and getattr(self.model, "_kept", None) == (B, x.shape[2], x.shape[3]):
                out = offer[2]
            else:
                out = self.model.forward_keep(x, mt)
            eps = out[:, :nc].contiguous()
            pred = torch.empty_like(x)
            ab = L.dev_tensor(torch.stack([ra, -rm], 1).contiguous(), torch.float32)
            L.check(lib.maua_axpby_rows(ctx, L.ptr(x), L.ptr(eps), L.ptr(ab), B, row, L.ptr(pred)))
            ab = L.dev_tensor(torch.stack([sigma, 1 - sigma], 1).contiguous(), torch.float32)
            L.check(lib.maua_axpby_rows(ctx, L.ptr(pred), L.ptr(x), L.ptr(ab), B, row, L.ptr(img)))
            g = self._sum_grads(img, ot)
            g_out = torch.zeros_like(out)
            g_out[:, :nc] = g
            jv = self.model.vjp(g_out)
            cf = L.dev_tensor(torch.stack([-(sigma * ra + 1 - sigma), sigma * rm], 1).contiguous(), torch.float32)
            res = torch.empty_like(x)
            L.check(lib.maua_axpby_rows(ctx, L.ptr(g), L.ptr(jv), L.ptr(cf), B, row, L.ptr(res)))
            return res
        cosine_t = torch.atan2(sigma, alpha) * 2 / math.pi                           # :252
        pred = self.model(x, cosine_t).pred                                          # :253
        ab = L.dev_tensor(torch.stack([sigma, 1 - sigma], 1).contiguous(), torch.float32)
        L.check(lib.maua_axpby_rows(ctx, L.ptr(pred), L.ptr(x), L.ptr(ab), B, row, L.ptr(img)))     # :254
        g = self._sum_grads(img, ot)
        a_c, s_c = torch.cos(cosine_t * math.pi / 2), torch.sin(cosine_t * math.pi / 2)
        jv = self.model.vjp(g)                                                       # (d v / d x)^T g
        cf = L.dev_tensor(torch.stack([-(sigma * a_c + 1 - sigma), sigma * s_c], 1).contiguous(), torch.float32)
        out = torch.empty_like(x)
        L.check(lib.maua_axpby_rows(ctx, L.ptr(g), L.ptr(jv), L.ptr(cf), B, row, L.ptr(out)))        # :268 (the minus sign folded in)
        return out


class MSEGuide:
    """A grad module that needs no un-vendored perceptor: loss = scale * mean((img - target)^2) per sample, so
    d loss / d img = 2 * scale * (img - target) / numel.  ``set_targets`` takes the prompts' ``target`` images (the
    audio-switched prompt schedule hands over one prompt per frame)."""

    def __init__(self, scale=1000.0):
        self.scale, self.target = scale, None

    def set_targets(self, prompts):
        ts = [p.target if hasattr(p, "target") else p for p in prompts]
        self.target = None if not ts else torch.stack([torch.as_tensor(t).float() for t in ts]).mean(0).cuda()

    def set_targets_per_sample(self, prompts):
        """One prompt per SAMPLE of the batch (the audio-switched schedule: frames either side of a prompt switch travel through
        the sampler together): target [B, C, H, W]."""
        ts = [p.target if hasattr(p, "target") else p for p in prompts]
        self.target = None if not ts else torch.stack([torch.as_tensor(t).float() for t in ts]).cuda()

    def factor(self, numel):
        return 2.0 * self.scale / numel

    def __call__(self, img, t):
        if self.target is None:
            return torch.zeros_like(img)
        img = L.dev_tensor(img, torch.float32)
        tgt = L.dev_tensor(self.target, torch.float32)
        if tuple(tgt.shape) not in (tuple(img.shape[1:]), tuple(img.shape)):
            raise ValueError(f"MSEGuide: target shape {tuple(tgt.shape)} does not fit images of shape {tuple(img.shape)}")
        row = img[0].numel()
        out = torch.empty_like(img)
        L.check(L.lib().maua_mse_guide_grad(L.ctx(img.device), L.ptr(img), L.ptr(tgt), C.c_long(row if tgt.dim() == img.dim() else 0),
                                            C.c_float(self.factor(row)), img.shape[0], C.c_long(row), L.ptr(out)))
        return out


class ImageTarget:
    """The prompt type MSEGuide consumes: a target image [C, H, W] in [-1, 1]."""

    def __init__(self, target):
        self.target = target

    def to(self, *_a, **_k):
        return self


class GuidedDiffusion(torch.nn.Module):
    """guided.py:277-339 with the reference's defaults (samplers "ddim", "p", "plms"; speed "fast" = the secondary model, or
    "hyper").  ``model_checkpoint`` / ``model`` + ``diffusion`` (+ ``secondary_model``): either the reference's checkpoint name
    (files must exist, see create_models) or ready objects (tests, bench)."""

    def __init__(self, grad_modules, sampler="ddim", timesteps=100, model_checkpoint="uncondImageNet512", device="cuda",
                 ddim_eta=0, plms_order=2, speed="fast", model=None, diffusion=None, secondary_model=None, allow_random_init=False,
                 dtype=torch.bfloat16, secondary_dtype=torch.float32, secondary_exact=False):
        super().__init__()
        if sampler not in ("ddim", "p", "plms"):
            raise NotImplementedError()
        mods = [gm for gm in grad_modules if gm.scale != 0]
        if model is None:
            model, diffusion, secondary_model = create_models(            # (:292: "ddimN" spacing only for DDIM)
                checkpoint=model_checkpoint, timestep_respacing=f"ddim{timesteps}" if sampler == "ddim" else str(timesteps),
                use_secondary=speed == "fast", allow_random_init=allow_random_init, dtype=dtype, secondary_dtype=secondary_dtype,
                secondary_exact=secondary_exact)
        elif speed == "fast" and secondary_model is None and mods:
            secondary_model = SecondaryDiffusionImageNet2(dtype=secondary_dtype, exact=secondary_exact) if allow_random_init else None
            if secondary_model is None:
                raise ValueError('speed="fast" with ready objects needs secondary_model= (or allow_random_init=True)')
        self.model, self.diffusion, self.ddim_eta = model, diffusion, ddim_eta
        self.secondary_model = secondary_model
        self.sampler, self.plms_order = sampler, plms_order
        self.conditioning = GradientGuidedConditioning(diffusion, secondary_model if speed == "fast" else model, mods,   # :297-302
                                                       speed=speed) if mods else None
        self.device = device
        self.use_graph = True   # guided loops whose every step is library work run as one hipGraph (False: step by step)
        self.original_num_steps = diffusion.original_num_steps
        self.timestep_map = diffusion.timestep_map
        self.image_size = model.image_size

    @torch.no_grad()
    def run(self, img, prompts, start_step, n_steps, noise=None, per_sample=False):
        """q_sample(img, start_step, noise), then n_steps DDIM updates with t = start_step, start_step - 1, ...
        -> the last step's pred_xstart.  Without grad modules, and with guidance that is all library work (speed "fast", one image-MSE
        module), the loop runs inside the library as one hipGraph.  ``per_sample``: ``prompts`` holds one prompt per sample."""
        img = L.dev_tensor(img, torch.float32)
        t = torch.tensor([start_step] * img.shape[0], dtype=torch.long)
        noise = torch.randn_like(img) if noise is None else L.dev_tensor(noise, torch.float32)
        x = self.diffusion.q_sample(img, t, noise)
        if n_steps <= 0:
            return None  # (the reference returns out["pred_xstart"] of out = None here, i.e. raises)
        if self.sampler == "ddim" and self.conditioning is None and self.ddim_eta == 0:
            return self.diffusion.ddim_sample_loop(self.model, x, start_step, n_steps)[1]
        if self.conditioning is not None:
            if per_sample and (len(prompts) != img.shape[0] or not self.conditioning.per_sample_prompts()):
                raise ValueError("per_sample needs one prompt per sample and grad modules with set_targets_per_sample")
            self.conditioning.set_targets([p.to(img) for p in prompts], noise, per_sample)
            if self.sampler == "ddim" and self.ddim_eta == 0 and self.use_graph and self.conditioning.graphable(x.shape):
                return self.diffusion.ddim_guided_loop(self.model, self.conditioning, x, start_step, n_steps)[1]
        out = None
        for _ in range(n_steps):   # guided.py:302-311, :333-337
            if self.sampler == "ddim":
                out = self.diffusion.ddim_sample(self.model, x, t, clip_denoised=False, cond_fn=self.conditioning,
                                                 model_kwargs={}, eta=self.ddim_eta)
            elif self.sampler == "p":
                out = self.diffusion.p_sample(self.model, x, t, clip_denoised=False, cond_fn=self.conditioning, model_kwargs={})
            else:
                out = self.diffusion.plms_sample(self.model, x, t, clip_denoised=False, cond_fn=self.conditioning,
                                                 model_kwargs={}, order=self.plms_order, old_out=out)
            x = out["sample"]
            t = t - 1
        return out["pred_xstart"]

    def forward(self, img, prompts, t_start, t_end=1, verbose=True, noise=None):
        """:322-339, arithmetic kept as written there (start_step counts from the NOISY end: t_start = 0.5 runs the last
        half of the schedule; t never reaches 0)."""
        n = len(self.timestep_map)
        return self.run(img, prompts, round(t_start * (n - 1)), round((t_end - t_start) * (n - 1)), noise)


# ------------------------------------------------------------------- audio-onset-switched prompts (configs[3])
def get_diffusion_model(diffusion="guided", timesteps: int = 50, sampler: str = "plms", guidance_speed: str = "fast", clip_scale: float = 0.0,
                        lpips_scale: float = 0.0, style_scale: float = 0.0, color_match_scale: float = 0.0, cfg_scale: float = 5.0, image=None,
                        guided_kwargs=None):
    """maua/diffusion/image.py:76-125 for the guided-diffusion processor: the grad-module list (CLIPGrads, LPIPSGrads, VGGGrads,
    ColorMatchGrads - each only when its scale is positive, in that order) around ``GuidedDiffusion``.  ``guided_kwargs`` (a dict) reach its
    constructor (``allow_random_init`` / ready ``model`` + ``diffusion`` objects: there are no checkpoints in the image; the perceptors
    follow ``allow_random_init``).  The latent / stable / glide processors are other networks and not part of this build."""
    if isinstance(diffusion, GuidedDiffusion):
        return diffusion
    if diffusion != "guided":
        raise NotImplementedError(f'get_diffusion_model("{diffusion}"): only the "guided" processor (maua/diffusion/processors/guided.py) is built')
    from .grad import CLIPGrads, ColorMatchGrads, LPIPSGrads, VGGGrads
    guided_kwargs = dict(guided_kwargs or {})
    rnd = dict(allow_random_init=True) if guided_kwargs.get("allow_random_init") else {}
    grad_modules = (
        ([CLIPGrads(scale=clip_scale, **rnd)] if clip_scale > 0 else [])
        + ([LPIPSGrads(scale=lpips_scale, **rnd)] if lpips_scale > 0 else [])
        + ([VGGGrads(scale=style_scale, **rnd)] if style_scale > 0 else [])
        + ([ColorMatchGrads(scale=color_match_scale)] if color_match_scale > 0 else [])
    )
    return GuidedDiffusion(grad_modules=grad_modules, sampler=sampler, timesteps=timesteps, speed=guidance_speed, **guided_kwargs)


def onset_prompt_schedule(audio, sr, fps, n_prompts, percentile=90):
    """Per video frame, which prompt is active: the prompt index advances at every onset PEAK of the clip (frames where
    the hop-aligned onset envelope - the same bit-exact path the StyleGAN2 render uses, audio.onsets - is a local maximum
    above its ``percentile``), cyclically.  -> int64 [n_frames] on the host.  The reference has no audio coupling in
    maua.diffusion; this is the coupling BASELINE configs[3] names, built from the path's own onset bins."""
    from . import audio as A
    from . import signal as SG
    if sr != 1024 * fps:
        raise ValueError("resample the clip to 1024 * fps first (audio_io.load_audio does): one STFT hop = one frame")
    env = A.onsets(torch.as_tensor(audio), sr).squeeze(-1)
    thr = SG.percentile(env, percentile)
    e = env.cpu()
    peak = torch.zeros_like(e, dtype=torch.bool)
    if len(e) > 2:
        peak[1:-1] = (e[1:-1] > e[:-2]) & (e[1:-1] >= e[2:]) & (e[1:-1] > float(thr))
    return (torch.cumsum(peak.long(), 0) % max(1, n_prompts)).long()


@torch.no_grad()
def sample(prompts: List, audio=None, sr=None, fps=30, n_frames=None, size=(256, 256), timesteps=100,
           t_start: Optional[float] = None, model=None, diffusion=None, grad_modules=None, seed=0, batch=4, init=None,
           verbose=False, speed="hyper", secondary_model=None):
    """configs[3]: one 100-step DDIM sample per video frame, the active prompt switched on the clip's onset bins.
    ``prompts``: list of prompt objects (what the grad modules' set_targets understands).  -> ([n_frames, 3, H, W] in
    [-1, 1], prompt index per frame)."""
    if model is None:
        model, diffusion, _ = create_models("uncondImageNet256", f"ddim{timesteps}")
    gd = GuidedDiffusion(grad_modules or [], timesteps=timesteps, model=model, diffusion=diffusion, speed=speed,
                         secondary_model=secondary_model)   # speed="fast" (the reference's default) needs the secondary model
    if audio is not None:
        idx = onset_prompt_schedule(audio, sr, fps, len(prompts))
        n_frames = len(idx) if n_frames is None else min(n_frames, len(idx))
        idx = idx[:n_frames]
    else:
        idx = torch.zeros(n_frames or 1, dtype=torch.long)
        n_frames = len(idx)
    g = torch.Generator().manual_seed(seed)
    H, W = size
    frames = torch.empty((n_frames, 3, H, W), dtype=torch.float32, device="cuda")
    n = len(gd.timestep_map)
    f = 0
    per_sample = bool(prompts) and gd.conditioning is not None and gd.conditioning.per_sample_prompts() and t_start is None
    while f < n_frames:
        # frames that share a prompt go through the sampler together (up to `batch`); grad modules that take one prompt per sample
        # (MSEGuide) let a batch run across the switches, so every batch is full
        e = f + 1
        while e < n_frames and e - f < batch and (per_sample or idx[e] == idx[f]):
            e += 1
        # drawn frame by frame (x0 then the q_sample noise, in frame order): a frame's draws do not depend on how the frames around it
        # were grouped into batches, so `batch` and the per-sample / per-prompt grouping leave the result for a seed unchanged
        draws = [(torch.randn((3, H, W), generator=g) if init is None else None, torch.randn((3, H, W), generator=g))
                 for _ in range(f, e)]
        x0 = torch.stack([d[0] for d in draws]) if init is None else torch.as_tensor(init).expand(e - f, 3, H, W)
        nz = torch.stack([d[1] for d in draws])
        if per_sample:
            frames[f:e] = gd.run(x0, [prompts[int(idx[j])] for j in range(f, e)], n - 1, n, noise=nz, per_sample=True)
            f = e
            continue
        active = [prompts[int(idx[f])]] if prompts else []
        if t_start is None:   # the whole respaced schedule: t = n - 1 ... 0 (`timesteps` DDIM steps)
            frames[f:e] = gd.run(x0, active, n - 1, n, noise=nz)
        else:                 # the reference's GuidedDiffusion.forward arithmetic
            frames[f:e] = gd.forward(x0, active, t_start, verbose=verbose, noise=nz)
        f = e
    return frames, idx

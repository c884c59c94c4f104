"""Oracle restatement of the guided-diffusion slice of BASELINE configs[3] (TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product).

What it restates (reference call sites): maua/diffusion/processors/guided.py:164-209 ``create_models`` (the OpenAI
guided-diffusion UNet: num_channels 256, num_res_blocks 2, attention at 32 / 16 / 8, 64-channel heads, learn_sigma,
resblock_updown, use_scale_shift_norm, linear noise schedule, "ddim100" respacing, rescale_timesteps) and :277-339
``GuidedDiffusion.forward`` (q_sample -> ddim_sample loop with a cond_fn).

The network and the sampler themselves live in the git submodule ``maua/submodules/guided_diffusion``
(github.com/crowsonkb/guided-diffusion, no pinned SHA, EMPTY in /root/reference): **PARITY UNPINNED**.  The code below
restates the published algorithm of guided_diffusion/unet.py (UNetModel, ResBlock, AttentionBlock, QKVAttentionLegacy,
timestep_embedding, GroupNorm32), gaussian_diffusion.py (linear betas, q_sample, p_mean_variance for epsilon /
learned-range models, condition_score, ddim_sample) and respace.py (space_timesteps, SpacedDiffusion, the wrapped
model's timestep map) in plain PyTorch-CPU fp32 (network) and numpy float64 (schedules), with the upstream state-dict
key names so that a released checkpoint would load unchanged.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ configuration
def default_channel_mult(image_size):
    """script_util.create_model: the channel multipliers chosen from the image size."""
    return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[image_size]


def unet_config(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions=(32, 16, 8),
                channel_mult=None, num_head_channels=64, in_channels=3, learn_sigma=True):
    """guided.py:171-190's model_config as the structural numbers the UNet needs.  ``attention_resolutions`` are
    feature-map sizes like the reference's string "32, 16, 8"; upstream turns them into down-sampling rates
    image_size // res."""
    cm = tuple(channel_mult) if channel_mult is not None else default_channel_mult(image_size)
    return dict(image_size=image_size, in_channels=in_channels, model_channels=model_channels,
                out_channels=(2 if learn_sigma else 1) * in_channels, num_res_blocks=num_res_blocks,
                attention_ds=tuple(image_size // int(r) for r in attention_resolutions), channel_mult=cm,
                num_head_channels=num_head_channels)


def unet_structure(cfg):
    """UNetModel.__init__ (unet.py): the module tree as lists of layer descriptions.
    -> dict(input=[[layer, ...], ...], middle=[...], output=[[...], ...]); a layer is ("conv", cin, cout),
    ("res", cin, cout, updown) with updown in (None, "down", "up"), or ("attn", ch)."""
    mc, cm, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    ch = int(cm[0] * mc)
    inp = [[("conv", cfg["in_channels"], ch)]]
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cm):
        for _ in range(nrb):
            layers = [("res", ch, int(mult * mc), None)]
            ch = int(mult * mc)
            if ds in cfg["attention_ds"]:
                layers.append(("attn", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            inp.append([("res", ch, ch, "down")])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, None), ("attn", ch), ("res", ch, ch, None)]
    out = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, int(mc * mult), None)]
            ch = int(mc * mult)
            if ds in cfg["attention_ds"]:
                layers.append(("attn", ch))
            if level and i == nrb:
                layers.append(("res", ch, ch, "up"))
                ds //= 2
            out.append(layers)
    return dict(input=inp, middle=mid, output=out, final_ch=ch)


def init_unet_params(cfg, generator=None, zero_out=False):
    """A state dict with guided-diffusion's key names.  Upstream zero-initialises the last conv of every ResBlock, every
    attention proj_out and the output conv (``zero_module``), which would make a random-init forward vacuous as a test;
    ``zero_out=False`` draws them like the other layers (fan-in scaled normal)."""
    g = generator or torch.Generator().manual_seed(0)
    p = {}
    emb = cfg["model_channels"] * 4

    def rn(*shape, fan_in, scale=1.0):
        return torch.randn(*shape, generator=g) * (scale / math.sqrt(fan_in))

    def lin(name, k, n):
        p[name + ".weight"], p[name + ".bias"] = rn(n, k, fan_in=k), rn(n, fan_in=1, scale=0.1)

    def conv3(name, ci, co, zero=False):
        p[name + ".weight"] = torch.zeros(co, ci, 3, 3) if zero else rn(co, ci, 3, 3, fan_in=ci * 9)
        p[name + ".bias"] = torch.zeros(co) if zero else rn(co, fan_in=1, scale=0.1)

    def gn(name, c):
        p[name + ".weight"] = 1 + 0.1 * torch.randn(c, generator=g)
        p[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    def layer(pfx, l):
        if l[0] == "conv":
            conv3(pfx, l[1], l[2])
        elif l[0] == "res":
            _, ci, co, _ud = l
            gn(pfx + ".in_layers.0", ci)
            conv3(pfx + ".in_layers.2", ci, co)
            lin(pfx + ".emb_layers.1", emb, 2 * co)
            gn(pfx + ".out_layers.0", co)
            conv3(pfx + ".out_layers.3", co, co, zero=zero_out)
            if ci != co:
                p[pfx + ".skip_connection.weight"] = rn(co, ci, 1, 1, fan_in=ci)
                p[pfx + ".skip_connection.bias"] = rn(co, fan_in=1, scale=0.1)
        else:
            c = l[1]
            gn(pfx + ".norm", c)
            p[pfx + ".qkv.weight"], p[pfx + ".qkv.bias"] = rn(3 * c, c, 1, fan_in=c), rn(3 * c, fan_in=1, scale=0.1)
            p[pfx + ".proj_out.weight"] = torch.zeros(c, c, 1) if zero_out else rn(c, c, 1, fan_in=c)
            p[pfx + ".proj_out.bias"] = torch.zeros(c) if zero_out else rn(c, fan_in=1, scale=0.1)

    s = unet_structure(cfg)
    lin("time_embed.0", cfg["model_channels"], emb)
    lin("time_embed.2", emb, emb)
    for i, layers in enumerate(s["input"]):
        for j, l in enumerate(layers):
            layer(f"input_blocks.{i}.{j}", l)
    for j, l in enumerate(s["middle"]):
        layer(f"middle_block.{j}", l)
    for i, layers in enumerate(s["output"]):
        for j, l in enumerate(layers):
            layer(f"output_blocks.{i}.{j}", l)
    gn("out.0", s["final_ch"])
    conv3("out.2", s["final_ch"], cfg["out_channels"], zero=zero_out)
    return p


# ------------------------------------------------------------------------------------------------------- network
def timestep_embedding(timesteps, dim, max_period=10000):
    """nn.py timestep_embedding: [N] -> [N, dim] (cos | sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(p, name, x):
    """GroupNorm32(32, C): computed in float32, eps 1e-5."""
    return F.group_norm(x.float(), 32, p[name + ".weight"], p[name + ".bias"], eps=1e-5)


def _resblock(p, pfx, x, emb, updown):
    """unet.py ResBlock._forward with use_scale_shift_norm and (for updown) Upsample / Downsample without conv:
    nearest x2 / avg_pool 2."""
    resample = {None: (lambda t: t), "down": (lambda t: F.avg_pool2d(t, 2, 2)),
                "up": (lambda t: F.interpolate(t, scale_factor=2, mode="nearest"))}[updown]
    h = F.silu(_gn(p, pfx + ".in_layers.0", x))
    h = resample(h)
    x = resample(x)
    h = F.conv2d(h, p[pfx + ".in_layers.2.weight"], p[pfx + ".in_layers.2.bias"], padding=1)
    emb_out = F.linear(F.silu(emb), p[pfx + ".emb_layers.1.weight"], p[pfx + ".emb_layers.1.bias"])[..., None, None]
    scale, shift = torch.chunk(emb_out, 2, dim=1)
    h = _gn(p, pfx + ".out_layers.0", h) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), p[pfx + ".out_layers.3.weight"], p[pfx + ".out_layers.3.bias"], padding=1)
    if pfx + ".skip_connection.weight" in p:
        x = F.conv2d(x, p[pfx + ".skip_connection.weight"], p[pfx + ".skip_connection.bias"])
    return x + h


def _attention(p, pfx, x, head_ch):
    """unet.py AttentionBlock._forward with QKVAttentionLegacy (use_new_attention_order False): the qkv channels are
    laid out [head][q | k | v][ch]; scale 1 / sqrt(sqrt(ch)) on both q and k; softmax in float32."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    n_heads = c // head_ch
    qkv = F.conv1d(_gn(p, pfx + ".norm", xf), p[pfx + ".qkv.weight"], p[pfx + ".qkv.bias"])
    length = qkv.shape[-1]
    q, k, v = qkv.reshape(b * n_heads, head_ch * 3, length).split(head_ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(head_ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, length)
    h = F.conv1d(a, p[pfx + ".proj_out.weight"], p[pfx + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def _run(p, pfx, layers, h, emb, cfg):
    for j, l in enumerate(layers):
        name = f"{pfx}.{j}"
        if l[0] == "conv":
            h = F.conv2d(h, p[name + ".weight"], p[name + ".bias"], padding=1)
        elif l[0] == "res":
            h = _resblock(p, name, h, emb, l[3])
        else:
            h = _attention(p, name, h, cfg["num_head_channels"])
    return h


def _unet_forward(p, cfg, x, timesteps, capture=None):
    s = unet_structure(cfg)
    emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = F.linear(F.silu(F.linear(emb, p["time_embed.0.weight"], p["time_embed.0.bias"])),
                   p["time_embed.2.weight"], p["time_embed.2.bias"])
    hs = []
    h = x.float()
    for i, layers in enumerate(s["input"]):
        h = _run(p, f"input_blocks.{i}", layers, h, emb, cfg)
        hs.append(h)
        if capture is not None:
            capture[f"input_blocks.{i}"] = h
    h = _run(p, "middle_block", s["middle"], h, emb, cfg)
    if capture is not None:
        capture["middle_block"] = h
    for i, layers in enumerate(s["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run(p, f"output_blocks.{i}", layers, h, emb, cfg)
        if capture is not None:
            capture[f"output_blocks.{i}"] = h
    h = F.silu(_gn(p, "out.0", h))
    return F.conv2d(h, p["out.2.weight"], p["out.2.bias"], padding=1)


@torch.no_grad()
def unet_forward(p, cfg, x, timesteps, capture=None):
    """UNetModel.forward: x [N, C, H, W], timesteps [N] (already scaled) -> [N, out_channels, H, W].
    ``capture``: optional dict that receives every block's output (for layer-by-layer parity)."""
    return _unet_forward(p, cfg, x, timesteps, capture)


# ------------------------------------------------------------------------------------------------------ diffusion
def linear_betas(num_diffusion_timesteps=1000):
    """gaussian_diffusion.get_named_beta_schedule("linear")."""
    scale = 1000 / num_diffusion_timesteps
    return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)


def space_timesteps(num_timesteps, section_counts):
    """respace.space_timesteps for "ddimN" and plain "N" / "a,b,c" strings."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx, all_steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac_stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur, taken = 0.0, []
        for _ in range(count):
            taken.append(start_idx + round(cur))
            cur += frac_stride
        all_steps += taken
        start_idx += size
    return set(all_steps)


class Schedule:
    """SpacedDiffusion(GaussianDiffusion): the float64 tables of the respaced process + the timestep map."""

    def __init__(self, diffusion_steps=1000, timestep_respacing="ddim100", rescale_timesteps=True):
        base = linear_betas(diffusion_steps)
        use = space_timesteps(diffusion_steps, timestep_respacing)
        ac = np.cumprod(1.0 - base, axis=0)
        last, betas, self.timestep_map = 1.0, [], []
        for i, a in enumerate(ac):
            if i in use:
                betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        self.original_num_steps, self.rescale_timesteps = diffusion_steps, rescale_timesteps
        self.betas = b = np.array(betas, dtype=np.float64)
        self.num_timesteps = len(b)
        self.alphas_cumprod = np.cumprod(1.0 - b, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(1.0 - b) / (1.0 - self.alphas_cumprod)

    def model_timesteps(self, t):
        """_WrappedModel.__call__: spaced index -> the timestep the network sees."""
        ts = torch.tensor(self.timestep_map)[t]
        return ts.float() * (1000.0 / self.original_num_steps) if self.rescale_timesteps else ts


def _ex(arr, t, shape):
    """_extract_into_tensor: float64 table -> float32 values broadcast to ``shape``."""
    v = torch.from_numpy(arr)[t].float()
    while v.dim() < len(shape):
        v = v[..., None]
    return v.expand(shape)


def q_sample(sch, x_start, t, noise):
    return _ex(sch.sqrt_alphas_cumprod, t, x_start.shape) * x_start + \
        _ex(sch.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise


def ddim_step(sch, model_output, x, t, cond_grad=None, eta=0.0, noise=None):
    """ddim_sample for an epsilon-predicting, learned-range model, clip_denoised False: ``model_output`` [N, 2C, H, W] is
    the network's output at (x, model_timesteps(t)); ``cond_grad`` = cond_fn(x, t) or None.
    -> (sample, pred_xstart)."""
    C = x.shape[1]
    eps_model = model_output[:, :C]
    c1, c2 = _ex(sch.sqrt_recip_alphas_cumprod, t, x.shape), _ex(sch.sqrt_recipm1_alphas_cumprod, t, x.shape)
    pred = c1 * x - c2 * eps_model                                  # _predict_xstart_from_eps
    alpha_bar = _ex(sch.alphas_cumprod, t, x.shape)
    if cond_grad is not None:                                       # condition_score
        eps = (c1 * x - pred) / c2
        eps = eps - (1 - alpha_bar).sqrt() * cond_grad
        pred = c1 * x - c2 * eps
    eps = (c1 * x - pred) / c2                                      # _predict_eps_from_xstart
    alpha_bar_prev = _ex(sch.alphas_cumprod_prev, t, x.shape)
    sigma = eta * torch.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * torch.sqrt(1 - alpha_bar / alpha_bar_prev)
    mean_pred = pred * torch.sqrt(alpha_bar_prev) + torch.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
    nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    if noise is None:
        noise = torch.zeros_like(x)
    return mean_pred + nonzero * sigma * noise, pred


def p_sample_step(sch, model_output, x, t, noise, cond_grad=None):
    """gaussian_diffusion.py p_sample + p_mean_variance (ModelVarType.LEARNED_RANGE, ModelMeanType.EPSILON, clip_denoised
    False) + condition_mean.  -> (sample, pred_xstart)."""
    C = x.shape[1]
    eps, var_values = model_output[:, :C], model_output[:, C:]
    min_log = _ex(sch.posterior_log_variance_clipped, t, x.shape)
    max_log = _ex(np.log(sch.betas), t, x.shape)
    frac = (var_values + 1) / 2
    log_variance = frac * max_log + (1 - frac) * min_log
    variance = torch.exp(log_variance)
    pred = _ex(sch.sqrt_recip_alphas_cumprod, t, x.shape) * x - _ex(sch.sqrt_recipm1_alphas_cumprod, t, x.shape) * eps
    mean = _ex(sch.posterior_mean_coef1, t, x.shape) * pred + _ex(sch.posterior_mean_coef2, t, x.shape) * x
    if cond_grad is not None:
        mean = mean.float() + variance * cond_grad.float()
    nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    return mean + nonzero * torch.exp(0.5 * log_variance) * noise, pred


def plms_sample(sch, model_fn, x, t, cond_fn=None, order=2, old_out=None):
    """plms_sample of the guided-diffusion fork the reference's (empty) submodule points at - un-vendored, restated from the
    published pseudo linear multistep sampler (Liu et al. 2022) in that fork's form: ``model_fn(x, t)`` -> network output at
    the spaced index t.  -> dict(sample, pred_xstart (unconditioned), old_eps)."""
    def get_model_output(xx, tt):
        out = model_fn(xx, tt)
        C = xx.shape[1]
        c1, c2 = _ex(sch.sqrt_recip_alphas_cumprod, tt, xx.shape), _ex(sch.sqrt_recipm1_alphas_cumprod, tt, xx.shape)
        pred_orig = c1 * xx - c2 * out[:, :C]
        pred = pred_orig
        if cond_fn is not None:
            e = (c1 * xx - pred) / c2
            e = e - (1 - _ex(sch.alphas_cumprod, tt, xx.shape)).sqrt() * cond_fn(xx, sch.model_timesteps(tt))
            pred = c1 * xx - c2 * e
        return (c1 * xx - pred) / c2, pred, pred_orig
    c1, c2 = _ex(sch.sqrt_recip_alphas_cumprod, t, x.shape), _ex(sch.sqrt_recipm1_alphas_cumprod, t, x.shape)
    abp = _ex(sch.alphas_cumprod_prev, t, x.shape)
    eps, pred, pred_orig = get_model_output(x, t)
    if order > 1 and old_out is None:
        old_eps = [eps]
        mean_pred = pred * torch.sqrt(abp) + torch.sqrt(1 - abp) * eps
        eps_2, _, _ = get_model_output(mean_pred, t - 1)
        eps_prime = (eps + eps_2) / 2
    else:
        old_eps = [] if old_out is None else list(old_out["old_eps"])
        old_eps.append(eps)
        cur = min(order, len(old_eps))
        if cur == 1:
            eps_prime = old_eps[-1]
        elif cur == 2:
            eps_prime = (3 * old_eps[-1] - old_eps[-2]) / 2
        elif cur == 3:
            eps_prime = (23 * old_eps[-1] - 16 * old_eps[-2] + 5 * old_eps[-3]) / 12
        else:
            eps_prime = (55 * old_eps[-1] - 59 * old_eps[-2] + 37 * old_eps[-3] - 9 * old_eps[-4]) / 24
    pred_prime = c1 * x - c2 * eps_prime
    mean_pred = pred_prime * torch.sqrt(abp) + torch.sqrt(1 - abp) * eps_prime
    if len(old_eps) >= order:
        old_eps.pop(0)
    nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    return {"sample": mean_pred * nonzero + pred * (1 - nonzero), "pred_xstart": pred_orig, "old_eps": old_eps}


@torch.no_grad()
def guided_diffusion_forward(p, cfg, sch, img, t_start, t_end=1, noise=None, cond_fn=None):
    """guided.py:322-339 GuidedDiffusion.forward with the "ddim" sampler (eta 0): q_sample to start_step, then n_steps
    ddim steps with t counting down; returns the last pred_xstart."""
    n = len(sch.timestep_map)
    start_step = round(t_start * (n - 1))
    n_steps = round((t_end - t_start) * (n - 1))
    t = torch.tensor([start_step] * img.shape[0], dtype=torch.long)
    if noise is None:
        noise = torch.randn_like(img)
    x = q_sample(sch, img, t, noise)
    pred = None
    for _ in range(n_steps):
        out = unet_forward(p, cfg, x, sch.model_timesteps(t))
        grad = None if cond_fn is None else cond_fn(x, sch.model_timesteps(t))
        x, pred = ddim_step(sch, out, x, t, grad)
        t = t - 1
    return pred


# ------------------------------------------------------------------------------------------------ secondary model ("fast" guidance)
# maua/diffusion/processors/guided.py:20-143 (in-tree, plain PyTorch): the small v-prediction UNet the reference's DEFAULT
# conditioning speed ("fast", guided.py:287) differentiates through, and :236-272 the conditioning itself.  Unlike the rest of
# this file this part IS pinned: tests/golden/g28_secondary.npz was generated by tests/golden/make_golden.py from the reference's
# own SecondaryDiffusionImageNet2 / GradientGuidedConditioning classes on weights from secondary_random_params() below.
SECONDARY_CS = (64, 128, 128, 256, 256, 512)


def secondary_conv_plan():
    """The 24 3x3 convolutions in execution order: (state-dict key prefix, Ci, Co, relu).  Key prefixes are what torch names
    the nested Sequential / SkipBlock modules of guided.py:77-134 ("net.2.main.1.0" = first ConvBlock inside the first SkipBlock)."""
    c = SECONDARY_CS
    plan = [("net.0.0", 3 + 16, c[0], True), ("net.1.0", c[0], c[0], True)]
    pfx = "net.2.main"
    down = [(c[0], c[1], c[1]), (c[1], c[2], c[2]), (c[2], c[3], c[3]), (c[3], c[4], c[4])]
    for ci, cm, co in down:      # [down, ConvBlock, ConvBlock, SkipBlock(...), ConvBlock, ConvBlock, up]
        plan += [(pfx + ".1.0", ci, cm, True), (pfx + ".2.0", cm, co, True)]
        pfx += ".3.main"
    plan += [(pfx + ".1.0", c[4], c[5], True), (pfx + ".2.0", c[5], c[5], True), (pfx + ".3.0", c[5], c[5], True),
             (pfx + ".4.0", c[5], c[4], True)]
    ups = [(c[4] * 2, c[4], c[3]), (c[3] * 2, c[3], c[2]), (c[2] * 2, c[2], c[1]), (c[1] * 2, c[1], c[0])]
    for ci, cm, co in ups:
        pfx = pfx[:-len(".3.main")]
        plan += [(pfx + ".4.0", ci, cm, True), (pfx + ".5.0", cm, co, True)]
    plan += [("net.3.0", c[0] * 2, c[0], True), ("net.4", c[0], 3, False)]
    return plan


def secondary_random_params(seed=0):
    """Seeded weights of the architecture, independent of torch's RNG stream (numpy PCG64, keys in plan order): the SAME
    function feeds the reference when the fixture is made and the HIP path / this restatement when it is checked."""
    rng = np.random.default_rng(seed)
    p = {"timestep_embed.weight": torch.from_numpy(rng.standard_normal((8, 1)).astype(np.float32))}
    for key, ci, co, _ in secondary_conv_plan():
        p[key + ".weight"] = torch.from_numpy((rng.standard_normal((co, ci, 3, 3)) * math.sqrt(2.0 / (9 * ci))).astype(np.float32))
        p[key + ".bias"] = torch.from_numpy((rng.standard_normal((co,)) * 0.05).astype(np.float32))
    return p


def secondary_forward(p, x, t):
    """guided.py:136-143: -> (v, pred, eps).  x [B, 3, H, W] (H, W multiples of 32), t [B] in [0, 1] (the cosine-schedule time)."""
    f = 2 * math.pi * t[:, None] @ p["timestep_embed.weight"].T                      # FourierFeatures :57-66
    emb = torch.cat([f.cos(), f.sin()], dim=-1)
    h = torch.cat([x, emb[:, :, None, None].expand(-1, -1, x.shape[2], x.shape[3])], dim=1)
    plan = secondary_conv_plan()

    def conv(i, h):
        key, _, _, relu = plan[i]
        h = F.conv2d(h, p[key + ".weight"], p[key + ".bias"], padding=1)
        return F.relu(h) if relu else h
    h = conv(1, conv(0, h))
    skips = []
    for lvl in range(4):
        skips.append(h)
        h = conv(3 + 2 * lvl, conv(2 + 2 * lvl, F.avg_pool2d(h, 2)))
    skips.append(h)
    h = F.avg_pool2d(h, 2)
    for i in range(10, 14):
        h = conv(i, h)
    for lvl in range(4):
        h = torch.cat([F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=False), skips.pop()], dim=1)
        h = conv(15 + 2 * lvl, conv(14 + 2 * lvl, h))
    h = torch.cat([F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=False), skips.pop()], dim=1)
    v = conv(23, conv(22, h))
    alphas, sigmas = torch.cos(t * math.pi / 2)[:, None, None, None], torch.sin(t * math.pi / 2)[:, None, None, None]
    return v, x * alphas - v * sigmas, x * sigmas + v * alphas


def fast_conditioning(p, sch, grad_fn, x, t):
    """guided.py:236-272 with speed="fast": t = the respaced model's timesteps (as cond_fn receives them); ``grad_fn(img, t)`` ->
    d loss / d img (the sum over the grad modules).  -> -J^T grad, J = d img / d x through the secondary model (torch
    autograd on the restatement above: this is the checker, the product evaluates the transposed network by hand)."""
    idx = torch.tensor([list(sch.timestep_map).index(int(v)) for v in t.long()])
    alpha = torch.from_numpy(sch.sqrt_alphas_cumprod).float()[idx]
    sigma = torch.from_numpy(sch.sqrt_one_minus_alphas_cumprod).float()[idx]
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_()
        cosine_t = torch.atan2(sigma, alpha) * 2 / math.pi
        pred = secondary_forward(p, xx, cosine_t)[1]
        s = sigma.reshape(-1, 1, 1, 1)
        img = pred * s + xx * (1 - s)
        g = grad_fn(img.detach(), t)
        return -torch.autograd.grad(img, xx, g)[0]


def unet_input_vjp(p, cfg, x, timesteps, g_out):
    """(d UNet(x, t) / d x)^T g_out by torch autograd on the restatement above - the checker of maua_unet_vjp (the product walks the
    network backwards by hand; the reference asks autograd, guided.py:268)."""
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_()
        out = _unet_forward(p, cfg, xx, timesteps)
        return torch.autograd.grad(out, xx, g_out)[0]


def regular_conditioning(p, cfg, sch, grad_fn, x, t):
    """guided.py:236-272 with speed="regular" (any speed other than "hyper" / "fast", :214-218): the loss is differentiated through
    the diffusion UNet itself.  self.model = partial(diffusion.p_mean_variance, model=model, clip_denoised=False) (:218);
    out = self.model(x=x, t=t)["pred_xstart"] (:251) with t the RESPACED index (:238) - p_mean_variance of the respaced process maps
    it back to the network's timestep and, for an epsilon model with learned variance, returns
    pred_xstart = sqrt_recip_alphas_cumprod[t] * x - sqrt_recipm1_alphas_cumprod[t] * eps, eps = the first half of the output
    (gaussian_diffusion.py p_mean_variance / _predict_xstart_from_eps); img = out * sigma + x * (1 - sigma) (:252)."""
    idx = torch.tensor([list(sch.timestep_map).index(int(v)) for v in t.long()])
    sigma = torch.from_numpy(sch.sqrt_one_minus_alphas_cumprod).float()[idx].reshape(-1, 1, 1, 1)
    ra, rm = _ex(sch.sqrt_recip_alphas_cumprod, idx, x.shape), _ex(sch.sqrt_recipm1_alphas_cumprod, idx, x.shape)
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_()
        out = _unet_forward(p, cfg, xx, sch.model_timesteps(idx))
        pred = ra * xx - rm * out[:, :x.shape[1]]
        img = pred * sigma + xx * (1 - sigma)
        g = grad_fn(img.detach(), t)
        return -torch.autograd.grad(img, xx, g)[0]

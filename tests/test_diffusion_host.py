"""CPU tests of the guided-diffusion slice (configs[3]): host logic of maua_amd.diffusion against the oracle's restatement, and
the oracle itself against analytic properties and torch's own modules (the guided_diffusion submodule is empty in the
reference checkout - PARITY UNPINNED - so the oracle is pinned on what CAN be checked here: torch.nn.GroupNorm / PReLU-free
building blocks, the closed forms of the DDIM update, the published schedule constants)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import diffusion as OD


def test_structure_and_parameter_names_agree_between_product_and_oracle():
    from maua_amd.diffusion import UNetModel, create_models
    cfg = OD.unet_config()                                      # guided.py:171-190 at image_size 256
    s = OD.unet_structure(cfg)
    assert len(s["input"]) == 18 and len(s["output"]) == 18 and s["final_ch"] == 256
    assert sum(l[0] == "attn" for b in s["input"] + s["output"] for l in b) + 1 == 16     # attention at 32 / 16 / 8 (+ middle)
    net = UNetModel.__new__(UNetModel)
    net.image_size, net.in_channels, net.model_channels, net.out_channels = 256, 3, 256, 6
    net.num_res_blocks, net.attention_resolutions, net.channel_mult, net.num_head_channels = 2, (8, 16, 32), (1, 1, 2, 2, 4, 4), 64
    net._structure = UNetModel._build_structure(net)
    shapes = UNetModel._param_shapes(net)
    want = {k: tuple(v.shape) for k, v in OD.init_unet_params(cfg, zero_out=True).items()}
    assert shapes == want and sum(int(np.prod(v)) for v in shapes.values()) == 552_814_086    # the released 256x256 model's size
    try:
        create_models("uncondImageNet256", "ddim100")
        assert False, "a missing checkpoint must raise"
    except FileNotFoundError:
        pass


def test_schedule_constants_and_respacing():
    from maua_amd.diffusion import SpacedDiffusion, space_timesteps
    b = OD.linear_betas(1000)
    assert b[0] == 1e-4 and abs(b[-1] - 0.02) < 1e-15 and len(b) == 1000
    assert sorted(OD.space_timesteps(1000, "ddim100")) == list(range(0, 1000, 10))
    assert OD.space_timesteps(10, "3") == {0, 4, 9} and OD.space_timesteps(300, "10,15,20") == space_timesteps(300, "10,15,20")
    sch = OD.Schedule(1000, "ddim100", True)
    sd = SpacedDiffusion(space_timesteps(1000, "ddim100"), b, rescale_timesteps=True)
    assert sd.timestep_map == sch.timestep_map == list(range(0, 1000, 10))
    # respaced alphas_cumprod = the base process's at the kept timesteps
    base_ac = np.cumprod(1 - b)
    assert np.allclose(sd.alphas_cumprod, base_ac[::10], rtol=1e-12) and np.array_equal(sd.alphas_cumprod, sch.alphas_cumprod)
    cf = sd.step_coefficients([99, 50, 0])
    assert cf.dtype == torch.float32 and float(cf[2, 3]) == 1.0 and float(cf[2, 5]) == 0.0     # alphas_cumprod_prev[0] = 1, sigma 0


def test_ddim_update_closed_forms():
    """If the model returns the TRUE noise, pred_xstart is x0 (to rounding) at every t, and the eta = 0 update lands exactly on
    q_sample(x0, t - 1, same noise); a conditioning gradient g shifts eps by -sqrt(1 - ac) g."""
    sch = OD.Schedule(1000, "ddim50", True)
    g = torch.Generator().manual_seed(0)
    x0, nz = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    t = torch.tensor([49, 30, 7, 1])
    xt = OD.q_sample(sch, x0, t, nz)
    mo = torch.cat([nz, torch.zeros_like(nz)], 1)
    sample, pred = OD.ddim_step(sch, mo, xt, t)
    assert float((pred - x0).abs().max()) <= 2e-4 * float(x0.abs().max())          # sqrt_recip ~ 144 at t = 49 amplifies f32 rounding
    assert float((sample - OD.q_sample(sch, x0, t - 1, nz)).abs().max()) <= 2e-4
    grad = 0.1 * torch.randn(4, 3, 8, 8, generator=g)
    _, pred_c = OD.ddim_step(sch, mo, xt, t, grad)
    ab = torch.from_numpy(sch.alphas_cumprod)[t].float().view(-1, 1, 1, 1)
    want = x0 + torch.from_numpy(sch.sqrt_recipm1_alphas_cumprod)[t].float().view(-1, 1, 1, 1) * (1 - ab).sqrt() * grad
    assert float((pred_c - want).abs().max()) <= 5e-4 * float(want.abs().max())


def test_oracle_blocks_against_torch_modules():
    """The oracle's ResBlock pieces written with torch modules (nn.GroupNorm(32, C), nn.Conv2d, nn.Linear, avg_pool /
    nearest) give the same numbers; its attention equals scaled-dot-product attention per head on the legacy channel layout."""
    g = torch.Generator().manual_seed(1)
    cfg = OD.unet_config(image_size=32, model_channels=32, num_res_blocks=1, attention_resolutions=(16,), channel_mult=(1, 2),
                         num_head_channels=32)
    p = OD.init_unet_params(cfg, g)
    x = torch.randn(2, 32, 16, 16, generator=g)
    emb = torch.randn(2, 128, generator=g)
    pfx = "input_blocks.2.0"                                                      # the 'down' ResBlock of level 0
    gn1 = torch.nn.GroupNorm(32, 32); gn1.weight.data, gn1.bias.data = p[pfx + ".in_layers.0.weight"], p[pfx + ".in_layers.0.bias"]
    gn2 = torch.nn.GroupNorm(32, 32); gn2.weight.data, gn2.bias.data = p[pfx + ".out_layers.0.weight"], p[pfx + ".out_layers.0.bias"]
    h = F.avg_pool2d(F.silu(gn1(x)), 2)
    h = F.conv2d(h, p[pfx + ".in_layers.2.weight"], p[pfx + ".in_layers.2.bias"], padding=1)
    eo = F.linear(F.silu(emb), p[pfx + ".emb_layers.1.weight"], p[pfx + ".emb_layers.1.bias"])
    h = gn2(h) * (1 + eo[:, :32, None, None]) + eo[:, 32:, None, None]
    h = F.conv2d(F.silu(h), p[pfx + ".out_layers.3.weight"], p[pfx + ".out_layers.3.bias"], padding=1)
    want = F.avg_pool2d(x, 2) + h
    with torch.no_grad():
        got = OD._resblock(p, pfx, x, emb, "down")
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # attention: legacy layout [head][q | k | v][ch] == per-head scaled dot-product attention
    pfx = "middle_block.1"
    xa = torch.randn(2, 64, 4, 4, generator=g)
    with torch.no_grad():
        got = OD._attention(p, pfx, xa, 32)
        xn = F.group_norm(xa, 32, p[pfx + ".norm.weight"], p[pfx + ".norm.bias"], eps=1e-5).reshape(2, 64, 16)
        qkv = F.conv1d(xn, p[pfx + ".qkv.weight"], p[pfx + ".qkv.bias"]).reshape(2, 2, 3, 32, 16)   # [B, head, qkv, ch, T]
        q, k, v = (qkv[:, :, i].transpose(-1, -2) for i in range(3))                                  # [B, head, T, ch]
        a = F.scaled_dot_product_attention(q, k, v).transpose(-1, -2).reshape(2, 64, 16)
        want = xa + F.conv1d(a, p[pfx + ".proj_out.weight"], p[pfx + ".proj_out.bias"]).reshape(2, 64, 4, 4)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # timestep embedding: cos | sin halves, unit frequency first
    e = OD.timestep_embedding(torch.tensor([0.0, 10.0]), 8)
    assert torch.equal(e[0], torch.tensor([1.0, 1, 1, 1, 0, 0, 0, 0])) and abs(float(e[1, 0]) - math.cos(10.0)) < 1e-6


def test_p_and_plms_restatements_closed_forms():
    """The oracle's p_sample / plms_sample restatements (un-vendored upstream: parity unpinned) against what can be said about
    them without the upstream code: (i) order-1 PLMS is the eta = 0 DDIM update (Liu et al. 2022, Song et al. 2021: the same
    transfer function); (ii) at order 2 the first call costs two model evaluations (improved Euler) and later ones one, the
    epsilon history holds order - 1 entries; (iii) p_sample with frac = 0 / 1 uses exactly the posterior / beta variance, its
    mean is the posterior mean of the predicted x0, and at t = 0 no noise is added; (iv) the Adams-Bashforth weights sum to 1."""
    sch = OD.Schedule(1000, "50", True)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 8, 8, generator=g)
    mo = torch.randn(2, 6, 8, 8, generator=g)
    t = torch.tensor([31, 7])
    calls = []

    def fn(xx, tt):
        calls.append(tt.clone())
        return mo
    o1 = OD.plms_sample(sch, fn, x, t, order=1, old_out={"old_eps": []})
    ws, wp = OD.ddim_step(sch, mo, x, t)
    assert float((o1["sample"] - ws).abs().max()) <= 2e-6 * float(ws.abs().max()) and torch.equal(o1["pred_xstart"], wp)
    calls.clear()
    o2 = OD.plms_sample(sch, fn, x, t, order=2, old_out=None)
    assert len(calls) == 2 and torch.equal(calls[1], t - 1) and len(o2["old_eps"]) == 1
    o3 = OD.plms_sample(sch, fn, o2["sample"], t - 1, order=2, old_out=o2)
    assert len(calls) == 3 and len(o3["old_eps"]) == 1
    for w, d in (([3, -1], 2), ([23, -16, 5], 12), ([55, -59, 37, -9], 24)):
        assert sum(w) == d
    nz = torch.randn(2, 3, 8, 8, generator=g)
    for frac_v, table in ((-1.0, sch.posterior_log_variance_clipped), (1.0, np.log(sch.betas))):
        m = mo.clone()
        m[:, 3:] = frac_v
        s, pred = OD.p_sample_step(sch, m, x, t, nz)
        c1 = torch.from_numpy(sch.posterior_mean_coef1)[t].float().view(-1, 1, 1, 1)
        c2 = torch.from_numpy(sch.posterior_mean_coef2)[t].float().view(-1, 1, 1, 1)
        sd = torch.exp(0.5 * torch.from_numpy(table)[t].float()).view(-1, 1, 1, 1)
        assert torch.allclose(s, c1 * pred + c2 * x + sd * nz, rtol=0, atol=1e-6)
    s0, p0 = OD.p_sample_step(sch, mo, x, torch.tensor([0, 0]), nz)
    c1 = float(sch.posterior_mean_coef1[0])
    assert abs(c1 - 1.0) < 1e-12 and torch.allclose(s0, p0, atol=1e-6)       # t = 0: the mean is the predicted x0, no noise

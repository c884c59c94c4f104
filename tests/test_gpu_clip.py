"""GPU parity tests of the text-prompt guidance path (maua/grad.py:96-165 CLIPGrads): every HIP piece through the C ABI against the
CPU oracle (oracle/clip.py) / the reference-generated fixture g33 / torch.autograd on the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import clip as OC

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def cos(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu().reshape(-1), torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


SMALL = dict(input_resolution=32, patch_size=8, width=64, layers=2, heads=2, output_dim=32)


def _tower(cfg, dt, seed=0):
    from maua_amd.clip import VisionTransformer
    p = OC.init_vit_params(cfg, torch.Generator().manual_seed(seed))
    vt = VisionTransformer(cfg["input_resolution"], cfg["patch_size"], cfg["width"], cfg["layers"], cfg["heads"], cfg["output_dim"], dtype=dt)
    vt.load_state_dict(p, strict=True)       # CLIP's own keys incl. the "visual." prefix
    return vt, p


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C_", [64, 768, 1024])
def test_layer_norm_and_its_vjp_match_torch(dt, C_):
    from maua_amd import _lib as L
    g = torch.Generator().manual_seed(C_)
    rows = 37
    x = (torch.randn(rows, C_, generator=g) * 2 + 0.5).to(dt)
    gam, bet = 1 + 0.2 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    dy, add = torch.randn(rows, C_, generator=g).to(dt), torch.randn(rows, C_, generator=g).to(dt)
    xd, gd, bd, dyd, addd = x.cuda(), gam.cuda(), bet.cuda(), dy.cuda(), add.cuda()
    y = torch.empty_like(xd)
    stats = torch.empty(rows, 2, device="cuda")
    lib, ctx = L.lib(), L.ctx()
    L.check(lib.maua_layer_norm(ctx, L.ptr(xd), L.ptr(gd), L.ptr(bd), C.c_long(rows), C_, L.dtype_id(dt), L.ptr(y), L.ptr(stats)))
    dx = torch.empty_like(xd)
    L.check(lib.maua_layer_norm_vjp(ctx, L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(dyd), L.ptr(addd), C.c_long(rows), C_, L.dtype_id(dt), L.ptr(dx)))
    with torch.enable_grad():
        xx = x.float().clone().requires_grad_()
        ref = F.layer_norm(xx, (C_,), gam, bet, 1e-5)
        want = torch.autograd.grad(ref, xx, dy.float())[0] + add.float()
    tol = 2e-6 if dt == torch.float32 else 1e-2
    assert rel(y, ref) <= tol and rel(dx, want) <= tol
    assert rel(stats[:, 0], x.float().mean(1)) <= 1e-5


# ------------------------------------------------------------------------------------------------ cutouts
def test_cutouts_match_the_reference_fixture(golden):
    """maua_cutouts on the rectangles the REFERENCE's random_cutouts took (g33) against its outputs (the resize inside being the
    restated resize_right algorithm): float32, <= 1e-5."""
    from maua_amd.grad import _run_cutouts
    g = golden("g33_cutouts")
    for k in (0, 1, 2, 4):
        H, W, cs, cutn, t, seed = (int(v) for v in g[f"cut{k}_cfg"])
        out = _run_cutouts(g[f"cut{k}_img"], g[f"cut{k}_rects"].numpy(), cs)
        assert tuple(out.shape) == tuple(g[f"cut{k}_out"].shape)
        assert rel(out, g[f"cut{k}_out"]) <= 1e-5, k
    # the full-size case (256^2 -> 32 cutouts of 224^2): per-image sums of the reference run
    H, W, cs, cutn, t, seed = (int(v) for v in g["cut3_cfg"])
    img = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(33))   # (not the fixture's image: shapes / finiteness only)
    out = _run_cutouts(img, g["cut3_rects"].numpy(), cs)
    assert tuple(out.shape) == (cutn * 2, 3, cs, cs) and bool(torch.isfinite(out).all())
    want = OC.cutouts_from_rects(img, [tuple(int(v) for v in r) for r in g["cut3_rects"].numpy()[8:12]], cs)
    assert rel(out[16:24], want) <= 1e-5


def test_cutouts_affine_normalize_and_vjp_match_autograd_on_the_oracle():
    """The form CLIPGrads uses - cutouts of (img + 1) / 2, Normalize(mean, std) - and its gradient: maua_cutouts_vjp against
    torch.autograd through the oracle's restatement, incl. up-scaling cutouts (image smaller than the cut size) and non-square images."""
    from maua_amd import _lib as L
    from maua_amd.grad import _run_cutouts
    for (H, W, cs, seed) in ((40, 52, 32, 1), (24, 24, 32, 2), (96, 96, 32, 3)):
        torch.manual_seed(seed)
        rects = OC.cutout_rects(H, W, cs, 8, OC.maua_cutouts_pow(700))
        g = torch.Generator().manual_seed(seed)
        img = torch.rand(2, 3, H, W, generator=g) * 2 - 1
        d = torch.randn(8 * 2, 3, cs, cs, generator=g)
        with torch.enable_grad():
            x = img.clone().requires_grad_()
            ref = OC.normalize(OC.cutouts_from_rects(x.add(1).div(2), rects, cs))
            want = torch.autograd.grad(ref, x, d)[0]
        out = _run_cutouts(img, rects, cs, 0.5, 0.5, OC.CLIP_MEAN, OC.CLIP_STD)
        assert rel(out, ref) <= 1e-5
        r = np.ascontiguousarray(np.asarray(rects, dtype=np.int32))
        dd = d.cuda()
        gi = torch.empty(2, 3, H, W, device="cuda")
        s = (C.c_float * 3)(*OC.CLIP_STD)
        L.check(L.lib().maua_cutouts_vjp(L.ctx(), L.ptr(dd), 2, H, W, r.ctypes.data_as(C.c_void_p), len(r), cs, C.c_float(0.5), s, L.ptr(gi)))
        assert rel(gi, want) <= 1e-5, (H, W)


# ------------------------------------------------------------------------------------------------ the image tower
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_vision_transformer_forward_and_input_gradient_match_the_oracle(dt):
    """VisionTransformer.forward = maua_clip_encode_image against the oracle's restatement of clip/model.py, and its vjp against
    torch.autograd on the oracle: exact-f32 mode <= 1e-4 / 2e-4 of the maximum; bf16 cosine >= 0.999 / 0.99."""
    vt, p = _tower(SMALL, dt)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 3, 32, 32, generator=g)
    de = torch.randn(5, SMALL["output_dim"], generator=g)
    with torch.enable_grad():
        xx = x.clone().requires_grad_()
        ref = OC.encode_image(p, SMALL, xx)
        want = torch.autograd.grad(ref, xx, de)[0]
    out = vt(x, keep=True)
    gx = vt.vjp(de)
    if dt == torch.float32:
        assert rel(out, ref) <= 1e-4 and rel(gx, want) <= 2e-4
    else:
        print("bf16 tower: embedding cosine", cos(out, ref), "input-gradient cosine", cos(gx, want))
        assert cos(out, ref) >= 0.999 and cos(gx, want) >= 0.99
    assert torch.equal(vt(x), out)          # (without kept activations the layers share one set of buffers: same result)


@pytest.mark.parametrize("width", [128, 256], ids=["256x128-tiles", "256x256-tiles"])
def test_tower_on_the_lds_direct_gemm_matches_the_register_staged_one(width):
    """A tower whose GEMMs are big enough for gemm_dma.hip (66 560 token rows; width 128: its 256 x 128 tile, width 256: the 256 x 256
    tile the ViT-B towers run on): embeddings and input gradients
    with the LDS-direct kernel (QuickGELU and its derivative riding on the epilogues) against the same tower on gemm.hip's kernels +
    the separate element-wise passes (ctx option "gemm_dma" = 0), and against the float32 oracle on a subset."""
    from maua_amd import _lib as L
    cfg = dict(input_resolution=64, patch_size=8, width=width, layers=2, heads=width // 64, output_dim=64)
    vt, p = _tower(cfg, torch.bfloat16, seed=3)
    g = torch.Generator().manual_seed(9)
    N = 1024
    x = torch.randn(N, 3, 64, 64, generator=g)
    de = torch.randn(N, 64, generator=g)
    lib = L.lib()
    res = {}
    for mode in (1, 0):
        L.check(lib.maua_ctx_set_option(L.ctx(), b"gemm_dma", mode))
        try:
            out = vt(x, keep=True)
            res[mode] = (out.clone(), vt.vjp(de).clone())
        finally:
            L.check(lib.maua_ctx_set_option(L.ctx(), b"gemm_dma", 1))
    assert cos(res[1][0], res[0][0]) >= 0.9999 and cos(res[1][1], res[0][1]) >= 0.999
    with torch.enable_grad():
        xx = x[:8].clone().requires_grad_()
        ref = OC.encode_image(p, cfg, xx)
        want = torch.autograd.grad(ref, xx, de[:8])[0]
    print("LDS-direct tower vs f32 oracle:", cos(res[1][0][:8], ref), cos(res[1][1][:8], want))
    assert cos(res[1][0][:8], ref) >= 0.999 and cos(res[1][1][:8], want) >= 0.99


# ------------------------------------------------------------------------------------------------ CLIPGrads
def _targets(E, P, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(P, E, generator=g), OC.normalise_weights(torch.rand(P, generator=g) + 0.2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_clip_guide_grad_matches_autograd_on_the_oracle(dt):
    """maua_clip_guide_grad = CLIPGrads.forward (cutouts -> Normalize -> image tower -> spherical distance to the targets -> weights,
    mean over cutouts -> gradient, averaged over the cutout batches) against torch.autograd.grad on the oracle's restatement of the
    same lines, same rectangles: exact-f32 <= 3e-4 of the gradient's maximum, bf16 cosine >= 0.98; clamp_gradient as :156-158."""
    from maua_amd import _lib as L
    vt, p = _tower(SMALL, dt, seed=1)
    E = SMALL["output_dim"]
    tgt, w = _targets(E, 3, 4)
    B, H, W, cutn, batches = 2, 40, 48, 8, 2
    g = torch.Generator().manual_seed(6)
    img = (torch.rand(B, 3, H, W, generator=g) * 2 - 1)
    torch.manual_seed(12)
    rects = [OC.cutout_rects(H, W, 32, cutn, OC.maua_cutouts_pow(620)) for _ in range(batches)]
    scale = 150.0
    want = OC.clip_grads(p, SMALL, img, rects, tgt, w, scale=scale)
    lib = L.lib()
    tn, wn = np.ascontiguousarray(tgt.numpy()), np.ascontiguousarray(w.numpy())
    L.check(lib.maua_clip_set_targets(vt._handle(), tn.ctypes.data_as(C.c_void_p), wn.ctypes.data_as(C.c_void_p), 1, 3, None, 0))
    r = np.ascontiguousarray(np.asarray(rects, dtype=np.int32))
    imgd = img.cuda()
    out = torch.empty_like(imgd)
    L.check(lib.maua_clip_guide_grad(vt._handle(), L.ptr(imgd), B, H, W, r.ctypes.data_as(C.c_void_p), None, cutn, batches, C.c_float(scale),
                                     C.c_float(0.0), L.ptr(out)))
    if dt == torch.float32:
        assert rel(out, want) <= 3e-4
        # the per-image losses of the last cutout batch (grad.py:153's dists.mul(weights).sum(2))
        losses = torch.empty(cutn * B, device="cuda")
        L.check(lib.maua_clip_last_image_losses(vt._handle(), cutn * B, L.ptr(losses)))
        cuts = OC.cutouts_from_rects(img.add(1).div(2), rects[-1], 32)
        d = OC.spherical_dist_loss(OC.encode_image(p, SMALL, OC.normalize(cuts)).unsqueeze(1), tgt.unsqueeze(0))
        assert rel(losses, d.mul(w).sum(1)) <= 1e-4
        # clamp_gradient
        mag = float(want.square().mean().sqrt())
        clamped = OC.clip_grads(p, SMALL, img, rects, tgt, w, scale=scale, clamp_gradient=0.5 * mag)
        L.check(lib.maua_clip_guide_grad(vt._handle(), L.ptr(imgd), B, H, W, r.ctypes.data_as(C.c_void_p), None, cutn, batches, C.c_float(scale),
                                         C.c_float(0.5 * mag), L.ptr(out)))
        assert rel(out, clamped) <= 3e-4
    else:
        print("bf16 CLIPGrads gradient cosine vs the f32 oracle:", cos(out, want))
        assert cos(out, want) >= 0.98


def test_clipgrads_module_prompts_per_sample_targets_and_seeded_cutouts():
    """The module as the reference's callers use it: set_targets with Embedding / Style prompts (:117-143), forward(img, t) drawing its
    cutouts from torch's global generator like MauaCutouts does (same seed -> the oracle's rectangles), per-sample prompts."""
    from maua_amd.clip import CLIPImageModel
    from maua_amd.grad import CLIPGrads, EmbeddingPrompt, StylePrompt, TextPrompt
    vt, p = _tower(SMALL, torch.float32, seed=2)
    E = SMALL["output_dim"]
    g = torch.Generator().manual_seed(8)
    e0, e1 = torch.randn(E, generator=g), torch.randn(E, generator=g)
    gm = CLIPGrads(scale=80.0, clip_models=[CLIPImageModel(vt)], cutout_kwargs=dict(cutn=8), cutout_batches=2)
    B, H, W = 2, 48, 48
    img = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    t = torch.tensor([437.0, 437.0])
    gm.set_targets([EmbeddingPrompt(e0, 1.0), EmbeddingPrompt(e1, 3.0)])
    torch.manual_seed(77)
    got = gm(img, t)
    torch.manual_seed(77)
    rects = [OC.cutout_rects(H, W, 32, 8, OC.maua_cutouts_pow(t[[0]].long())) for _ in range(2)]
    want = OC.clip_grads(p, SMALL, img, rects, torch.stack([e0, e1]), OC.normalise_weights([1.0, 3.0]), scale=80.0)
    assert rel(got, want) <= 3e-4
    # the square image's identical whole-image cutouts went through the tower once (merge_identical): the same gradient as the
    # reference's full list (that is what `want` was computed from), and as the unmerged run
    r0, m0 = CLIPGrads.merge_identical(np.asarray(rects, dtype=np.int32))
    assert m0 is not None and r0.shape == (2, 7, 3) and m0.tolist() == [[2.0] + [1.0] * 6] * 2
    gm.merge_cutouts = False
    torch.manual_seed(77)
    assert rel(gm(img, t), got) <= 1e-5
    gm.merge_cutouts = True
    # per-sample prompts: sample 0 -> e0, sample 1 -> e1
    p0, p1 = EmbeddingPrompt(e0), EmbeddingPrompt(e1)
    gm.set_targets_per_sample([p0, p1])
    torch.manual_seed(78)
    got = gm(img, t)
    torch.manual_seed(78)
    rects = [OC.cutout_rects(H, W, 32, 8, OC.maua_cutouts_pow(437)) for _ in range(2)]
    for b, e in enumerate((e0, e1)):
        one = OC.clip_grads(p, SMALL, img[b:b + 1], rects, e[None], torch.ones(1), scale=80.0)
        assert rel(got[b:b + 1], one) <= 3e-4
    # a StylePrompt is embedded through the image tower's own cutouts (t = 0); a TextPrompt needs the text tower
    gm.set_targets([StylePrompt(torch.rand(1, 3, 40, 40, generator=g), weight=2.0)])
    assert tuple(gm.targets[0].shape) == (1, 2 * 8, E) and abs(float(gm.weights.sum()) - 1) < 1e-5
    assert bool(torch.isfinite(gm(img, t)).all())
    with pytest.raises(NotImplementedError):
        gm.set_targets([TextPrompt("a fractal city")])
    gm2 = CLIPGrads(scale=1.0, clip_models=[CLIPImageModel(vt, text_encoder=lambda s: e0[None])], cutout_kwargs=dict(cutn=8), cutout_batches=1)
    gm2.set_targets([TextPrompt("a fractal city")])
    assert torch.equal(gm2.targets[0][0, 0], e0)


def test_text_guided_sampler_loop_graph_equals_step_by_step():
    """configs[3] with text-prompt guidance: GuidedDiffusion(speed "fast") with CLIPGrads as its grad module - the captured loop
    (maua_ddim_guided_loop + maua_unet_set_clip_guide: UNet, secondary model, cutouts, image tower forward + backward, DDIM update, one
    hipGraph) against the step-by-step path under the same seed (same cutouts), twice (the second call replays the graph with new
    rectangles)."""
    from maua_amd.clip import CLIPImageModel
    from maua_amd.diffusion import GuidedDiffusion, SecondaryDiffusionImageNet2, SpacedDiffusion, UNetModel, space_timesteps
    from maua_amd.grad import CLIPGrads, EmbeddingPrompt
    from oracle import diffusion as OD
    vt, _ = _tower(SMALL, torch.bfloat16, seed=2)
    E = SMALL["output_dim"]
    g = torch.Generator().manual_seed(3)
    net = UNetModel(image_size=64, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1, attention_resolutions=(4, 8),
                    channel_mult=(1, 2, 2), num_head_channels=32, use_scale_shift_norm=True, resblock_updown=True, dtype=torch.bfloat16,
                    generator=g)
    sec = SecondaryDiffusionImageNet2(dtype=torch.float32, generator=g, exact=False)
    sd = SpacedDiffusion(space_timesteps(1000, "ddim6"), OD.linear_betas(1000), rescale_timesteps=True)
    prompts = [EmbeddingPrompt(torch.randn(E, generator=g)), EmbeddingPrompt(torch.randn(E, generator=g), 0.5)]
    outs = {}
    for use_graph in (True, False):
        gm = CLIPGrads(scale=500.0, clip_models=[CLIPImageModel(vt)], cutout_kwargs=dict(cutn=8), cutout_batches=2, clamp_gradient=0.05)
        gd = GuidedDiffusion([gm], timesteps=6, model=net, diffusion=sd, speed="fast", secondary_model=sec)
        gd.use_graph = use_graph
        res = []
        for rep in range(2):
            gg = torch.Generator().manual_seed(40 + rep)
            x0, nz = torch.randn(2, 3, 64, 64, generator=gg), torch.randn(2, 3, 64, 64, generator=gg)
            torch.manual_seed(90 + rep)
            res.append(gd.run(x0, prompts, 5, 6, noise=nz).clone())
        outs[use_graph] = res
        if use_graph:
            assert net.guided_graph_active()
    for a, b in zip(outs[True], outs[False]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    assert not torch.equal(outs[True][0], outs[True][1])
    # the guidance does something: the unguided sampler ends elsewhere
    gd0 = GuidedDiffusion([], timesteps=6, model=net, diffusion=sd)
    gg = torch.Generator().manual_seed(40)
    x0, nz = torch.randn(2, 3, 64, 64, generator=gg), torch.randn(2, 3, 64, 64, generator=gg)
    assert not torch.allclose(gd0.run(x0, [], 5, 6, noise=nz), outs[True][0])

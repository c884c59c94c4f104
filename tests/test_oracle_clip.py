"""CPU tests of the text-prompt guidance oracle (oracle/clip.py) against the reference-generated fixture g33 and published
implementations that ARE in the image (torch.nn.MultiheadAttention), and of the product's host-side cutout draws."""
import numpy as np
import torch

from oracle import clip as OC


def test_spherical_dist_loss_matches_the_reference(golden):
    g = golden("g33_cutouts")
    got = OC.spherical_dist_loss(g["sd_x"], g["sd_y"])
    assert got.shape == g["sd_out"].shape and float((got - g["sd_out"]).abs().max()) <= 1e-6
    from maua_amd.grad import spherical_dist_loss
    assert torch.equal(spherical_dist_loss(g["sd_x"], g["sd_y"]), got)


def test_cutout_rectangles_and_outputs_match_the_reference(golden):
    """The reference's MauaCutouts under a seed: the oracle AND the product's host-side draw (maua_amd.grad.cutout_rects /
    MauaCutouts.rects) take the same rectangles from torch's global generator; the oracle's cutouts equal the reference's outputs."""
    from maua_amd.grad import MauaCutouts
    g = golden("g33_cutouts")
    for k in range(5):
        H, W, cs, cutn, t, seed = (int(v) for v in g[f"cut{k}_cfg"])
        want = g[f"cut{k}_rects"].numpy()
        torch.manual_seed(seed)
        rects = OC.cutout_rects(H, W, cs, cutn, OC.maua_cutouts_pow(t))
        assert np.array_equal(np.asarray(rects), want), k
        torch.manual_seed(seed)
        assert np.array_equal(np.asarray(MauaCutouts(cs, cutn).rects(H, W, torch.tensor([float(t)])[[0]].long())), want), k
        if f"cut{k}_img" in g:
            out = OC.cutouts_from_rects(g[f"cut{k}_img"], rects, cs)
            assert float((out - g[f"cut{k}_out"]).abs().max()) <= 1e-6
    assert (g["cut3_rects"].numpy()[:8] == np.array([256, 0, 0])).all()     # cutn // 4 cutouts cover the whole square image


def test_resize_restatement_properties():
    """resize_right's algorithm as restated: identity at equal size, weights sum to one (a constant stays constant away from the zero
    padding), shrinking widens the kernel (taps = ceil(4 / scale)), linear ramps are reproduced in the interior when up-scaling."""
    x = torch.rand(1, 3, 20, 20)
    assert torch.equal(OC.resize(x, (20, 20)), x)
    for (i, o) in ((40, 32), (256, 224), (24, 32), (100, 32)):
        left, w = OC.resize_tables(i, o)
        assert w.shape[1] == (4 if o >= i else int(np.ceil(4 * i / o - 1e-7)))
        assert float((w.sum(1) - 1).abs().max()) <= 1e-6
        c = OC.resize(torch.ones(1, 1, i, i), (o, o))
        m = max(2, int(np.ceil(2 * max(i / o, 1) * o / i)) + 1)
        assert float((c[..., m:-m, m:-m] - 1).abs().max()) <= 1e-5
    ramp = torch.arange(24.0).reshape(1, 1, 1, 24).expand(1, 1, 24, 24)
    up = OC.resize(ramp, (48, 48))
    assert float((up[0, 0, 10, 8:40] - (torch.arange(8, 40) / 2 - 0.25)).abs().max()) <= 1e-4


def test_attention_restatement_matches_torch_multihead_attention():
    """oracle.clip.attention against torch.nn.MultiheadAttention with the same packed parameters (what clip/model.py's
    ResidualAttentionBlock calls): <= 1e-5."""
    g = torch.Generator().manual_seed(0)
    d, heads, N, T = 64, 2, 3, 17
    mha = torch.nn.MultiheadAttention(d, heads)
    with torch.no_grad():
        for p in mha.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    x = torch.randn(N, T, d, generator=g)
    want = mha(x.transpose(0, 1), x.transpose(0, 1), x.transpose(0, 1), need_weights=False)[0].transpose(0, 1)
    got = OC.attention(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, heads)
    assert float((got - want).abs().max()) <= 1e-5


def test_encode_image_and_clip_grads_shapes_and_weight_rules():
    cfg = OC.vit_config(32, 8, 64, 2, 2, 32)
    p = OC.init_vit_params(cfg)
    assert set(p) == set(OC.vit_param_shapes(cfg)) and all(k.startswith("visual.") for k in p)
    x = torch.randn(3, 3, 32, 32)
    assert tuple(OC.encode_image(p, cfg, x).shape) == (3, 32)
    w = OC.normalise_weights([1.0, -3.0])
    assert torch.allclose(w, torch.tensor([0.5, -1.5]))
    try:
        OC.normalise_weights([1.0, -1.0])
        assert False
    except RuntimeError:
        pass
    img = torch.rand(2, 3, 40, 40) * 2 - 1
    torch.manual_seed(0)
    rects = [OC.cutout_rects(40, 40, 32, 8, 1.0)]
    g = OC.clip_grads(p, cfg, img, rects, torch.randn(2, 32), torch.tensor([0.5, 0.5]), scale=10.0)
    assert g.shape == img.shape and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    gc = OC.clip_grads(p, cfg, img, rects, torch.randn(2, 32), torch.tensor([0.5, 0.5]), scale=10.0, clamp_gradient=1e-6)
    assert abs(float(gc.square().mean().sqrt()) - 1e-6) <= 1e-8


def test_product_tower_module_keys_and_loader_rules():
    """maua_amd.clip.VisionTransformer carries CLIP's state-dict keys (the oracle's, minus "visual."); load() refuses towers this build
    does not have and refuses to invent weights unless asked."""
    import pytest
    from maua_amd import clip as CL
    cfg = OC.vit_config(32, 8, 64, 2, 2, 32)
    vt = CL.VisionTransformer(32, 8, 64, 2, 2, 32)
    assert {"visual." + k for k in vt.state_dict()} == set(OC.vit_param_shapes(cfg))
    p = OC.init_vit_params(cfg)
    vt.load_state_dict(p)
    assert torch.equal(vt.state_dict()["proj"], p["visual.proj"])
    with pytest.raises(NotImplementedError):
        CL.load("RN50")
    with pytest.raises(FileNotFoundError):
        CL.load("ViT-B/16")
    with pytest.raises(NotImplementedError):
        CL.load("ViT-B/16", allow_random_init=True)[0].encode_text("x")

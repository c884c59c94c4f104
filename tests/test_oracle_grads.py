"""CPU tests of the image-prompt grad-module oracle (oracle/grads.py) against the reference-generated fixture g34 (the reference's own
differentiable_histogram / ColorMatchGrads / loss functions / Perceptor hooks / DangoCutouts / resample, tests/golden/make_golden.py
``golden_grads``)."""
import numpy as np
import torch

from oracle import clip as OC
from oracle import grads as OG


def test_differentiable_histogram_matches_the_reference(golden):
    g = golden("g34_grads")
    x, w = g["hist_x"], g["hist_w"]
    for got, want in ((OG.differentiable_histogram(x, w, 255), g["hist_out_w"]), (OG.differentiable_histogram(x, None, 255), g["hist_out"]),
                      (OG.differentiable_histogram(x, w, 17), g["hist_out_17"])):
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 2e-7, float((got - want).abs().max())
        assert float((got.sum(-1) - 1).abs().max()) <= 1e-5


def test_colormatch_histogram_and_gradient_match_the_reference(golden):
    g = golden("g34_grads")
    img, style = g["cm_img"], g["cm_style"]
    for sw in (1, 0):
        target = OG.colormatch_histogram(style, bool(sw))
        assert float((target - g[f"cm_target_{sw}"]).abs().max()) <= 2e-7
        assert float((OG.colormatch_histogram(img, bool(sw)) - g[f"cm_hist_{sw}"]).abs().max()) <= 2e-7
        grad, _ = OG.colormatch_grads(img, target, 3.0, bool(sw))
        want = g[f"cm_grad_{sw}"]
        assert float((grad - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-12


def test_loss_functions_match_the_reference(golden):
    g = golden("g34_grads")
    a, b = g["loss_a"], g["loss_b"]
    # the reference folds the batch into the channel axis: [2, 6, 5, 7] -> 12 x 12
    gm = OG.gram_matrix(a.reshape(1, 12, 5, 7))
    assert float((gm - g["gram_a"]).abs().max()) <= 1e-5
    assert abs(float(OG.scaled_mse_loss(gm, b) - g["scaled_mse"])) <= 1e-6 * abs(float(g["scaled_mse"]))
    assert abs(float(OG.feature_loss(gm, b) - g["feature_loss"])) <= 1e-6 * abs(float(g["feature_loss"]))


def test_vgg_grads_match_the_reference_hooks(golden):
    """The reference's Perceptor hooks + get_loss + torch.autograd.grad (VGGGrads.forward) around the restated vgg19.features."""
    g = golden("g34_grads")
    p = OG.init_vgg_params(OG.VGG19_CFG, 29, generator=torch.Generator().manual_seed(int(g["vgg_seed"])))
    targets = OG.kbc_style_embeddings(p, g["vgg_style"])
    for k, t in enumerate(targets):
        if f"vgg_target{k}" in g:
            want = g[f"vgg_target{k}"]
            assert float((t[0] - want).abs().max()) <= 1e-4 * float(want.abs().max())
        else:
            assert float((t[0].diagonal() - g[f"vgg_target{k}_diag"]).abs().max()) <= 1e-4 * float(g[f"vgg_target{k}_diag"].abs().max())
    grad, losses = OG.vgg_grads(p, g["vgg_img"], targets, float(g["vgg_strength"]))
    assert abs(float(losses[0] - g["vgg_loss"])) <= 1e-4 * abs(float(g["vgg_loss"]))
    want = g["vgg_grad"]
    assert float((grad - want).abs().max()) <= 1e-3 * float(want.abs().max())


def test_vgg_plan_indices():
    ops, idx = OG.vgg_plan(OG.VGG19_CFG, 29)
    assert [i for (k, _), i in zip(ops, idx) if k == "conv"][:5] == [1, 3, 6, 8, 11] and idx[-1] == 29 and len(ops) == 17
    assert all(t in idx for t in OG.KBC_STYLE_LAYERS)
    ops16, idx16 = OG.vgg_plan(OG.VGG16_CFG, 29)
    assert all(t in idx16 for t in OG.LPIPS_TAPS) and idx16[-1] == 29
    assert list(OG.vgg_param_shapes(OG.VGG16_CFG, 29))[-2:] == ["28.weight", "28.bias"]


def test_dango_cutouts_match_the_reference(golden):
    """DangoCutouts(skip_augs=True) under a seed: the oracle AND the product's host-side plan draw the same crops from torch's global
    generator; the oracle's cutouts equal the reference's outputs."""
    from maua_amd.grad import DangoCutouts
    g = golden("g34_grads")
    for k in range(4):
        H, W, cs, t, seed, overview, inner = (int(v) for v in g[f"dango{k}_cfg"])
        grey_p = float(g[f"dango{k}_grey_p"])
        torch.manual_seed(seed)
        plan = OG.dango_plan(H, W, cs, overview, inner, grey_p)
        sizes = g[f"dango{k}_sizes"].numpy()
        # the reference resizes the padded square once (+ once more when overview > 4), then every inner crop
        crops = [pl for pl in plan if pl[0] >= 0]
        assert [tuple(s) for s in sizes[-len(crops):]] == [(pl[0], pl[0]) for pl in crops], k
        torch.manual_seed(seed)
        dc = DangoCutouts(cs, skip_augs=True)
        assert dc.plan(H, W, t) == plan, k
        if f"dango{k}_img" in g:
            out = OG.dango_cutouts(g[f"dango{k}_img"], plan, cs, OC.resize)
            want = g[f"dango{k}_out"]
            assert out.shape == want.shape and float((out - want).abs().max()) <= 1e-6, k


def test_normal_cutouts_draws_match_the_reference(golden):
    """Cutouts(skip_augs=True) ("normal", cutouts.py:53-98) under a seed: the product's host-side draw takes the reference's crops of the
    zero-padded image from torch's global generator; the oracle's resize of them equals the reference's outputs."""
    from maua_amd.grad import Cutouts
    g = golden("g34_grads")
    for k in range(2):
        S, cs, cutn, seed = (int(v) for v in g[f"normal{k}_cfg"])
        cu = Cutouts(cs, cutn, skip_augs=True)
        p = cu.pad_of(S)
        torch.manual_seed(seed)
        rects = cu.rects(S + 2 * p, S + 2 * p)
        assert np.array_equal(np.asarray(rects), g[f"normal{k}_rects"].numpy()), k
        img = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(350 + k))
        padded = torch.nn.functional.pad(img, (p,) * 4)
        out = OC.cutouts_from_rects(padded, rects, cs)
        assert float((out - g[f"normal{k}_out"]).abs().max()) <= 1e-6, k


def test_resample_is_the_identity_at_256(golden):
    assert float(golden("g34_grads")["resample256_maxdiff"]) == 0.0


def test_lpips_oracle_properties():
    """The LPIPS restatement: zero distance and zero gradient at the target, non-negative, symmetric."""
    gen = torch.Generator().manual_seed(5)
    p = OG.init_vgg_params(OG.VGG16_CFG, 29, generator=gen)
    lins = OG.init_lpips_lins(gen)
    a = torch.rand(1, 3, 32, 32, generator=gen) * 2 - 1
    b = torch.rand(1, 3, 32, 32, generator=gen) * 2 - 1
    assert float(OG.lpips_distance(p, lins, a, a)) == 0.0
    dab, dba = float(OG.lpips_distance(p, lins, a, b)), float(OG.lpips_distance(p, lins, b, a))
    assert dab > 0 and abs(dab - dba) <= 1e-6 * dab
    grad, d = OG.lpips_grads(p, lins, a, b, 2.0)
    assert grad.shape == a.shape and float(grad.abs().max()) > 0 and abs(float(d) - dab) <= 1e-6 * dab


def test_host_side_plans_of_the_perceptors_and_resample_sizes():
    """Host logic of the image-prompt modules that needs no device: torchvision's features indices -> the library's plan (convolution +
    ReLU = one entry, indexed by the ReLU), the weight-key -> convolution map, and resample's output size (image.py:217-223) against the
    oracle's resample."""
    from maua_amd.ops import resample_size
    from maua_amd.perceptors import LPIPS_TAPS, VGG16_CFG, VGG19_CFG, features_plan
    from oracle import ops as OO
    plan, idx, convs = features_plan(VGG19_CFG, 29)
    assert plan == [64, 64, 0, 128, 128, 0, 256, 256, 256, 256, 0, 512, 512, 512, 512, 0, 512] and idx[-1] == 29
    assert [idx.index(t) for t in OG.KBC_STYLE_LAYERS] == [0, 3, 6, 11, 16] and convs["28"] == 12 and len(convs) == 13
    ops, oidx = OG.vgg_plan(OG.VGG19_CFG, 29)
    assert idx == oidx and [0 if k == "pool" else c for k, c in ops] == plan
    plan16, idx16, _ = features_plan(VGG16_CFG, 29)
    assert all(t in idx16 for t in LPIPS_TAPS) and plan16.count(0) == 4
    for (h, w, size) in ((512, 512, 256), (384, 640, 256), (100, 130, 256), (48, 80, (32, 80))):
        assert tuple(OO.resample(torch.zeros(1, 1, h, w), size).shape[-2:]) == resample_size(h, w, size)

"""RealESRGAN x4 generator (RRDBNet) on the HIP device vs the oracle's PyTorch-CPU restatement of the published
architecture (first slice).  basicsr / realesrgan are un-vendored: parity unpinned; random-init
weights of the right shapes (there is no network for the released checkpoints)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    return 10 * torch.log10(1.0 / ((a - b) ** 2).mean().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dt,blocks", [(torch.float32, 2), (torch.bfloat16, 3)])
def test_rrdbnet_matches_oracle(dt, blocks):
    from maua_amd.super import RRDBNet
    from oracle import super as OSR
    net = RRDBNet(num_block=blocks, dtype=dt, generator=torch.Generator().manual_seed(3))
    p = net.state_dict()
    g = torch.Generator().manual_seed(4)
    for k in p:  # non-trivial biases
        if k.endswith(".bias"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    p["conv_last.bias"] = torch.full((3,), 0.5)  # keep the output inside (0, 1) so that the clamp is not the whole test
    p["conv_last.weight"] = p["conv_last.weight"] * 0.1
    net.load_state_dict(p)
    x = torch.rand(2, 3, 24, 40, generator=g)
    y = net(x).cpu()
    ref = OSR.rrdbnet(p, x, blocks)
    assert y.shape == ref.shape == (2, 3, 96, 160)
    inside = ((ref > 0) & (ref < 1)).float().mean()
    assert 0.02 < float(ref.std()) and float(inside) > 0.6, (float(ref.std()), float(inside))
    if dt == torch.float32:
        assert float((y - ref).abs().max()) <= 2e-5
    else:
        assert _psnr(y, ref) >= 40.0, _psnr(y, ref)
    # the u8 frame of the same call: round(255 y), HWC
    u8 = torch.empty((2, 96, 160, 3), dtype=torch.uint8, device="cuda")
    y2 = torch.empty_like(y, device="cuda")
    net(x, out=y2, rgb8_out=u8)
    assert torch.equal(y2.cpu(), y)
    want = y.mul(255).round().byte().permute(0, 2, 3, 1)
    assert int((u8.cpu().int() - want.int()).abs().max()) <= (1 if dt == torch.bfloat16 else 0)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_full_depth_rrdbnet_matches_oracle(dt):
    """configs[4]'s generator at its released depth (RealESRGAN_x4plus: 23 RRDB blocks = 345 trunk convolutions) on a
    non-square 96 x 136 image -> 384 x 544, against the oracle (a few seconds of CPU)."""
    from maua_amd.super import RRDBNet
    from oracle import super as OSR
    net = RRDBNet(num_block=23, dtype=dt, generator=torch.Generator().manual_seed(7))
    p = net.state_dict()
    g = torch.Generator().manual_seed(8)
    for k in p:
        if k.endswith(".bias"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    p["conv_last.bias"] = torch.full((3,), 0.5)
    p["conv_last.weight"] = p["conv_last.weight"] * 0.1
    net.load_state_dict(p)
    x = torch.rand(1, 3, 96, 136, generator=g)
    y = net(x, clamp=False).cpu()        # the raw output (random weights of this depth leave [0, 1]; RealESRGANer clamps later)
    with torch.no_grad():
        ref = OSR.rrdbnet_raw(p, x, 23)
    assert y.shape == ref.shape == (1, 3, 384, 544) and bool(torch.isfinite(ref).all()) and float(ref.std()) > 1e-3
    scale = float(ref.abs().max())
    if dt == torch.float32:
        assert float((y - ref).abs().max()) <= 1e-4 * scale, float((y - ref).abs().max()) / scale
    else:
        mse = float(((y - ref) ** 2).mean())
        rng = float(ref.max() - ref.min())
        assert 10 * np.log10(rng * rng / mse) >= 38.0, 10 * np.log10(rng * rng / mse)


def test_realesrgan_wrapper_and_render_pipeline():
    """The reference's call shapes (load_model / upscale, realesrgan.py:22-49) and the configs[4] pipeline: StyleGAN2
    frames -> [0,1] -> 4x up-scaler -> u8, per frame on the device.  A missing checkpoint is an error (the reference
    downloads it), not a silent random network."""
    from maua.super.image.models.realesrgan import load_model, upscale
    from maua_amd.stylegan2 import SynthesisNetwork
    with pytest.raises(FileNotFoundError):
        load_model("x4plus-anime")
    with pytest.raises(KeyError):
        load_model("no-such-model", allow_random_init=True)
    model = load_model("x4plus-anime", allow_random_init=True)                       # 6 RRDB blocks, seeded random init
    assert model.model.num_block == 6 and model.pre_pad == 10 and model.tile_size == 0
    net = SynthesisNetwork(64, 64, 3, channel_base=2048, channel_max=64, generator=torch.Generator().manual_seed(0))
    ws = torch.randn(2, net.num_ws, 64, generator=torch.Generator().manual_seed(1))
    frames = net(ws).add(1).div(2).clamp(0, 1)               # [2, 3, 64, 64] in [0, 1]
    outs = list(upscale([f[None] for f in frames], model))
    assert len(outs) == 2 and tuple(outs[0].shape) == (1, 3, 256, 256)
    assert float(outs[0].min()) >= 0.0 and float(outs[0].max()) <= 1.0


@pytest.mark.parametrize("name,tile", [("x4plus-anime", 0), ("x4plus-anime", 24), ("xsx4-animevideo", 0), ("xsx4-animevideo", 20)])
def test_realesrganer_enhance_matches_published_chain(name, tile):
    """RealESRGANer.enhance as the reference calls it - /255, BGR2RGB flip, reflect pre_pad 10 (right, bottom), network on
    the whole image or on tile_pad-padded tiles, crop, clamp, flip back, round - against the oracle's restatement of the
    published code (realesrgan un-vendored: parity unpinned), exact-f32 mode: u8 images differ by at most 1 on <= 0.1 % of
    the values.  The channel flip is folded into the first / last convolution on the device."""
    from maua_amd.super import load_model
    from oracle import super as OSR
    model = load_model(name, dtype=torch.float32, allow_random_init=True, tile=tile)
    p = model.model.state_dict()
    g = torch.Generator().manual_seed(11)
    img = (torch.rand(37, 50, 3, generator=g) * 255).numpy()          # RGB HWC in [0, 255], odd sizes
    got, mode = model.enhance(img)
    assert mode == "RGB" and got.shape == (148, 200, 3) and got.dtype == np.uint8
    if name == "xsx4-animevideo":
        ref_net = lambda x: OSR.srvgg_compact(p, x)
    else:
        ref_net = lambda x: OSR.rrdbnet_raw(p, x, 6)
    want = OSR.realesrganer_enhance(ref_net, img, tile=tile)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-3, (d.max(), (d > 0).mean())
    # the flips matter (a network is not colour-symmetric) and so does the pre-pad
    plain = OSR.realesrganer_enhance(lambda x: ref_net(x[:, [2, 1, 0]])[:, [2, 1, 0]], img, tile=tile)
    assert np.abs(plain.astype(int) - want.astype(int)).max() > 1
    assert np.abs(OSR.realesrganer_enhance(ref_net, img, tile=tile, pre_pad=0).astype(int) - want.astype(int)).max() > 0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_srvgg_compact_matches_oracle(dt):
    """SRVGGNetCompact ("xsx4-animevideo": 16 convolutions + PReLU, PixelShuffle 4, + nearest base): f32 <= 2e-5, bf16
    PSNR >= 40 dB; on the LDS-direct narrow tiles (64 x 32 k) and on the generic kernel (odd sizes)."""
    from maua_amd.super import SRVGGNetCompact
    from oracle import super as OSR
    net = SRVGGNetCompact(dtype=dt, generator=torch.Generator().manual_seed(4))
    p = net.state_dict()
    assert len(p) == 18 * 2 + 17 and tuple(p["body.34.weight"].shape) == (48, 64, 3, 3)
    for shape in [(2, 3, 16, 32), (1, 3, 21, 19)]:
        x = torch.rand(shape, generator=torch.Generator().manual_seed(5))
        want = OSR.srvgg_compact(p, x)
        got = net(x, clamp=False).cpu()
        if dt == torch.float32:
            assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), shape
        else:
            mse = float(((got - want) ** 2).mean())
            rng = float(want.max() - want.min())
            assert 10 * np.log10(rng * rng / mse) >= 40.0, shape
        u8 = torch.empty((shape[0], 4 * shape[2], 4 * shape[3], 3), dtype=torch.uint8, device="cuda")
        net(x, rgb8_out=u8)
        want8 = got.clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)
        assert int((u8.cpu().int() - want8.int()).abs().max()) <= 1


def test_rrdb_trunk_on_lds_direct_kernel_matches_generic_and_oracle(monkeypatch):
    """bf16, image sizes the LDS-direct kernel takes (H % 8 == 0, W % 32 == 0): every 3x3 convolution of the trunk and the
    up-sampling tail runs on modconv_dma's narrow N tiles (32 / 64 output channels, channel-sliced dense-block operands,
    residual in the copy-out) - against the generic MFMA kernel on the same weights and the fp32 oracle."""
    from maua_amd.super import RRDBNet
    from oracle import super as OSR
    g = torch.Generator().manual_seed(8)
    blocks = 2
    ref_net = RRDBNet(num_block=blocks, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(5))
    p = ref_net.state_dict()
    for k in p:
        if k.endswith(".bias"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    p["conv_last.bias"] = torch.full((3,), 0.5)
    p["conv_last.weight"] = p["conv_last.weight"] * 0.1
    x = torch.rand(2, 3, 32, 64, generator=g)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MAUA_RRDB_DMA", mode)
        net = RRDBNet(num_block=blocks, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(5))
        net.load_state_dict(p)
        outs[mode] = net(x).cpu()
        del net
    ref = OSR.rrdbnet(p, x, blocks)
    assert outs["1"].shape == ref.shape == (2, 3, 128, 256)
    assert _psnr(outs["1"], ref) >= 40.0 and _psnr(outs["0"], ref) >= 40.0, (_psnr(outs["1"], ref), _psnr(outs["0"], ref))
    assert _psnr(outs["1"], outs["0"]) >= 50.0, _psnr(outs["1"], outs["0"])   # same bf16 operands, residual rounded once more

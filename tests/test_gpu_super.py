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


def test_realesrgan_wrapper_and_render_pipeline():
    """The reference's call shapes (load_model / upscale, realesrgan.py:22-49) and the configs[4] pipeline: StyleGAN2
    frames -> [0,1] -> 4x up-scaler -> u8, per frame on the device."""
    from maua.super.image.models.realesrgan import load_model, upscale
    from maua_amd.stylegan2 import SynthesisNetwork
    model = load_model("x4plus-anime")                       # 6 RRDB blocks, random init (no checkpoint on disk)
    assert model.model.num_block == 6
    net = SynthesisNetwork(64, 64, 3, channel_base=2048, channel_max=64, generator=torch.Generator().manual_seed(0))
    ws = torch.randn(2, net.num_ws, 64, generator=torch.Generator().manual_seed(1))
    frames = net(ws).add(1).div(2).clamp(0, 1)               # [2, 3, 64, 64] in [0, 1]
    outs = list(upscale([f[None] for f in frames], model))
    assert len(outs) == 2 and tuple(outs[0].shape) == (1, 3, 256, 256)
    assert float(outs[0].min()) >= 0.0 and float(outs[0].max()) <= 1.0
    big = model.model(frames)                                # batched, on-device form of the same thing
    assert float((big[0].cpu() - outs[0][0]).abs().max()) <= 1 / 255 + 1e-3   # upscale() goes through u8 images


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

"""Layer-level modules (maua_amd/modules.py <-> maua/GAN/wrappers/inference/stylegan2.py:29-384) on the C-ABI operator layer:
each class against the oracle's restatement of the same layer, and a hand-stacked network of SynthesisBlocks against the
one-call SynthesisNetwork with the same state dict."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) / max(1e-20, float(b.abs().max()))


def test_fully_connected_and_conv2d_layers():
    from maua_amd.modules import Conv2dLayer, FullyConnectedLayer
    from oracle import stylegan2 as OS
    torch.manual_seed(0)
    x = torch.randn(5, 48)
    fc = FullyConnectedLayer(48, 32, bias_init=1)                        # the affine of a synthesis layer
    want = OS.fully_connected(x, fc.weight.detach(), fc.bias.detach())
    assert rel(fc(x.cuda()), want) <= 2e-6
    sq = FullyConnectedLayer(48, 48, activation="lrelu", lr_multiplier=0.01)       # a mapping layer (quirk Q3: x @ w)
    with torch.no_grad():
        sq.bias.add_(torch.randn(48))
    want = OS.fully_connected(x, sq.weight.detach(), sq.bias.detach(), "lrelu", 0.01)
    assert rel(sq(x.cuda()), want) <= 2e-6
    ref = torch.nn.functional.leaky_relu(x @ (sq.weight.detach() * sq.weight_gain) + sq.bias.detach() * 0.01, 0.2) * math.sqrt(2)
    assert rel(sq(x.cuda()), ref) <= 2e-6
    conv = Conv2dLayer(16, 24, kernel_size=3, activation="lrelu", conv_clamp=3.0)
    with torch.no_grad():
        conv.bias.add_(0.3 * torch.randn(24))
    img = torch.randn(2, 16, 12, 20)
    w = conv.weight.detach() * conv.weight_gain
    ref = torch.nn.functional.conv2d(img, w, padding=1) + conv.bias.detach().view(1, -1, 1, 1)
    ref = (torch.nn.functional.leaky_relu(ref, 0.2) * math.sqrt(2) * 0.7).clamp(-3.0 * 0.7, 3.0 * 0.7)
    assert rel(conv(img.cuda(), gain=0.7), ref) <= 2e-5
    assert tuple(Conv2dLayer(16, 8, 1, bias=False, up=2)(img.cuda()).shape) == (2, 8, 24, 40)
    with pytest.raises(NotImplementedError):
        Conv2dLayer(4, 4, 3, down=2)


@pytest.mark.parametrize("up", [1, 2])
def test_synthesis_and_torgb_layers_match_oracle(up):
    from maua_amd.modules import SynthesisLayer, ToRGBLayer
    from oracle import stylegan2 as OS
    torch.manual_seed(1)
    res = 16
    layer = SynthesisLayer(32, 48, w_dim=24, resolution=res, up=up, conv_clamp=256.0)
    with torch.no_grad():
        layer.bias.add_(0.1 * torch.randn(48))
    x = torch.randn(2, 32, res // up, res // up)
    w = torch.randn(2, 24)
    p = {"l." + k: v.detach() for k, v in layer.state_dict().items()}
    want = OS.synthesis_layer(p, "l", x, w, up=up, noise=p["l.noise_const"])
    assert rel(layer(x.cuda(), w.cuda()), want) <= 2e-5
    assert rel(layer(x.cuda(), w.cuda(), noise_mode="none"), OS.synthesis_layer(p, "l", x, w, up=up, noise=torch.zeros(res, res))) <= 2e-5
    rgb = ToRGBLayer(48, 3, w_dim=24, conv_clamp=256.0)
    with torch.no_grad():
        rgb.bias.add_(0.1 * torch.randn(3))
    pr = {"t." + k: v.detach() for k, v in rgb.state_dict().items()}
    assert rel(rgb(want.cuda(), w.cuda()), OS.torgb_layer(pr, "t", want, w)) <= 2e-5


def test_stacked_blocks_equal_the_synthesis_network():
    """SynthesisBlock by SynthesisBlock (one library call per layer) == SynthesisNetwork (one library call per forward) on the
    same parameters, and both == the oracle."""
    from maua_amd.modules import SynthesisBlock
    from maua_amd.stylegan2 import SynthesisNetwork
    from oracle import stylegan2 as OS
    w_dim, res = 32, 32
    net = SynthesisNetwork(w_dim, res, 3, channel_base=1024, channel_max=64, dtype=torch.float32, generator=torch.Generator().manual_seed(2))
    sd = net.state_dict()
    g = torch.Generator().manual_seed(3)
    for k in sd:
        if k.endswith(".bias") and "affine" not in k:
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    net.load_state_dict(sd)
    ch = OS.channels_dict(res, 1024, 64)
    blocks, prev = [], 0
    for i, r in enumerate(OS.block_resolutions(res)):
        b = SynthesisBlock(prev, ch[r], w_dim=w_dim, resolution=r, img_channels=3, is_last=r == res)
        mine = {k[len(f"bs.{i}."):]: v for k, v in sd.items() if k.startswith(f"bs.{i}.")}
        assert set(mine) == set(b.state_dict()), (set(mine) ^ set(b.state_dict()))
        b.load_state_dict(mine)
        blocks.append(b)
        prev = ch[r]
    ws = torch.randn(2, net.num_ws, w_dim, generator=g)
    x = img = None
    w_idx = 0
    for b in blocks:                                   # SynthesisNetwork.forward, inference/stylegan2.py:429-436
        cur = ws[:, w_idx: w_idx + b.num_conv + b.num_torgb].cuda()
        x, img = b(x, img, cur)
        w_idx += b.num_conv
    want = OS.synthesis_network(sd, ws)
    assert rel(img, want) <= 5e-5
    assert rel(net(ws), want) <= 5e-5
    assert rel(img, net(ws)) <= 5e-5

"""Pin the oracle: every oracle function vs fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
from math import sqrt

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import stylegan2 as S


def close(a, b, tol=1e-6):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a.double() - b.double()).abs().max())
    assert err <= tol * scale, f"max-abs {err} > {tol}*{scale}"


def test_setup_filter(golden):
    g = golden("g01_setup_filter")
    f = O.setup_filter([1, 3, 3, 1])
    assert torch.equal(f, g["f"])
    assert torch.equal(f[0], torch.tensor([1., 3., 3., 1.]) / 64)


def test_bias_act(golden):
    g = golden("g02_bias_act")
    x, b = g["x"], g["b"]
    for act in ["linear", "lrelu"]:
        for gi, gain in enumerate([1.0, sqrt(2)]):
            for ci, clamp in enumerate([None, 2.5]):
                close(O.bias_act(x, b, act=act, gain=gain, clamp=clamp), g[f"y_{act}_g{gi}_c{ci}"])
    close(O.bias_act(x, b, act="relu"), g["y_relu_default"])
    close(O.bias_act(x, None, act="sigmoid"), g["y_sigmoid_nobias"])
    close(O.bias_act(x, b, act="tanh"), g["y_tanh"])
    close(O.bias_act(x, b, act="swish"), g["y_swish"])
    close(O.bias_act(x * 100, b, act="lrelu", gain=sqrt(2), clamp=256.0), g["y_lrelu_clamp256"])


def test_upfirdn2d(golden):
    g = golden("g03_upfirdn2d")
    f = g["f"]
    close(O.upfirdn2d(g["x"], f, up=2, padding=(2, 1, 2, 1), gain=4), g["y_up"])
    close(O.upfirdn2d(g["x17"], f, padding=(1, 1, 1, 1), gain=4), g["y_fir"])
    close(O.upfirdn2d(g["x"], f, down=2, padding=(1, 1, 1, 1)), g["y_down"])
    close(O.upfirdn2d(g["x"], f, padding=(2, -1, -1, 3)), g["y_crop"])
    close(O.upfirdn2d(g["xr"], f, up=2, padding=(2, 1, 2, 1), gain=4), g["y_rect"])


def test_upsample2d(golden):
    g = golden("g04_upsample2d")
    close(O.upsample2d(g["x"], g["f"]), g["y"])


def test_modconv_up1(golden):
    g = golden("g05_modconv_up1")
    close(O.modulated_conv2d(g["x"], g["w3"], g["s"], noise=g["noise"], up=1, padding=1), g["y_demod"], 2e-6)
    close(O.modulated_conv2d(g["x"], g["w3"], g["s"], up=1, padding=1), g["y_demod_nonoise"], 2e-6)
    close(O.modulated_conv2d(g["x"], g["w1"], g["s"], demodulate=False), g["y_1x1"], 2e-6)


def test_modconv_up2(golden):
    g = golden("g06_modconv_up2")
    y = O.modulated_conv2d(g["x"], g["w3"], g["s"], noise=g["noise"], up=2, padding=1, resample_filter=g["f"])
    close(y, g["y"], 2e-6)


def test_norm2nd_fc_mapping(golden):
    g = golden("g07_norm2nd")
    close(O.normalize_2nd_moment(g["z"]), g["y"])
    g = golden("g07_fc_linear")
    close(S.fully_connected(g["x"], g["weight"], g["bias"]), g["y"])
    g = golden("g07_mapping")
    p = {k.replace("__", "."): v for k, v in g.items() if k.startswith("fcs") or k == "w_avg"}
    close(S.mapping_network(p, g["z"], 1.0, num_ws_=6), g["y_psi1"], 2e-6)
    close(S.mapping_network(p, g["z"], 0.7, num_ws_=6), g["y_psi07"], 2e-6)


def test_mapping_init_order(golden):
    g = golden("g07_mapping512")
    p = S.init_mapping_params(generator=torch.Generator().manual_seed(11))
    assert np.isclose(p["fcs.0.weight"].double().sum().item(), g["w0_sum"], rtol=0, atol=1e-6 * abs(g["w0_sum"]) + 1e-3)
    assert np.isclose(p["fcs.7.weight"].double().sum().item(), g["w7_sum"], rtol=0, atol=1e-6 * abs(g["w7_sum"]) + 1e-3)
    close(S.mapping_network(p, g["z"])[:, 0], g["w"], 1e-5)


def test_synth_layer_and_torgb(golden):
    g = golden("g08_synth_layer")
    p = {"L." + k.replace("__", "."): v for k, v in g.items() if k not in ("x", "w", "y")}
    close(S.synthesis_layer(p, "L", g["x"], g["w"], up=1), g["y"], 2e-6)
    g = golden("g08_torgb")
    p = {"R." + k.replace("__", "."): v for k, v in g.items() if k not in ("x", "w", "y")}
    close(S.torgb_layer(p, "R", g["x"], g["w"]), g["y"], 2e-6)


def test_synth_init_order(golden):
    g = golden("g08_synth_init")
    p = S.init_synthesis_params(32, w_dim=16, channel_base=256, channel_max=16,
                                generator=torch.Generator().manual_seed(21))
    ref = {k.replace("__", "."): v for k, v in g.items()}
    assert set(ref) == set(p)
    for k in ref:
        assert torch.equal(ref[k], p[k]), k
    g = golden("g08_synth_init1024_sums")
    p = S.init_synthesis_params(1024, generator=torch.Generator().manual_seed(22))
    assert int(g["num_ws"]) == S.num_ws(1024) == 18
    for k, v in g.items():
        if k == "num_ws":
            continue
        s = p[k.replace("__", ".")].double().sum().item()
        assert abs(s - float(v)) <= 1e-9 * max(1.0, abs(float(v))), k

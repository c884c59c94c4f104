"""Pin the oracle: every oracle function vs fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import math
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


def test_fp16_ops(golden):
    """g30: the reference's operator layer on float16 tensors, incl. the FP16 pre-normalisation branch (ops.py:161-165).  Same
    functions, same dtype, same CPU kernels underneath: the oracle reproduces the reference to the bit."""
    g = golden("g30_fp16_ops")
    h = torch.float16
    assert g["x"].dtype == h and g["y_demod"].dtype == h
    y = O.modulated_conv2d(g["x"], g["w3"], g["s"], noise=g["noise"], up=1, padding=1)
    assert y.dtype == h and torch.equal(y, g["y_demod"])
    s_small = (g["s"].float() / 300).to(h)
    assert torch.equal(O.modulated_conv2d(g["x"], g["w3"], s_small, up=1, padding=1), g["y_small"])
    # the branch is what keeps this case finite: |x * s| exceeds the half range
    assert float((g["x"].float().abs().amax() * g["s"].float().abs().amax())) > 65504.0
    assert torch.equal(O.bias_act(g["xb"], g["b"], act="lrelu", gain=sqrt(2), clamp=256.0), g["y_ba"])
    assert torch.equal(O.upfirdn2d(g["xu"], g["f"].to(h), up=2, padding=[2, 1, 2, 1], gain=4), g["y_up"])


def test_modconv_up2(golden):
    g = golden("g06_modconv_up2")
    y = O.modulated_conv2d(g["x"], g["w3"], g["s"], noise=g["noise"], up=2, padding=1, resample_filter=g["f"])
    close(y, g["y"], 2e-6)


def test_offpath_operator_arguments(golden):
    """g31: conv2d_resample at paddings other than k // 2 (the reference's own call) and up-layers with other filters / factors
    (composed of the reference's pieces like g06) - the oracle's restatement of ops.py:189-233 covers them as written."""
    g = golden("g31_offpath_ops")
    for p in (0, 2, 3):
        close(O.conv2d_resample(g["x"], g["w3"], padding=p), g[f"y3_p{p}"], 2e-6)
    close(O.conv2d_resample(g["x"], g["w1"], padding=2), g["y1_p2"], 2e-6)
    close(O.conv2d_resample(g["xg"], g["wg"], padding=0, groups=3), g["yg_p0"], 2e-6)
    for name, up in (("f121_up2", 2), ("f11_up2", 2), ("f14641_up2", 2), ("f1331_up4", 4), ("f8_up4", 4), ("f6_up3", 3)):
        y = O.modulated_conv2d(g["x"], g["w3"], g["s"], up=up, padding=1, resample_filter=g["f_" + name])
        close(y, g["y_" + name], 2e-6)


def test_norm2nd_fc_mapping(golden):
    g = golden("g07_norm2nd")
    close(O.normalize_2nd_moment(g["z"]), g["y"])
    g = golden("g07_fc_linear")
    close(S.fully_connected(g["x"], g["weight"], g["bias"]), g["y"])
    g = golden("g07_mapping")
    p = {k.replace("__", "."): v for k, v in g.items() if k.startswith("fcs") or k == "w_avg"}
    close(S.mapping_network(p, g["z"], 1.0, num_ws_=6), g["y_psi1"], 2e-6)
    close(S.mapping_network(p, g["z"], 0.7, num_ws_=6), g["y_psi07"], 2e-6)
    close(S.mapping_network(p, g["z"], 0.5, num_ws_=6, truncation_cutoff=2), g["y_cut2"], 2e-6)


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


# ------------------------------------------------------------------------------------------ audio chain
from oracle import audio as A
from oracle import io as OIO
from oracle import latent as OL
from oracle import noise as ON
from oracle import quantile as OQ
from oracle import signal as OSG


def test_stft_mel_onset(golden):
    a = golden("g09_audio_clip")["audio"]
    sr = 30720
    g = golden("g09_stft")
    D = A.stft(a)
    assert D.shape[1] == int(g["n_cols"]) == 1 + len(a) // 1024
    cols = g["cols"].long()
    close(D.real[:, cols], g["D_re"], 2e-6)
    close(D.imag[:, cols], g["D_im"], 2e-6)
    S1 = A.spectrogram(a)
    assert list(S1.shape) == g["S1_shape"].tolist()  # frame count = N // 1024 (int-exact)
    close(S1[:, cols[:-1]], g["S1_cols"], 2e-6)
    g = golden("g09_mel")
    close(A.mel_frequencies(130), g["mel_f"], 1e-6)
    basis = A.mel_basis(sr, fmax=11025.0)
    close(basis.sum(1), g["basis_rowsum"], 1e-5)
    close(basis.sum(0), g["basis_colsum"], 1e-5)
    close(basis[[0, 1, 63, 127]], g["basis_rows"], 1e-6)
    M = A.melspectrogram(a, sr, fmax=11025.0)
    close(M[:, cols[:-1]], g["M_cols"], 1e-5)
    db = A.power_to_db(M)
    close(db[:, cols[:-1]], g["db_cols"], 1e-5)
    env = A.onset_strength(a, sr)
    assert env.shape == g["env"].shape
    assert float(env[0]) == 0.0 and float(env[1]) == 0.0  # the 2-frame left pad (bin <-> frame mapping)
    close(env, g["env"], 2e-5)


def test_hpss_istft_onsets_rms(golden):
    a = golden("g09_audio_clip")["audio"]
    g = golden("g10_hpss")
    a1 = a[: int(g["n"])].contiguous()
    D = A.stft(a1)
    mag = D.abs()
    med_t = A.median_filter2d(mag[None, None], (1, 31), (15, 15, 0, 0))[0, 0]
    med_f = A.median_filter2d(mag[None, None], (31, 1), (0, 0, 15, 15))[0, 0]
    close(med_t[:, [0, 7, 30]], g["med_t_f32_cols"], 2e-6)
    close(med_f[:, [0, 7, 30]], g["med_f_f32_cols"], 2e-6)
    close(med_t, g["med_t"].float(), 2e-3)  # stored as f16
    Hh, Hp = A.hpss(D, margin=8.0)
    close(Hp.real[:, [0, 7, 30]], g["Hp_re"], 1e-5)
    close(Hp.imag[:, [0, 7, 30]], g["Hp_im"], 1e-5)
    close(Hh.real[:, [0, 7, 30]], g["Hh_re"], 1e-5)
    close(Hh.imag[:, [0, 7, 30]], g["Hh_im"], 1e-5)
    H1, P1 = A.hpss(D, margin=1.0)
    close(P1.real[:, [0, 7, 30]], g["P1_re"], 1e-5)
    close(H1.real[:, [0, 7, 30]], g["H1_re"], 1e-5)
    close(A.percussive(a1), g["perc"], 1e-5)
    close(A.harmonic(a1), g["harm"], 1e-5)
    g = golden("g10_onsets_rms")
    close(A.onsets(a, 30720), g["onsets"], 5e-5)
    close(A.rms(a), g["rms"], 1e-6)


def test_processing(golden):
    g = golden("g11_processing")
    e, e2, e4, short = g["e"], g["e2"], g["e4"], g["short"]
    for sg in [1, 2, 5]:
        close(A.gaussian_filter(e, sg), g[f"p_circ_s{sg}"], 1e-6)
        close(A.gaussian_filter(e, sg, mode="reflect"), g[f"p_refl_s{sg}"], 1e-6)
    close(A.gaussian_filter(e2, 2), g["p_2d_s2"], 1e-6)
    close(A.gaussian_filter(e4, 1), g["p_4d_s1"], 1e-6)
    close(A.gaussian_filter(short, 2), g["p_short_s2"], 1e-6)
    close(A.normalize(e2), g["p_normalize"], 1e-6)
    close(A.standardize(e), g["p_standardize"], 1e-6)
    g2 = golden("g11_salience")
    close(A.normalize(A.salience_weighted(A.gaussian_filter(g2["env"], 2))), g2["feat"], 1e-5)


def test_quantile_c_vs_reference(golden):
    g = golden("g11_processing")
    qs = [0.025, 0.25, 0.5, 0.75, 0.975]
    e = g["e"]
    big = torch.randn(100001, generator=torch.Generator().manual_seed(5))
    # regenerate the 'big' vector the golden used: it followed e, e2, e4, short draws from the same generator
    gg = torch.Generator().manual_seed(5)
    torch.rand(200, generator=gg); torch.rand(200, 3, generator=gg); torch.rand(40, 2, 3, 4, generator=gg)
    torch.rand(6, 2, generator=gg)
    big = torch.randn(100001, generator=gg)
    assert torch.equal(big[:4], g["big_seed_check"])
    withnan = e.clone()
    withnan[::7] = float("nan")
    for i, q in enumerate(qs):
        assert OQ.quantile(e, q).item() == g["q_small"][i].item()      # bit-exact (order statistics + midpoint)
        assert OQ.quantile(big, q).item() == g["q_big"][i].item()
        assert OQ.quantile(withnan, q).item() == g["q_nan"][i].item()
    assert np.isnan(OQ.quantile(torch.tensor([float("nan")]), 0.5).item())
    # float32-q rounding moves the order-statistic pair away from torch.quantile's at the tails
    v, lo, hi = OQ.quantile_with_indices(big, 0.975)
    assert (lo, hi) == (int(np.float64(np.float32(0.975)) * 100000), int(np.ceil(np.float64(np.float32(0.975)) * 100000)))


def test_quantile_c_vs_built_reference():
    from oracle.build_ref import load_reference_quantile
    ext = load_reference_quantile()
    if ext is None:
        pytest.skip("oracle/_ref not built (reference not mounted)")
    g = torch.Generator().manual_seed(77)
    for n in [1, 2, 3, 10, 1001, 65536]:
        x = torch.randn(n, generator=g)
        for q in [0.0, 0.01, 0.025, 0.3, 0.5, 0.75, 0.975, 1.0]:
            ref = ext._efficient_quantile(x, torch.FloatTensor([q]), True, 3).squeeze().item()
            assert OQ.quantile(x, q).item() == ref, (n, q)


def test_signal(golden):
    g = golden("g11_signal")
    e, e2 = g["e"], g["e2"]
    close(OSG.gaussian_filter(e, 2), g["s_circ_s2"], 1e-6)
    close(OSG.gaussian_filter(e, 2, causal=0), g["s_causal0_s2"], 1e-6)
    close(OSG.gaussian_filter(e, 2, causal=0.5), g["s_causal05_s2"], 1e-6)
    close(OSG.gaussian_filter(e2, 5, mode="reflect"), g["s_refl_s5"], 1e-6)
    close(OSG.percentile_clip(e.clone(), 95), g["s_percentile_clip95"], 1e-6)
    close(OSG.percentile_clip(e2.clone(), 80), g["s_percentile_clip80_2d"], 1e-6)
    assert np.float32(OSG.percentile(e, 50)) == g["s_percentile_50"]  # k-th value: bit-exact
    assert np.float32(OSG.percentile(e, 95)) == g["s_percentile_95"]
    assert OSG.percentile_index(200, 50) == 1 + round(0.5 * 199) == 101  # half-even: 99.5 -> 100
    close(OSG.resample(e, 333), g["s_resample_1d"], 1e-6)
    close(OSG.resample(e2, 77), g["s_resample_2d"], 1e-6)
    close(OSG.normalize(e2), g["s_normalize"], 1e-6)


def test_latents(golden):
    g = golden("g12_latents")
    y, env, envs = g["y"], g["env"], g["envs"]
    close(OL.slerp_loops(y, 64, 2), g["slerp_loops"], 2e-6)
    close(OL.single_weighted(y[0], y[1], env), g["single_weighted"], 1e-6)
    close(OL.multi_weighted(y, envs), g["multi_weighted"], 2e-6)
    assert torch.equal(OL.select_modulo_indices(len(y), env), g["select_modulo_idx"])  # bit-exact indices
    close(OL.select_modulo(y, env), g["select_modulo"], 1e-6)
    g = golden("g12_spline")
    close(OL.spline_loops(g["y"], 50, 3).double(), g["classic_size50_loops3"], 2e-6)
    close(OL.spline_loop_latents(g["y"], 50, 2.5).double(), g["selfsup_size50_loops2p5"], 2e-6)
    g = golden("g12_seeds")
    z = OIO.get_z_latents("0-3,7")
    assert z.dtype == torch.float64 and z.shape == (4, 512)
    assert torch.equal(z[:, :8], g["z"])  # MT19937 values bit-exact
    assert OIO.parse_seeds("0-3,7") == [0, 1, 2, 7]


def test_noise(golden):
    g = golden("g13_noise")
    close(ON.loop(g["loop_noise"], g["loop_idx"], 0, 16, 5), g["loop_y_0_16"], 2e-6)
    close(ON.loop(g["loop_noise"], g["loop_idx"], 40, 8, 5), g["loop_y_40_8"], 2e-6)
    mod = g["mod"]
    bl = ON.blend(g["blend_noise"], mod, 8, 4)
    mu = ON.multiply(g["mul_noise"], mod, 8, 4)
    close(bl, g["blend_y"], 2e-6)
    close(mu, g["mul_y"], 2e-6)
    lp = ON.loop(g["loop_noise"], g["loop_idx"], 8, 4, 5)
    close(ON.average(lp, mu), g["avg_y"], 2e-6)
    md = ON.modulate(lp, mu, mod, 8, 4)
    close(md, g["modulate_y"], 2e-6)
    close(ON.scale_bias(md, 0.7, 0.1), g["scalebias_y"], 2e-6)


PATCH_KEYS = ("patch_type", "seq_feat", "merge_type", "merge_depth")


def _patch_features(g):
    return {k[5:]: g[k] for k in g if k.startswith("feat_")}


def test_latent_patch_graphs(golden):
    """L6: selfsupervised/latent.py:16-80 on explicit selections (fixture = the reference's latent_patch run here)."""
    g = golden("g20_patches")
    feats = _patch_features(g)
    segs = {(k, 4): g["seg"] for k in feats}
    for i, case in enumerate(g["cases"]):
        kw = dict(zip(PATCH_KEYS, str(case).split("|")))
        out = OL.latent_patch(g[f"perm{i}"], g["base"], g["palette"], segs, feats, tempo=120.0, fps=24, segments=4,
                              loop_bars=4, seq_feat_weight=0.8, mod_feat="rms", mod_feat_weight=0.6, **kw)
        close(out, g[f"lat{i}"], 2e-6)


def patch_subs(g):
    return [dict(eval(str(s))) for s in g["subs"]]  # repr(sorted(dict.items())) written by make_golden.py


def test_noise_patch_graphs(golden):
    """N-2: selfsupervised/noise.py:89-140 - three stacked sub-patches over the 17 base Loop modules."""
    g = golden("g20_patches")
    feats = _patch_features(g)
    T_ = len(g["base"])
    sizes = [tuple(int(v) for v in s) for s in g["nsizes"]]
    idx = torch.linspace(0, 2 * 2 * torch.pi, T_)
    layers = (0, 7, 13, 16)
    noise = [(lambda l: lambda i, b: ON.loop(g[f"nbase_planes{l}"], idx, i, b, 3 + l % 4))(l) for l in range(17)]
    planes = {l: iter([g[k] for k in sorted((k for k in g if k.startswith(f"nplanes{l}_")),
                                            key=lambda k: int(k.split("_")[1]))][1:]) for l in layers}
    # layer 0 ends on an "overwrite" sub-patch: only that last module survives in the reference's graph (its planes are
    # entry 0 of the fixture); the blend module it replaces gets placeholder planes
    planes[0] = iter([torch.zeros(2, 4, *sizes[0]), g["nplanes0_0"]])
    for sub in patch_subs(g):
        noise = ON.noise_patch(planes, noise, sizes, feats, 120.0, 24, only=layers, **sub)
    for l in layers:
        close(noise[l](5, 6), g[f"ny{l}"], 3e-6)


def test_tensor2bytes(golden):
    g = golden("g14_tensor2bytes")
    assert np.array_equal(OIO.tensor2bytes(g["img"]), g["bytes"].numpy())
    g = golden("g26_tensor2bytes_ranges")
    for k in range(4):
        mn, mx = g[f"range{k}"].tolist()
        assert np.array_equal(OIO.tensor2bytes(g[f"img{k}"], (mn, mx)), g[f"bytes{k}"].numpy()), k


def test_features_n3(golden):
    """SURVEY 8(f) N3 first batch: oracle restatements vs the reference's own outputs (g16)."""
    from maua_amd.pipeline import synthetic_audio
    g = golden("g16_features")
    a = golden("g09_audio_clip")["audio"]
    sr = int(g["sr"])
    close(A.dct(g["dct_in"]), g["dct_none"], 1e-6)
    close(A.dct(g["dct_in"], norm="ortho"), g["dct_ortho"], 1e-6)
    close(A.emphasize(g["emph_in"], 10, 50), g["emph_10_50"], 1e-6)
    close(A.emphasize(g["emph_in"], 3, 80), g["emph_3_80"], 1e-6)
    close(A.mfcc(a, sr), g["mfcc"], 2e-5)
    close(A.spectral_flatness(a), g["flatness"], 2e-5)
    close(A.spectral_contrast(a, sr), g["contrast"], 2e-4)  # dB of near-zero valley bins amplifies 1e-6 STFT differences
    close(A.spectral_contrast(a, sr, linear=True), g["contrast_linear"], 2e-5)
    close(A.tonnetz_from_chroma(g["chroma"]), g["tonnetz"], 1e-6)
    a12 = synthetic_audio(int(g["n12"]), sr, int(g["seed12"]))
    close(A.drop_strength(a12), g["drop_strength"], 2e-5)


def test_processing_clamps_and_filters(golden):
    """processing.py clamp_*_percentile vs the reference's outputs; the Butterworth passes vs the reference's scipy call (g27);
    closed forms of the published biquads / contrast (torchaudio un-vendored: parity unpinned)."""
    g = golden("g27_processing")
    for name, fn, arg in [("peaks_1d_90", A.clamp_peaks_percentile, 90), ("upper_1d_75", A.clamp_upper_percentile, 75),
                          ("lower_1d_30", A.clamp_lower_percentile, 30)]:
        assert torch.equal(fn(g["e1"], arg), g[name]), name
    for name, fn, arg in [("peaks_3_50", A.clamp_peaks_percentile, 50), ("upper_3_20", A.clamp_upper_percentile, 20),
                          ("lower_3_95", A.clamp_lower_percentile, 95)]:
        assert torch.equal(fn(g["e3"], arg), g[name]), name
    y, sr = g["y"].numpy(), int(g["sr"])
    assert np.array_equal(A.butter_pass(y, sr, 200, "low"), g["low_200_12"].numpy())
    assert np.array_equal(A.butter_pass(y, sr, 100, "low", 24), g["low_100_24"].numpy())
    assert np.array_equal(A.butter_pass(y, sr, 3000, "high"), g["high_3000_12"].numpy())
    assert np.array_equal(A.butter_pass(y, sr, [200, 3000], "band"), g["band_200_3000_12"].numpy())
    # biquads: unit DC gain of the low pass, zero DC gain of the high pass, -3 dB at the corner (Q = 0.707)
    n = 1 << 15
    t = torch.arange(n) / 16000.0
    assert abs(float(A.low_pass(torch.full((n,), 0.5), 16000, 300)[-1]) - 0.5) < 1e-6
    assert abs(float(A.high_pass(torch.full((n,), 0.5), 16000, 300)[-1])) < 1e-6
    tone = 0.5 * torch.sin(2 * math.pi * 300 * t)
    for f in (A.low_pass, A.high_pass):
        amp = f(tone, 16000, 300)[n // 2:].abs().max()
        assert abs(float(amp) / 0.5 - 10 ** (-3 / 20)) < 2e-3
    x = torch.linspace(-1, 1, 101)
    c = A.contrast_enhance(x, 75)
    assert float(c[0]) == pytest.approx(-1, abs=1e-6) and abs(float(c[50])) < 1e-6 and float(c[-1]) == pytest.approx(1, abs=1e-6)
    assert torch.allclose(A.contrast_enhance(x, 0), torch.sin(x * (math.pi / 2)))


def test_resample(golden):
    """maua/ops/image.py resample (SURVEY 8(f) N2 post-process): oracle vs the reference's outputs."""
    g = golden("g17_resample")
    x = g["x"]
    close(O.resample(x, (16, 24)), g["down"], 1e-6)
    close(O.resample(x, (9, 30)), g["down_h"], 1e-6)
    close(O.resample(x, (25, 40)), g["up"], 1e-6)
    close(O.resample(x, (28, 17)), g["mixed"], 1e-6)
    close(O.resample(x, 12), g["short12"], 1e-6)


def test_pulse(golden):
    """rosa/beat.py plp / features pulse: oracle vs the reference (g18, 40 s clip rebuilt from its seed)."""
    from maua_amd.pipeline import synthetic_audio
    g = golden("g18_pulse")
    sr = int(g["sr"])
    a = synthetic_audio(int(g["n"]), sr, int(g["seed"]))
    close(A.onset_strength(A.percussive(a), sr, aggregate="median"), g["env_median"], 2e-4)
    close(A.pulse(a, sr), g["pulse"], 2e-3)


def test_pulse_short_clips(golden):
    """clips shorter than the tempogram window: the transform length is the envelope length (even 300, odd 277) - g24."""
    from maua_amd.pipeline import synthetic_audio
    g = golden("g24_pulse_short")
    a = synthetic_audio(int(g["n"]), int(g["sr"]), int(g["seed"]))
    for tag in ("even", "odd"):
        p = A.pulse(a[: int(g[f"n_{tag}"])], int(g["sr"]))
        assert p.shape == g[f"pulse_{tag}"].shape
        close(p, g[f"pulse_{tag}"], 2e-3)


def test_cqt_pieces(golden):
    """N3 constant-Q chain: every piece the reference can run here (g21, make_golden.py:golden_cqt)."""
    from oracle import cqt as OC
    g = golden("g21_cqt")
    y = golden("g09_audio_clip")["audio"]
    sr = 30720
    fmin = torch.tensor(OC.C1_HZ).float()
    top = OC.cqt_frequencies(252, fmin, 36)[-36:]
    close(top, g["top_freqs"], 1e-6)
    filters, lengths = OC.constant_q(sr, top.min(), 36, 36)
    close(lengths, g["lengths"], 1e-6)
    close(torch.view_as_real(filters[[0, 17, 35]]), g["filt_rows"], 1e-5)
    close(filters.abs().sum(1), g["filt_abs_sum"], 1e-5)
    basis, n_fft, _ = OC.cqt_filter_fft(sr, top.min(), 36, 36)
    assert n_fft == int(g["n_fft"]) == 1024
    assert torch.equal((basis != 0).sum(1), g["basis_nnz"])          # the sparsity pattern: integer-exact
    close(torch.view_as_real(basis[[0, 17, 35]]), g["basis_rows"], 1e-5)
    close(basis.abs().sum(1), g["basis_abs_sum"], 1e-5)
    resp = basis @ OC.stft_rect(y, n_fft, 1024)[:, :-1]
    close(torch.view_as_real(resp), g["resp"], 2e-5)
    pitch, mag = OC.piptrack(y, sr)
    nz = torch.nonzero(pitch)
    assert torch.equal(nz, g["pitch_idx"])                            # which bins are peaks: exact
    close(pitch[nz[:, 0], nz[:, 1]], g["pitch_val"], 1e-6)
    close(mag[nz[:, 0], nz[:, 1]], g["mag_val"], 1e-6)
    assert abs(float(OC.estimate_tuning(y, sr, bins_per_octave=36)) - float(g["tuning"])) < 1e-7
    assert torch.equal(OC.cq_to_chroma(252, 36, 12), g["cq_to_chroma"])
    close(OC.constant_q_lengths(sr, fmin, 252, 36), g["lengths_full"], 1e-6)
    # the un-vendored pieces, against independent statements of what they are: the resampler halves a band-limited tone
    # (amplitude kept, frequency kept), the quantiser's spline passes through its knots
    t = torch.arange(8192) / 8192.0
    tone = torch.sin(2 * torch.pi * 200 * t)
    half = OC.resample(tone, 8192, 4096)
    want = torch.sin(2 * torch.pi * 200 * torch.arange(4096) / 4096.0)
    assert half.shape == (4096,) and float((half[64:-64] - want[64:-64]).abs().max()) < 2e-3
    xs, coef = OC.quantiser_coeffs()
    _, ys = OC.quantiser_knots()
    close(torch.from_numpy(coef[0]), ys[:-1], 1e-6)
    out = OC.chromagram(y, sr)
    assert out.shape == (len(y) // 1024, 12) and bool(torch.isfinite(out).all())
    close(out.norm(dim=1), torch.ones(len(out)), 1e-5)                # CENS rows are L2-normalised


def test_cqt_host_setup_matches_oracle_and_fixture(golden):
    """the host-built constants of maua_amd/cqt.py (no device needed): filter FFT basis against the reference fixture, the
    half-band resampling taps and the quantiser's spline rows against the oracle's restatements."""
    from maua_amd import cqt as Q
    from oracle import cqt as OC
    g = golden("g21_cqt")
    basis, n_fft, lengths = Q.cqt_filter_fft(30720, float(g["top_freqs"].min()), 36, 36)
    assert n_fft == 1024 and torch.equal((basis != 0).sum(1), g["basis_nnz"])
    close(torch.view_as_real(basis[[0, 17, 35]]), g["basis_rows"], 1e-5)
    close(lengths, g["lengths"], 1e-6)
    assert torch.equal(Q.cq_to_chroma(252, 36, 12), g["cq_to_chroma"])
    taps, width = Q._kaiser_half_band()
    k, w, _, _ = OC.sinc_resample_kernel(2, 1)
    assert width == w and taps.numel() == k.numel()
    close(taps, k.reshape(-1), 1e-6)
    xs, coef = Q._quantiser()
    oxs, ocoef = OC.quantiser_coeffs()
    close(xs, oxs, 0)
    close(coef, torch.as_tensor(ocoef, dtype=torch.float32), 1e-6)


def test_vqt_pieces(golden):
    """variable-Q (gamma != 0, constantq.py:29-115): filter lengths, sparsified basis and the top octave's response of the
    oracle AND the host set-up of maua_amd/cqt.py against the reference fixture g23 (make_golden.py:golden_vqt)."""
    from maua_amd import cqt as Q
    from oracle import cqt as OC
    g = golden("g23_vqt")
    y = golden("g09_audio_clip")["audio"]
    sr = 30720
    fmin = torch.tensor(OC.C1_HZ).float()
    for tag in ("erb", "g5"):
        bpo, gamma = int(g[f"{tag}_bpo"]), float(g[f"{tag}_gamma"])
        top = OC.cqt_frequencies(7 * bpo, fmin, bpo)[-bpo:]
        for M in (OC, Q):
            basis, n_fft, lengths = M.cqt_filter_fft(sr, top.min(), bpo, bpo, 1, 0.01, gamma)
            assert n_fft == int(g[f"{tag}_n_fft"])
            close(lengths, g[f"{tag}_lengths"], 1e-6)
            assert torch.equal((basis != 0).sum(1), g[f"{tag}_basis_nnz"])
            close(torch.view_as_real(basis[[0, bpo // 2, bpo - 1]]), g[f"{tag}_basis_rows"], 1e-5)
            close(basis.abs().sum(1), g[f"{tag}_basis_abs_sum"], 1e-5)
            close(M.constant_q_lengths(sr, fmin, 7 * bpo, bpo, 1, gamma), g[f"{tag}_lengths_full"], 1e-6)
        basis, n_fft, _ = OC.cqt_filter_fft(sr, top.min(), bpo, bpo, 1, 0.01, gamma)
        close(torch.view_as_real(basis @ OC.stft_rect(y, n_fft, 1024)[:, :-1]), g[f"{tag}_resp"], 2e-5)
    # gamma = None is the ERB default; gamma = 0 is the CQT
    assert abs(float(g["erb_gamma"]) - 24.7 * (2.0 ** (1.0 / 36) - 1) / 0.108) < 1e-12
    a = OC.vqt(y, sr, 1024, n_bins=72, gamma=0, bins_per_octave=12)
    assert torch.equal(a, OC.cqt(y, sr, 1024, n_bins=72, bins_per_octave=12))
    v = OC.vqt(y, sr, 1024, n_bins=252, bins_per_octave=36)
    assert v.shape == (252, len(y) // 1024) and bool(torch.isfinite(torch.view_as_real(v)).all())


def test_sinc_resample_oracle_properties():
    """oracle.audio.sinc_resample = torchaudio.functional.resample's published algorithm (un-vendored: parity unpinned) -
    checked by what it must do: length ceil(n new / orig), a band-limited tone survives with its amplitude and frequency,
    and the kaiser / 2:1 case equals the constant-Q chain's own half-band restatement (oracle/cqt.py)."""
    from oracle import audio as OA
    from oracle import cqt as OC
    t = torch.arange(44100) / 44100.0
    tone = torch.sin(2 * torch.pi * 1000 * t)
    out = OA.sinc_resample(tone, 44100, 30720)
    ref = torch.sin(2 * torch.pi * 1000 * torch.arange(30720) / 30720.0)
    assert out.shape == (30720,) and float((out[64:-64] - ref[64:-64]).abs().max()) < 2e-3
    x = torch.randn(4097, generator=torch.Generator().manual_seed(1))
    close(OA.sinc_resample(x, 2, 1, resampling_method="sinc_interp_kaiser"), OC.resample(x, 2, 1), 1e-6)
    assert OA.sinc_resample(x, 8000, 10240).shape == (int(np.ceil(10240 * 4097 / 8000)),)


def _segment_inputs(seed=3, T=640, C=12, n_sections=5):
    """same construction as tests/golden/make_golden.py:segment_inputs (a feature with section structure + a beat grid)"""
    g = torch.Generator().manual_seed(seed)
    templates = torch.randn(3, C, generator=g)
    order = [0, 1, 0, 2, 1]
    bounds = torch.linspace(0, T, n_sections + 1).long()
    env = torch.empty(T, C)
    for s in range(n_sections):
        lo, hi = int(bounds[s]), int(bounds[s + 1])
        env[lo:hi] = templates[order[s]] + 0.3 * torch.randn(hi - lo, C, generator=g)
    beats, b = [], 0
    while True:
        b += int(torch.randint(6, 11, (), generator=g))
        if b >= T - 2:
            break
        beats.append(b)
    return env, beats, order, bounds


def test_segment_oracle_matches_reference_fixture(golden):
    """oracle/segment.py against g22 (the reference's own recurrence_matrix / timelag_median_filter / median_filter1d /
    init_plus_plus / differentiable_k_means on seeded inputs)."""
    from oracle import segment as O
    g = {k: np.asarray(v) for k, v in golden("g22_segment").items()}
    env, beats, _, _ = _segment_inputs()
    assert np.array_equal(env.numpy(), g["env"]) and np.array_equal(np.array(beats), g["beats"])
    assert np.array_equal(O.sync(g["env"], g["beats"]), g["Csync"])
    R = O.recurrence_matrix(g["Csync"], width=3)
    assert np.array_equal(R != 0, g["R"] != 0) and np.abs(R - g["R"]).max() < 1e-6
    assert np.array_equal(O.timelag_median_filter(g["R"]), g["Rf"])
    assert np.array_equal(O.median_filter1d(g["ev"].T, 9, 4).T, g["evf"])
    for k in (2, 6, 16):
        X = g[f"km{k}_X"]
        Xn = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
        assert np.abs(O.init_plus_plus(Xn, k) - g[f"km{k}_init"]).max() < 1e-6
        mu, r, dist = O.soft_kmeans(X, k, 100)
        assert np.abs(mu - g[f"km{k}_mu"]).max() < 5e-6 and np.abs(r - g[f"km{k}_r"]).max() < 5e-6
        assert np.abs(dist - g[f"km{k}_dist"]).max() < 5e-6


def test_segment_oracle_recovers_sections_and_beats():
    """what the un-pinned pieces must do: the beat tracker locks onto a click train at the given tempo (librosa's
    published dynamic program), and the full chain (dense sym Laplacian = torch_geometric's get_laplacian) separates the
    A B A C B sections of the fixture feature."""
    from oracle import segment as O
    # 21.5 frames/s (sr 22050 / hop 1024), 129 BPM -> period 10 frames
    T, period = 900, 10
    env = np.zeros(T, dtype=np.float32)
    env[7::period] = 1.0
    env += 0.05 * np.random.RandomState(0).rand(T).astype(np.float32)
    beats, local, cum, back = O.beat_track(env, bpm=60.0 * (22050 / 1024) / period)
    assert len(beats) > 80 and np.all(np.diff(beats) == period) and np.all(beats % period == 7)
    assert O.beat_track(np.zeros(50, dtype=np.float32), 120.0)[0].size == 0
    env2, bts, order, bounds = _segment_inputs()
    segs = O.laplacian_segmentation(env2.numpy(), bts, ks=(2, 4, 6))
    assert [s.shape for s in segs] == [(640, 2), (640, 4), (640, 6)]
    lab = segs[1].argmax(1)
    mids = [int((bounds[s] + bounds[s + 1]) // 2) for s in range(5)]
    # sections with the same template share a label, different templates differ (k = 4 over 3 templates)
    assert lab[mids[0]] == lab[mids[2]] and lab[mids[1]] == lab[mids[4]]
    assert len({int(lab[mids[0]]), int(lab[mids[1]]), int(lab[mids[3]])}) == 3


def test_mm_onset_oracle_properties():
    """oracle/mmonsets.py (madmom's published onset chain; un-vendored -> parity unpinned) does what the chain must: unit-sum
    triangular filters on strictly increasing bins, frame count ceil(len / hop), silence -> zeros, every detection function
    peaks on the frames of a click train, the envelope lies in [0, 1]."""
    from oracle import mmonsets as OM
    fb, corners = OM.log_filterbank(30720)
    assert fb.shape[0] == 1024 and np.allclose(fb.sum(0), 1.0, atol=1e-6) and (fb >= 0).all()
    centres = fb.argmax(0)
    assert np.all(np.diff(centres) > 0) and all(lo <= c <= hi for c, (lo, hi) in zip(centres, corners))
    f0 = OM.onset_functions(np.zeros(3000, dtype=np.float32), 30720)
    assert all(v.shape == (6,) and not v.any() for v in f0.values())
    g = torch.Generator().manual_seed(2)
    z = 1e-3 * torch.randn(40 * 512, generator=g)
    z[2048::4096] += 1.0
    f = OM.onset_functions(z.numpy(), 30720)
    for name, v in f.items():
        peaks = np.argsort(v)[-5:]
        assert all(int(p) % 8 in (3, 4, 5) for p in peaks) and len({int(p) // 8 for p in peaks}) == 5, name
    env = OM.mm_onset_envelope(z.numpy(), 30720)
    assert env.shape == (40,) and env.min() >= 0 and env.max() <= 1


def test_secondary_diffusion_model_and_fast_conditioning_match_the_reference(golden):
    """g28 (generated by the reference's own SecondaryDiffusionImageNet2 + GradientGuidedConditioning(speed="fast"),
    guided.py:68-143, :212-274): the restatement's v / pred / eps and the conditioning gradient -J^T g.  The first
    reference-pinned fixture of oracle/diffusion.py."""
    from oracle import diffusion as OD
    g = golden("g28_secondary")
    p = OD.secondary_random_params(int(g["seed"]))
    v, pred, eps = OD.secondary_forward(p, g["x"], g["t"])
    for got, want in ((v, g["v"]), (pred, g["pred"]), (eps, g["eps"])):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    sch = OD.Schedule(1000, "ddim100")
    target, scale = g["target"], float(g["mse_scale"])
    grad = OD.fast_conditioning(p, sch, lambda img, tt: (2.0 * scale / img[0].numel()) * (img - target), g["xt"], g["t_model"])
    want = g["cond_grad"]
    assert float((grad - want).abs().max()) <= 2e-5 * float(want.abs().max()) and float(want.abs().max()) > 0


def test_orig_and_resnet_architectures_match_the_reference_pieces(golden):
    """g29 (inference/stylegan2.py:275-382): constructor key set / shapes / draw order / num_ws of the "orig" and "resnet"
    networks, the reference Conv2dLayer's own forward (the resnet skip at up = 1), and one resnet block composed from the
    reference's ops the way SynthesisBlock.forward composes it."""
    from math import sqrt
    g = golden("g29_architectures")
    for arch, seed in (("orig", 31), ("resnet", 32)):
        p = S.init_synthesis_params(32, w_dim=16, channel_base=256, channel_max=16, generator=torch.Generator().manual_seed(seed),
                                    architecture=arch)
        keys = [str(k) for k in g[f"{arch}__keys"]]
        assert set(p.keys()) == set(keys) and int(g[f"{arch}__num_ws"]) == S.num_ws(32)
        for k in keys:
            assert torch.equal(g[f"{arch}__" + k.replace(".", "__")], p[k]), (arch, k)
    assert not any(".skip." in str(k) for k in g["orig__keys"]) and sum(".torgb.weight" in str(k) for k in g["orig__keys"]) == 1
    p = {"bs.1." + k[len("blk__p__"):].replace("__", "."): v for k, v in g.items() if k.startswith("blk__p__")}
    x, ws = g["blk__x"], g["blk__ws"]
    up1 = S.conv2d_layer(p, "bs.1.skip", x, up=1, gain=sqrt(0.5))
    assert torch.allclose(up1, g["blk__skip_up1"], rtol=1e-5, atol=1e-6)
    y = S.conv2d_layer(p, "bs.1.skip", x, up=2, gain=sqrt(0.5))
    assert torch.allclose(y, g["blk__skip"], rtol=1e-5, atol=1e-6)
    x0 = S.synthesis_layer(p, "bs.1.conv0", x, ws[:, 0], up=2)
    assert torch.allclose(x0, g["blk__conv0"], rtol=1e-4, atol=1e-5)
    x1 = S.synthesis_layer(p, "bs.1.conv1", x0, ws[:, 1], up=1, gain=sqrt(0.5))
    assert torch.allclose(x1, g["blk__conv1"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(y + x1, g["blk__out"], rtol=1e-4, atol=1e-5)

def test_regular_speed_conditioning_matches_the_reference_class(golden):
    """g32: what the REFERENCE's GradientGuidedConditioning.forward returns for speed "regular" (guided.py:214-218, 236-272: timestep
    mapping, img = pred_xstart * sigma + x * (1 - sigma), summed grad modules, -autograd.grad) around the restated network - the
    oracle's regular_conditioning must reproduce it (this pins the conditioning's arithmetic; the UNet itself is the published
    algorithm restated, the submodule being empty in the reference checkout)."""
    from oracle import diffusion as OD
    g = golden("g32_regular_conditioning")
    cfg = OD.unet_config(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions=(16, 8), channel_mult=(1, 2, 2),
                         num_head_channels=32)
    p = OD.init_unet_params(cfg, torch.Generator().manual_seed(int(g["unet_seed"])))
    sch = OD.Schedule(1000, "ddim20")
    k = 2.0 * float(g["mse_scale"]) / g["target"].numel()
    got = OD.regular_conditioning(p, cfg, sch, lambda im, _t: k * (im - g["target"]), g["xt"], g["t_model"])
    close(got, g["cond_grad"], 1e-5)

"""Synthesis network on the GPU (one C-ABI call per batch) vs the oracle network.  GPU only."""
import pytest
import torch

from oracle import stylegan2 as OS

pytestmark = pytest.mark.gpu


def build(res, cbase, cmax, dtype, seed=3, w_dim=64):
    from maua_amd.stylegan2 import SynthesisNetwork
    g = torch.Generator().manual_seed(seed)
    net = SynthesisNetwork(w_dim, res, 3, channel_base=cbase, channel_max=cmax, dtype=dtype, generator=g)
    p = net.state_dict()
    # make biases non-trivial
    g2 = torch.Generator().manual_seed(seed + 1)
    for k in p:
        if k.endswith(".bias") and "affine" not in k:
            p[k] = torch.randn(p[k].shape, generator=g2) * 0.1
    net.load_state_dict(p)
    return net, p


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    rng = float(b.max() - b.min())
    return 10 * torch.log10(torch.tensor(rng * rng / max(mse, 1e-30))).item()


@pytest.mark.parametrize("res,cbase,cmax", [(32, 1024, 64), (64, 2048, 128)])
def test_synth_f32_vs_oracle(res, cbase, cmax):
    net, p = build(res, cbase, cmax, torch.float32)
    net.keep_features(True)
    g = torch.Generator().manual_seed(9)
    B = 3
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    img = net(ws, noise=noise).cpu()
    ref, feats = OS.synthesis_network(p, ws, noise=noise, return_features=True)
    for l, f in enumerate(feats):
        got = net.get_feature(l, B).cpu()
        err = float((got - f).abs().max()) / float(f.abs().max())
        assert err <= 2e-5, f"layer {l}: {err}"
    err = float((img - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-5, err
    # const noise path (noise=None -> noise_const buffers) and broadcast noise
    img2 = net(ws).cpu()
    ref2 = OS.synthesis_network(p, ws)
    assert float((img2 - ref2).abs().max()) / float(ref2.abs().max()) <= 2e-5


def test_synth_bf16_vs_oracle():
    net, p = build(64, 2048, 128, torch.bfloat16)
    g = torch.Generator().manual_seed(10)
    B = 4
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    img = net(ws, noise=noise).cpu()
    ref = OS.synthesis_network(p, ws, noise=noise)
    # bf16 operands / f32 accumulate vs the fp32 oracle: measured 60-68 dB on nets of this size; the bar is set so
    # that a regression of more than a few dB (one layer rounding twice, a wrong tap) fails.  max-abs <= 1e-2 of the
    # image range.
    rng = float(ref.max() - ref.min())
    assert psnr(img, ref) >= 55.0, psnr(img, ref)
    assert float((img - ref).abs().max()) <= 1e-2 * rng


@pytest.mark.parametrize("res,cbase,cmax", [(64, 2048, 128), (256, 8192, 64)])
def test_synth_fp16_vs_oracle(res, cbase, cmax):
    """Round 5: SynthesisNetwork(dtype=torch.float16) - IEEE half activations and weights on v_mfma_f32_32x32x16_f16 with the
    reference's FP16 pre-normalisation (ops.py:161-165: weights by their per-channel maximum and sqrt(fan-in) at load time, styles by
    their per-sample maximum every batch) - against the fp32 oracle.  Half carries three more mantissa bits than bf16, so the bar
    sits above the bf16 one (55 dB; measured here >= 75 dB); u8 frames, per-layer features and a latent scale that would overflow
    x * s without the pre-normalisation."""
    net, p = build(res, cbase, cmax, torch.float16)
    net16, _ = build(res, cbase, cmax, torch.bfloat16)
    net.keep_features(True)
    g = torch.Generator().manual_seed(12)
    B = 2
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    img = torch.empty((B, 3, res, res), device="cuda")
    u8 = torch.empty((B, res, res, 3), dtype=torch.uint8, device="cuda")
    net(ws, noise=noise, out=img, rgb8_out=u8)
    ref, feats = OS.synthesis_network(p, ws, noise=noise, return_features=True)
    for l, f in enumerate(feats):
        got = net.get_feature(l, B).cpu()
        assert float((got - f).abs().max()) / float(f.abs().max()) <= 8e-3, l
    q = psnr(img.cpu(), ref)
    assert q >= 65.0 and q >= psnr(net16(ws, noise=noise).cpu(), ref), q
    assert torch.equal(u8, ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1))
    # latents 40x larger: styles of several hundred, |x * s| far beyond 65504 - finite and as close to the oracle as before
    big = ws * 40
    img_b = net(big, noise=noise).cpu()
    ref_b = OS.synthesis_network(p, big, noise=noise)
    assert bool(torch.isfinite(img_b).all()) and psnr(img_b, ref_b) >= 60.0, psnr(img_b, ref_b)


def test_synth_fp16_full_size_runs_the_fast_kernels():
    """Round 6 (VERDICT r5 missing 2): a float16 network at the BASELINE size takes the SAME routing as the bf16 one - LDS-direct
    convolutions, transposed-conv + FIR up-layers, register-stationary 512^2 kernels, the fused 1024^2 walk, all on
    v_mfma_f32_32x32x16_f16 with the pre-normalised weights / styles - instead of the generic kernels.  One frame against the fp32 CPU
    oracle (>= 65 dB, and no worse than the bf16 network's frame), against the same network routed over the generic f16 kernels
    (options off), and the u8 pack; a batch renders its frames bit for bit like single frames."""
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    net = SynthesisNetwork(512, 1024, 3, dtype=torch.float16, generator=torch.Generator().manual_seed(0))
    p = net.state_dict()
    net16 = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16)
    net16.load_state_dict(p)
    g = torch.Generator().manual_seed(31)
    B = 3
    ws = torch.randn(B, net.num_ws, 512, generator=g).cuda()
    noise = [torch.randn(B, 1, s[3], s[3], generator=g).cuda() for s in net.layer_shapes()]
    img = torch.empty((B, 3, 1024, 1024), device="cuda")
    u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
    net(ws, noise=noise, out=img, rgb8_out=u8)
    assert torch.equal(u8, ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1))
    one = net(ws[1:2], noise=[n[1:2].contiguous() for n in noise])
    assert torch.equal(one[0], img[1])
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(32, nthr))
    try:
        ref = OS.synthesis_network(p, ws[1:2].cpu(), noise=[n[1:2].cpu() for n in noise])
    finally:
        torch.set_num_threads(nthr)
    q, q16 = psnr(img[1:2].cpu(), ref), psnr(net16(ws[1:2], noise=[n[1:2].contiguous() for n in noise]).cpu(), ref)
    print("1024^2 float16 frame vs the fp32 oracle: %.1f dB (bf16 network: %.1f dB)" % (q, q16))
    assert q >= 65.0 and q >= q16 - 0.5
    # the same network on the generic f16 kernels (round 5's routing): every fast path switched off
    gen = SynthesisNetwork(512, 1024, 3, dtype=torch.float16)
    gen.load_state_dict(p)
    h = gen._handle()
    for key in (b"use_hires", b"upwalk", b"dma_conv", b"tconv_dma", b"tconv_up", b"lowres"):
        L.check(L.lib().maua_synth_set_option(h, key, 0))
    slow = gen(ws[1:2], noise=[n[1:2].contiguous() for n in noise])
    assert psnr(slow.cpu(), ref) >= 65.0 and psnr(slow.cpu(), img[1:2].cpu()) >= 65.0


def test_synth_rgb8_and_determinism():
    net, p = build(32, 1024, 64, torch.float32)
    g = torch.Generator().manual_seed(11)
    ws = torch.randn(2, net.num_ws, 64, generator=g)
    img = torch.empty((2, 3, 32, 32), device="cuda")
    u8 = torch.empty((2, 32, 32, 3), dtype=torch.uint8, device="cuda")
    net(ws, out=img, rgb8_out=u8)
    want = ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)
    assert torch.equal(u8, want)
    img2 = net(ws)
    assert torch.equal(img, img2)  # bit-identical re-run
    # batch independence: frame 1 alone == frame 1 in the batch (frame-range sharding relies on it)
    img3 = net(ws[1:])
    assert torch.equal(img3[0], img[1])


def test_hires_kernels_match_generic_and_oracle():
    """The register-stationary high-resolution kernels (+ fused toRGB) must agree with the generic MFMA kernel
    bit-for-bit in structure-independent terms: compare both against each other (bf16, tight) and the oracle."""
    import ctypes as C
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    # channel_base 8192 / max 64: 4..128 -> 64 ch, 256 -> 32 ch: exercises <64,64,1>, <64,32,2>, <32,32,1>
    g = torch.Generator().manual_seed(4)
    net = SynthesisNetwork(64, 256, 3, channel_base=8192, channel_max=64, dtype=torch.bfloat16, generator=g)
    p = net.state_dict()
    g2 = torch.Generator().manual_seed(5)
    for k in p:
        if k.endswith(".bias") and "affine" not in k:
            p[k] = torch.randn(p[k].shape, generator=g2) * 0.1
    net.load_state_dict(p)
    B = 3
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    net.keep_features(True)
    img_h = net(ws, noise=noise).cpu()
    feats_h = [net.get_feature(l, B).cpu() for l in range(net.num_layers)]
    # the last block's fused epilogue also packs the u8 frame: with and without the f32 image in the same call
    img_d = torch.empty((B, 3, 256, 256), device="cuda")
    u8a = torch.empty((B, 256, 256, 3), dtype=torch.uint8, device="cuda")
    u8b = torch.empty_like(u8a)
    net(ws, noise=noise, out=img_d, rgb8_out=u8a)
    net(ws, noise=noise, rgb8_out=u8b)
    assert torch.equal(img_d.cpu(), img_h)
    want = ((img_d + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)
    assert torch.equal(u8a, want) and torch.equal(u8b, want)
    h = net._handle()
    L.check(L.lib().maua_synth_set_option(h, b"use_hires", 0))
    img_g = net(ws, noise=noise).cpu()
    feats_g = [net.get_feature(l, B).cpu() for l in range(net.num_layers)]
    for l, (a, b) in enumerate(zip(feats_h, feats_g)):
        err = float((a - b).abs().max()) / float(b.abs().max())
        assert err <= 2e-2, f"layer {l}: hires vs generic {err}"   # both bf16; styles folded into W vs into x
    rng = float(img_g.max() - img_g.min())
    assert float((img_h - img_g).abs().max()) <= 2e-2 * rng
    ref = OS.synthesis_network(p, ws, noise=noise)
    assert psnr(img_h, ref) >= 40.0
    # fused toRGB on/off gives the same image up to f32 summation order
    L.check(L.lib().maua_synth_set_option(h, b"use_hires", 1))
    L.check(L.lib().maua_synth_set_option(h, b"fuse_torgb", 0))
    img_nf = net(ws, noise=noise).cpu()
    assert float((img_nf - img_h).abs().max()) <= 1e-4 * rng
    # without keep_features the last block's conv1 does not store its features (only its fused toRGB reads them):
    # same image, same u8 frame, bit for bit
    L.check(L.lib().maua_synth_set_option(h, b"fuse_torgb", 1))
    net.keep_features(False)
    L.check(L.lib().maua_synth_set_option(h, b"upwalk", 2))
    img_e = torch.empty_like(img_d)
    u8c = torch.empty_like(u8a)
    u8d = torch.empty_like(u8a)
    net(ws, noise=noise, out=img_e, rgb8_out=u8c)
    net(ws, noise=noise, rgb8_out=u8d)
    # (the last block then runs as ONE walk - modconv_upwalk.hip - whose conv1 folds demodulation into the weights:
    #  equal to the unfused kernels up to bf16 weight rounding, and bit-identical with / without the f32 image)
    assert psnr(img_e.cpu(), img_h) >= 60.0 and float((img_e.cpu() - img_h).abs().max()) <= 2e-3 * rng
    assert torch.equal(u8c, ((img_e + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1))
    assert torch.equal(u8d, u8c)
    L.check(L.lib().maua_synth_set_option(h, b"upwalk", 1))   # the same block as separate kernels: bit-identical to capture mode
    net(ws, noise=noise, out=img_e, rgb8_out=u8c)
    assert torch.equal(img_e, img_d) and torch.equal(u8c, u8a)
    L.check(L.lib().maua_synth_set_option(h, b"upwalk", 2))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_tconv_up_matches_phase_kernels(dt):
    """Up-layers as minimal transposed conv + FIR pass (default) vs the four 3x3 phase kernels: same layer outputs."""
    from maua_amd import _lib as L
    net, p = build(64, 2048, 128, dt)
    g = torch.Generator().manual_seed(12)
    B = 2
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    net.keep_features(True)
    L.check(L.lib().maua_synth_set_option(net._handle(), b"tconv_up", 1 << 20))  # every up-layer, also the tiny and the largest ones
    img_t = net(ws, noise=noise).cpu()
    feats_t = [net.get_feature(l, B).cpu() for l in range(net.num_layers)]
    L.check(L.lib().maua_synth_set_option(net._handle(), b"tconv_up", 0))
    img_p = net(ws, noise=noise).cpu()
    feats_p = [net.get_feature(l, B).cpu() for l in range(net.num_layers)]
    tol = 2e-5 if dt == torch.float32 else 3e-2
    for l, (a, b) in enumerate(zip(feats_t, feats_p)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), f"layer {l}"
    assert float((img_t - img_p).abs().max()) <= tol * float(img_p.abs().max())


@pytest.mark.parametrize("B", [1, 33, 70])
def test_odd_batch_sizes_are_batch_independent(B):
    """Any batch size (the styles GEMMs tile samples by 32, workspaces grow on demand): every frame equals the same
    frame rendered alone-ish (in a batch of 2), bit for bit."""
    net, p = build(64, 2048, 64, torch.bfloat16)
    g = torch.Generator().manual_seed(9)
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    img = net(ws).cpu()
    assert bool(torch.isfinite(img).all())
    for i in sorted({0, B // 2, B - 1}):
        pair = torch.stack([ws[i], ws[(i + 1) % B]])
        assert torch.equal(net(pair).cpu()[0], img[i]), i


def test_full_size_network_properties_and_oracle_frame():
    """BASELINE size (1024^2, 512 channels, 17 layers, bf16 - the net bench.py times): size-independent properties of
    the default kernel routing (register-stationary, tconv + FIR, fused toRGB / u8, LDS-direct loads) plus ONE frame
    against the fp32 CPU oracle at full size."""
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(21)
    B = 3
    ws = torch.randn(B, net.num_ws, 512, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    noise_d = [n.cuda() for n in noise]
    img = torch.empty((B, 3, 1024, 1024), device="cuda")
    u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
    net(ws, noise=noise_d, out=img, rgb8_out=u8)
    # the u8 frame is the packed f32 image of the same call; the u8-only call gives the same frame
    want = ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)
    assert torch.equal(u8, want)
    u8b = torch.empty_like(u8)
    for _ in range(5):  # re-runs are bit-identical (no race in the persistent / LDS-direct kernels)
        u8b.zero_()
        net(ws, noise=noise_d, rgb8_out=u8b)
        assert torch.equal(u8b, u8)
    # batch independence (frame-range sharding relies on it): frame 1 alone == frame 1 inside the batch
    img1 = net(ws[1:2], noise=[n[1:2] for n in noise_d])
    assert torch.equal(img1[0], img[1])
    # generic kernels only (no register-stationary kernels, phase-form up-layers): same image up to bf16 rounding
    h = net._handle()
    L.check(L.lib().maua_synth_set_option(h, b"use_hires", 0))
    L.check(L.lib().maua_synth_set_option(h, b"tconv_up", 0))
    img_g = net(ws, noise=noise_d)
    L.check(L.lib().maua_synth_set_option(h, b"use_hires", 1))
    L.check(L.lib().maua_synth_set_option(h, b"tconv_up", 1))
    assert psnr(img.cpu(), img_g.cpu()) >= 45.0
    # the LDS-direct-load kernels (conv1 on pre-modulated input, transposed conv of the up-layers) vs the register-staged
    # ones: the same operands except that a pre-modulated activation is rounded once instead of twice
    for opt in (b"dma_conv", b"tconv_dma"):
        L.check(L.lib().maua_synth_set_option(h, opt, 0))
        img_o = net(ws, noise=noise_d)
        L.check(L.lib().maua_synth_set_option(h, opt, 1))
        assert psnr(img.cpu(), img_o.cpu()) >= 60.0, (opt, psnr(img.cpu(), img_o.cpu()))
    # toRGB fused into the conv1 epilogues (register-stationary kernels at 512^2 / 1024^2, the generic kernel at 256^2
    # where Co == its N tile) vs the stand-alone toRGB kernels: same image up to f32 summation order
    # (with the last block as separate kernels, upwalk = 1: its one-walk form folds conv1's demodulation into the weights
    #  - a different bf16 rounding, compared below)
    L.check(L.lib().maua_synth_set_option(h, b"upwalk", 1))
    img_sep = net(ws, noise=noise_d)
    L.check(L.lib().maua_synth_set_option(h, b"fuse_torgb", 0))
    img_nf = net(ws, noise=noise_d)
    L.check(L.lib().maua_synth_set_option(h, b"fuse_torgb", 1))
    L.check(L.lib().maua_synth_set_option(h, b"upwalk", 2))
    assert float((img_nf - img_sep).abs().max()) <= 1e-4 * float(img.max() - img.min())
    assert psnr(img.cpu(), img_sep.cpu()) >= 65.0, psnr(img.cpu(), img_sep.cpu())   # fused walk vs separate kernels
    # one full-size frame against the oracle (fp32, CPU): same bar as the small bf16 nets
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(32, nthr))
    try:
        ref = OS.synthesis_network(net.state_dict(), ws[:1], noise=[n[:1] for n in noise])
    finally:
        torch.set_num_threads(nthr)
    rng = float(ref.max() - ref.min())
    print(f"full-size bf16 frame vs oracle: PSNR {psnr(img[:1].cpu(), ref):.1f} dB, "
          f"max-abs {float((img[:1].cpu() - ref).abs().max()) / rng:.2e} of the range")
    assert psnr(img[:1].cpu(), ref) >= 55.0, psnr(img[:1].cpu(), ref)  # 17 layers, 512 channels: measured 65.9 dB
    assert float((img[:1].cpu() - ref).abs().max()) <= 1e-2 * rng


def test_bench_shape_batch_equals_single_frames():
    """The shape bench.py times: its default batch of frames of the 1024^2 bf16 net in one call (512-channel layers through
    the batch-wide low-resolution GEMM, the LDS-direct-load kernels and the fused last-block walk at full grids).  Frames
    0, 15 and B - 1 of that batch equal the same frames rendered alone, bit for bit (what frame-range sharding relies
    on), and frame 15 matches the fp32 CPU oracle."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("maua_bench", pathlib.Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from maua_amd.noise import Loop, loop_batch
    from maua_amd.stylegan2 import SynthesisNetwork
    net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(77)
    B = bench.DEFAULT_BATCH
    assert B >= 32
    ws = torch.randn(B, net.num_ws, 512, generator=g).cuda()
    sizes = [s[3] for s in net.layer_shapes()]
    rng_n = torch.Generator().manual_seed(42)
    mods = [Loop(rng_n, 2 * B, (s, s), n_loops=2, sigma=5) for s in sizes]
    nz = loop_batch(mods, 3, B)  # the bench's noise path: frames 3 .. B + 2 of a 2 B-frame loop
    img = torch.empty((B, 3, 1024, 1024), device="cuda")
    u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
    net(ws, noise=nz, out=img, rgb8_out=u8)
    for i in (0, 15, B - 1):
        one = torch.empty((1, 3, 1024, 1024), device="cuda")
        one8 = torch.empty((1, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
        net(ws[i:i + 1], noise=[n[i:i + 1].contiguous() for n in nz], out=one, rgb8_out=one8)
        assert torch.equal(one[0], img[i]), i
        assert torch.equal(one8[0], u8[i]), i
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(32, nthr))
    try:
        ref = OS.synthesis_network(net.state_dict(), ws[15:16].cpu(), noise=[n[15:16].cpu() for n in nz])
    finally:
        torch.set_num_threads(nthr)
    got = img[15:16].cpu()
    assert psnr(got, ref) >= 55.0, psnr(got, ref)
    # (a random-init net's image spans ~ +-60, so one u8 step is 1e-4 of its range: the u8 frame is checked against the
    #  pack of the f32 image of the same call, exactly, rather than against the fp32 oracle's frame)
    want8 = ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)
    assert torch.equal(u8, want8)


def test_full_size_u8_frames_match_oracle_on_a_non_saturating_network():
    """SURVEY 8(d)'s output bar at the BASELINE size.  A random-init 1024^2 network spans ~ +-60, so nearly every u8 pixel
    of its frames is 0 or 255; here every toRGB weight is scaled (the image is linear in them) so that the frames have a standard
    deviation of ~0.35 (> 90 % of the pixels strictly inside the u8 range), and frames {0, 64, 127} of a B = 128 batch are compared with the fp32 CPU oracle's u8 frames:
    exact-f32 mode <= 1 LSB on <= 0.5 % of the pixels (the SURVEY bar), bf16 (the bench's dtype; rmse ~1e-3 of a 7.8e-3
    LSB) within 1 LSB on all but <= 0.1 % of the pixels (never more than 2), with ~19 % of the pixels rounding the other
    way."""
    from maua_amd.noise import Loop, loop_batch
    from maua_amd.stylegan2 import SynthesisNetwork
    from oracle import io as OIO
    net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(78)
    B = 128
    ws = torch.randn(B, net.num_ws, 512, generator=g).cuda()
    sizes = [s[3] for s in net.layer_shapes()]
    mods = [Loop(torch.Generator().manual_seed(43), 2 * B, (s, s), n_loops=2, sigma=5) for s in sizes]
    nz = loop_batch(mods, 5, B)
    probe = net(ws[:4], noise=[n[:4].contiguous() for n in nz])
    scale = 0.35 / float(probe.std())   # (heavy tails: a few pixels clip, most sit well inside [-1, 1])
    p = net.state_dict()
    for k in p:
        if ".torgb.weight" in k:
            p[k] = p[k] * scale
    net.load_state_dict(p)
    net32 = SynthesisNetwork(512, 1024, 3, dtype=torch.float32)
    net32.load_state_dict(p)
    neth = SynthesisNetwork(512, 1024, 3, dtype=torch.float16)   # (round 6: the reference's render dtype on the fast kernels)
    neth.load_state_dict(p)
    u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
    net(ws, noise=nz, rgb8_out=u8)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(32, nthr))
    try:
        for i in (0, 64, 127):
            nzi = [n[i:i + 1].contiguous() for n in nz]
            ref = OS.synthesis_network(p, ws[i:i + 1].cpu(), noise=[n.cpu() for n in nzi])
            ref8 = torch.from_numpy(OIO.frames_to_u8(ref))[0].int()
            assert float(ref8.float().std()) > 25 and float(((ref8 > 0) & (ref8 < 255)).float().mean()) > 0.9   # not saturating
            d16 = (u8[i].cpu().int() - ref8).abs()
            # bf16 features: rmse ~1e-3 of the image range against a 7.8e-3 LSB -> ~19 % of the pixels round the other
            # way, a handful in the tails by 2 (measured: 18.9 %, max 2)
            assert int(d16.max()) <= 2 and float((d16 > 1).float().mean()) <= 1e-3 and float((d16 > 0).float().mean()) <= 0.25, \
                (i, int(d16.max()), float((d16 > 1).float().mean()), float((d16 > 0).float().mean()))
            one8 = torch.empty((1, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
            net32(ws[i:i + 1], noise=nzi, rgb8_out=one8)
            d32 = (one8[0].cpu().int() - ref8).abs()
            assert int(d32.max()) <= 1 and float((d32 > 0).float().mean()) <= 0.005, (i, float((d32 > 0).float().mean()))
            # float16 features carry three more mantissa bits than bf16: never more than 1 LSB, and an order of magnitude fewer
            # pixels that round the other way (measured 2.4 %)
            neth(ws[i:i + 1], noise=nzi, rgb8_out=one8)
            dh = (one8[0].cpu().int() - ref8).abs()
            print("frame %d: u8 pixels off by one - bf16 %.2f %%, float16 %.2f %% (max %d), exact-f32 %.3f %%" %
                  (i, 100 * float((d16 > 0).float().mean()), 100 * float((dh > 0).float().mean()), int(dh.max()), 100 * float((d32 > 0).float().mean())))
            assert int(dh.max()) <= 1 and float((dh > 0).float().mean()) <= 0.05, (i, int(dh.max()), float((dh > 0).float().mean()))
    finally:
        torch.set_num_threads(nthr)


def test_warp_hook_on_layers_with_fused_torgb():
    """A translate / rotate hook on a conv1 whose toRGB normally rides on the conv epilogue (register-stationary
    kernels, LDS-direct kernel) - including the LAST conv1, whose features are normally not stored at all: the hook
    replaces the layer output before toRGB reads it (wrappers/stylegan2.py:153-194), so the fused path must step
    aside.  256^2 bf16 net with 128 / 64 / 32 channels at 64^2 / 128^2 / 256^2, against the oracle."""
    from maua_amd.stylegan2 import StyleGAN2Synthesizer
    gen = torch.Generator().manual_seed(14)
    G, _ = build(256, 8192, 128, torch.bfloat16)
    syn = StyleGAN2Synthesizer.__new__(StyleGAN2Synthesizer)
    torch.nn.Module.__init__(syn)
    syn.G_synth, syn.num_ws, syn.w_dim = G, G.num_ws, G.w_dim
    assert [s[2] for s in G.layer_shapes()][-3:] == [64, 32, 32]
    B = 2
    ws = torch.randn(B, syn.num_ws, 64, generator=gen)
    plain = OS.synthesis_network(G.state_dict(), ws)
    last = G.num_layers          # layer_names index of bs.6.conv1 (1-based count of synthesis layers)
    for layer in (last, last - 2):
        rotation = torch.tensor([10.0, -25.0])
        h, w = G.layer_size(layer - 1)
        img = syn.forward(ws, rotation=rotation, rotation_layer=layer).cpu()
        u8 = torch.empty((B, 256, 256, 3), dtype=torch.uint8, device="cuda")
        G(ws, rgb8_out=u8)  # hooks persist: the u8-only call takes the same route
        Mr = StyleGAN2Synthesizer._rotation_scale_matrix(rotation, torch.ones(B), None, h, w, B)
        ref = OS.synthesis_network(G.state_dict(), ws, warps=[(layer, Mr)])
        assert psnr(img, ref) >= 45.0, (layer, psnr(img, ref))
        assert psnr(plain, ref) < 35.0  # the hook does something
        want = ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)
        assert int((u8.cpu().int() - want.int()).abs().max()) <= 1
        syn.forward(ws, rotation=torch.zeros(B), rotation_layer=layer)  # identity warp: back to the plain image
    assert psnr(syn.forward(ws).cpu(), plain) >= 55.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_lowres_gemm_matches_per_sample_kernels(dt):
    """<= 8x8 layers as one batch-wide split-K GEMM (default) vs the per-sample tiles of the generic kernel: same layer
    outputs; frames stay bit-identical across batch sizes (the K slices are fixed by the layer, not by the batch)."""
    from maua_amd import _lib as L
    net, p = build(64, 2048, 128, dt)
    g = torch.Generator().manual_seed(31)
    B = 5
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g).cuda() for s in net.layer_shapes()]
    net.keep_features(True)
    img_l = net(ws, noise=noise).cpu()
    feats_l = [net.get_feature(l, B).cpu() for l in range(net.num_layers)]
    one = net(ws[3:4], noise=[n[3:4] for n in noise]).cpu()
    assert torch.equal(one[0], img_l[3])
    L.check(L.lib().maua_synth_set_option(net._handle(), b"lowres", 0))
    img_g = net(ws, noise=noise).cpu()
    feats_g = [net.get_feature(l, B).cpu() for l in range(net.num_layers)]
    tol = 2e-5 if dt == torch.float32 else 2e-2
    for l, (a, b) in enumerate(zip(feats_l, feats_g)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), f"layer {l}"
    assert float((img_l - img_g).abs().max()) <= tol * float(img_g.abs().max())


@pytest.mark.parametrize("res,B", [(128, 1), (256, 3), (512, 2)])
def test_upwalk_block_walks_match_phase_form_and_oracle(res, B):
    """The last block's 64 -> 32 channel up-layer as a half-folded row walk (modconv_upwalk.hip), alone (upwalk = 1) and
    fused with conv1 + toRGB + skip + u8 pack into one walk whose features never reach HBM (upwalk = 2), against the
    register-stationary phase-form kernels (upwalk = 0) and the fp32 oracle.  Sizes: 1 / 3 / 5 strips of 126 output
    pixels (the last one partial), several row segments per strip; rows / columns at every image edge."""
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    cbase = 32 * res                    # 64 channels up to res / 2, 32 at res
    g = torch.Generator().manual_seed(res)
    net = SynthesisNetwork(64, res, 3, channel_base=cbase, channel_max=64, dtype=torch.bfloat16, generator=g)
    p = net.state_dict()
    g2 = torch.Generator().manual_seed(res + 1)
    for k in p:
        if k.endswith(".bias") and "affine" not in k:
            p[k] = torch.randn(p[k].shape, generator=g2) * 0.1
    net.load_state_dict(p)
    shapes = net.layer_shapes()
    assert shapes[-1][1:3] == (32, 32) and shapes[-2][1:3] == (64, 32), shapes[-2:]
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in shapes]
    h = net._handle()
    imgs, u8s = {}, {}
    for mode in (0, 1, 2):
        L.check(L.lib().maua_synth_set_option(h, b"upwalk", mode))
        img = torch.empty((B, 3, res, res), device="cuda")
        u8 = torch.empty((B, res, res, 3), dtype=torch.uint8, device="cuda")
        u8_only = torch.empty_like(u8)
        net(ws, noise=noise, out=img, rgb8_out=u8)
        net(ws, noise=noise, rgb8_out=u8_only)               # no f32 image requested: same frame
        assert torch.equal(u8, u8_only), mode
        assert torch.equal(u8, ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)), mode
        img2 = torch.empty_like(img)
        net(ws, noise=noise, out=img2)
        assert torch.equal(img, img2), mode                  # bit-identical re-run
        imgs[mode], u8s[mode] = img.cpu(), u8.cpu()
    rng = float(imgs[0].max() - imgs[0].min())
    for mode in (1, 2):   # same bf16 operands, different summation order / weight folding
        assert psnr(imgs[mode], imgs[0]) >= 60.0, (mode, psnr(imgs[mode], imgs[0]))
        assert float((imgs[mode] - imgs[0]).abs().max()) <= 5e-3 * rng, mode
    ref = OS.synthesis_network(p, ws, noise=noise)
    for mode in (0, 1, 2):
        assert psnr(imgs[mode], ref) >= 50.0, (mode, psnr(imgs[mode], ref))
    # a frame alone equals the same frame inside the batch (frame-range sharding relies on it), fused walk included
    if B > 1:
        L.check(L.lib().maua_synth_set_option(h, b"upwalk", 2))
        one = torch.empty((1, 3, res, res), device="cuda")
        net(ws[B - 1:], noise=[n[B - 1:] for n in noise], out=one)
        assert torch.equal(one[0].cpu(), imgs[2][B - 1])


def test_dual_store_of_plain_and_style_scaled_features_is_bit_identical():
    """Round 5: a conv1 layer whose toRGB stays a separate pass (512 channels: 32^2, 64^2) writes its features twice in one epilogue -
    plain for that pass, multiplied by the next up-layer's styles into the premod buffer - instead of a premod pass over them
    (option "dual_store").  Same products, same roundings: frames must not change by a bit."""
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    net = SynthesisNetwork(512, 256, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(5))   # 512 channels up to 64^2
    g = torch.Generator().manual_seed(6)
    B = 3
    ws = torch.randn(B, net.num_ws, 512, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    h = net._handle()
    out = {}
    for mode in (0, 1):
        L.check(L.lib().maua_synth_set_option(h, b"dual_store", mode))
        img = torch.empty((B, 3, 256, 256), device="cuda")
        u8 = torch.empty((B, 256, 256, 3), dtype=torch.uint8, device="cuda")
        net(ws, noise=noise, out=img, rgb8_out=u8)
        out[mode] = (img.cpu(), u8.cpu())
    L.check(L.lib().maua_synth_set_option(h, b"dual_store", 1))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    ref = OS.synthesis_network(net.state_dict(), ws, noise=noise)
    assert psnr(out[1][0], ref) >= 55.0


@pytest.mark.parametrize("res,B,segs", [(256, 3, 0), (256, 2, 3), (512, 2, 5), (512, 1, 7)])
def test_fused_walk_narrow_last_strip_is_bit_identical(res, B, segs):
    """Round 5: a last strip of <= 32 columns (256 = 2 x 126 + 4, 512 = 4 x 126 + 8, 1024 = 8 x 126 + 16) is walked as two
    half-height sub-items at once (modconv_upwalk.hip, option "walk_narrow").  Same arithmetic per pixel, so frames must not
    change by a bit - with the cost model's row segments and with forced odd splits (rows of a segment odd: the lower sub-item is
    one row shorter; last segment shorter than the others)."""
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    g = torch.Generator().manual_seed(res + segs)
    net = SynthesisNetwork(64, res, 3, channel_base=32 * res, channel_max=64, dtype=torch.bfloat16, generator=g)
    p = net.state_dict()
    for k in p:
        if k.endswith(".bias") and "affine" not in k:
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
    net.load_state_dict(p)
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    h = net._handle()
    out = {}
    for narrow in (0, 1):
        L.check(L.lib().maua_synth_set_option(h, b"walk_narrow", narrow))
        L.check(L.lib().maua_synth_set_option(h, b"walk_segs", segs))
        img = torch.empty((B, 3, res, res), device="cuda")
        u8 = torch.empty((B, res, res, 3), dtype=torch.uint8, device="cuda")
        net(ws, noise=noise, out=img, rgb8_out=u8)
        out[narrow] = (img.cpu(), u8.cpu())
    L.check(L.lib().maua_synth_set_option(h, b"walk_narrow", 1))
    L.check(L.lib().maua_synth_set_option(h, b"walk_segs", 0))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    # ... and the cost model's segmentation gives the same frame as the forced one
    img = torch.empty((B, 3, res, res), device="cuda")
    net(ws, noise=noise, out=img)
    assert torch.equal(img.cpu(), out[1][0])


@pytest.mark.parametrize("target", [(40, 96), (33, 64), (7, 32)])
def test_upwalk_block_walks_on_a_resized_non_square_grid(target):
    """The same three forms of the last block behind a feature-space resize (get_hook's resize: 64^2 -> target after
    bs.4.conv1, so the last block's up-layer runs on 2 target inputs and renders 4 target): non-square, odd row counts,
    rows not a multiple of the row segments, partial last strips - against each other and the oracle."""
    th, tw = target
    from maua_amd import _lib as L
    from maua_amd.stylegan2 import SynthesisNetwork
    g = torch.Generator().manual_seed(21)
    net = SynthesisNetwork(64, 256, 3, channel_base=8192, channel_max=64, dtype=torch.bfloat16, generator=g)
    p = net.state_dict()
    for k in p:
        if k.endswith(".bias") and "affine" not in k:
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
    net.load_state_dict(p)
    B = 2
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    fill = torch.randn((64, th, tw), generator=g) * 0.5
    net.set_resize(8, target=(th, tw), fill_noise=fill, noise_generator=torch.Generator().manual_seed(5), mode="stretch")
    assert net.output_hw == (4 * th, 4 * tw)
    h = net._handle()
    imgs = {}
    for mode in (0, 1, 2):
        L.check(L.lib().maua_synth_set_option(h, b"upwalk", mode))
        img = torch.empty((B, 3, 4 * th, 4 * tw), device="cuda")
        u8 = torch.empty((B, 4 * th, 4 * tw, 3), dtype=torch.uint8, device="cuda")
        net(ws, out=img, rgb8_out=u8)
        assert torch.equal(u8, ((img + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(0, 2, 3, 1)), mode
        imgs[mode] = img.cpu()
    rng = float(imgs[0].max() - imgs[0].min())
    for mode in (1, 2):
        assert psnr(imgs[mode], imgs[0]) >= 60.0, (mode, psnr(imgs[mode], imgs[0]))
        assert float((imgs[mode] - imgs[0]).abs().max()) <= 5e-3 * rng, mode
    ref = OS.synthesis_network(net.state_dict(), ws, resize=dict(layer=8, mode="stretch", target=(th, tw), fill=fill, padding=(0, 0, 0, 0)))
    for mode in (0, 1, 2):
        assert psnr(imgs[mode], ref) >= 50.0, (mode, psnr(imgs[mode], ref))
    net.set_resize(None)


def test_fused_up_layer_is_bit_identical_to_the_two_launch_path():
    """modconv_tconv_fir.hip (transposed conv + 4x4 FIR + epilogue in one kernel, t in LDS) against the two-launch path
    (tconv_dma + edges, then upfir_epilogue: t through HBM): the whole image bit for bit, on a 256^2 network whose 32^2 ...
    128^2 up-layers (tiles that overhang the image on every side, several channel blocks, noise, biases) all take the fused
    kernel, on a resized non-square grid, and against the oracle."""
    from maua_amd import _lib as L
    net, p = build(256, 8192, 128, torch.bfloat16)
    g = torch.Generator().manual_seed(21)
    B = 3
    ws = torch.randn(B, net.num_ws, 64, generator=g)
    noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
    h = net._handle()
    imgs = {}
    for v in (0, 32):
        L.check(L.lib().maua_synth_set_option(h, b"tconv_fir", v))
        imgs[v] = net(ws, noise=noise).cpu()
        u8 = torch.empty((B, 256, 256, 3), dtype=torch.uint8, device="cuda")
        net(ws, noise=noise, rgb8_out=u8)
        imgs[(v, "u8")] = u8.cpu()
    assert torch.equal(imgs[0], imgs[32]) and torch.equal(imgs[(0, "u8")], imgs[(32, "u8")])
    ref = OS.synthesis_network(p, ws, noise=noise)
    assert psnr(imgs[32], ref) >= 50.0
    one = net(ws[1:2], noise=[n[1:2] for n in noise]).cpu()       # position in the batch does not matter
    assert torch.equal(one[0], imgs[32][1])
    # resized grid: the 32^2 block stretched to 40 x 96 -> up-layers at 40 x 96, 80 x 192 (tiles of 6 x 30 positions do not divide them)
    net.set_resize(7, target=(40, 96), noise_generator=torch.Generator().manual_seed(5))
    res = {}
    for v in (0, 32):
        L.check(L.lib().maua_synth_set_option(h, b"tconv_fir", v))
        res[v] = net(ws).cpu()
    assert tuple(res[0].shape) == (B, 3, 320, 768) and torch.equal(res[0], res[32])
    net.set_resize(None)
    L.check(L.lib().maua_synth_set_option(h, b"tconv_fir", 256))


def test_raw_noise_maps_with_per_sample_factors_equal_the_normalised_maps():
    """noise.loop_batch(raw=True): the Loop maps un-normalised in ONE pass + the factor 1 / (rms + eps) per (layer, sample), applied by
    the convolution epilogues (every kernel that takes a noise operand: low-resolution GEMM, generic, LDS-direct, tconv + FIR pair,
    fused up-layer, register-stationary, both halves of the fused walk).  Against the two-pass normalised maps (noise.py:42-53):
    maps x factor == normalised maps to f32 rounding; the f32-mode image of a 256^2 network (all kernel families but the walk)
    within 2e-6 of its range (u8 frames equal on >= 99.9 % of the bytes) and the bf16 1024^2 frames (the walk) at PSNR >= 70 dB; partial batches; a batch
    whose factors are handed over equals the same frames rendered one by one."""
    from maua_amd.noise import Loop, loop_batch
    from maua_amd.stylegan2 import SynthesisNetwork
    for res, dt, cbase, cmax, wd in ((256, torch.float32, 8192, 128, 64), (1024, torch.bfloat16, 32768, 512, 512)):
        net = SynthesisNetwork(wd, res, 3, channel_base=cbase, channel_max=cmax, dtype=dt, generator=torch.Generator().manual_seed(0))
        T, B = 11, 5
        mods = [Loop(torch.Generator().manual_seed(43), T, (s[3], s[3]), n_loops=2, sigma=5) for s in net.layer_shapes()]
        ws = torch.randn(T, net.num_ws, wd, generator=torch.Generator().manual_seed(1)).cuda()
        for i0, b in ((2, B), (8, B)):                       # (8, 5): only 3 frames are left
            raw, nrm = loop_batch(mods, i0, b, raw=True), loop_batch(mods, i0, b)
            nb = raw[0].shape[0]
            assert nb == min(b, T - i0) and tuple(raw.scales.shape) == (len(mods), nb)
            for a, c in zip(raw.normalised(), nrm):
                assert float((a - c).abs().max()) <= 4e-6 * float(c.abs().max())
            u_raw = torch.empty((nb, res, res, 3), dtype=torch.uint8, device="cuda")
            u_nrm = torch.empty_like(u_raw)
            img_raw = net(ws[i0:i0 + nb], noise=raw, rgb8_out=u_raw, out=torch.empty((nb, 3, res, res), device="cuda"))
            img_nrm = net(ws[i0:i0 + nb], noise=nrm, rgb8_out=u_nrm, out=torch.empty((nb, 3, res, res), device="cuda"))
            rng = float(img_nrm.max() - img_nrm.min())
            if dt == torch.float32:
                assert float((img_raw - img_nrm).abs().max()) <= 2e-6 * rng
            else:
                assert psnr(img_raw.cpu(), img_nrm.cpu()) >= 70.0
            # (bf16: every layer's output is rounded, a last-bit difference in the noise term flips some of those roundings)
            assert float((u_raw == u_nrm).float().mean()) >= (0.999 if dt == torch.float32 else 0.95)
            one = loop_batch(mods, i0 + 1, 1, raw=True)
            u_one = torch.empty((1, res, res, 3), dtype=torch.uint8, device="cuda")
            net(ws[i0 + 1:i0 + 2], noise=one, rgb8_out=u_one)
            assert torch.equal(u_one[0], u_raw[1])          # position in the batch does not matter
            # the factors cannot be dropped silently - whatever treats a RawNoise as a sequence gets NORMALISED maps
            # (iteration, indexing, a comprehension), and the library forgets a batch's factors when its render call returns
            assert not isinstance(raw, list) and len(raw) == len(mods)
            as_list = [m[:, None] for m in raw]
            assert all(float((a[:, 0] - c).abs().max()) <= 4e-6 * float(c.abs().max()) for a, c in zip(as_list, nrm))
            u_lst = torch.empty_like(u_raw)
            net(ws[i0:i0 + nb], noise=as_list, rgb8_out=u_lst)     # right after a raw call: no stale factors on the handle
            assert float((u_lst == u_nrm).float().mean()) >= (0.999 if dt == torch.float32 else 0.95)
            assert torch.equal(raw[3], raw.normalised()[3])


@pytest.mark.parametrize("arch", ["orig", "resnet"])
def test_orig_and_resnet_architectures_match_the_oracle(arch):
    """inference/stylegan2.py:340-382 for the two non-default block architectures (layer-at-a-time on the operator kernels:
    maua_modconv2d, maua_upfirdn2d, maua_add) against the oracle restatement that g29 pins on the reference's own pieces."""
    from maua_amd.stylegan2 import SynthesisNetwork
    g = torch.Generator().manual_seed(17)
    B = 3
    for dtype in (torch.float32, torch.bfloat16):
        net = SynthesisNetwork(64, 64, 3, channel_base=2048, channel_max=128, dtype=dtype, architecture=arch,
                               generator=torch.Generator().manual_seed(5))
        p = net.state_dict()
        assert (sum(k.endswith("torgb.weight") for k in p) == 1) and (any(".skip." in k for k in p) == (arch == "resnet"))
        for k in p:
            if k.endswith(".bias") and "affine" not in k:
                p[k] = torch.randn(p[k].shape, generator=g) * 0.1
        net.load_state_dict(p)
        ws = torch.randn(B, net.num_ws, 64, generator=g)
        noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in net.layer_shapes()]
        u8 = torch.empty((B, 64, 64, 3), dtype=torch.uint8, device="cuda")
        img = net(ws, noise=noise).cpu()
        assert torch.equal(net(ws, noise=noise, rgb8_out=u8).cpu(), u8.cpu())
        ref = OS.synthesis_network(p, ws, noise=noise, architecture=arch)
        if dtype == torch.float32:
            assert float((img - ref).abs().max()) / float(ref.abs().max()) <= 2e-5
        else:
            assert psnr(img, ref) >= 50.0, psnr(img, ref)
        # the layers' own noise_const when no noise is handed in; clone() keeps the architecture
        img2 = net.clone()(ws).cpu()
        ref2 = OS.synthesis_network(p, ws, architecture=arch)
        assert (float((img2 - ref2).abs().max()) / float(ref2.abs().max()) <= 2e-5) if dtype == torch.float32 else psnr(img2, ref2) >= 50.0
        # frames: the same packing as the one-call network's (ops/io.py:47-70)
        want = ((img.clamp(-1, 1) + 1) / 2 * 255).round().to(torch.uint8).permute(0, 2, 3, 1)
        assert float((u8.cpu().int() - want.int()).abs().max()) <= 1
        with pytest.raises(NotImplementedError):
            net.keep_features(True)
            net.get_feature(0, B)


def test_resnet_block_module_matches_the_reference_composition(golden):
    """g29's resnet block (composed from the reference's own ops: Conv2dLayer skip, transposed convolution + upfirdn2d, conv1
    through the reference SynthesisLayer, `y + x`) through maua_amd.modules.SynthesisBlock on the device."""
    from maua_amd import modules as M, ops
    g = golden("g29_architectures")
    blk = M.SynthesisBlock(8, 6, w_dim=16, resolution=16, img_channels=3, is_last=False, architecture="resnet")
    sd = {k[len("blk__p__"):].replace("__", "."): v for k, v in g.items() if k.startswith("blk__p__")}
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda()
    x, img = blk(g["blk__x"].cuda(), None, g["blk__ws"].cuda())
    assert img is None
    ref = g["blk__out"]
    assert float((x.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    y = blk.skip(g["blk__x"].cuda(), gain=2 ** -0.5).cpu()
    assert float((y - g["blk__skip"]).abs().max()) <= 1e-5 * float(g["blk__skip"].abs().max())
    # maua_add itself: both dtypes, a length that is not a multiple of the vector width, unaligned views, in place
    gen = torch.Generator().manual_seed(1)
    for dt in (torch.float32, torch.bfloat16):
        a, b = torch.randn(100003, generator=gen).to(dt).cuda(), torch.randn(100003, generator=gen).to(dt).cuda()
        want = (a.float() + b.float()).to(dt)
        assert torch.equal(ops.add(a, b), want)
        assert torch.equal(ops.add(a[1:], b[1:]), want[1:])
        assert torch.equal(ops.add(a, b, out=a), want)

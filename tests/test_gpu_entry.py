"""BASELINE configs[0] plumbing on the GPU box: 32-frame 256x256 random-init render through the drop-in
maua.audiovisual entry points (synthetic clip: the MP3 named by configs[0] is missing upstream)."""
import json
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_wav(path, n_frames):
    from maua_amd.pipeline import synthetic_audio
    sr = 30720
    a = synthetic_audio(n_frames * 1024, sr)
    pcm = (a.clamp(-1, 1) * 32767).short().numpy()
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    return str(path)


@pytest.fixture(scope="module")
def wav(tmp_path_factory):
    return _write_wav(tmp_path_factory.mktemp("audio") / "clip.wav", 32)


@pytest.fixture(scope="module")
def wav_long(tmp_path_factory):
    # the sampler's salience weighting filters with sigma=80 (radius 320, reflect): like the reference it needs
    # clips longer than 320 frames
    return _write_wav(tmp_path_factory.mktemp("audio") / "clip352.wav", 352)


def test_generate_from_patch_memmap_and_ffmpeg(wav, tmp_path, monkeypatch):
    from maua_amd.audiovisual.generate import generate_audiovisal_from_patch, main
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    common = dict(audio_file=wav, model_file="None", patch_file="maua_amd/audiovisual/patches/examples/stylegan2.py",
                  patch_name=None, fps=30, out_size=(256, 256), resize_strategy="pad-zero", resize_layer=0)
    import os, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(repo)
    os.symlink(os.path.join(repo, "maua_amd"), tmp_path / "maua_amd")
    torch.manual_seed(0)
    video, (audio, sr) = generate_audiovisal_from_patch(renderer="memmap", renderer_kwargs={}, **common)
    assert video.shape == (32, 3, 256, 256) and video.dtype == np.uint8 and sr == 30720
    assert video.std() > 1.0  # not a constant image
    torch.manual_seed(0)
    out = generate_audiovisal_from_patch(renderer="ffmpeg", renderer_kwargs=dict(output_file=str(tmp_path / "o.mp4")),
                                         **common)[0]
    import shutil
    if not shutil.which("ffmpeg"):
        meta = json.loads(open(out + ".json").read())
        assert meta["frames"] == 32 and (meta["width"], meta["height"]) == (256, 256)
        raw = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(32, 256, 256, 3)
        # same seed, same frames: ffmpeg path (round-half-even pack) vs memmap path (truncating astype) differ <= 1
        d = np.abs(raw.transpose(0, 3, 1, 2).astype(int) - np.asarray(video).astype(int))
        assert d.max() <= 1


def test_sample_generate(wav, wav_long, tmp_path):
    """selfsupervised sampler at 256^2 (downscale 4), f32 parity mode."""
    from maua_amd._lib import MauaHipError
    from maua_amd.audiovisual.sample import generate
    with pytest.raises(MauaHipError):  # 32 frames < reflect padding of the sigma=80 filter (torch raises too)
        generate(wav, None, seed=5, out_dir=str(tmp_path))
    wav = wav_long
    out_file, frames = generate(wav, None, seed=5, fps=30, downscale_factor=4, batch_size=8, out_dir=str(tmp_path),
                                dtype=torch.float32)
    assert frames.shape == (352, 256, 256, 3) and frames.dtype == torch.uint8
    meta = json.loads(open(out_file.replace(".mp4", ".json")).read())
    assert meta["seed"] == 5 and len(meta["latent_patches"]) >= 2
    # reference-tail mode drops the tail like sample.py:90 (352 frames, B=8 -> 344)
    _, fr2 = generate(wav, None, seed=5, fps=30, downscale_factor=4, batch_size=8, out_dir=str(tmp_path),
                      dtype=torch.float32, reference_tail=True)
    assert fr2.shape[0] == 344 and torch.equal(fr2, frames[:344])
    # a saved patch file reproduces the render (sample.py:62-66 Patch.load)
    _, fr3 = generate(wav, None, patch_file=out_file.replace(".mp4", ".json"), seed=5, fps=30, downscale_factor=4,
                      batch_size=8, out_dir=str(tmp_path / "again"), dtype=torch.float32)
    assert torch.equal(fr3, frames)


def test_sample_generate_aspect_ratio(wav_long, tmp_path):
    """aspect_ratio != 1 (sample.py:53): the 1024 net resized at layer 0 ("stretch"), 768 x 512 at downscale 2."""
    import shutil
    from maua_amd.audiovisual.sample import generate
    out_file, frames = generate(wav_long, None, seed=3, fps=30, downscale_factor=2, aspect_ratio=1.5, batch_size=16,
                                out_dir=str(tmp_path))
    assert tuple(frames.shape) == (352, 512, 768, 3) and "768x512" in out_file and float(frames.float().std()) > 1.0
    if not shutil.which("ffmpeg"):
        meta = json.loads(open(out_file + ".json").read())
        assert (meta["width"], meta["height"], meta["frames"]) == (768, 512, 352)


def test_generate_non_native_size(wav, tmp_path, monkeypatch):
    """The CLI entry point at a size the network does not produce natively: the synthesizer resizes its features at
    resize_layer, rounds to the layer's multiple and force_output_size resamples to the request."""
    from maua_amd.audiovisual.generate import generate_audiovisal_from_patch
    monkeypatch.chdir(tmp_path)
    import os, warnings
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(repo)
    os.symlink(os.path.join(repo, "maua_amd"), tmp_path / "maua_amd")
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "resizes to multiples of ..." (200 is not a multiple of 1024 // 64)
        video, _ = generate_audiovisal_from_patch(
            audio_file=wav, model_file="None", patch_file="maua_amd/audiovisual/patches/examples/stylegan2.py",
            patch_name=None, renderer="memmap", renderer_kwargs={}, fps=30, out_size=(200, 136), resize_strategy="stretch",
            resize_layer=9)
    assert video.shape == (32, 3, 136, 200) and video.std() > 1.0
    # the ffmpeg renderer: frames are resampled by the patch's postprocess before the u8 pack
    import json, shutil
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, _ = generate_audiovisal_from_patch(
            audio_file=wav, model_file="None", patch_file="maua_amd/audiovisual/patches/examples/stylegan2.py",
            patch_name=None, renderer="ffmpeg", renderer_kwargs=dict(output_file=str(tmp_path / "n.mp4")), fps=30,
            out_size=(200, 136), resize_strategy="stretch", resize_layer=9)
    if not shutil.which("ffmpeg"):
        meta = json.loads(open(out + ".json").read())
        assert meta["frames"] == 32 and (meta["width"], meta["height"]) == (200, 136)
        raw = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(32, 136, 200, 3)
        assert raw.std() > 1.0


def test_ffmpeg_renderer_runs_the_patch_postprocess(wav, tmp_path, monkeypatch):
    """(advisor, round 1) A patch that overrides process_outputs is honoured at the native size too: the renderer
    packs u8 inside the synthesis call only when the postprocess is known to be the identity (the reference always
    calls postprocess(frame_batch) before writing, render/ffmpeg.py:72-73)."""
    import os
    import shutil
    from maua_amd.audiovisual.generate import generate_audiovisal_from_patch
    if shutil.which("ffmpeg"):
        pytest.skip("compares the raw-frame fallback of the writer")
    monkeypatch.chdir(tmp_path)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(repo)
    os.symlink(os.path.join(repo, "maua_amd"), tmp_path / "maua_amd")
    (tmp_path / "inv_patch.py").write_text(
        "from maua_amd.audiovisual.patches.examples.stylegan2 import ExampleSG2Patch\n"
        "class Inverted(ExampleSG2Patch):\n"
        "    def process_outputs(self, video):\n"
        "        return 1 - video\n")
    common = dict(audio_file=wav, model_file="None", patch_name=None, fps=30, out_size=(256, 256),
                  resize_strategy="pad-zero", resize_layer=0, renderer="ffmpeg")
    torch.manual_seed(0)
    plain = generate_audiovisal_from_patch(patch_file="maua_amd/audiovisual/patches/examples/stylegan2.py",
                                           renderer_kwargs=dict(output_file=str(tmp_path / "a.mp4")), **common)[0]
    torch.manual_seed(0)
    inv = generate_audiovisal_from_patch(patch_file=str(tmp_path / "inv_patch.py"),
                                         renderer_kwargs=dict(output_file=str(tmp_path / "b.mp4")), **common)[0]
    a = np.fromfile(plain + ".rgb24", dtype=np.uint8).astype(int)
    b = np.fromfile(inv + ".rgb24", dtype=np.uint8).astype(int)
    assert a.shape == b.shape and a.std() > 1.0
    assert np.abs((255 - a) - b).max() <= 1


def test_python_m_maua_audiovisual_generate(wav, tmp_path):
    """`python -m maua.audiovisual.generate ...` (the reference's command line, generate.py:57-98) runs from any
    directory with the repo on PYTHONPATH, with the reference's default patch file."""
    import os
    import shutil
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "maua.audiovisual.generate", "--audio_file", wav, "--model_file", "None",
                        "--out_size", "256,256", "--fps", "30", "--resize_strategy", "stretch", "--out_dir", str(tmp_path)],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [f for f in os.listdir(tmp_path) if f.startswith("clip_None_stretch_256x256")]
    assert outs, os.listdir(tmp_path)
    if not shutil.which("ffmpeg"):
        meta = json.loads(open(tmp_path / "clip_None_stretch_256x256.mp4.json").read())
        assert meta["frames"] == 32 and (meta["width"], meta["height"]) == (256, 256)


def test_sample_generate_with_fused_upscale(wav_long, tmp_path):
    """BASELINE configs[4] as a product call: generate(..., upscale=...) renders 256^2 frames and up-scales every batch x4 on the
    device (RealESRGANer.enhance's arithmetic per frame, several frames per network call) straight into this rank's writer - no
    gather (a 4096^2 frame is 48 MiB).  Without ffmpeg the raw part is inspected: frame count, geometry, and frame 5 is
    enhance() of the frame the plain render produces for the same seed (bf16, another batch size -> another summation order in
    the convolutions: PSNR >= 35 dB between the u8 images)."""
    import shutil
    from maua_amd.audiovisual.sample import generate
    from maua_amd.super import load_model
    out, frames = generate(wav_long, None, seed=5, fps=30, downscale_factor=4, batch_size=8, out_dir=str(tmp_path / "a"),
                           upscale="x4plus-anime", upscale_batch=3, upscale_random_init=True)
    assert frames is None and "x4plus-anime_1024x1024" in out
    plain_file, plain = generate(wav_long, None, seed=5, fps=30, downscale_factor=4, batch_size=8, out_dir=str(tmp_path / "b"))
    assert tuple(plain.shape) == (352, 256, 256, 3)
    if not shutil.which("ffmpeg"):
        assert out.endswith("_parts.txt")
        part = str(tmp_path / "a") + "/" + open(out).read().split("'")[1]
        meta = json.loads(open(part + ".json").read())
        assert meta["frames"] == 352 and (meta["width"], meta["height"]) == (1024, 1024)
        raw = np.memmap(part + ".rgb24", dtype=np.uint8, mode="r").reshape(352, 1024, 1024, 3)
        model = load_model("x4plus-anime", allow_random_init=True)
        want = model.enhance(plain[5].float().cpu().numpy())[0]
        d = np.asarray(raw[5]).astype(np.float64) - want.astype(np.float64)
        assert 10 * np.log10(255.0 ** 2 / max(float((d ** 2).mean()), 1e-12)) >= 35.0
    else:
        assert out.endswith(".mp4")


def test_device_counter_rng_matches_its_oracle_twin_and_seeds_the_synthetic_network():
    """Round 5: the build-owned counter RNG.  maua_philox_u32 is bit-exact against oracle/rng.py
    (itself pinned to the published known-answer vectors, CPU suite) at every offset; normals agree to the last ulps of the float32
    log / sin / cos (4e-6 absolute); a tensor filled in pieces equals the tensor filled at once.  The benchmark's synthetic
    generator drawn on the device (init_synthesis_params_device) holds exactly the twin's numbers - so any rank, any device and
    the host agree on the network without exchanging it - and renders the same frames as a network fed the same numbers from the
    host."""
    import numpy as np
    from maua_amd.rng import philox_normal, philox_u32
    from maua_amd.stylegan2 import SynthesisNetwork, init_synthesis_params_device
    from oracle import rng as OR
    for seed, stream, n, off in ((0, 0, 1, 0), (5, 7, 1000, 0), (5, 7, 999, 3), (2 ** 40 + 1, 2 ** 33 + 5, 4097, 2 ** 34 + 2), (9, 1, 6, 1)):
        got = philox_u32(seed, stream, n, off).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, OR.u32(seed, stream, n, off)), (seed, stream, n, off)
        z = philox_normal((n,), seed, stream, off, mean=0.25, std=2.0).cpu().numpy()
        assert float(np.abs(z - OR.normal(seed, stream, n, off, mean=0.25, std=2.0)).max()) <= 8e-6
    whole = philox_normal((3, 50, 7), 11, 4)
    parts = torch.cat([philox_normal((k,), 11, 4, offset=o) for o, k in ((0, 13), (13, 500), (513, 537))])
    assert torch.equal(whole.reshape(-1), parts)
    assert abs(float(whole.mean())) < 0.1 and abs(float(whole.std()) - 1) < 0.1
    # the synthetic generator: the k-th random tensor of the construction order = stream k of the seed
    p = init_synthesis_params_device(64, 64, channel_base=2048, channel_max=64, seed=3)
    rand_keys = [k for k, v in p.items() if v.is_cuda]
    assert rand_keys[0] == "bs.0.const" and rand_keys[1] == "bs.0.conv1.affine.weight" and len(rand_keys) == 1 + 3 * 9 + 2 * 5
    for j in (0, 1, 7, len(rand_keys) - 1):
        k = rand_keys[j]
        twin = torch.from_numpy(OR.normal(3, j, p[k].numel())).reshape(p[k].shape)
        assert float((p[k].cpu() - twin).abs().max()) <= 4e-6, k
    q = init_synthesis_params_device(64, 64, channel_base=2048, channel_max=64, seed=3)
    assert all(torch.equal(p[k], q[k]) for k in p)
    assert not torch.equal(p["bs.2.conv0.weight"], init_synthesis_params_device(64, 64, channel_base=2048, channel_max=64, seed=4)["bs.2.conv0.weight"])
    net_dev = SynthesisNetwork(64, 64, 3, channel_base=2048, channel_max=64, dtype=torch.bfloat16, _params=p)
    net_host = SynthesisNetwork(64, 64, 3, channel_base=2048, channel_max=64, dtype=torch.bfloat16, _params={k: v.cpu() for k, v in p.items()})
    ws = torch.randn(3, net_dev.num_ws, 64, generator=torch.Generator().manual_seed(1))
    a, b = (torch.empty((3, 64, 64, 3), dtype=torch.uint8, device="cuda") for _ in range(2))
    net_dev(ws, rgb8_out=a)
    net_host(ws, rgb8_out=b)
    assert torch.equal(a, b)


def test_device_drawn_clip_waveform_and_latent_schedule_match_their_host_twins():
    """The benchmark clip's waveform drawn on the device (maua_philox_clip_audio: SURVEY 8(d)'s tone + 2 Hz clicks + noise floor from
    the counter RNG) equals oracle/rng.py clip_audio to float32 rounding, has the signal model's structure (clicks only in the first 5 %
    of every half second), and the latent schedule bench.py builds from it with a device-initialised mapper
    (pipeline.synthetic_clip_latents(device_rng=True)) equals the ORACLE's composition - onset envelope, mapping network, spline loops,
    blend, gaussian - of the same waveform and mapper weights."""
    import numpy as np
    from maua_amd import pipeline
    from maua_amd.rng import clip_audio
    from maua_amd.stylegan2 import get_z_latents
    from oracle import audio as OA, latent as OL, rng as OR, stylegan2 as OSG
    sr = 30720
    for n, seed in ((1, 3), (4 * 15360 + 3, 1234), (100001, 7)):
        got = clip_audio(n, sr, seed=seed).cpu().numpy()
        assert float(np.abs(got - OR.clip_audio(seed, n, sr)).max()) <= 2e-6, (n, seed)
    y = clip_audio(4 * 15360, sr, seed=1234).cpu()
    tone = 0.3 * torch.sin(2 * torch.pi * 220 * torch.arange(4 * 15360, dtype=torch.float64) / sr).float()
    resid = (y - tone).reshape(4, 15360)                       # clicks + noise floor per half second
    assert float(resid[:, 768:].abs().max()) < 0.06 and float(resid[:, :768].abs().max()) > 0.08      # |0.01 n| stays small; clicks reach 0.1
    assert abs(float(resid[:, 768:].std()) - 0.01) < 1e-3
    T, fps, num_ws, w_dim = 600, 30, 18, 512
    keep = {}
    lat, info = pipeline.synthetic_clip_latents(T, fps, num_ws, w_dim, device_rng=True, keep=keep)
    assert info["audio_and_mapper"] == "device counter RNG" and keep["wav"].is_cuda and tuple(lat.shape) == (T, num_ws, w_dim)
    wav = keep["wav"].cpu()
    assert float((wav - torch.from_numpy(OR.clip_audio(1234, T * 1024, 1024 * fps))).abs().max()) <= 2e-6
    mp = {k: v.cpu() for k, v in keep["mapper"].state_dict().items()}
    twin = torch.from_numpy(OR.normal(0, (1 << 20) + 3, 512 * 512)).reshape(512, 512) / 0.01            # the 4th matrix = stream 2^20 + 3
    assert float((mp["fcs.3.weight"] - twin).abs().max()) <= 1e-3                                     # (values ~ 100: 4e-6 relative)
    env = OA.onsets(wav, 1024 * fps).squeeze(-1)
    pal = OSG.mapping_network(mp, get_z_latents("0-60", w_dim).float(), num_ws_=num_ws)
    half = pal.shape[0] // 2
    sub = (slice(None), slice(0, num_ws, 6), slice(0, w_dim, 64))
    low, high = OL.spline_loops(pal[:half][sub], T, 4), OL.spline_loops(pal[half:2 * half][sub], T, 4)
    want = OA.gaussian_filter(low * (1 - env[:, None, None]) + high * env[:, None, None], 2)
    got = lat.cpu()[sub]
    err = float((got - want).abs().max()) / float(want.abs().max())
    assert err <= 3e-4, err

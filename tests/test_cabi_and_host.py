"""CPU-only checks: the C-ABI library loads and exports what include/maua_hip.h declares; host-side logic of the
drop-in layer; the product refuses to run without a HIP device (no silent CPU fallback)."""
import ctypes
import json
import os
import wave

import numpy as np
import pytest
import torch

from maua_amd import _lib as L


def test_library_exports_every_declared_symbol():
    from maua_amd.build import build
    lib = build()
    l = ctypes.CDLL(str(lib))
    syms = L.declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(l, s)]
    assert not missing, missing
    l.maua_version.restype = ctypes.c_char_p
    assert b"gfx950" in l.maua_version()


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import maua_amd.audio as A
    import maua_amd.ops as M
    from maua_amd.stylegan2 import SynthesisNetwork
    with pytest.raises(L.MauaHipError):
        M.bias_act(torch.zeros(1, 1, 2, 2))
    with pytest.raises(L.MauaHipError):
        M.modulated_conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 3, 3), torch.ones(1, 8), padding=1)
    with pytest.raises(L.MauaHipError):
        A.stft(torch.zeros(4096))
    net = SynthesisNetwork(16, 16, channel_base=512, channel_max=32)
    with pytest.raises(L.MauaHipError):
        net(torch.zeros(1, net.num_ws, 16))
    # ctx creation itself fails loudly on a box without a HIP device
    p = ctypes.c_void_p()
    assert L.lib().maua_ctx_create(0, None, ctypes.byref(p)) != 0
    assert len(L.lib().maua_last_error()) > 0


def test_product_does_not_import_oracle():
    import pathlib
    root = pathlib.Path(L.__file__).parent
    for f in root.rglob("*.py"):
        txt = f.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, f


def test_frame_range_partition():
    from maua_amd.pipeline import frame_range
    for T in [0, 1, 7, 3600, 3601]:
        for W in [1, 2, 3, 8]:
            rs = [frame_range(T, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == T
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    assert frame_range(3600, 3, 8) == (1350, 1800)


def test_seeds_and_init_order(golden):
    from maua_amd.stylegan2 import MappingNetwork, get_z_latents, init_synthesis_params, parse_seeds
    assert parse_seeds("0-3,7") == [0, 1, 2, 7]
    g = golden("g12_seeds")
    assert torch.equal(get_z_latents("0-3,7")[:, :8], g["z"])
    g = golden("g08_synth_init")
    p = init_synthesis_params(32, w_dim=16, channel_base=256, channel_max=16, generator=torch.Generator().manual_seed(21))
    ref = {k.replace("__", "."): v for k, v in g.items()}
    assert set(ref) == set(p)
    assert all(torch.equal(ref[k], p[k]) for k in ref)
    g = golden("g07_mapping512")
    m = MappingNetwork(512, 0, 512, 18, generator=torch.Generator().manual_seed(11))
    assert abs(m.state_dict()["fcs.0.weight"].double().sum().item() - float(g["w0_sum"])) <= 1e-3


def test_host_constants_match_reference(golden):
    import maua_amd.audio as A
    import maua_amd.ops as M
    g = golden("g01_setup_filter")
    assert torch.equal(M.setup_filter([1, 3, 3, 1]), g["f"])
    g = golden("g09_mel")
    basis = A.mel(30720, 2048, fmax=11025.0)
    assert torch.allclose(basis[[0, 1, 63, 127]], g["basis_rows"], atol=1e-7)
    assert torch.allclose(basis.sum(1), g["basis_rowsum"], atol=1e-5)
    assert torch.allclose(A.mel_frequencies(130), g["mel_f"], rtol=1e-6)
    taps, r = A.gaussian_taps(2, 200)
    assert r == 8 and abs(float(taps.sum()) - 1) < 1e-6
    taps, r = A.gaussian_taps(5, 3)  # radius limited to 3 * len
    assert r == 9
    taps, _ = A.gaussian_taps(2, 200, causal=0, classic=True)
    assert float(taps[9:].abs().sum()) == 0.0
    taps, _ = A.gaussian_taps(2, 200, causal=0.5, classic=True)
    assert float(taps[9]) > 0


def test_wrappers_structure():
    from maua_amd.stylegan2 import StyleGAN2, get_generator_class
    assert get_generator_class("stylegan2") is StyleGAN2
    G = StyleGAN2(model_file=None, output_size=(64, 64), generator=torch.Generator().manual_seed(0))
    assert G.res == 64 and G.num_ws == 10 and G.synthesizer.output_size == (64, 64)
    assert G.synthesizer.layer_names[:3] == ["bs.0.conv1", "bs.0.conv1", "bs.1.conv0"]
    assert G.get_z_latents("0-4").shape == (4, 512)
    G1024 = StyleGAN2(model_file=None)
    assert G1024.res == 1024 and G1024.num_ws == 18 and len(G1024.synthesizer.layer_names) == 18
    # other sizes go through the feature-space resize, which needs the HIP device (no CPU fallback)
    from maua_amd._lib import MauaHipError
    from maua_amd.stylegan2 import resize_strategy
    if not torch.cuda.is_available():
        with pytest.raises(MauaHipError):
            StyleGAN2(model_file=None, output_size=(1920, 1080))
    assert resize_strategy(4, (4, 8), "stretch") == dict(mode="stretch")
    assert resize_strategy(16, (17, 30), "pad-0.5-left") == dict(mode="pad", padding=(14, 0, 0, 1), pad_how="constant",
                                                                 pad_value=0.5)
    assert resize_strategy(4, (5, 7), "pad-reflect-out")["padding"] == (1, 2, 0, 1)
    assert resize_strategy(4, (3, 4), "pad-0-out")["padding"] == (0, 0, -1, 0)       # negative = crop, as F.pad (:256-259, :294)
    with pytest.raises(ValueError):
        resize_strategy(4, (4, 8), "pad-zero")  # the reference's CLI default does not parse there either (Q7)
    sd = G.synthesizer.G_synth.state_dict()
    assert sd["bs.0.const"].shape == (512, 4, 4) and sd["bs.4.conv1.weight"].shape == (512, 512, 3, 3)


def test_patch_loading_and_audio_io(tmp_path):
    from maua_amd.audio_io import load_audio
    from maua_amd.audiovisual.patches.base import MauaPatch, get_patch_from_file
    cls = get_patch_from_file("maua_amd/audiovisual/patches/examples/stylegan2.py")
    assert cls.__name__ == "ExampleSG2Patch" and issubclass(cls, MauaPatch)
    with pytest.raises(Exception):
        get_patch_from_file("maua_amd/audiovisual/patches/examples/stylegan2.py", "Nope")
    # 16-bit stereo wav -> mono float, sliced, resampled to 1024*fps
    sr = 8000
    t = np.arange(sr * 2) / sr
    pcm = (np.stack([np.sin(2 * np.pi * 440 * t), np.zeros_like(t)], 1) * 32767).astype(np.int16)
    f = tmp_path / "a.wav"
    with wave.open(str(f), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    a, s = load_audio(str(f), offset=0.5, duration=1.0)      # (resampling to 1024 * fps runs on the device: test_gpu_audio)
    assert s == sr and len(a) == sr and a.dtype == torch.float32
    assert 0.2 < float(a.abs().max()) <= 0.6  # mono mean of (sine, 0)
    p = MauaPatch(str(f), fps=24)
    assert p.n_frames == round(2.0 * 24) and p.sr == sr


def test_compressed_audio_goes_through_ffmpeg(tmp_path, monkeypatch):
    """sample.py:17 (`torchaudio.load` of the .mp3 configs[0] names): formats scipy cannot read are piped through the ffmpeg /
    ffprobe executables.  The image has neither, so stand-ins on PATH play them: this pins the plumbing (argument order,
    channel de-interleave, mono mean, slice) and the error paths, not a decoder."""
    import os
    import stat
    from maua_amd.audio_io import load_audio, read_audio
    mp3 = tmp_path / "clip.mp3"
    mp3.write_bytes(b"not really an mp3")
    monkeypatch.setenv("PATH", str(tmp_path / "nobin"))
    with pytest.raises(NotImplementedError, match="ffmpeg"):
        read_audio(mp3)
    sr, n = 22050, 4410
    pcm = np.stack([np.linspace(-1, 1, n), np.linspace(1, -1, n) * 0.5], 1).astype("<f4")    # interleaved stereo
    (tmp_path / "pcm.bin").write_bytes(pcm.tobytes())
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "ffprobe").write_text(f"#!/bin/sh\necho '{{\"streams\": [{{\"sample_rate\": \"{sr}\", \"channels\": 2}}]}}'\n")
    (bindir / "ffmpeg").write_text(f"#!/bin/sh\necho \"$@\" > {tmp_path}/args.txt\ncat {tmp_path}/pcm.bin\n")
    for f in ("ffprobe", "ffmpeg"):
        os.chmod(bindir / f, os.stat(bindir / f).st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{bindir}:/usr/bin:/bin")
    a, got_sr = read_audio(mp3)
    assert got_sr == sr and tuple(a.shape) == (2, n) and np.array_equal(a.numpy(), pcm.T)
    args = (tmp_path / "args.txt").read_text().split()
    assert args[args.index("-i") + 1] == str(mp3) and args[args.index("-f") + 1] == "f32le" and args[args.index("-ac") + 1] == "2"
    mono, s2 = load_audio(str(mp3), offset=0.05, duration=0.1)
    lo = int(0.05 * sr)
    assert s2 == sr and np.allclose(mono.numpy(), pcm.mean(1)[lo: lo + int(0.1 * sr)])
    with pytest.raises(FileNotFoundError):
        read_audio(tmp_path / "missing.mp3")
    (bindir / "ffmpeg").write_text("#!/bin/sh\necho 'Invalid data found' >&2\nexit 1\n")
    with pytest.raises(RuntimeError, match="Invalid data"):
        read_audio(mp3)


def test_video_writer_raw_fallback(tmp_path):
    import shutil
    if shutil.which("ffmpeg"):
        pytest.skip("ffmpeg present")
    from maua_amd.video import VideoWriter
    out = tmp_path / "v.mp4"
    fr = torch.arange(2 * 4 * 6 * 3, dtype=torch.uint8).reshape(2, 4, 6, 3)
    with VideoWriter(str(out), (6, 4), 30) as v:
        v.write(fr)
        v.write(fr[0])
    raw = np.fromfile(str(out) + ".rgb24", dtype=np.uint8)
    assert raw.size == 3 * 4 * 6 * 3 and np.array_equal(raw[: fr.numel()], fr.numpy().ravel())
    meta = json.loads(open(str(out) + ".json").read())
    assert meta["frames"] == 3 and meta["width"] == 6
    with pytest.raises(TypeError):
        with VideoWriter(str(out), (6, 4), 30) as v:
            v.write(torch.zeros(4, 6, 3, dtype=torch.int32))


def test_generate_cli_argument_surface():
    """same flags/defaults as maua/audiovisual/generate.py:60-71"""
    import inspect
    from maua_amd.audiovisual import generate as G
    from maua_amd.audiovisual import sample as S
    sig = inspect.signature(G.generate_audiovisal_from_patch)
    assert list(sig.parameters) == ["audio_file", "model_file", "patch_file", "patch_name", "renderer", "renderer_kwargs",
                                    "fps", "out_size", "resize_strategy", "resize_layer"]
    sig = inspect.signature(S.generate)
    for name, default in [("fps", 30), ("downscale_factor", 4), ("batch_size", 32), ("aspect_ratio", 1)]:
        assert sig.parameters[name].default == default
    with pytest.raises(SystemExit):
        G.main([])  # --audio_file / --model_file are required


def test_pass_filters_have_no_host_path():
    """audio.py:96-112 low / high / band pass: the recurrence runs on the device (maua_sosfilt; parity in
    test_gpu_audio.py::test_iir_filters_and_percentile_clamps) - without one the call fails instead of filtering on the host."""
    import torch
    from maua_amd._lib import MauaHipError
    from maua_amd.audiovisual import audioreactive as ar
    if not torch.cuda.is_available():
        with pytest.raises(MauaHipError):
            ar.low_pass(np.zeros(1000), 30720, 200)


def test_maua_namespace_resolves_to_the_native_package():
    """the reference's module paths for this path import unchanged and name the maua_amd objects."""
    import importlib
    import pkgutil
    import maua
    for m in pkgutil.walk_packages(maua.__path__, "maua."):
        importlib.import_module(m.name)
    import maua_amd.ops as native_ops
    import maua_amd.stylegan2 as native_sg
    from maua.GAN.wrappers.inference import ops
    from maua.GAN.wrappers.stylegan2 import StyleGAN2, StyleGAN2Synthesizer
    from maua.audiovisual import audioreactive as ar
    from maua.audiovisual.audioreactive.selfsupervised.features import audio as FA, processing as FP
    from maua.audiovisual.audioreactive.selfsupervised.sample import generate
    from maua.audiovisual.generate import generate_audiovisal_from_patch
    from maua.audiovisual.patches.base import MauaPatch, get_patch_from_file
    from maua.audiovisual.patches.base.stylegan2 import StyleGAN2Patch
    assert ops.modulated_conv2d is native_ops.modulated_conv2d and ops.upfirdn2d is native_ops.upfirdn2d
    assert StyleGAN2 is native_sg.StyleGAN2 and StyleGAN2Synthesizer is native_sg.StyleGAN2Synthesizer
    assert callable(ar.onsets) and callable(ar.spline_loops) and callable(FA.mfcc) and callable(FP.gaussian_filter)
    assert callable(generate) and callable(generate_audiovisal_from_patch)
    import maua_amd.diffusion as native_df
    from maua.diffusion.processors.guided import GuidedDiffusion, create_models
    from maua.diffusion.sample import sample
    from maua.super.image.models.realesrgan import SRVGGNetCompact, load_model
    assert GuidedDiffusion is native_df.GuidedDiffusion and create_models is native_df.create_models and callable(sample)
    assert callable(load_model) and SRVGGNetCompact.__module__ == "maua_amd.super"
    # round 6: the text-prompt guidance modules under the reference's paths
    import maua_amd.grad as native_gr
    from maua.grad import CLIPGrads, GradModule
    from maua.loss import spherical_dist_loss
    from maua.ops.cutouts import MauaCutouts, make_cutouts, random_cutouts
    from maua.prompt import ContentPrompt, StylePrompt, TextPrompt
    assert CLIPGrads is native_gr.CLIPGrads and GradModule is native_gr.GradModule and MauaCutouts is native_gr.MauaCutouts
    assert callable(spherical_dist_loss) and callable(make_cutouts) and callable(random_cutouts)
    assert issubclass(StylePrompt, native_gr.ImagePrompt) and issubclass(ContentPrompt, native_gr.ImagePrompt) and TextPrompt("x", 2.0)()[1] == 2.0
    # the reference's default patch file resolves through the namespace to a class defined in that module
    cls = get_patch_from_file("maua/audiovisual/patches/examples/stylegan2.py")
    assert issubclass(cls, StyleGAN2Patch) and issubclass(cls, MauaPatch)
    assert cls.__module__ == "maua.audiovisual.patches.examples.stylegan2"


def test_video_writer_pipes_frames_to_ffmpeg(tmp_path, monkeypatch):
    """maua/ops/video.py:15-128: producer -> bounded queue -> writer thread -> ffmpeg stdin.  Neither image has an
    ffmpeg binary, so a stand-in named `ffmpeg` on PATH records its command line and copies stdin to the output file:
    the pipe path (rawvideo rgb24 on stdin, -s WxH, -r fps, audio input, back-pressure through a 2-slot queue, close /
    wait on exit) runs for real."""
    import os
    import stat
    import numpy as np
    from maua_amd.video import VideoWriter
    fake = tmp_path / "bin" / "ffmpeg"
    fake.parent.mkdir()
    fake.write_text("#!/bin/sh\nfor a in \"$@\"; do out=\"$a\"; done\necho \"$@\" > \"$out.cmd\"\nsleep 0.2\ncat > \"$out\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + os.environ["PATH"])
    out = tmp_path / "clip.mp4"
    rng = np.random.default_rng(0)
    frames = [torch.from_numpy(rng.integers(0, 256, (3, 8, 12, 3), dtype=np.uint8)) for _ in range(7)]
    with VideoWriter(str(out), (12, 8), 30, audio_file="song.wav", audio_offset=1.5, audio_duration=2.0,
                     ffmpeg_preset="fast", max_queue=2) as vw:
        for f in frames:
            vw.write(f)               # blocks while the 2-slot queue is full (the consumer sleeps first)
        with pytest.raises(TypeError):
            vw.write(torch.zeros(8, 12, 3, dtype=torch.int32))
    got = np.frombuffer(out.read_bytes(), dtype=np.uint8).reshape(21, 8, 12, 3)
    assert np.array_equal(got, torch.cat(frames).numpy()) and vw.frames_written == 21
    cmd = (tmp_path / "clip.mp4.cmd").read_text().split()
    assert cmd[cmd.index("-s") + 1] == "12x8" and cmd[cmd.index("-r") + 1] == "30" and cmd[cmd.index("-f") + 1] == "rawvideo"
    assert cmd[cmd.index("-ss") + 1] == "1.5" and cmd[cmd.index("-t") + 1] == "2.0" and "song.wav" in cmd
    assert cmd[cmd.index("-preset") + 1] == "fast" and cmd[-1] == str(out)


def test_bench_layer_table_matches_the_profile_slots():
    """bench.py's per-launch accounting (kernel name, algorithmic FLOPs / bytes per profile slot) must line up with what
    synth.hip launches for the BASELINE network: 1 styles slot + (conv slots incl. the two-slot tconv layers) + 9 toRGB
    slots + the u8 pack, the fused last-block walk carrying the whole block's work with its conv1 slot empty."""
    import importlib.util
    import pathlib
    from maua_amd.stylegan2 import SynthesisNetwork
    spec = importlib.util.spec_from_file_location("maua_bench", pathlib.Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Shapes:  # layer_table only needs the layer shapes of the 1024^2 network
        block_resolutions = [4, 8, 16, 32, 64, 128, 256, 512, 1024]

        def layer_shapes(self):
            ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
            out = [("bs.0.conv1", 512, 512, 4, 1)]
            for i, r in enumerate(self.block_resolutions[1:], 1):
                out += [(f"bs.{i}.conv0", ch[r // 2], ch[r], r, 2), (f"bs.{i}.conv1", ch[r], ch[r], r, 1)]
            return out

    rows = bench.layer_table(Shapes())
    names = [r[0] for r in rows]
    assert names[0] == "styles" and names[-1] == "pack_rgb8"
    assert len(rows) == 1 + 17 + 4 + 9 + 1            # 4 up-layers occupy two slots (tconv + FIR pass)
    kern = {r[0]: r[1] for r in rows}
    assert kern["bs.8.conv0"].startswith("upwalk_fused") and kern["bs.8.conv1"] == "(in the fused walk)"
    assert kern["bs.7.conv1"] == "modconv_hires_kernel<64,64,1>" and kern["bs.6.conv1"].startswith("modconv_dma_kernel<4,2,2,2,2,64>")
    total_gflop = sum(r[2] for r in rows)
    assert abs(total_gflop - 148.5) < 1.5, total_gflop   # 148.5 GFLOP per frame on minimal MACs
    assert bench.DEFAULT_BATCH == 128
    # the real class agrees with the stand-in on the shapes (cheap: no weights are drawn)
    assert [s[1:] for s in Shapes().layer_shapes()] == [tuple(s[1:]) for s in SynthesisNetwork.layer_shapes_for(1024)] \
        if hasattr(SynthesisNetwork, "layer_shapes_for") else True


def test_host_layer_signatures_match_the_reference():
    """The drop-in surface as data: argument names, their order and their defaults of every reference function / method the
    host layer mirrors (tests/golden/g25_signatures.json, read from the reference's source with ast by make_golden.py) against
    inspect.signature of ours.  Extra trailing keyword arguments of ours are allowed (dtype, generator, allow_random_init ...);
    the stated exceptions are defaults that name a CPU device (there is no CPU path) and container spellings."""
    import importlib
    import inspect
    import json
    from pathlib import Path
    sigs = json.loads((Path(__file__).parent / "golden" / "g25_signatures.json").read_text())
    assert len(sigs) >= 35
    immaterial = {"device", "postprocess_fn"}                      # cuda-if-available / torch.device("cpu") / lambda x: x
    for e in sigs:
        obj = importlib.import_module(e["ours"][0])
        for part in e["ours"][1].split("."):
            obj = getattr(obj, part)
        ours = inspect.signature(obj).parameters
        names = [n for n, p in ours.items() if p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
        ref_names = [a[0] for a in e["args"]]
        assert names[:len(ref_names)] == ref_names, (e["name"], e["reference"], ref_names, names)
        for n, d in e["args"]:
            p = ours[n]
            if d is None:   # (required in the reference; ours may add a default - random init for a missing checkpoint, say)
                continue
            assert p.default is not inspect._empty, (e["name"], n, "the reference has a default")
            if n in immaterial:
                continue
            try:
                want = eval(d, {"torch": torch})
            except Exception:
                continue                                             # (defaults that are expressions over other names)
            got = p.default
            if callable(want):   # torch.hann_window, torch.mean, a lambda: ours names the same callable, or None for "that default"
                assert got is None or got is want or (callable(got) and getattr(got, "__name__", "") == getattr(want, "__name__", "?")), \
                    (e["name"], e["reference"], n, d, got)
                continue
            if isinstance(want, (list, tuple)) or (torch.is_tensor(want) and want.dim() > 0):
                try:
                    want, got = [float(v) for v in want], [float(v) for v in got]
                except (TypeError, ValueError):     # (lists of names: perceptors=["ViT-B/16"])
                    want, got = list(want), list(got)
            elif torch.is_tensor(want):
                want, got = float(want), float(got)
            assert got == want, (e["name"], e["reference"], n, d, p.default)


def test_benchmark_setup_helpers_are_deterministic():
    """The set-up shortcuts of bench.py (round 4): per-tensor parallel weight draws and the fast benchmark waveform depend on their
    seed only (not on thread timing), keep the parameter set / the signal model, and leave the fixture waveform untouched."""
    import torch
    from maua_amd.pipeline import synthetic_audio
    from maua_amd.stylegan2 import init_synthesis_params, init_synthesis_params_parallel
    a = init_synthesis_params_parallel(64, 64, channel_base=2048, channel_max=64, seed=3, workers=4)
    b = init_synthesis_params_parallel(64, 64, channel_base=2048, channel_max=64, seed=3, workers=2)
    c = init_synthesis_params(64, 64, channel_base=2048, channel_max=64, generator=torch.Generator().manual_seed(3))
    assert set(a) == set(c) and all(a[k].shape == c[k].shape for k in a)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(a["bs.2.conv0.bias"], c["bs.2.conv0.bias"]) and not torch.equal(a["bs.2.conv0.weight"], c["bs.2.conv0.weight"])
    assert abs(float(a["bs.3.conv1.weight"].std()) - 1.0) < 0.02
    d = init_synthesis_params_parallel(64, 64, channel_base=2048, channel_max=64, seed=4, workers=4)
    assert not torch.equal(a["bs.1.conv1.weight"], d["bs.1.conv1.weight"])
    n, sr = 5 * 15360 + 77, 30720     # not a whole number of click periods: the slow placement path
    w1, w2 = synthetic_audio(n, sr, seed=9, fast=True), synthetic_audio(n, sr, seed=9, fast=True)
    assert w1.dtype == torch.float32 and tuple(w1.shape) == (n,) and torch.equal(w1, w2)
    whole = synthetic_audio(4 * 15360, sr, seed=9, fast=True)             # whole periods: the reshaped placement path
    tone = 0.3 * torch.sin(2 * torch.pi * 220 * torch.arange(4 * 15360, dtype=torch.float64) / sr).float()
    dev = (whole - tone).abs().reshape(4, 15360)
    assert float(dev[:, :768].mean()) > 3 * float(dev[:, 768:].mean())    # clicks sit in the first 5 % of every half second
    ref = synthetic_audio(4 * 15360, sr, seed=9)                          # the fixture waveform: same model, same level
    assert abs(float(ref.std()) - float(whole.std())) < 5e-3 and not torch.equal(ref, whole)


def test_counter_rng_twin_matches_the_published_known_answer_vectors():
    """oracle/rng.py (the CPU twin of csrc/rng.hip, the build-owned counter RNG of SURVEY 8(d)) against the Philox4x32-10
    known-answer vectors of the Random123 distribution (Salmon et al., SC'11) - the pin of an oracle that has no reference file to
    follow - and the stream addressing both sides share: element i = word i % 4 of counter i / 4, any offset, any length."""
    import numpy as np
    from oracle import rng
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = rng.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert [int(v) for v in got] == list(want)
    whole = rng.u32(5, 7, 41)
    assert all(np.array_equal(rng.u32(5, 7, n, offset=o), whole[o:o + n]) for o, n in ((0, 41), (3, 6), (4, 8), (17, 24), (40, 1)))
    assert not np.array_equal(rng.u32(5, 8, 41), whole) and not np.array_equal(rng.u32(6, 7, 41), whole)
    # stream 2^32 differs from stream 0 (the high counter words carry the stream), seeds use both key words
    assert not np.array_equal(rng.u32(1, 1 << 32, 8), rng.u32(1, 0, 8)) and not np.array_equal(rng.u32(1 << 32, 0, 8), rng.u32(0, 0, 8))
    z = rng.normal(1, 2, 400000)
    assert z.dtype == np.float32 and abs(float(z.mean())) < 6e-3 and abs(float(z.std()) - 1.0) < 5e-3 and bool(np.isfinite(z).all())
    assert np.array_equal(rng.normal(1, 2, 10, offset=6), z[6:16])
    assert np.allclose(rng.normal(1, 2, 10, mean=3.0, std=0.5), 3.0 + 0.5 * z[:10], atol=1e-6)

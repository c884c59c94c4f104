"""Beat tracking + Laplacian segmentation (maua_amd/segment.py, csrc/segment.hip; selfsupervised/mir.py:31-41) on the device
against oracle/segment.py and the reference fixture g22."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs():
    from test_oracle_golden import _segment_inputs
    return _segment_inputs()


def test_beat_dp_matches_oracle_on_clicks_and_on_the_clip_envelope():
    from maua_amd import audio as A
    from maua_amd import segment as SG
    from maua_amd.pipeline import synthetic_audio
    from oracle import segment as O
    T, period = 900, 10
    env = np.zeros(T, dtype=np.float32)
    env[7::period] = 1.0
    env += 0.05 * np.random.RandomState(0).rand(T).astype(np.float32)
    bpm = 60.0 * (22050 / 1024) / period
    want, *_ = O.beat_track(env, bpm)
    got = SG.beat_track(torch.from_numpy(env), bpm)
    assert got.dtype == np.int64 and np.array_equal(got, want) and np.all(np.diff(got) == period)
    # the 3600-frame BASELINE clip: 2 Hz clicks at 30 fps, tempo estimated like mir.py:27-30
    wav = synthetic_audio(3600 * 1024, 30720)
    onset = A.onsets(wav, 30720).reshape(-1)
    tempo = A.tempo(onset)
    want, local, cum, back = O.beat_track(onset.cpu().numpy(), tempo)
    got = SG.beat_track(onset, tempo)
    assert len(got) > 50 and np.array_equal(got, want)
    # slow tempo: a look-back window longer than one wave (period 43 -> 66 candidates), trim=True
    want, *_ = O.beat_track(onset.cpu().numpy(), 30.0, trim=True)
    assert np.array_equal(SG.beat_track(onset, 30.0, trim=True), want)
    assert SG.beat_track(torch.zeros(64), 120.0).size == 0
    with pytest.raises(Exception):
        SG.beat_track(onset, 1e6)          # period 0


def test_segment_kernels_match_reference_fixture(golden):
    from maua_amd import segment as SG
    g = {k: torch.as_tensor(np.asarray(v)) for k, v in golden("g22_segment").items()}
    env, beats = g["env"], [int(b) for b in g["beats"]]
    assert torch.equal(SG.sync(env, beats).cpu(), g["Csync"])
    R = SG.recurrence_matrix(g["Csync"], width=3, sym=True).cpu()
    assert torch.equal(R != 0, g["R"] != 0) and float((R - g["R"]).abs().max()) < 2e-6
    assert torch.equal(SG.timelag_median_filter(g["R"]).cpu(), g["Rf"])
    assert torch.equal(SG.median_filter_rows(g["ev"], 9).cpu(), g["evf"])
    for k in (2, 6, 16):
        X = g[f"km{k}_X"]
        mu, r, dist = SG.differentiable_k_means(X, k, 100)
        assert float((mu.cpu() - g[f"km{k}_mu"]).abs().max()) < 1e-5
        assert float((r.cpu() - g[f"km{k}_r"]).abs().max()) < 1e-5
        assert float((dist.cpu() - g[f"km{k}_dist"]).abs().max()) < 1e-5
    # mean aggregation (the "rosa" variant's MFCC path) and ragged spans
    x = torch.randn(50, 3, generator=torch.Generator().manual_seed(0))
    got = SG.sync(x, [1, 2, 30], "mean").cpu()
    want = torch.stack([x[0:1].mean(0), x[1:2].mean(0), x[2:30].mean(0), x[30:].mean(0)])
    assert float((got - want).abs().max()) < 1e-6
    with pytest.raises(ValueError):
        SG.sync(x, [5, 5])


def test_laplacian_segmentation_matches_oracle_and_finds_the_sections():
    from maua_amd import segment as SG
    from oracle import segment as O
    env, beats, order, bounds = _inputs()
    ks = (2, 4, 6, 8)
    got = SG.laplacian_segmentation(env, beats, ks=ks)
    want = O.laplacian_segmentation(env.numpy(), beats, ks=ks)
    for k, a, b in zip(ks, got, want):
        assert tuple(a.shape) == (640, k)
        # the same PARTITION of the frames: two centres that start in one section converge to the same mean, and which of
        # the twins carries the argmax is decided by the last bit (in the reference too), so labels are compared up to a
        # relabelling: the confusion matrix has one entry per row and per column
        la, lb = a.argmax(1).cpu().numpy(), b.argmax(1)
        conf = np.zeros((k, k), dtype=np.int64)
        np.add.at(conf, (la, lb), 1)
        assert conf.max(1).sum() >= 0.98 * len(la) and conf.max(0).sum() >= 0.98 * len(la), (k, conf)
    lab = got[1].argmax(1).cpu()
    mids = [int((bounds[s] + bounds[s + 1]) // 2) for s in range(5)]
    assert lab[mids[0]] == lab[mids[2]] and lab[mids[1]] == lab[mids[4]]
    assert len({int(lab[mids[0]]), int(lab[mids[1]]), int(lab[mids[3]])}) == 3
    with pytest.raises(ValueError):
        SG.laplacian_segmentation(env[:40], [10, 20, 30], ks=(2,))


def test_retrieve_music_information_returns_segmentations_and_patch_uses_them():
    from maua_amd.audiovisual import sample as S
    from maua_amd.pipeline import synthetic_audio
    n_frames, fps = 360, 30
    wav = synthetic_audio(n_frames * 1024, 1024 * fps, seed=5)
    feats, segs, tempo = S.retrieve_music_information(wav, 1024 * fps, ks=[2, 4, 6])
    assert set(feats) == set(S.ALLFEATS) and tempo > 0
    assert set(segs) == {(n, k) for n in S.ALLFEATS + ["rosa"] for k in (2, 4, 6)}
    for (name, k), s in segs.items():
        assert s.dtype == torch.int64 and tuple(s.shape) == (n_frames,) and 0 <= int(s.min()) and int(s.max()) < k
    patch = S.Patch(feats, segs, tempo, fps=fps, seed=3)
    assert patch.ks == [2, 4, 6]
    patch.latent_patches = [dict(patch_type="segmentation", segments=4, loop_bars=4, seq_feat="chromagram", seq_feat_weight=1,
                                 mod_feat="rms", mod_feat_weight=1, merge_type="average", merge_depth="all")]
    palette = torch.randn(20, 18, 512, generator=torch.Generator().manual_seed(0)).cuda()
    lat, noise = patch.forward(palette, downscale_factor=16)
    assert tuple(lat.shape) == (n_frames, 18, 512) and bool(torch.isfinite(lat).all())


def test_short_clips_skip_segmentations_and_saved_patches_fall_back(monkeypatch):
    """(advisor, round 2) a clip with <= 7 beat-synchronous frames gets no segmentations (and does not call the
    segmentation code, which raises on it); a clip with few beats drops the k it cannot be cut into, and a saved patch
    that asks for a dropped k renders with the nearest available one (or the "feature" form when there is none)."""
    from maua_amd import segment as SG
    from maua_amd.audiovisual import sample as S
    from maua_amd.pipeline import synthetic_audio
    n_frames, fps = 352, 30
    wav = synthetic_audio(n_frames * 1024, 1024 * fps, seed=5)
    palette = torch.randn(20, 18, 512, generator=torch.Generator().manual_seed(0)).cuda()
    sub = dict(patch_type="segmentation", segments=16, loop_bars=4, seq_feat="chromagram", seq_feat_weight=1,
               mod_feat="rms", mod_feat_weight=1, merge_type="average", merge_depth="all")
    # (a) three beats: nothing to segment
    monkeypatch.setattr(SG, "beat_track", lambda env, tempo: [60, 150, 260])
    feats, segs, tempo = S.retrieve_music_information(wav, 1024 * fps)
    assert segs == {} and set(feats) == set(S.ALLFEATS)
    patch = S.Patch(feats, segs, tempo, fps=fps, seed=3)
    assert patch.ks == [] and all(p["patch_type"] in ("feature", "loop") for p in patch.latent_patches)
    patch.latent_patches = [dict(sub)]  # what Patch.load of a JSON written for a longer clip installs
    lat, _ = patch.forward(palette, downscale_factor=16)
    assert tuple(lat.shape) == (n_frames, 18, 512) and bool(torch.isfinite(lat).all())
    # (b) nine beats: k <= 10 only; segments=16 falls back to k=8 (the nearest; 12 was dropped too)
    monkeypatch.setattr(SG, "beat_track", lambda env, tempo: list(range(30, 330, 34))[:9])
    feats, segs, tempo = S.retrieve_music_information(wav, 1024 * fps)
    assert {k for (_, k) in segs} == {2, 4, 6, 8}
    patch = S.Patch(feats, segs, tempo, fps=fps, seed=3)
    patch.latent_patches = [dict(sub)]
    a, _ = patch.forward(palette, downscale_factor=16)
    patch.latent_patches = [dict(sub, segments=8)]
    b, _ = patch.forward(palette, downscale_factor=16)
    assert torch.equal(a, b)

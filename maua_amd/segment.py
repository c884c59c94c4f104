"""Beat tracking and beat-synchronous Laplacian segmentation on the HIP device (drop-in for
maua/audiovisual/audioreactive/selfsupervised/features/rosa/segment.py: distance_matrix / recurrence_matrix :7-58,
median_filter1d :60-64, timelag_median_filter :74-82, init_plus_plus :85-103, differentiable_k_means :106-131,
laplacian_segmentation :134-209, laplacian_segmentation_rosa :220-267; and for the librosa call of mir.py:31
``rosa.beat.beat_track(onset_envelope=..., trim=False, hop_length=1024, bpm=tempo)``).

librosa, torch_geometric and scikit-learn are un-vendored: the beat tracker restates librosa's published dynamic program
(Ellis 2007; librosa.beat.__beat_tracker 0.8 - 0.10), the graph Laplacian is torch_geometric's
``get_laplacian(normalization="sym")`` written densely (I - D^-1/2 A D^-1/2, self loops removed, isolated nodes -> 0),
``laplacian_segmentation_rosa`` runs the same chain on CQT / MFCC features with a seeded k-means (the reference's
sklearn KMeans is unseeded, i.e. not reproducible even by itself).  Parity unpinned for those three pieces; everything
else follows the in-tree torch code line by line and is checked against oracle/segment.py.

Device work: maua_beat_dp, maua_segment_reduce, maua_recurrence_affinity, maua_timelag_median,
maua_median_filter_rows, maua_soft_kmeans (csrc/segment.hip); the symmetric eigendecomposition of the n_beats x n_beats
Laplacian is the one library call (torch.linalg.eigh on the device, as in the reference).  Host work is what the
reference also does on the host: the k-means++ seeding with numpy's RandomState (segment.py:85-103) and the backtrace
through the beat links."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import audio as A


# ---------------------------------------------------------------------------------------------------- beat tracker
def beat_track(onset_envelope, bpm, sr=22050, hop_length=1024, tightness=100.0, trim=False):
    """librosa.beat.beat_track(onset_envelope=env, bpm=bpm, hop_length=hop, trim=trim)[1] -> int64 numpy array of beat
    frames.  Reference quirk kept (Q11, as in audio.tempo): mir.py:31 does not pass ``sr``, so the beat period is
    round(60 * (22050 / hop) / bpm) frames whatever the true frame rate."""
    env = A._f32(onset_envelope).reshape(-1)
    T = env.numel()
    if T == 0 or not bool((env != 0).any()):
        return np.zeros(0, dtype=np.int64)                      # librosa: "no onsets -> no beats"
    if bpm <= 0:
        raise ValueError("bpm must be strictly positive")
    period = int(round(60.0 * (float(sr) / hop_length) / bpm))
    norm = env.std(unbiased=True)                               # __normalize_onsets: onsets / onsets.std(ddof=1)
    if float(norm) > 0:
        env = (env / norm).contiguous()
    local = torch.empty(T, dtype=torch.float64, device=env.device)
    cum = torch.empty(T, dtype=torch.float64, device=env.device)
    back = torch.empty(T, dtype=torch.int32, device=env.device)
    L.check(L.lib().maua_beat_dp(L.ctx(env.device), L.ptr(env), T, period, C.c_double(tightness), L.ptr(local), L.ptr(cum),
                                 L.ptr(back)))
    return beats_from_links(local.cpu().numpy(), cum.cpu().numpy(), back.cpu().numpy(), trim)


def beats_from_links(localscore, cumscore, backlink, trim=False):
    """librosa __last_beat + the backtrace + __trim_beats (host: a walk through <= T links)."""
    pad = np.pad(cumscore, 1, mode="edge")
    maxes = (cumscore > pad[:-2]) & (cumscore >= pad[2:])       # util.localmax
    med = np.median(cumscore[maxes])
    beats = [int(np.argwhere(cumscore * maxes * 2 > med).max())]
    while backlink[beats[-1]] >= 0:
        beats.append(int(backlink[beats[-1]]))
    beats = np.array(beats[::-1], dtype=np.int64)
    smooth = np.convolve(localscore[beats], np.array([0.0, 0.5, 1.0, 0.5, 0.0]), "same")   # scipy.signal.hann(5)
    if len(beats) < 5:                                          # np.convolve "same" is relative to the longer operand
        full = np.convolve(localscore[beats], np.array([0.0, 0.5, 1.0, 0.5, 0.0]), "full")
        smooth = full[2:2 + len(beats)]
    threshold = 0.5 * math.sqrt(float((smooth ** 2).mean())) if trim else 0.0
    valid = np.argwhere(smooth > threshold)
    return beats[int(valid.min()):int(valid.max())]            # (librosa's slice drops the last valid beat)


# ---------------------------------------------------------------------------------------------------- segment.py
def _rows(x):
    x = A._f32(x)
    return x.reshape(x.shape[0], -1).contiguous()


def sync(envelope, beats, aggregate="median"):
    """segment.py:152-155 (torch.median per inter-beat span = lower median) / librosa.util.sync(aggregate=np.mean)."""
    x = _rows(envelope)
    T, Cn = x.shape
    bounds = [0] + [int(b) for b in beats] + [T]
    if any(b1 >= b2 for b1, b2 in zip(bounds[:-1], bounds[1:])):
        raise ValueError("beats must be strictly increasing frames inside (0, len(envelope))")
    bd = torch.tensor(bounds, dtype=torch.int32, device=x.device)
    out = torch.empty((len(bounds) - 1, Cn), dtype=torch.float32, device=x.device)
    L.check(L.lib().maua_segment_reduce(L.ctx(x.device), L.ptr(x), T, Cn, L.ptr(bd), len(bounds) - 1,
                                        {"median": 0, "mean": 1}[aggregate], L.ptr(out)))
    return out


def recurrence_matrix(data, k=None, width=1, sym=False, bandwidth=None):
    """segment.py:23-57 (affinity mode): k nearest rows per column outside the +-width band, optionally symmetrised by
    the minimum, exp(-d / bandwidth) with bandwidth = median of the row maxima."""
    if not sym or bandwidth is not None:
        raise NotImplementedError("the segmentation chain calls recurrence_matrix(sym=True, bandwidth=None)")
    x = _rows(data)
    t, d = x.shape
    if k is None:
        k = 2 * np.ceil(np.sqrt(t - 2 * width + 1)) if t > 2 * width + 1 else 2
    k = int(k)
    rec = torch.empty((t, t), dtype=torch.float32, device=x.device)
    L.check(L.lib().maua_recurrence_affinity(L.ctx(x.device), L.ptr(x), t, d, k, width, L.ptr(rec)))
    return rec


def timelag_median_filter(rec):
    rec = A._f32(rec).contiguous()
    out = torch.empty_like(rec)
    L.check(L.lib().maua_timelag_median(L.ctx(rec.device), L.ptr(rec), rec.shape[0], L.ptr(out)))
    return out


def median_filter_rows(x, k):
    """median_filter1d(x.T, k, 1, k // 2).T of segment.py:60-64 (the way :193 uses it on the eigenvectors)."""
    x = A._f32(x).contiguous()
    out = torch.empty_like(x)
    L.check(L.lib().maua_median_filter_rows(L.ctx(x.device), L.ptr(x), x.shape[0], x.shape[1], k, L.ptr(out)))
    return out


def init_plus_plus(ds, k):
    """segment.py:85-103 k-means++ seeding on the host (float32 rows, numpy RandomState(42 + idx) draws)."""
    ds = np.asarray(ds, dtype=np.float32)
    picked = [0]
    d2 = ((ds - ds[0]) ** 2).sum(1)
    for idx in range(1, k):
        probs = d2 / (d2.sum() + 1e-8)
        cum = probs.cumsum()
        r = np.random.RandomState(42 + idx).rand()
        i = int(np.searchsorted(cum, r, side="right"))          # first j with r < cum[j]
        i = min(i, len(cum) - 1)
        picked.append(i)
        d2 = np.minimum(d2, ((ds - ds[i]) ** 2).sum(1))
    return ds[picked]


def differentiable_k_means(data, k, num_iter, cluster_temp=5):
    """segment.py:106-131 -> (mu [k, d], r [n, k], dist [n, k]) with d == k columns (the chain's use)."""
    x = _rows(data)
    n, d = x.shape
    if d != k:
        raise NotImplementedError("soft k-means runs on the first k eigenvectors (d == k)")
    x = (x / torch.linalg.vector_norm(x, dim=1, keepdim=True)).contiguous()
    mu0 = torch.from_numpy(init_plus_plus(x.cpu().numpy(), k)).to(x.device).contiguous()
    r = torch.empty((n, k), dtype=torch.float32, device=x.device)
    mu = torch.empty((k, k), dtype=torch.float32, device=x.device)
    L.check(L.lib().maua_soft_kmeans(L.ctx(x.device), L.ptr(x), n, k, L.ptr(mu0), num_iter, C.c_float(cluster_temp), L.ptr(r),
                                     L.ptr(mu)))
    return mu, r, x @ mu.t()


def sym_laplacian(A_):
    """torch_geometric.utils.get_laplacian(edge_index, edge_weight, normalization="sym") as a dense matrix."""
    A0 = A_ - torch.diag(torch.diagonal(A_))                    # remove_self_loops
    deg = A0.sum(1)
    dinv = deg.pow(-0.5)
    dinv = torch.where(torch.isinf(dinv), torch.zeros_like(dinv), dinv)
    return torch.eye(A_.shape[0], device=A_.device) - dinv[:, None] * A0 * dinv[None, :]


def _spectral_clusters(Rf, path_distance, ks, n_frames, soft=True):
    """segment.py:166-208 from the filtered recurrence matrix and the squared path distances on."""
    sigma = torch.median(path_distance)
    if not float(sigma) > 0:   # more than half of the beat-to-beat steps are exactly zero (a constant feature): the
        pos = path_distance[path_distance > 0]   # reference divides 0 / 0 here and raises out of eigh; keep going instead
        sigma = pos.mean() if pos.numel() else torch.ones((), device=path_distance.device)
    path_sim = torch.exp(-path_distance / sigma)
    R_path = torch.diag(path_sim, diagonal=1) + torch.diag(path_sim, diagonal=-1)
    deg_path, deg_rec = R_path.sum(1), Rf.sum(1)
    mu = deg_path.dot(deg_path + deg_rec) / torch.sum((deg_path + deg_rec) ** 2)
    A_ = mu * Rf + (1 - mu) * R_path
    Lm = sym_laplacian(A_)
    try:
        _, evecs = torch.linalg.eigh(Lm)
    except RuntimeError:           # segment.py:187-191: the general solver's real parts when the symmetric one gives up
        _, evecs = torch.linalg.eig(Lm)
        evecs = evecs.real.contiguous()
    evecs = median_filter_rows(evecs, 9)
    Cnorm = torch.cumsum(evecs ** 2, dim=1) ** 0.5
    if max(ks) > evecs.shape[1]:
        raise ValueError(f"{max(ks)} segments need at least as many beats (have {evecs.shape[1]})")
    out = []
    for k in ks:
        X = (evecs[:, :k] / Cnorm[:, k - 1:k]).contiguous()
        _, r, _ = differentiable_k_means(X, k, 100)
        if not soft:
            r = r.argmax(1)[:, None].float()
        out.append(torch.nn.functional.interpolate(r.T[None], size=n_frames, mode="nearest")[0].T)
    return out


def laplacian_segmentation(envelope, beats, ks=(2, 4, 6, 8, 12, 16)):
    """segment.py:134-209 -> list of soft one-hot segmentations [T, k] (mir.py:38 takes their argmax)."""
    env = _rows(envelope)
    Csync = sync(env, beats, "median")
    if Csync.shape[0] <= 7:
        raise ValueError("laplacian_segmentation needs more than 7 beat-synchronous frames")
    Rf = timelag_median_filter(recurrence_matrix(Csync, width=3, sym=True))
    path_distance = torch.sum(torch.diff(Csync, dim=0) ** 2, dim=1)
    return _spectral_clusters(Rf, path_distance, list(ks), env.shape[0])


def laplacian_segmentation_rosa(audio, sr, out_size, ks=(2, 4, 6, 8, 16), beats=None):
    """segment.py:220-267: the same chain on a beat-synchronous 36-bins-per-octave CQT (dB re max, median) with the
    MFCC path matrix (mean) -> int64 [out_size, len(ks)].  Deviations (librosa / sklearn un-vendored): the CQT, MFCC
    and recurrence matrix are this package's (the in-tree torch ports of the same librosa functions), ``beats``
    defaults to beat_track on the onset envelope at the estimated tempo (librosa re-estimates both from its own
    onset strength), and the hard assignment is the argmax of the seeded soft k-means instead of sklearn's unseeded
    KMeans."""
    from . import cqt as Q
    y = A._f32(audio).reshape(-1)
    Cq = Q.cqt(y, sr, hop_length=1024, bins_per_octave=36, n_bins=7 * 36)   # [252, T] magnitudes
    Cdb = 20.0 * torch.log10(torch.clamp(Cq, min=1e-5))         # amplitude_to_db(ref=np.max, amin=1e-5, top_db=80)
    Cdb = Cdb - 20.0 * torch.log10(torch.clamp(Cq.max(), min=1e-5))
    Cdb = torch.maximum(Cdb, Cdb.max() - 80.0)
    feat = Cdb.T.contiguous()
    if beats is None:
        env = A.onsets(y, sr).reshape(-1)
        beats = [int(b) for b in beat_track(env, A.tempo(env))]
        if beats and beats[0] == 0:
            del beats[0]
    beats = [b for b in beats if 0 < b < feat.shape[0]]
    Csync = sync(feat, beats, "median")
    Rf = timelag_median_filter(recurrence_matrix(Csync, width=3, sym=True))
    Msync = sync(A.mfcc(y, sr)[: feat.shape[0]], beats, "mean")
    path_distance = torch.sum(torch.diff(Msync, dim=0) ** 2, dim=1)
    segs = _spectral_clusters(Rf, path_distance, list(ks), out_size, soft=False)
    return torch.stack([s[:, 0] for s in segs], dim=1).long()

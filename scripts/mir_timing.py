"""Time retrieve_music_information (features + tempo + beats + segmentations) on the 3600-frame BASELINE clip:
python scripts/mir_timing.py"""
import sys, time, torch
sys.path.insert(0, ".")
from maua_amd import audio as A, segment as SG
from maua_amd.audiovisual import sample as S
from maua_amd.pipeline import synthetic_audio
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
wav = synthetic_audio(3600 * 1024, 30720)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feats, segs, tempo = S.retrieve_music_information(wav, 30720)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"pass {rep}: retrieve_music_information {t1 - t0:.3f} s, tempo {tempo:.2f}, {len(segs)} segmentations")
env = A.onsets(wav, 30720).reshape(-1)
torch.cuda.synchronize(); t0 = time.perf_counter()
beats = SG.beat_track(env, tempo)
t1 = time.perf_counter()
print(f"beat_track {1e3 * (t1 - t0):.1f} ms, {len(beats)} beats, median spacing {int(torch.as_tensor(beats).diff().median())} frames")
x = feats["chromagram"]
torch.cuda.synchronize(); t0 = time.perf_counter()
SG.laplacian_segmentation(x, [int(b) for b in beats if b > 0])
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"laplacian_segmentation (one feature, 6 ks) {1e3 * (t1 - t0):.1f} ms")
def lap(msg, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{msg:40s} {1e3 * (time.perf_counter() - t0):8.1f} ms"); return r
for fn in S.AFEATFNS:
    lap("feature " + fn.__name__, lambda: fn(wav, 30720))
bl = [int(b) for b in beats if b > 0]
lap("laplacian_segmentation_rosa", lambda: SG.laplacian_segmentation_rosa(wav, 30720, 3600, ks=(2, 4, 6, 8, 12, 16), beats=bl))
from maua_amd import cqt as Q
lap("  cqt 252 bins", lambda: Q.cqt(wav, 30720, hop_length=1024, bins_per_octave=36, n_bins=252))
lap("  mfcc", lambda: A.mfcc(wav, 30720))
lap("tempo", lambda: A.tempo(env))
lap("post-processing of 8 features", lambda: {k: A.normalize(A.salience_weighted(A.gaussian_filter(v, sigma=2))) for k, v in feats.items()})

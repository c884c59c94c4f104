#!/usr/bin/env python
"""Benchmark of the audio-reactive StyleGAN2 render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): random-init 1024x1024 StyleGAN2 SynthesisNetwork (seed 0), bf16 operands /
f32 accumulate, 3600-frame clip @30 fps (120 s of synthetic audio at 30 720 Hz), frames sharded by contiguous
range over the ranks.  One "step" = one batch of B frames through the per-batch hot path of
selfsupervised/sample.py:90-98 (reference): 17 index-addressed Loop noise maps -> StyleGAN2 synthesis forward ->
(x+1)/2 -> u8 HWC pack.  Latents (spline-loop schedule blended by the onset envelope) and network weights are
resident in HBM before the timed region, as the reference has them resident before its render loop.
After the K timed steps every rank renders its WHOLE frame range once (the "clip" leg: sustained rate over
3600 / N frames, then the one RCCL gather of the u8 shards to rank 0, timed) - reported as extra keys `sustained`,
`e2e` (SURVEY 8(d) metric (ii): set-up + render of the whole clip) and `gather_ms`; `value` stays the K-step rate.
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

T_FRAMES, FPS, RES, W_DIM = 3600, 30, 1024, 512
NOISE_SIZES = [4, 8, 8, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 512, 1024, 1024]  # patch.py:142-151
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


# what tests/ assert for this dtype against the fp32 CPU oracle (stated here so that the number is never read without its bar)
TOLERANCE = {
    "u8_frames_bf16": "the bench dtype: every u8 value within 1 LSB of the fp32 oracle's on all but <= 0.1 % of the pixels (never more "
                      "than 2), and <= 25 % of the pixels off by one (measured 18.9 %: bf16 feature rmse ~1e-3 of the image range "
                      "against a 7.8e-3 LSB) - tests/test_gpu_synth.py::test_full_size_u8_frames_match_oracle_on_a_non_saturating_network; PSNR of the f32 "
                      "image >= 60 dB (measured 65.9)",
    "u8_frames_exact_f32_mode": "SURVEY 8(d)'s bar: <= 1 LSB on <= 0.5 % of the pixels (same test, dtype float32)",
    "integers": "frame indices, onset-bin assignments, peak masks, order statistics: bit-exact",
    "audio_float": "onset envelope |err| <= 5e-4 of its [0, 1] range; STFT / mel / dB per tests/test_gpu_audio.py",
}
DEFAULT_BATCH = 128
CLIP_LEG_TIMEOUT_S = 240   # watchdog of the clip leg's exchange at N > 1 (the leg itself takes about a second)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH,
                    help="frames per step per GPU (the reference's sample.py defaults to 32 for 8-24 GB cards; 128 frames of "
                         "activations are ~10 GB of the MI355X's 288 GB and amortise the low-resolution layers: +8 %% frames/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--gather", choices=("cabi", "torch"), default=None,
                    help="transport of the clip leg's frame exchange at N > 1: the library's own RCCL communicator (default) or "
                         "torch.distributed's point-to-point calls on torch's RCCL group (also the automatic fallback)")
    ap.add_argument("--strict", action="store_true",
                    help="exit with status 3 (after printing the headline line) when the clip leg's exchange fails or hangs at N > 1; "
                         "by default the status stays 0 and the line carries \"ok\": false")
    ap.add_argument("--diffusion-arms", action="store_true",
                    help="configs[3] leg: also time the secondary model's other arithmetic modes (exact f32, bf16): four more 100-step loops")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the configs[3] (guided-diffusion DDIM) and configs[4] (render -> RealESRGAN x4) legs (extra keys)")
    return ap.parse_args()


def kernel_of(ci, co, res, up):
    """Which kernel instantiation synth.hip launches for a layer (mirrors launch_modconv_t / hires_supported /
    dma_conv_supported); names match the rocprofv3 kernel-trace rows."""
    hin = res // up
    if (ci, co, up) == (64, 32, 2) and hin % 64 == 0:
        # the last block as one walk (modconv_upwalk.hip): conv0 up -> conv1 -> toRGB + skip -> u8, features stay in LDS;
        # its conv1 / toRGB profile slots measure ~0
        return "upwalk_fused_kernel<64,32>"
    if (ci, co, up) in ((32, 32, 1), (64, 64, 1), (64, 32, 2)) and hin % 32 == 0:
        return f"modconv_hires_kernel<{ci},{co},{up}>"
    if up == 2 and 256 <= hin <= 512:
        # transposed conv + FIR + epilogue in one kernel, t in LDS (modconv_tconv_fir.hip; synth option tconv_fir = 256)
        return "tconv_fir_kernel"
    if up == 2 and 32 <= hin <= 512:
        # main position block on LDS-direct loads; the profile slot also holds the thin last row / column
        # (tconv2_kernel) and, where the producing conv1 could not pre-scale its output, the premod pass;
        # + upfir_epilogue_kernel in a second profile slot
        return "tconv_dma_kernel (+edges, +premod)"
    if up == 1 and 32 <= hin <= 512 and ci % 64 == 0 and co % 128 == 0:  # conv1 behind an up-layer that pre-scales its output
        if co % 256 == 0:
            return "modconv_dma_kernel<2,4,4,2,1,128>"   # 256-channel N tile, 128-byte K rows, one workgroup per CU
        return "modconv_dma_kernel<4,2,2,2,2,64>"        # 128-channel N tile, 64-byte K rows, two workgroups per CU
    cov = co * up * up
    if hin * hin <= 64 and ci % 64 == 0 and cov % 128 == 0:
        return "lowres_conv_kernel<bf16> (+premod, +epilogue)"  # one profile slot covers the three launches
    if up == 1 and hin * hin <= 256 and cov % 128 == 0:
        return "modconv3x3_kernel<bf16,4,1,2,1,9,64>"
    if cov % 128 == 0:
        k128 = ci % 64 == 0
        if hin * hin >= 4096:
            return "modconv3x3_kernel<bf16,4,4,2,1,3,%d>" % (128 if k128 else 64)
        return "modconv3x3_kernel<bf16,2,4,2,1,3,%d>" % (128 if (k128 and hin * hin < 256) else 64)
    return "modconv3x3_kernel<bf16,4,1,2,2,9,64>" if cov % 64 == 0 else "modconv3x3_kernel<bf16,4,1,2,1,9,64>"


def rgb_fused(c, r):
    """does the block's toRGB ride on its conv1 epilogue?  (register-stationary kernels; the LDS-direct kernel and the
    generic kernel when all channels sit in one N tile)"""
    k = kernel_of(c, c, r, 1)
    return k.startswith("modconv_hires") or (k.startswith("modconv_dma") and c in (128, 256)) or \
        (k.startswith("modconv3x3") and c == 128 and r * r >= 4096)


def layer_table(net):
    """per-launch algorithmic work of one forward (per frame): name, kernel, GFLOP (tconv-minimal MACs*2,
    SURVEY 8(a) table), algorithmic bytes (input read once + output written once + noise).  Fused toRGB work is
    accounted to the conv1 kernel that carries it; its profile slot then measures ~0."""
    rows = [("styles", "styles", 0.0, 0.0)]
    shapes = net.layer_shapes()
    li = 0
    walk_fused = False
    for i, r in enumerate(net.block_resolutions):
        for _ in range(1 if i == 0 else 2):
            pfx, ci, co, res, up = shapes[li]
            li += 1
            hin = res // up
            gflop = 2 * hin * hin * 9 * ci * co / 1e9
            byts = (hin * hin * ci + res * res * co) * 2 + res * res * 4  # bf16 in/out + f32 noise
            kern = kernel_of(ci, co, res, up)
            last = i == len(net.block_resolutions) - 1
            if last and kern.startswith("upwalk_fused"):
                # whole block: + conv1 (co -> co at res) + toRGB; HBM: input, both noise maps, skip image, u8 frame
                gflop += 2 * res * res * 9 * co * co / 1e9 + 2 * res * res * co * 3 / 1e9
                byts = hin * hin * ci * 2 + 2 * res * res * 4 + (res // 2) ** 2 * 12 + res * res * 3
                rows.append((pfx, kern, gflop, byts))
                walk_fused = True
                continue
            if last and up == 1 and walk_fused:
                rows.append((pfx, "(in the fused walk)", 0.0, 0.0))
                continue
            if up == 1 and rgb_fused(co, res):  # + fused toRGB: img write + upsampled skip read
                gflop += 2 * r * r * co * 3 / 1e9
                byts += r * r * 12 + (r // 2) ** 2 * 12
                if i == len(net.block_resolutions) - 1:  # last block: the features are not stored and the image
                    byts += r * r * 3 - res * res * co * 2 - r * r * 12  # leaves as u8 (no f32 image, no pack pass)
            if kern.startswith("tconv_fir"):  # two profile slots like the two-launch path; everything happens in the first
                rows.append((pfx + ".tconv", kern, gflop, byts))
                rows.append((pfx + ".upfir", "(in tconv_fir)", 0.0, 0.0))
            elif kern.startswith("tconv_dma"):  # two profile slots: MACs on the first, the output write on the second
                t_bytes = (res + 1) * (res + 1) * co * 2
                rows.append((pfx + ".tconv", kern, gflop, hin * hin * ci * 2 + t_bytes))
                rows.append((pfx + ".upfir", "upfir_epilogue_kernel<bf16>", 0.0, t_bytes + res * res * co * 2 + res * res * 4))
            else:
                rows.append((pfx, kern, gflop, byts))
        c = shapes[li - 1][2]
        fused = rgb_fused(c, r)
        rows.append((f"bs.{i}.torgb", "torgb(fused)" if fused else "torgb_kernel",
                     0.0 if fused else 2 * r * r * c * 3 / 1e9,
                     0.0 if fused else r * r * c * 2 + r * r * 12 + (r // 2) ** 2 * 12))
    rows.append(("pack_rgb8", "(u8 pack in the fused walk)" if walk_fused else "pack_rgb8_kernel", 0.0,
                 0.0 if walk_fused else RES * RES * 15))
    return rows


def build_inputs(device, rank, world, seed=0, host_rng=False):
    """Everything that is resident in HBM before the render loop.  Round 5: the network's random init (23.6 M values) and the 17
    noise modules' planes (8.4 M) are drawn ON THE DEVICE from the build-owned counter RNG (maua_philox_normal, SURVEY 8(d):
    identical on every rank / device, reproducible on the host through oracle/rng.py) - kernels instead of host draws + uploads.
    So are the clip's waveform and the mapper's random init (pipeline.synthetic_clip_latents(device_rng=True)): the whole set-up is
    device work behind a few milliseconds of host calls, on one thread.
    ``host_rng=True``: round 4's set-up (torch's host generators on helper threads), kept for A/B."""
    import threading
    from maua_amd.noise import Loop
    from maua_amd.stylegan2 import SynthesisNetwork
    from maua_amd import pipeline

    from maua_amd import _lib as L
    L.ctx(device)   # the per-device library context exists before the two threads use it
    box = {}

    def make_net():
        torch.cuda.set_device(device)
        from maua_amd.stylegan2 import init_synthesis_params_device, init_synthesis_params_parallel
        p = init_synthesis_params_parallel(RES, W_DIM, seed=seed) if host_rng else init_synthesis_params_device(RES, W_DIM, seed=seed)
        net = SynthesisNetwork(W_DIM, RES, 3, dtype=torch.bfloat16, _params=p)
        net._handle()
        box["net"] = net
    def make_noise():   # 17 Loop modules: three planes each
        torch.cuda.set_device(device)
        if host_rng:
            rng = torch.Generator().manual_seed(42)
            box["noise"] = [Loop(rng, T_FRAMES, (s, s), n_loops=4, sigma=5) for s in NOISE_SIZES]
        else:   # module j's planes = Philox stream j of seed 42 + seed (streams 0 .. 16; the network's tensors use their own seed)
            from maua_amd.rng import philox_normal
            if os.environ.get("MAUA_BENCH_SETUP_TRACE"):
                box["noise"] = []
                for j, s in enumerate(NOISE_SIZES):
                    ta = time.perf_counter()
                    pl = philox_normal((3, s, s), 42 + seed, j, device=device)
                    tb = time.perf_counter()
                    box["noise"].append(Loop(None, T_FRAMES, (s, s), n_loops=4, sigma=5, noise=pl))
                    print("[bench set-up]   plane %d (%d^2): draw %.2f ms, module %.2f ms" % (j, s, (tb - ta) * 1e3, (time.perf_counter() - tb) * 1e3),
                          file=sys.stderr)
                return
            box["noise"] = [Loop(None, T_FRAMES, (s, s), n_loops=4, sigma=5, noise=philox_normal((3, s, s), 42 + seed, j, device=device))
                            for j, s in enumerate(NOISE_SIZES)]
    if host_rng:
        ths = [threading.Thread(target=make_net), threading.Thread(target=make_noise)]
        for th in ths:
            th.start()
        latents, info = pipeline.synthetic_clip_latents(T_FRAMES, FPS, 18, W_DIM)
        for th in ths:
            th.join()
    else:   # everything is device work behind a few milliseconds of host calls: one thread, one stream
        trace = os.environ.get("MAUA_BENCH_SETUP_TRACE")
        t0 = time.perf_counter()
        make_net()
        if trace:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        make_noise()
        if trace:
            torch.cuda.synchronize(); t2 = time.perf_counter()
        latents, info = pipeline.synthetic_clip_latents(T_FRAMES, FPS, 18, W_DIM, device_rng=True)
        if trace:
            torch.cuda.synchronize()
            print("[bench set-up] network %.1f ms, noise planes %.1f ms, clip chain %.1f ms" %
                  ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3), file=sys.stderr)
    net, noise = box["net"], box["noise"]
    info = dict(info, weights_and_noise_planes="torch host generators" if host_rng else
                "device counter RNG (Philox4x32-10, maua_philox_normal): network = streams of seed %d, noise planes = streams of seed %d" % (seed, 42 + seed))
    assert net.num_ws == 18
    return net, latents.to(device), noise, info


def cpu_baseline(seconds):
    """The oracle (CPU restatement of the reference, fp32, PyTorch-CPU convs composed like ops.py) on the same
    workload, bounded: whole 1024^2 frames (B=1) until `seconds` of CPU time are spent (at least 1)."""
    from oracle import audio as OA
    from oracle import noise as ON
    from oracle import stylegan2 as OS
    from maua_amd.pipeline import synthetic_audio
    # oneDNN convolutions stop scaling (and regress) far below the 256 hardware threads of the GPU box's host
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    # the clip's audio pre-pass (once per clip - STFT, HPSS medians, iSTFT, mel, onset envelope of all
    # 3 686 400 samples), timed on its own: it belongs to the CPU path's whole-clip time, not to its per-frame rate
    ta = time.time()
    from maua_amd.rng import clip_audio
    wav = clip_audio(T_FRAMES * 1024, 1024 * FPS, seed=1234).cpu()   # the clip the device leg rendered (drawn on the device)
    env = OA.onsets(wav, 1024 * FPS)
    audio_s = time.time() - ta
    assert env.shape[0] == T_FRAMES
    p = OS.init_synthesis_params(RES, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    ws = torch.randn(1, OS.num_ws(RES), W_DIM, generator=g)
    planes = [torch.randn(3, s, s, generator=g) for s in NOISE_SIZES]
    idx = torch.linspace(0, 4 * 2 * torch.pi, T_FRAMES)
    n, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            noise = [ON.loop(pl, idx, n, 1, 5)[:, None] for pl in planes]
            img = OS.synthesis_network(p, ws, noise=noise)
            img.add(1).div(2).clamp(0, 1).mul(255).round().byte()
            n += 1
            if time.time() - t0 >= seconds:
                break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "kind_note": "the oracle/ CPU restatement of the reference's path (PyTorch-CPU convolutions composed like ops.py, fp32), "
                         "not the reference's own module (which cannot travel to the GPU box)",
            "host_cpu_count": os.cpu_count(), "audio_prepass_s": audio_s,
            "clip_seconds_extrapolated": audio_s + T_FRAMES / (n / dt),
            "threads_capped": torch.get_num_threads() < (os.cpu_count() or 1),
            "threads_note": "oneDNN convolutions stop scaling (and regress) well below the host's hardware threads; "
                            "32 threads measured fastest on the 256-thread host",
            "sample": f"the whole clip's audio pre-pass once ({audio_s:.1f} s) + {n} frame(s) of the same 1024x1024 workload "
                      f"(B=1, fp32 oracle: noise + synthesis + u8), {dt:.1f} s"}


def leg_traffic(leg):
    """HBM bytes per unit of work of an extra leg (profiles/rNN_<leg>_traffic.json of the newest round, scripts/collect_leg_traffic.py: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the leg's own script, all kernels summed) -> (bytes, note) or (None, why)."""
    cands = sorted((Path(__file__).resolve().parent / "profiles").glob(f"r*_{leg}_traffic.json"))   # the newest round's file
    p = cands[-1] if cands else Path(__file__).resolve().parent / "profiles" / f"{leg}_traffic.json"
    try:
        v = json.loads(p.read_text())
        return float(v["bytes_per_unit"]), f"replayed from profiles/{p.name}: {v['unit']} ({v['units_in_run']} units in the profiled run)"
    except Exception as e:
        return None, f"unmeasured ({type(e).__name__}: no PMC file for this leg)"


def live_leg_traffic(leg, batch, timeout_s=150):
    """HBM bytes per unit of an extra leg measured NOW, at the leg's own batch: two counter passes (rocprofv3
    --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE; separate child runs, the guide's unit and gfx950 corrections) over
    scripts/leg_probe.py - a few launch-by-launch steps of the leg - summed over EVERY kernel between the last two marker launches
    (diffusion: one ddim_step_kernel per guided step; upscale: one fused last-block walk per synthesis call).  (None, why) when the
    profiler is unavailable - the caller then replays the newest profiles/ file and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("MAUA_BENCH_NO_LIVE_TRAFFIC"):
        return None, "live counter passes disabled (MAUA_BENCH_NO_LIVE_TRAFFIC)"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself under rocprofv3"
    root = os.path.dirname(os.path.abspath(__file__))
    if leg == "diffusion":
        cmd = [sys.executable, os.path.join(root, "scripts", "leg_probe.py"), "diffusion", "--batch", str(batch), "--steps", "3"]
        marker, per, unit = "ddim_step_kernel", 1, f"one guided step (UNet forward + secondary forward / VJP + update) at batch {batch}"
    else:
        cmd = [sys.executable, os.path.join(root, "scripts", "leg_probe.py"), "upscale", "--frames", str(batch), "--steps", "2"]
        marker, per, unit = "upwalk_fused_kernel", batch, "one 1024^2 frame rendered and up-scaled x4 to 4096^2 u8"
    t0 = time.perf_counter()
    try:
        tot = {}
        with tempfile.TemporaryDirectory(prefix="maua_legtraffic_") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                env = dict(os.environ, TMPDIR="/tmp", MAUA_BENCH_NO_LIVE_TRAFFIC="1")
                subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", counter, "--", *cmd],
                               check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=root, timeout=timeout_s)
                f = glob.glob(os.path.join(tmp, "**", f"{counter}_counter_collection.csv"), recursive=True)[0]
                per_disp, names = {}, {}
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter:
                        d = int(r["Dispatch_Id"])
                        per_disp[d] = per_disp.get(d, 0.0) + float(r["Counter_Value"])
                        names[d] = r["Kernel_Name"]
                marks = sorted(d for d, nm in names.items() if marker in nm)
                if len(marks) < 2:
                    return None, f"the counter pass saw {len(marks)} launches of {marker}"
                lo, hi = marks[-2], marks[-1]
                tot[counter] = sum(v for d, v in per_disp.items() if lo < d <= hi)
        by = (2.0 * 1024 * tot["FETCH_SIZE"] + 1024 * tot["WRITE_SIZE"]) / per
        return by, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two separate child passes of "
                    "scripts/leg_probe.py at this leg's batch, every kernel between the last two %s launches summed; per %s "
                    "(FETCH_SIZE x 2 and KiB -> B per the guide's gfx950 corrections; %.0f s)" % (marker, unit, time.perf_counter() - t0))
    except Exception as e:   # noqa: BLE001
        return None, "live counter passes failed: " + repr(e)[:200]


def measured_traffic(kernel_name, batch=None):
    """HBM bytes per launch measured with rocprofv3 --pmc in an EARLIER run of this same command: profiles/traffic.json,
    written by scripts/collect_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes with the corrections of
    MI355X_MICROARCH.md section HBM (FETCH_SIZE x2 on gfx950, KiB -> B).  The file records the frames per step it was
    collected at; weights and halos do not scale with the batch, so a file collected at another batch is NOT used.
    (None, reason) if absent."""
    p = Path(__file__).resolve().parent / "profiles" / "traffic.json"
    if not p.exists():
        return None, "profiles/traffic.json absent"
    try:
        v = json.loads(p.read_text()).get(kernel_name)
        if v is None:
            return None, "kernel not in profiles/traffic.json"
        if batch and v.get("batch") and int(v["batch"]) != int(batch):
            return None, f"profiles/traffic.json was collected at {v['batch']} frames per step, this run uses {batch}"
        return float(v["bytes_per_launch"]), \
            "replayed from profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command at " \
            f"{v.get('batch', '?')} frames per step)"
    except Exception as e:
        return None, f"profiles/traffic.json unreadable: {e}"


def live_traffic(kernel_name, batch, timeout_s=90):
    """HBM bytes per launch of `kernel_name` measured NOW: the two counter passes of scripts/collect_traffic.py (rocprofv3
    --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE - separate runs, the guide's unit and gfx950 corrections) over a 2-step child run of
    this same command at this run's batch, so that `roofline.traffic` is evidence of THIS run and not a replay (VERDICT r4,
    measurement hygiene).  None + the reason when rocprofv3 is missing, the run is itself profiled, or a pass fails / times out - the
    caller then replays profiles/traffic.json and says so."""
    import shutil
    import tempfile
    if os.environ.get("MAUA_BENCH_NO_LIVE_TRAFFIC"):
        return None, "live counter passes disabled (MAUA_BENCH_NO_LIVE_TRAFFIC)"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself under rocprofv3"
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
    try:
        import collect_traffic as CT
        t0 = time.perf_counter()
        with tempfile.TemporaryDirectory(prefix="maua_traffic_") as tmp:
            fetch = CT.run("FETCH_SIZE", batch=batch, steps=2, timeout=timeout_s, out=os.path.join(tmp, "f"))
            write = CT.run("WRITE_SIZE", batch=batch, steps=2, timeout=timeout_s, out=os.path.join(tmp, "w"))
        f = [v for k, vals in fetch.items() if CT.bench_name(k) == kernel_name for v in vals]
        w = [v for k, vals in write.items() if CT.bench_name(k) == kernel_name for v in vals]
        if not f:
            return None, "the counter passes saw no launch of " + kernel_name
        # (drop the first launch of each pass: the warm-up step's cold caches)
        f, w = (f[1:] or f), (w[1:] or w)
        by = 2.0 * 1024 * sum(f) / len(f) + (1024 * sum(w) / len(w) if w else 0.0)
        return by, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two separate child passes of this command "
                    "at %d frames per step (%d + %d launches; FETCH_SIZE x 2 and KiB -> B per the guide's gfx950 corrections; %.0f s)"
                    % (batch, len(f), len(w), time.perf_counter() - t0))
    except Exception as e:   # noqa: BLE001 - the bench line must not depend on the profiler
        return None, "live counter passes failed: " + repr(e)[:200]


def extra_diffusion(batch=32, steps=100, size=256, clip_batch=8, full_arms=False, traffic=True):
    """configs[3] as BASELINE states it: guided-diffusion UNet (guided.py:171-190's architecture, random init), `steps`-step DDIM at
    `size`^2 with the reference's DEFAULT guidance (speed "fast": secondary-model forward + the gradient back through it every step,
    guided.py:236-272) towards image targets that switch with the clip's onset peaks (onset_prompt_schedule; text prompts need CLIP
    weights, which the image does not have: DESIGN section 2).  The whole guided loop - UNet, secondary forward, grad module,
    secondary VJP, DDIM update x `steps` - is one hipGraph (maua_ddim_guided_loop); timed after one untimed loop.  `value` is the
    guided rate; the unguided loop (rounds 3-4's number) is the `unguided` sub-key.  Roofline: algorithmic FLOPs of both networks."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
    from bench_diffusion import secondary_gflop, unet_gflop
    from maua_amd.diffusion import (GuidedDiffusion, ImageTarget, MSEGuide, SecondaryDiffusionImageNet2, create_models,
                                    onset_prompt_schedule)
    from maua_amd.pipeline import synthetic_audio
    model, diffusion, secondary = create_models("uncondImageNet256", f"ddim{steps}", allow_random_init=True, use_secondary=True,
                                                generator=torch.Generator().manual_seed(0))
    gf = unet_gflop(model, size, size)
    gf_sec = secondary_gflop(size, size, vjp=True)
    # ---- the unguided loop (no cond_fn): what rounds 3-4 reported for this leg
    x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(1)).cuda()
    xs = x.clone()
    diffusion.ddim_sample_loop(model, xs)          # capture + first replay (untimed)
    torch.cuda.synchronize()
    best_u = None
    for _ in range(1):   # (a replay of the captured loop: run to run within 0.3 %)
        xs.copy_(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, pred = diffusion.ddim_sample_loop(model, xs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best_u = dt if best_u is None else min(best_u, dt)
    unguided = {"value": batch / best_u, "unit": "samples/s", "seconds_per_batch": best_u, "hipgraph": model.graph_active(),
                "finite": bool(torch.isfinite(pred).all()), "tflops": gf * batch * steps / best_u / 1e3}
    # ---- the guided loop: 2 batches of the onset-switched schedule (one prompt per frame; frames either side of a switch share a batch)
    fps, n_frames = 30, 2 * batch
    wav = synthetic_audio(n_frames * 1024, 1024 * fps, seed=2)
    idx = onset_prompt_schedule(wav, 1024 * fps, fps, 2)
    g = torch.Generator().manual_seed(3)
    prompts = [ImageTarget(torch.randn(3, size, size, generator=g).clamp(-1, 1) * 0.5 + s_) for s_ in (-0.4, 0.4)]
    n = diffusion.num_timesteps

    def guided_leg(sec):
        gd = GuidedDiffusion([MSEGuide(1000.0)], timesteps=steps, model=model, diffusion=diffusion, secondary_model=sec)
        gr = torch.Generator().manual_seed(4)
        x0 = [torch.randn(batch, 3, size, size, generator=gr).cuda() for _ in range(2)]
        nz = [torch.randn(batch, 3, size, size, generator=gr).cuda() for _ in range(2)]

        def one(k):
            return gd.run(x0[k], [prompts[int(idx[k * batch + j])] for j in range(batch)], n - 1, n, noise=nz[k], per_sample=True)
        one(0)                                      # capture + first replay (untimed)
        torch.cuda.synchronize()
        best, out = None, None
        for k in range(1, 2):   # one timed replay, on the batch that straddles the prompt switch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = one(k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, bool(torch.isfinite(out).all()), model.guided_graph_active()
    # the default (create_models): the secondary model in float32 like the reference keeps it, its products as bf16 split products
    best, finite, graphed = guided_leg(secondary)
    arms = {}
    if full_arms:   # (--diffusion-arms: the secondary model's other arithmetic modes, rounds 4-5's sub-keys; 2 more 100-step loops each)
        sec32 = SecondaryDiffusionImageNet2(dtype=torch.float32, exact=True)   # every product on the exact-f32 matrix path
        sec32.load_state_dict(secondary.state_dict())
        best32, finite32, graphed32 = guided_leg(sec32)
        del sec32
        sec16 = SecondaryDiffusionImageNet2(dtype=torch.bfloat16)
        sec16.load_state_dict(secondary.state_dict())
        best16, finite16, graphed16 = guided_leg(sec16)
        del sec16
        arms = {"guided_exact_f32_secondary": {"value": batch / best32, "unit": "samples/s", "seconds_per_batch": best32, "hipgraph": graphed32,
                                               "finite": finite32, "over_unguided": best_u / best32,
                                               "note": "every product of the secondary model on the exact-f32 matrix path (v_mfma_f32_32x32x2_f32)"},
                "guided_bf16_secondary": {"value": batch / best16, "unit": "samples/s", "seconds_per_batch": best16, "hipgraph": graphed16,
                                          "finite": finite16, "over_unguided": best_u / best16,
                                          "note": "opt-in: secondary model in bf16 (guidance gradient 3.5 % off the reference's in L2 norm)"}}
    # ---- speed "regular" (guided.py:250-252): the loss gradient through the diffusion UNet itself - a kept forward + the network
    # walked backwards every step (maua_unet_forward_keep / maua_unet_vjp), the loop one hipGraph like the "fast" one
    # (maua_ddim_guided_loop without a secondary model); the whole 100-step loop is timed (round 6; round 5 extrapolated from 10 steps)
    def regular_leg():
        gd = GuidedDiffusion([MSEGuide(1000.0)], timesteps=steps, model=model, diffusion=diffusion, speed="regular")
        gr = torch.Generator().manual_seed(5)
        x0, nz = (torch.randn(batch, 3, size, size, generator=gr).cuda() for _ in range(2))
        pr = [prompts[int(idx[j])] for j in range(batch)]
        gd.run(x0, pr, n - 1, 2, noise=nz, per_sample=True)    # weights' transposed copies, arena (untimed, 2 steps)
        gd.run(x0, pr, n - 1, n, noise=nz, per_sample=True)    # capture + first replay of the 100-step loop (untimed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gd.run(x0, pr, n - 1, n, noise=nz, per_sample=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": batch / dt, "unit": "samples/s", "seconds_per_batch": dt, "ms_per_step": dt / steps * 1e3, "steps_timed": steps,
                "finite": bool(torch.isfinite(out).all()), "hipgraph": model.guided_graph_active(),
                "note": "speed 'regular': UNet forward (kept) + input gradient through the UNet + DDIM update per step, all %d steps timed" % steps}
    try:
        regular = regular_leg()
    except Exception as e:   # (an extra of an extra: never takes the leg down)
        regular = {"error": repr(e)[:300]}
    # ---- text-prompt guidance, configs[3] as BASELINE words it: CLIPGrads (maua/grad.py:96-165, the reference's defaults: ViT-B/16,
    # MauaCutouts cutn = 32, 8 cutout batches per step) as the grad module of the default "fast" conditioning - per step 8 x 32 cutouts
    # of the image estimate through the CLIP image tower forward AND backward (random init like the UNet: no weights in the image),
    # spherical distance to the prompts' embeddings (two prompts switched by the clip's onsets, handed in as embeddings: the text tower
    # runs once, off the loop), one hipGraph for the whole 100-step loop.  Smaller batch than the other arms: a step is ~9 x the UNet's FLOPs
    def clip_leg(cb):
        from bench_clip import vit_gflop
        from maua_amd.clip import load as clip_load
        from maua_amd.grad import CLIPGrads, EmbeddingPrompt
        cm, _ = clip_load("ViT-B/16", allow_random_init=True, generator=torch.Generator().manual_seed(6))
        gm = CLIPGrads(scale=1000.0, clip_models=[cm], clamp_gradient=0.05)
        gd = GuidedDiffusion([gm], timesteps=steps, model=model, diffusion=diffusion, secondary_model=secondary)
        gr = torch.Generator().manual_seed(7)
        tp = [EmbeddingPrompt(torch.randn(512, generator=gr)) for _ in range(2)]
        x0, nz = (torch.randn(cb, 3, size, size, generator=gr).cuda() for _ in range(2))
        pr = [tp[int(idx[batch - cb // 2 + j])] for j in range(cb)]   # (frames either side of the clip's prompt switch)
        torch.manual_seed(8)
        gd.run(x0, pr, n - 1, n, noise=nz, per_sample=True)      # eager step + capture + first replay (untimed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gd.run(x0, pr, n - 1, n, noise=nz, per_sample=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r, m = gm.merge_identical(gm.draw_rects(0, size, size, torch.tensor([500.0])))
        imgs = r.shape[0] * r.shape[1]                           # images through the tower per sample and step
        gf_clip = vit_gflop() * imgs
        tfc = (gf + gf_sec + gf_clip) * cb * steps / dt / 1e3
        del gd, gm, cm
        return {"value": cb / dt, "unit": "samples/s", "guidance": "clip", "batch": cb, "seconds_per_batch": dt, "ms_per_step": dt / steps * 1e3,
                "steps_timed": steps, "hipgraph": model.guided_graph_active(), "finite": bool(torch.isfinite(out).all()),
                "perceptor": "ViT-B/16 image tower, random init, bf16; cutn 32 x 8 cutout batches per step (the reference's defaults); the %d "
                             "identical whole-image cutouts of a cutout batch pass once with their joint weight: %d tower images per sample "
                             "and step instead of 256, same gradient (tests/test_gpu_clip.py)" % (32 // 4, imgs),
                "gflop_per_sample_step": {"unet_forward": gf, "secondary_forward_and_vjp": gf_sec, "clip_forward_and_input_gradient": gf_clip},
                "roofline": {"bound": "mfma", "achieved": tfc, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tfc / MFMA_BF16_PEAK_TF,
                             "note": "algorithmic FLOPs of the EXECUTED work (merged cutouts) of all three networks over the loop's time"},
                "over_unguided_rate": (cb / dt) / (batch / best_u)}
    try:
        clip_arm = clip_leg(clip_batch)
    except Exception as e:
        clip_arm = {"error": repr(e)[:300]}
    # ---- image prompts, the other prompt kind of configs[3]'s sentence (get_diffusion_model's list, maua/diffusion/image.py:92-97):
    # VGGGrads (style: Gram matrices of five vgg19 layers), ColorMatchGrads (hue histogram) and LPIPSGrads (content: vgg16 + lpips
    # heads) as the grad modules of the default "fast" conditioning - every loss AND its gradient inside the library (round 6:
    # csrc/perceptor.hip, colormatch.hip), the module LIST evaluated and summed inside the captured guided loop (maua_unet_set_guides,
    # csrc/guides.hip): one hipGraph for the whole loop like the other arms, all `steps` timed
    def image_prompt_leg(ib):
        from maua_amd.grad import ColorMatchGrads, ContentPrompt, LPIPSGrads, StylePrompt, VGGGrads
        from maua_amd.perceptors import VGG16_CFG, VGG19_CFG
        gr = torch.Generator().manual_seed(9)
        mods = [VGGGrads(scale=100.0, allow_random_init=True, generator=gr), ColorMatchGrads(scale=1e4),
                LPIPSGrads(scale=10.0, allow_random_init=True, generator=gr)]
        gd = GuidedDiffusion(mods, timesteps=steps, model=model, diffusion=diffusion, secondary_model=secondary)
        pr = [StylePrompt(img=torch.rand(1, 3, size, size, generator=gr)), ContentPrompt(img=torch.rand(1, 3, size, size, generator=gr))]
        x0, nz = (torch.randn(ib, 3, size, size, generator=gr).cuda() for _ in range(2))
        gd.run(x0, pr, n - 1, n, noise=nz)                       # workspaces, targets, capture + first replay (untimed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gd.run(x0, pr, n - 1, n, noise=nz)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0

        def vgg_gf(cfg, last):   # forward + input gradient (2 x) of features[: last + 1] at size^2, GFLOP per image
            f, i, cin, hw = 0.0, 0, 3, size * size
            for v in cfg:
                if i > last:
                    break
                if v == "M":
                    hw //= 4; i += 1
                else:
                    f += 2.0 * 9 * cin * v * hw; cin = v; i += 2
            return 2 * f / 1e9
        gf_p = vgg_gf(VGG19_CFG, 29) + vgg_gf(VGG16_CFG, 29)
        tfi = (gf + gf_sec + gf_p) * ib * steps / dt / 1e3
        return {"value": ib / dt, "unit": "samples/s", "guidance": "image prompts: VGGGrads (style) + ColorMatchGrads + LPIPSGrads (content)",
                "batch": ib, "seconds_per_batch": dt, "ms_per_step": dt / steps * 1e3, "steps_timed": steps, "hipgraph": model.guided_graph_active(),
                "finite": bool(torch.isfinite(out).all()),
                "perceptors": "vgg19.features[:30] and vgg16.features[:30] + lpips heads, random init, bf16 (no checkpoints in the image)",
                "gflop_per_sample_step": {"unet_forward": gf, "secondary_forward_and_vjp": gf_sec, "vgg_forward_and_input_gradient": gf_p},
                "roofline": {"bound": "mfma", "achieved": tfi, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tfi / MFMA_BF16_PEAK_TF},
                "over_unguided_rate": (ib / dt) / (batch / best_u),
                "note": "the three modules' gradients are summed per step inside the captured loop (guided.py:258-266)"}
    try:
        image_arm = image_prompt_leg(batch)
    except Exception as e:
        image_arm = {"error": repr(e)[:300]}
    tf = (gf + gf_sec) * batch * steps / best / 1e3
    del model, secondary
    torch.cuda.empty_cache()
    tr, tr_note = live_leg_traffic("diffusion", batch) if traffic else (None, "not collected (traffic=False)")
    if tr is None and traffic:
        why = tr_note
        tr, tr_note = leg_traffic("diffusion")
        tr_note += " - fallback (" + why + "); that file: UNet forward only at batch 8"
    res = {"metric": "samples/sec, guided-diffusion 256x256, 100-step DDIM, onset-switched prompts (configs[3])", "value": batch / best,
           "unit": "samples/s", "guided": True,
           "headline_arm": "guided_mse: image-MSE targets (speed 'fast'); the TEXT-PROMPT arm - configs[3] as BASELINE words it, CLIPGrads "
                           "through a ViT-B/16 image tower - is `guided_clip` below with its own FLOP count and roofline",
           "guidance": "speed 'fast' (reference default): secondary model forward + VJP every step, image-MSE grad module, one target "
                       "per frame switched at the clip's onset peaks",
           "secondary_dtype": "f32 tensors (the reference keeps the secondary model in fp32), products as three bf16 split products on "
                              "the bf16 matrix cores (MAUA_F32_SPLIT, ~2^-17 per product; guidance gradient within 1e-3 of the reference's "
                              "float32 autograd, tests/test_gpu_diffusion.py)",
           "prompt_switches_in_timed_frames": int((idx[batch + 1:n_frames] != idx[batch:n_frames - 1]).sum()),
           "batch": batch, "steps": steps, "seconds_per_batch": best, "ms_per_step": best / steps * 1e3, "dtype": "bf16",
           "data": "synthetic (random-init UNet of the reference's configuration, 552.8 M parameters; random-init secondary model, 13.9 M; "
                   "random-init CLIP ViT-B/16 image tower, 86 M)",
           "hipgraph": graphed, "finite": finite,
           "guided_clip": clip_arm,
           "guided_image_prompts": image_arm,
           "guided_regular": regular,
           "unguided": unguided, "guided_over_unguided": best_u / best,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TF,
                        "gflop_per_forward_per_sample": gf, "gflop_secondary_forward_and_vjp_per_sample": gf_sec,
                        "note": "both networks' algorithmic FLOPs over the guided loop's time against the bf16 peak (the secondary model's "
                                "share executes 3 bf16 products per algorithmic product)",
                        "traffic": tr, "traffic_note": tr_note}}
    res.update(arms)
    return res


def extra_f16(batch, steps=10):
    """The reference's own render dtype (render/ffmpeg.py:45, wrappers/__init__.py fp16=True) on the headline's kernels (round 6: the
    LDS-direct, transposed-conv + FIR, register-stationary and fused-walk kernels have float16 forms): `steps` synthesis calls of
    `batch` 1024^2 frames -> u8, float16 network, timed beside the SAME loop on the bf16 network (same parameters, same noise
    tensors; the noise-map kernels of the headline step are not part of either)."""
    from maua_amd.stylegan2 import SynthesisNetwork, init_synthesis_params_device
    p = init_synthesis_params_device(RES, W_DIM, seed=0)
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = torch.randn(batch, 18, W_DIM, generator=g, device="cuda")
    noise = [torch.randn(batch, 1, s_, s_, generator=g, device="cuda") for s_ in NOISE_SIZES]
    u8 = torch.empty((batch, RES, RES, 3), dtype=torch.uint8, device="cuda")
    out = {}
    for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        net = SynthesisNetwork(W_DIM, RES, 3, dtype=dt, _params=p)
        for _ in range(2):
            net(ws, noise=noise, rgb8_out=u8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(ws, noise=noise, rgb8_out=u8)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / steps
        net._destroy()
    return {"metric": "frames/sec, 1024^2 StyleGAN2 synthesis -> u8, float16 network (the reference's render dtype)", "value": batch / out["f16"],
            "unit": "frames/s", "dtype": "f16", "batch": batch, "steps": steps, "ms_per_step": out["f16"] * 1e3,
            "bf16_same_loop_ms_per_step": out["bf16"] * 1e3, "f16_over_bf16_rate": out["bf16"] / out["f16"],
            "note": "synthesis calls only (caller-supplied noise tensors); parity: tests/test_gpu_synth.py "
                    "test_synth_fp16_full_size_runs_the_fast_kernels (>= 65 dB against the fp32 oracle)"}


def extra_upscale(steps=3, frames=8, upscale_batch=4, traffic=True):
    """configs[4], one GPU's slice, through the product path (audiovisual/sample.py generate(upscale=...) runs exactly this per
    batch): `frames` 1024^2 StyleGAN2 frames rendered in one call -> RealESRGANer.enhance_frames (x4plus: 23 RRDB blocks, random
    init; the reference's enhance arithmetic per frame: / 255, reflect pre_pad 10 -> 1034^2, network, crop, clamp, round),
    `upscale_batch` frames per network call -> 4096^2 u8 frames on the device."""
    from maua_amd.stylegan2 import SynthesisNetwork
    from maua_amd.super import load_model
    G = SynthesisNetwork(W_DIM, RES, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
    up = load_model("x4plus", dtype=torch.bfloat16, allow_random_init=True)
    ws = torch.randn(frames, G.num_ws, W_DIM, generator=torch.Generator().manual_seed(1)).cuda()
    u8 = torch.empty((frames, RES, RES, 3), dtype=torch.uint8, device="cuda")

    def step():
        G(ws, rgb8_out=u8)
        for k in range(0, frames, upscale_batch):
            big = up.enhance_frames(u8[k:k + upscale_batch])
        return big
    big = step()
    assert tuple(big.shape) == (min(upscale_batch, frames), 4 * RES, 4 * RES, 3) and big.dtype == torch.uint8
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps * frames)
    f, g = 64, 32
    rdb = 9 * sum((f + k * g) * (f if k == 4 else g) for k in range(5))
    macs_px = 23 * 3 * rdb + 9 * (3 * f + f * f) + 9 * f * f * 4 + 9 * f * f * 16 + 9 * f * f * 16 + 9 * f * 3 * 16
    tf = 2 * macs_px * RES * RES / 1e12 / dt     # algorithmic: the 1024^2 frame (the pre_pad border's 2 % extra pixels are not counted)
    tr, tr_note = live_leg_traffic("upscale", 4) if traffic else (None, "not collected (traffic=False)")
    if tr is None and traffic:
        why = tr_note
        tr, tr_note = leg_traffic("upscale")
        tr_note += " - fallback (" + why + ")"
    return {"metric": "frames/sec per GPU, 1024^2 StyleGAN2 render -> RealESRGAN x4 -> 4096^2 u8 (configs[4], one GPU's slice)",
            "value": 1.0 / dt, "unit": "frames/s", "ms_per_frame": dt * 1e3, "dtype": "bf16", "data": "synthetic",
            "frames_per_render_call": frames, "frames_per_upscaler_call": upscale_batch,
            "path": "SynthesisNetwork(rgb8_out) -> RealESRGANer.enhance_frames (pre_pad 10), what generate(upscale='x4plus') runs per batch",
            "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TF,
                         "traffic": tr, "traffic_note": tr_note,
                         "frac_hbm_of_measured_traffic": None if tr is None else tr / dt / 1e9 / HBM_PEAK_GBS}}


def main():
    a = parse()
    if a.gather:
        os.environ["MAUA_GATHER"] = a.gather
    t_start = time.perf_counter()
    # host side = small tensors (seeds, 512 x 512 matrices, one 3.7 M-sample waveform): torch's intra-op pool sized for
    # the box's 128+ hardware threads costs more in fork/join than it saves (set-up 0.48 s -> 0.29 s on the GPU box;
    # with N ranks on one node the ranks would also oversubscribe each other); the CPU-baseline leg sets its own count
    torch.set_num_threads(min(torch.get_num_threads(), 8))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (maua_amd has no CPU path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MAUA_BENCH_BACKEND", "nccl")   # ("gloo": scripts/two_ranks_one_gpu.py, control flow on a 1-GPU box)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from maua_amd import _lib as L
    from maua_amd import pipeline
    from maua_amd.noise import loop_batch
    torch.zeros(1, device=device)
    torch.cuda.synchronize()
    # process warm-up, timed on its own and NOT part of the clip's set-up: a 16-frame 64^2 clip through the same code path makes
    # the HIP runtime allocate what it allocates once per process (staging buffers of pageable copies, the caching allocator's
    # first pools, code objects of the library's translation units) - a first call into libmaua_hip.so measured 20-60 ms for
    # 0.3 ms of work.  A process that renders clips pays this once, like the context itself.
    t_w = time.perf_counter()
    pipeline.warm_up(device)
    torch.cuda.synchronize()
    # (part of the once-per-process work: a full collection now, and the interpreter's long-lived objects - torch's import alone leaves
    #  over a million - out of the collector's way: a generation-2 pass over them, 35 ms, otherwise lands wherever the allocation
    #  counters trip, e.g. in the middle of a clip's 15 ms set-up)
    import gc
    gc.collect()
    gc.freeze()
    warmup_s = time.perf_counter() - t_w
    t_ctx = time.perf_counter()  # HIP context up and warm; everything after this is the clip's own set-up
    net, latents, noise, info = build_inputs(device, rank, world)
    net._handle()
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_ctx
    B = a.batch
    lo, hi = pipeline.frame_range(T_FRAMES, rank, world)
    # every timed step packs its frames into its own slot; the slots form a ring so that a long run (--steps in the
    # thousands) stays within HBM: 64 slots x 32 frames x 3 MiB = 6 GiB per rank, 8x that on rank 0 for the gather
    keep = max(1, min(a.steps, 64))
    out_u8 = torch.empty((keep, B, RES, RES, 3), dtype=torch.uint8, device=device)
    scratch_u8 = torch.empty((B, RES, RES, 3), dtype=torch.uint8, device=device)

    def step(k, u8):
        i = lo + (k * B) % max(1, (hi - lo) - B + 1)  # batches walk this rank's frame range
        nz = loop_batch(noise, i, B, raw=True)  # 17 Loop modules: the maps in one pass + per-sample factors (two launches)
        net(latents[i:i + B], noise=nz, rgb8_out=u8)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(a.warmup):
        step(k, scratch_u8)
    h = net._handle()
    lib = L.lib()
    L.check(lib.maua_synth_set_option(h, b"profile", 1))
    stamps = torch.zeros((2, 2), dtype=torch.int64, device=device)
    fence()
    L.check(lib.maua_ctx_clock_stamp(L.ctx(device), L.ptr(stamps[0])))
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k, out_u8[k % keep])
    L.check(lib.maua_ctx_clock_stamp(L.ctx(device), L.ptr(stamps[1])))
    fence()
    elapsed = time.perf_counter() - t0
    # average shader clock over the timed steps: shader-cycle counter over the constant 100 MHz counter (the part runs at its
    # power limit under this load, so the sustained clock - not the 2.4 GHz nominal - is what the MFMA fraction is paid in)
    d = (stamps[1] - stamps[0]).tolist()
    sclk_mhz = (d[0] / d[1] * 100.0) if d[1] > 0 else None
    # per-launch HIP-event durations recorded on the kernels' stream during the timed steps
    cnt = C.c_int()
    L.check(lib.maua_synth_get_profile(h, None, 0, C.byref(cnt)))
    ms = (C.c_float * max(1, cnt.value))()
    L.check(lib.maua_synth_get_profile(h, ms, cnt.value, C.byref(cnt)))
    L.check(lib.maua_synth_set_option(h, b"profile", 0))

    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_local = hi - lo

    def build_result(clip_s, gather_ms):
        """rank 0's JSON line from the timed steps (+ the clip leg's numbers when it ran)"""
        frames = world * B * a.steps
        rows = layer_table(net)
        per_fwd = len(rows)
        groups = {}
        nfwd = cnt.value // per_fwd if per_fwd else 0
        for f in range(nfwd):
            for j, (name, grp, gflop, byts) in enumerate(rows):
                g = groups.setdefault(grp, {"ms": 0.0, "gflop": 0.0, "bytes": 0.0, "launches": 0})
                g["ms"] += ms[f * per_fwd + j]
                g["gflop"] += gflop * B
                g["bytes"] += byts * B
                g["launches"] += 1
        # dominant kernel = the instantiation with the largest share of the GPU time of a step
        dom = max((g for g in groups if groups[g]["gflop"] > 0 or groups[g]["bytes"] > 0), key=lambda g: groups[g]["ms"])
        gd = groups[dom]
        roof_all = {}
        fused_rows = sorted(gn for gn in groups if gn.startswith("(") or gn == "torgb(fused)")
        for gname, g in groups.items():
            if g["ms"] <= 0 or gname in fused_rows:   # placeholder slots of work fused into another launch: no rate of their own
                continue
            roof_all[gname] = {"ms_per_launch": g["ms"] / g["launches"], "launches_per_step": g["launches"] // max(1, nfwd),
                               "tflops": g["gflop"] / g["ms"], "gbs": g["bytes"] / g["ms"] / 1e6,
                               "share_of_gpu_time": g["ms"] / sum(x["ms"] for x in groups.values())}
        # the roof that binds is the one the kernel is closer to (algorithmic flops vs dense bf16 MFMA peak,
        # algorithmic bytes vs HBM peak); both fractions are reported
        tf, gbs = gd["gflop"] / gd["ms"], gd["bytes"] / gd["ms"] / 1e6  # GFLOP/ms = TFLOP/s
        f_mfma, f_hbm = tf / MFMA_BF16_PEAK_TF, gbs / HBM_PEAK_GBS
        if f_mfma >= f_hbm:
            roof = {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": f_mfma}
        else:
            roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f_hbm}
        roof.update({"frac_mfma": f_mfma, "frac_hbm": f_hbm})
        traffic, traffic_source = (None, "N > 1") if world > 1 else live_traffic(dom, B)
        if traffic is None:
            replayed, src = measured_traffic(dom, B)
            traffic, traffic_source = replayed, src + " [" + traffic_source + "]"
        roof.update({"kernel": dom, "avg_launch_ms": gd["ms"] / gd["launches"], "launches_timed": gd["launches"],
                     "traffic": traffic, "traffic_source": traffic_source})
        # the whole step against both roofs: algorithmic FLOPs / bytes of every launch of the forward (+ the noise maps the
        # step writes and the synthesis reads: 2 x 4 B x sum of the 17 map sizes per frame) over the step's wall time
        step_gflop = sum(r[2] for r in rows) * B
        step_bytes = (sum(r[3] for r in rows) + 8 * sum(sz * sz for sz in NOISE_SIZES)) * B
        step_ms = elapsed / a.steps * 1e3
        roof["whole_step"] = {"tflops": step_gflop / step_ms, "frac_mfma": step_gflop / step_ms / MFMA_BF16_PEAK_TF,
                              "gbs": step_bytes / step_ms / 1e6, "frac_hbm": step_bytes / step_ms / 1e6 / HBM_PEAK_GBS,
                              "gflop_per_frame": step_gflop / B, "bytes_per_frame": step_bytes / B}
        res = {
            "metric": "frames/sec (whole node), 1024x1024 StyleGAN2 audio-reactive render",
            "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: 1024x1024 random-init StyleGAN2 (seed 0), 3600-frame clip @30fps "
                                   "(120 s synthetic audio @30720 Hz), per step: 17 Loop noise maps + synthesis + u8 pack",
                       "frames_per_step_per_gpu": B, "clip_frames": T_FRAMES, "frame_sharding": f"contiguous x{world}",
                       "latents": info},
            "ok": True,
            "tolerance": TOLERANCE,
            "roofline": roof, "kernels": roof_all, "fused_into_other_launches": fused_rows, "gather_ms": gather_ms,
            "sustained_sclk_mhz": sclk_mhz,
            # sustained: every rank's whole shard once, after the timed steps (max over ranks); e2e adds the set-up of
            # rank 0 (weight init + upload, synthetic audio, audio pre-pass, latent schedule, mapper, noise planes)
            "sustained": None if clip_s is None else {"clip_frames": T_FRAMES, "frames_per_gpu": n_local, "seconds": clip_s,
                                                      "fps": T_FRAMES / clip_s},
            "e2e": None if clip_s is None else {"clip_frames": T_FRAMES, "setup_s": setup_s, "process_warmup_s": warmup_s, "render_s": clip_s,
                    "gather_s": (gather_ms or 0.0) / 1e3,
                    "seconds": setup_s + clip_s + (gather_ms or 0.0) / 1e3,
                    "fps": T_FRAMES / (setup_s + clip_s + (gather_ms or 0.0) / 1e3),
                    # the same clip as a COLD process sees it (the first clip a process renders also pays process_warmup_s)
                    "seconds_cold": setup_s + (warmup_s or 0.0) + clip_s + (gather_ms or 0.0) / 1e3,
                    "fps_cold": T_FRAMES / (setup_s + (warmup_s or 0.0) + clip_s + (gather_ms or 0.0) / 1e3),
                    # definition marker: 1 = rounds 1-3 (one number, process warm-up inside, seeded reference draw order for the
                    # weights); 2 = round 4 on (warm-up timed apart, parallel weight init, fast synthetic audio, raw noise path) -
                    # `seconds` / `fps` of different versions are not comparable; `seconds_cold` is the closest to version 1
                    "definition_version": 2,
                    "includes": "weight init + upload, synthetic audio, HPSS onset pre-pass, latent schedule, mapper, "
                                "noise planes, render + u8 pack of every frame, gather (N > 1); excludes what a process pays once: the "
                                "HIP context, the first import and process_warmup_s (a 16-frame 64^2 clip through the same path)"},
        }
        res["config"]["gather"] = "streamed: finished chunks of frames_per_step frames travel to rank 0 on a side stream during the render"
        return res

    # ---- clip leg: this rank's whole frame range once (3600 / N frames, ceil((hi - lo) / B) steps) into a resident
    # buffer, the u8 shards travelling to rank 0 CHUNK BY CHUNK on a side stream while the next batch renders (RCCL
    # point-to-point over xGMI, distributed.StreamingGather) - the real end of a sharded render.  `sustained` = until
    # every rank has rendered its range; `gather_ms` = what is left of the exchange after that (the non-overlapped rest)
    del out_u8
    from maua_amd.distributed import StreamingGather
    # The exchange of the clip leg is the one part of this file that a 1-GPU box cannot exercise (N > 1: RCCL point-to-point
    # rounds on a side stream).  A watchdog keeps a stuck exchange from taking the measured headline down with it: after
    # CLIP_LEG_TIMEOUT_S every rank leaves, rank 0 first prints the line of the timed steps with the clip keys set to null.
    import threading

    def give_up():
        if rank == 0:
            res = build_result(None, None)
            res["clip_leg"] = f"no result within {CLIP_LEG_TIMEOUT_S} s - the streamed gather did not finish"
            res["ok"], res["clip_leg_failed"] = False, True
            print("\n" + json.dumps(res), flush=True)   # (own line even behind a partly flushed RCCL message)
        print(f"[bench rank {rank}] clip leg: no result within {CLIP_LEG_TIMEOUT_S} s, leaving", file=sys.stderr, flush=True)
        os._exit(3 if a.strict else 0)
    watchdog = threading.Timer(CLIP_LEG_TIMEOUT_S, give_up)
    watchdog.daemon = True
    if world > 1:
        watchdog.start()
    clip_s = gather_ms = full = sg = None
    from maua_amd.distributed import clip_batch
    # frames per call of the clip leg: a batch that divides the longest shard (3600 / N frames: 150 at N = 1, 2, 4, 8) - no ragged
    # last call - measured against the timed steps' batch B, the faster one is the leg's; every rank takes the
    # same one (the streamed gather's rounds are chunks of equal size)
    cb_div = clip_batch(pipeline.frame_range(T_FRAMES, 0, world)[1] - pipeline.frame_range(T_FRAMES, 0, world)[0], B)
    clip_runs = {}

    def clip_once(cb):
        nonlocal sg
        fence()
        sg = StreamingGather(T_FRAMES, (RES, RES, 3), cb, dtype=torch.uint8, device=device, rank=rank, world=world)
        fence()
        tc = time.perf_counter()
        for off, b in sg.chunks():
            i = lo + off
            net(latents[i:i + b], noise=loop_batch(noise, i, b, raw=True), rgb8_out=sg.local[off:off + b])
            sg.chunk_done()
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(device))
        done.synchronize()                      # the render stream only: the side stream may still be sending
        c_s = time.perf_counter() - tc
        out = sg.finish()
        fence()
        t_s = time.perf_counter() - tc
        if dist is not None:
            t = torch.tensor([c_s, t_s], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c_s, t_s = float(t[0].item()), float(t[1].item())
        return c_s, t_s, out
    try:
        if cb_div != B:   # the alternative first (untimed warm-up of its workspace size included in its own first calls)
            c_s, t_s, full = clip_once(B)
            clip_runs[B] = {"render_s": c_s, "total_s": t_s}
            del full, sg
            full = sg = None
        clip_s, total_s, full = clip_once(cb_div)
        clip_runs[cb_div] = {"render_s": clip_s, "total_s": total_s}
        clip_b = cb_div
        if cb_div != B and clip_runs[B]["total_s"] < total_s:   # the divisor batch lost: quote the timed steps' batch
            clip_s, total_s, clip_b = clip_runs[B]["render_s"], clip_runs[B]["total_s"], B
        gather_ms = max(0.0, total_s - clip_s) * 1e3 if dist is not None else None
    except Exception as e:   # N > 1 only: a failing exchange must not take the measured headline with it
        if world == 1:
            raise
        if rank == 0:
            res = build_result(None, None)
            res["clip_leg"] = f"failed: {type(e).__name__}: {e}"
            res["ok"], res["clip_leg_failed"] = False, True
            print("\n" + json.dumps(res), flush=True)   # (own line even behind a partly flushed RCCL message)
        import traceback
        print(f"[bench rank {rank}] clip leg failed: {type(e).__name__}: {e}\n{traceback.format_exc()}", file=sys.stderr, flush=True)
        os._exit(3 if a.strict else 0)
    watchdog.cancel()
    if rank == 0:
        assert tuple(full.shape) == (T_FRAMES, RES, RES, 3)
    gather_transport, gather_nranks = sg.transport, sg.nranks
    del full, sg

    if rank == 0:
        res = build_result(clip_s, gather_ms)
        res["config"]["gather"] += f"; transport: {gather_transport}"
        # the metric BASELINE states - the 3600-frame clip, sharded across the node's GPUs, WITH its gather - as a first-class key
        # beside `value` (which stays the K-step rate the driver's consistency check reads; at N > 1 that one is weak-scaling by
        # construction).  rccl_nranks is read back from the communicator that moved the frames, not echoed from the launcher.
        res["strong"] = {"metric": "frames/sec (whole node), the 3600-frame clip rendered and gathered once", "clip_frames": T_FRAMES,
                         "seconds": total_s, "fps": T_FRAMES / total_s, "render_s": clip_s, "gather_s_not_overlapped": max(0.0, total_s - clip_s),
                         "n_gpus": world, "rccl_nranks": gather_nranks, "transport": gather_transport, "scaling": "strong",
                         "frames_per_call": clip_b,
                         "frames_per_call_measured": {str(k): v for k, v in clip_runs.items()},
                         "frames_per_call_rule": "a batch that divides the shard (distributed.clip_batch), kept when it is not slower than the "
                                                 "timed steps' batch on this run"}
        if world == 1 and not a.no_extras:
            # the other single-GPU BASELINE configs, as extra keys (each with its own metric / roofline; never part of `value`)
            del latents, noise
            net._destroy()
            torch.cuda.empty_cache()
            for key, fn in (("f16", lambda: extra_f16(B)), ("diffusion", lambda: extra_diffusion(full_arms=a.diffusion_arms)),
                            ("upscale", extra_upscale)):
                try:
                    res[key] = fn()
                except Exception as e:   # an extra leg must not take the headline line down with it
                    res[key] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
        sys.stdout.flush()
        print(("\n" if world > 1 else "") + json.dumps(res), flush=True)   # (N > 1: own line even behind a partly flushed RCCL message)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

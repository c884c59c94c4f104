/* libmaua_hip.so — C ABI of the MI355X (gfx950) audio-reactive StyleGAN2 render path.
 *
 * This is the drop-in boundary.  The reference (maua-maua-maua/maua) has no compiled plugin
 * interface in-tree; its innermost replaceable surface is the Python operator layer
 * maua/GAN/wrappers/inference/ops.py (upstream equivalent: the CUDA plugins of
 * nv/torch_utils/ops, bound through torch_utils.custom_ops.get_plugin).  Each entry point below
 * names the reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types.  All tensor pointers are DEVICE pointers in the
 *    address space of ctx's device; the caller owns every buffer it passes in, the library owns only
 *    the context workspaces and the parameters uploaded with maua_synth_load().
 *  - every call returns MAUA_OK (0) or a negative error; maua_last_error() returns a thread-local text.
 *  - launches are asynchronous on the ctx stream; the only synchronising calls are maua_ctx_sync(),
 *    maua_synth_load() (host->device upload) and the *_host helpers that say so.
 *  - a ctx is confined to one host thread at a time; different ctxs are independent.
 *  - dtype: MAUA_F32 (exact-f32 MFMA path, parity mode), MAUA_BF16 (bf16 operands, f32 accumulate: the bench dtype, every fast
 *    kernel) or MAUA_F16 (IEEE half operands, f32 accumulate - the reference's own render dtype, render/ffmpeg.py:45 and
 *    wrappers/__init__.py fp16=True; the operator layer and maua_synth_* take it on the generic MFMA kernels, with ops.py:161-165's
 *    pre-normalisation of weights and styles; the diffusion / up-scaler networks do not).
 */
#ifndef MAUA_HIP_H
#define MAUA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAUA_OK 0
#define MAUA_ERR (-1)

enum maua_dtype { MAUA_F32 = 0, MAUA_BF16 = 1, MAUA_F16 = 2,
                  /* float32 tensors whose products run on the bf16 matrix cores as three split products (x = hi + lo, both bf16:
                   * hi_w hi_x + lo_w hi_x + hi_w lo_x, f32 accumulate; ~2^-17 per product - between TF32's 2^-11, the CUDA default for
                   * the reference's fp32 convolutions, and exact f32).  Accepted by maua_secondary_create only. */
                  MAUA_F32_SPLIT = 3 };
/* activation ids: reference ops.py:9-19 / :44-62 */
enum maua_act {
  MAUA_ACT_LINEAR = 0, MAUA_ACT_RELU = 1, MAUA_ACT_LRELU = 2, MAUA_ACT_TANH = 3, MAUA_ACT_SIGMOID = 4,
  MAUA_ACT_ELU = 5, MAUA_ACT_SELU = 6, MAUA_ACT_SOFTPLUS = 7, MAUA_ACT_SWISH = 8
};
/* gaussian_filter padding modes: reference signal.py:108-157 (F.pad modes) */
enum maua_pad_mode { MAUA_PAD_CIRCULAR = 0, MAUA_PAD_REFLECT = 1, MAUA_PAD_REPLICATE = 2, MAUA_PAD_CONSTANT = 3 };

typedef struct maua_ctx maua_ctx;
typedef struct maua_synth maua_synth;
typedef struct maua_rrdbnet maua_rrdbnet;
typedef struct maua_unet maua_unet;
typedef struct maua_srvgg maua_srvgg;

/* ---- context (library plumbing: no reference counterpart — the reference relies on torch's current device and
 * stream; a maua_ctx carries exactly that: device ordinal, HIP stream, a scratch arena) --------------------------- */
const char* maua_version(void);
const char* maua_last_error(void);
/* stream: a hipStream_t (NULL = the device's default stream); pass torch's current stream handle. */
int maua_ctx_create(int device, void* stream, maua_ctx** out);
int maua_ctx_set_stream(maua_ctx* ctx, void* stream);
int maua_ctx_sync(maua_ctx* ctx);
/* kernel-selection switches for A/B measurements and parity tests (no reference counterpart).
 * "dma_conv" (default 1): maua_modconv2d runs eligible shapes (bf16, 3x3, up 1, Ci % 64 == 0, Co % 128 == 0,
 * H % 8 == 0, W % 32 == 0) on the LDS-direct-load kernel of the synthesis hot path. */
int maua_ctx_set_option(maua_ctx* ctx, const char* key, int value);
/* Measurement aid (bench.py `sustained_sclk_mhz`): enqueue a one-thread kernel on the ctx stream that writes two device counters
 * to stamp_dev[0..1] (device memory, 2 x u64): [0] the shader-cycle counter (ticks with the power-managed shader clock), [1] the
 * constant-rate 100 MHz counter.  Two stamps around a region: (d[0] / d[1]) x 100 MHz = the average shader clock inside it.
 * No reference counterpart (the reference publishes no throughput measurement). */
int maua_ctx_clock_stamp(maua_ctx* ctx, unsigned long long* stamp_dev);
void maua_ctx_destroy(maua_ctx* ctx);

/* ---- B1: operator layer (NCHW contiguous, like the reference tensors) --------------------------- */
/* replaces the tensor additions of SynthesisBlock.forward, inference/stylegan2.py:360 (`x = y + x`, "resnet") and :373
 * (`img = img + y`) for callers that run the network one layer at a time: out = a + b over n elements, summed in f32 and
 * rounded to dtype (F32 / BF16); out may alias a or b. */
int maua_add(maua_ctx* ctx, const void* a, const void* b, void* out, long n, int dtype);
/* replaces ops.py:65-84 bias_act (upstream plugin nv/torch_utils/ops/bias_act).
 * y = clamp(act(x + b[c]) * gain); b may be NULL; clamp < 0 disables clamping. */
int maua_bias_act(maua_ctx* ctx, const void* x, const float* b, void* y, int N, int C, int H, int W, int dtype,
                  int act, float alpha, float gain, float clamp);
/* replaces ops.py:87-114 upfirdn2d (upstream plugin nv/torch_utils/ops/upfirdn2d).
 * f: [fh,fw] f32 2-D filter (correlation, no flip); gain multiplies f (caller passes the 2-D gain,
 * i.e. reference gain**(f.ndim/2) already resolved).  Output size
 * Ho = (H*up + py0 + py1 - fh) / down + 1 (likewise Wo); y must hold [N,C,Ho,Wo]. */
int maua_upfirdn2d(maua_ctx* ctx, const void* x, const float* f, int fh, int fw, void* y, int N, int C, int H, int W,
                   int dtype, int up, int down, int px0, int px1, int py0, int py1, float gain);
/* replaces ops.py:146-186 modulated_conv2d + :189-233 conv2d_resample (+ optionally the bias_act that
 * always follows it, stylegan2.py:238-250 / :270-271), fused.
 * x [N,Ci,H,W], weight [Co,Ci,k,k] f32 (k = 1 or 3), styles [N,Ci] f32, noise [N|1,1,H*up,W*up] f32 or NULL
 * (noise_batch_stride = 0 broadcasts), bias [Co] f32 or NULL, y [N,Co,H*up,W*up].
 * up in {1,2}; for up == 2 the filter is the 4x4 [1,3,3,1] outer product /64 with gain 4 (ops.py:211-225).
 * flip_weight = 1 gives the upstream-NVIDIA kernel flip.  act/gain/clamp as maua_bias_act;
 * pass act = MAUA_ACT_LINEAR, gain = 1, clamp = -1, bias = NULL for the bare modulated_conv2d. */
int maua_modconv2d(maua_ctx* ctx, const void* x, const float* weight, const float* styles, const float* noise,
                   long noise_batch_stride, float noise_strength, const float* bias, void* y, int N, int Ci, int Co,
                   int H, int W, int k, int up, int demodulate, int flip_weight, int act, float alpha, float gain,
                   float clamp, int dtype);
/* replaces render/ffmpeg.py:72 (.add(1).div(2)) + ops/io.py:47-70 tensor2bytes:
 * u8 = round_half_even(clamp((x+1)/2, 0, 1) * 255), NCHW f32 [B,3,H,W] -> HWC u8 [B,H,W,3]. */
int maua_pack_rgb8(maua_ctx* ctx, const float* img, uint8_t* out_hwc, int B, int H, int W);
/* ops/io.py:47-70 tensor2bytes itself, any value range and channel count (ops/video.py:66 feeds every frame of a
 * VideoWriter through it): u8 = round_half_even((clamp(x, mn, mx) - mn) / (mx - mn) * 255), NCHW f32 [B,C,H,W] -> HWC u8.
 * The range is double like the Python scalars: mx - mn is formed in double and then rounded to f32, every other step is f32. */
int maua_tensor2bytes(maua_ctx* ctx, const float* img, uint8_t* out_hwc, int B, int C, int H, int W, double value_min,
                      double value_max);

/* ---- B2: synthesis network (parameters uploaded once, batched forwards) ------------------------- */
/* replaces inference/stylegan2.py:385-436 SynthesisNetwork(w_dim, img_resolution, img_channels=3,
 * channel_base, channel_max), 'skip' architecture, conv_clamp 256.
 * dtype: activation/operand type.  nv_compat: bit0 flip up-layer kernels (Q2), bit1 scale noise by
 * the loaded noise_strength instead of 1 (Q4). */
int maua_synth_create(maua_ctx* ctx, int img_resolution, int w_dim, int channel_base, int channel_max, int dtype,
                      int nv_compat, maua_synth** out);
void maua_synth_destroy(maua_synth* net);
int maua_synth_num_ws(const maua_synth* net);      /* SynthesisNetwork.num_ws, inference/stylegan2.py:409-427 */
int maua_synth_num_layers(const maua_synth* net);  /* synthesis layers in execution order (17 at 1024^2) */
/* options: "keep_features" (0/1) keeps every layer's activation for maua_synth_get_feature (parity/debug);
 * "profile" (0/1) records HIP events on the ctx stream around every launch of a forward;
 * "tconv_up" (default 1) runs up-layers with 32^2..512^2 inputs as the minimal stride-2 transposed convolution + a
 * FIR/epilogue pass (0 = four 3x3 phase kernels everywhere, 4x the MACs; v > 1 = every up-layer with input size <= v);
 * "use_hires" (default 1) / "fuse_torgb" (default 1) select the register-stationary high-resolution kernels and
 * the toRGB fusion (0 = generic kernels everywhere, for A/B comparisons and parity tests). */
/* ---- arbitrary output sizes (maua/GAN/wrappers/stylegan2.py:104-151 change_output_resolution and
 * :216-340 get_hook).  The feature map is resized at ONE layer and every later layer runs at the scaled size.
 *  layer:  the reference's index into layer_names (0 = pre-hook on bs.0.conv1's input, L >= 1 = forward hook on
 *          the L-th entry, i.e. synthesis layer L-1 in execution order); -1 removes the resize.
 *  mode 0 "stretch": bicubic (align_corners False) to (target_h, target_w); the block's toRGB output is resized back
 *          and the block's image forward again, as the reference's rgb / img hooks do.
 *  mode 1 "pad-<how>-<where>": F.pad by (pad_left, pad_right, pad_top, pad_bottom) with pad_how (maua_pad_mode) /
 *          pad_value; inverse = crop.  Negative entries crop, as F.pad does - at layer 0 only (the pre-hook, :294): behind a
 *          later layer the reference's toRGB inverse slices with negative bounds and its forward fails (:278, :313-323).
 *  fill_noise_host: optional [C][target_h][target_w] f32 added to the resized FEATURES (the reference draws it once
 *          per hook from the per-channel mean/std of the first resized batch, :233-248; the caller owns the RNG).
 * Later layers need noise of their new size: maua_synth_layer_size reports it, maua_synth_load accepts a noise_const of
 * exactly that size. */
int maua_synth_set_resize(maua_synth* net, int layer, int mode, int target_h, int target_w, int pad_left, int pad_right,
                          int pad_top, int pad_bottom, int pad_how, float pad_value, const float* fill_noise_host);
/* geometric transform hooks (wrappers/stylegan2.py:153-194 apply_translation / apply_zoom / apply_rotation = kornia
 * translate / scale / rotate with padding_mode="reflection"): after layer `layer` (1-based index into layer_names) the
 * features are warped per sample, bilinear, reflection-padded, align_corners=True.  inv_matrices_dev [B][6] maps an
 * OUTPUT pixel (x, y) to its SOURCE pixel (row-major 2x3, device, owned by the caller and read by the next forwards);
 * NULL clears the slot.  Slots 0..2 on the same layer run in slot order (the reference registers translate, zoom,
 * rotate in that order). */
int maua_synth_set_warp(maua_synth* net, int slot, int layer, const float* inv_matrices_dev);
/* output size (h, w) of synthesis layer `layer` (execution order), or of the final image for layer == -1 */
int maua_synth_layer_size(const maua_synth* net, int layer, int* h, int* w);
/* the torch ops the hooks are made of, on NCHW tensors (dtype f32 / bf16): mode 0 = F.interpolate(x, (out_h, out_w),
 * mode="bicubic", align_corners=False); mode 1 = F.pad with left/top offsets (negative = crop), out size given;
 * mode 2 = bicubic with align_corners=True (maua/ops/image.py:240, the post-render resample). */
int maua_resize2d(maua_ctx* ctx, const void* x, void* y, int N, int C, int H, int W, int out_h, int out_w, int mode,
                  int pad_left, int pad_top, int pad_how, float pad_value, int dtype);
/* depthwise 1-D correlation of planar f32 [planes][H][W] along rows (axis 0) or columns (axis 1) with reflect padding
 * `radius` (taps [2*radius+1], device): the lanczos pre-filter of maua/ops/image.py:226-236 resample. */
int maua_conv1d_reflect(maua_ctx* ctx, const float* x, float* y, const float* taps, int radius, int axis, long planes,
                        int H, int W);
/* the adjoints of the two steps of `resample` (maua/ops/image.py:214-240) - what autograd gives a loss taken on a resampled image
 * (LPIPSGrads, maua/grad.py:191-192): gy -> gx through the reflect-padded 1-D correlation / through the bicubic interpolation
 * (gy device f32 [N][C][out_h][out_w] -> gx [N][C][H][W]; align_corners as maua_resize2d's mode 2 / 0) */
int maua_conv1d_reflect_vjp(maua_ctx* ctx, const float* gy, float* gx, const float* taps, int radius, int axis, long planes, int H, int W);
int maua_resize2d_bicubic_vjp(maua_ctx* ctx, const float* gy, float* gx, int N, int C, int H, int W, int out_h, int out_w, int align_corners);
int maua_synth_set_option(maua_synth* net, const char* key, int value);
/* the load_state_dict of the inference modules (inference/stylegan2.py:195-436 parameter / buffer names).
 * name: the reference state_dict key ("bs.3.conv0.weight", "bs.0.const", "bs.2.torgb.affine.bias",
 * "bs.1.conv1.noise_const", optional "bs.1.conv1.noise_strength").  host_data: HOST f32 array.
 * Synchronous.  Unknown names return MAUA_ERR ("resample_filter" buffers are accepted and checked). */
int maua_synth_load(maua_synth* net, const char* name, const float* host_data, size_t count);
/* the same with the values in DEVICE memory (e.g. drawn by maua_philox_normal): no host round trip for the large tensors */
int maua_synth_load_device(maua_synth* net, const char* name, const float* dev_values, size_t count);
/* replaces wrappers/stylegan2.py:65-102 StyleGAN2Synthesizer.forward + G_synth.forward(noise_mode="const").
 * ws [B,num_ws,w_dim] f32; noise: NULL or array of num_layers pointers, entry l = f32 [B|1, h_l, w_l] (or NULL
 * for that layer's noise_const), noise_batch_stride[l] in elements (0 = broadcast); img_out f32 [B,3,R,R]. */
int maua_synth_forward(maua_synth* net, const float* ws, const float* const* noise, const long* noise_batch_stride,
                       int B, float* img_out);
/* same, plus the u8 pack fused behind it (render/ffmpeg.py:72 + ops/io.py:47-70); img_out may be NULL. */
int maua_synth_render_rgb8(maua_synth* net, const float* ws, const float* const* noise,
                           const long* noise_batch_stride, int B, float* img_out, uint8_t* rgb8_out);
/* profile mode: per-launch durations (ms) of every forward since the last read, in launch order per forward:
 * styles, then per block [conv0 (two entries — transposed conv, FIR pass — when it runs as tconv_up),] conv1, torgb,
 * then pack_rgb8 if requested.  Synchronises on the last event;
 * a call with ms_out != NULL resets the recording (ms_out == NULL only returns the count). */
int maua_synth_get_profile(maua_synth* net, float* ms_out, int capacity, int* count);
/* Per-sample factors on the noise inputs of the NEXT maua_synth_render_rgb8 call - that one call only: it forgets them when it
 * returns (also on an error), so one batch's factors cannot leak into a later forward.  scales = device [num_layers][layer_stride >=
 * B] f32, row l for synthesis layer l (maua_noise_loop_batch_raw's output), or NULL for plain noise (x + noise * strength,
 * ops.py:184-185). */
int maua_synth_set_noise_scale(maua_synth* net, const float* scales, long layer_stride);
/* debug/parity (what a torch forward hook on SynthesisLayer would capture): copy layer l's activation (NHWC, net dtype) converted to f32 NCHW [B,C,h,w] after a forward. */
int maua_synth_get_feature(maua_synth* net, int layer, int B, float* out_nchw);

/* ---- audio pre-pass (once per clip).  n_fft = 2048, hop = 1024, periodic Hann, centre/reflect padding ---- */
/* Spectra are FRAME-major: spec[frame][bin] complex64 (re, im interleaved), 1025 bins. */
int maua_stft_num_frames(int n_samples); /* 1 + n_samples / 1024: torch.stft centre framing, rosa/spectral.py:10-21 */
/* replaces rosa/spectral.py:10-21 stft (torch.stft).  y [n_samples] f32 -> spec [frames][1025] complex. */
int maua_stft(maua_ctx* ctx, const float* y, int n_samples, float* spec);
/* replaces rosa/spectral.py:24-32 istft (torch.istft, center=True, length=...).  -> y [length] f32. */
int maua_istft(maua_ctx* ctx, const float* spec, int n_frames, int length, float* y);
/* |spec|**power over n complex values (rosa/spectral.py:59-62 spectrogram, :113-117 magphase). */
int maua_magnitude(maua_ctx* ctx, const float* spec, long n, float power, float* mag);
/* replaces processing.py:75-85 median_filter2d as hpss calls it: window 31, reflect padding 15, along time
 * (axis 0, "harmonic") or along frequency (axis 1, "percussive") of mag [n_frames][n_bins]. */
int maua_median31(maua_ctx* ctx, const float* mag, int n_frames, int n_bins, int axis, float* out);
/* replaces rosa/spectral.py:145-161 hpss (+ :120-142 softmask): complex spec in, harmonic / percussive complex
 * spectra out (either may be NULL). */
int maua_hpss(maua_ctx* ctx, const float* spec, int n_frames, float margin, float power, float* harm_out,
              float* perc_out);
/* replaces rosa/spectral.py:65-70 melspectrogram(power=2): mel[m][f] = sum_k basis[m][k] |spec[f][k]|^2 for
 * f < n_frames_used (the reference drops the last STFT column).  basis [n_mels][1025], mel [n_mels][n_frames_used]. */
int maua_mel_power(maua_ctx* ctx, const float* spec, int n_frames_used, const float* basis, int n_mels, float* mel);
/* replaces rosa/convert.py:7-12 power_to_db (in place on mel) + rosa/beat.py:13-21: env[f] = mean_m relu(dB lag-1
 * difference), left-padded by pad_width zeros, cropped to T. */
int maua_onset_from_mel(maua_ctx* ctx, float* mel_inout, int n_mels, int T, float amin, float top_db, int pad_width,
                        int aggregate /* 0 = torch.mean (beat.py:10), 1 = torch.median(...).values (:44) */, float* env);
/* the same transforms with any power-of-two n_fft <= 2048, any hop and the caller's window (device, [n_fft]):
 * rosa/beat.py:33-39 fourier_tempogram = stft(onset_envelope, n_fft=win_length, hop_length=1) and the istft of
 * :67.  spec [1 + n_samples / hop][n_fft / 2 + 1] complex64, frame-major. */
int maua_stft_general(maua_ctx* ctx, const float* y, int n_samples, int n_fft, int hop, const float* window_dev,
                      float* out_frames_bins_complex);
int maua_istft_general(maua_ctx* ctx, const float* spec_frames_bins_complex, int n_frames, int n_fft, int hop,
                       const float* window_dev, int length, float* y);
/* rosa/beat.py:50-64 (plp steps 3-4) in place on a frame-major tempogram: zero the bins whose tempo (tempo_freq_dev
 * [n_bins], BPM) lies outside [tempo_min, tempo_max], keep only the bins at each frame's peak of log1p(1e6 |z|), divide
 * by finfo.tiny ** 0.5 + the frame's largest magnitude. */
int maua_plp_select(maua_ctx* ctx, float* tempogram_frames_bins_complex, int n_frames, int n_bins,
                    const float* tempo_freq_dev, float tempo_min, float tempo_max);
/* the autocorrelation tempogram behind selfsupervised/mir.py:27-30 (rosa.beat.tempo -> librosa.feature.tempogram,
 * un-vendored: published algorithm): ac[f][l] = sum_n (w[n] e[f+n]) (w[n+l] e[f+n+l]) for l < n_lags <= win, over the
 * n_frames hop-1 frames of env_padded [n_frames + win - 1] (the caller pads the envelope by win/2 ramp samples). */
int maua_autocorr_frames(maua_ctx* ctx, const float* env_padded, const float* window, int n_frames, int win,
                         int n_lags, float* ac_frames_lags);
/* ---- beats and Laplacian segmentation (selfsupervised/mir.py:31-41; features/rosa/segment.py) ----
 * mir.py:31 rosa.beat.beat_track(onset_envelope, trim=False, hop_length=1024, bpm=tempo): librosa is un-vendored, this is
 * its published dynamic program (librosa.beat.__beat_local_score + __beat_track_dp).  onset_norm [T] = envelope / its
 * std(ddof=1); period = round(60 * frame_rate / bpm) frames.  Outputs (device): localscore [T] f64 = the envelope
 * smoothed by exp(-0.5 (32 m / period)^2), cumscore [T] f64, backlink [T] i32 (best predecessor frame, negative = none).
 * The caller walks the links back from the last beat (host; maua_amd.segment.beats_from_links). */
int maua_beat_dp(maua_ctx* ctx, const float* onset_norm, int T, int period, double tightness, double* localscore,
                 double* cumscore, int* backlink);
/* segment.py:152-155 beat-synchronous feature: out[s][c] = lower median (mode 0, torch.median) or mean (mode 1,
 * librosa.util.sync of :241) of x[bounds[s] .. bounds[s+1])[c]; x [T][C] f32, bounds [n_segments + 1] i32 on the device */
int maua_segment_reduce(maua_ctx* ctx, const float* x, int T, int C, const int* bounds, int n_segments, int mode,
                        float* out);
/* segment.py:23-57 recurrence_matrix(data, k, width, sym=True): rec [n][n] = exp(-d / bandwidth) on the k nearest rows of
 * every column (|i - j| < width excluded), symmetrised by the minimum, bandwidth = lower median of the row maxima */
int maua_recurrence_affinity(maua_ctx* ctx, const float* data, int n, int d, int k, int width, float* rec);
/* segment.py:74-82 timelag_median_filter: median of 7 along the diagonals (time-lag domain), out != rec */
int maua_timelag_median(maua_ctx* ctx, const float* rec, int n, float* out);
/* segment.py:60-64 median_filter1d as :193 applies it: window k (odd, <= 15) along the rows of x [n][m], reflect padding */
int maua_median_filter_rows(maua_ctx* ctx, const float* x, int n, int m, int k, float* out);
/* segment.py:106-131 differentiable_k_means on unit-norm rows data [n][k] from centres mu0 [k][k] (k <= 16): `iters`
 * updates mu = (r^T data) / sum r, r = softmax(temp * data mu^T); r_out [n][k], mu_out [k][k] */
int maua_soft_kmeans(maua_ctx* ctx, const float* data, int n, int k, const float* mu0, int iters, float temp, float* r_out,
                     float* mu_out);
/* ---- constant-Q features (rosa/constantq.py:13-115, rosa/spectral.py:164-325, rosa/pitch.py) ----
 * out[i] = scale * sum_k taps[k] x[i * stride + k - left] (zero outside the signal): one phase of torchaudio's sinc
 * resampler - what constantq.py:92 `resample(y, sr, sr / 2, "kaiser_window")` computes (torchaudio un-vendored:
 * published kernel, built by the caller). */
int maua_fir_decimate(maua_ctx* ctx, const float* x, long n, const float* taps, int ntaps, int stride, int left,
                      float scale, float* out, long n_out);
/* replaces audioreactive/audio.py:96-112 low_pass / high_pass / band_pass (scipy.signal.sosfilt over the decoded clip, float64)
 * and the lfilter under the biquads of selfsupervised/features/processing.py:142-151.  sos_host [n_sections][6] = b0 b1 b2 a0 a1 a2
 * per section (HOST array, scipy's layout), x / y [n] device float64 (y may be x); zero initial state.  Every section is scipy's
 * direct form II transposed sample for sample; the clip is cut into 128-sample chunks whose start states come from a scan. */
int maua_sosfilt(maua_ctx* ctx, const double* sos_host, int n_sections, const double* x, long n, double* y);
/* replaces processing.py:154-155 contrast_enhance (torchaudio.functional.contrast, un-vendored: the published formula)
 * y = sin(t + enhancement_amount / 750 * sin(4 t)), t = x pi / 2;  enhancement_amount in [0, 100]. */
int maua_contrast(maua_ctx* ctx, const float* x, long n, float enhancement_amount, float* y);
/* torch.istft's overlap-add (rosa/spectral.py:24-32) for already-windowed time frames [n_frames][W]: y[t] =
 * sum_f frames[f][t + start - f hop] / sum_f window[t + start - f hop]^2, t < length.  Used by the tempogram of clips
 * shorter than win_length (rosa/beat.py:48-49, 66: a non-power-of-two transform, run as a DFT GEMM). */
int maua_overlap_add(maua_ctx* ctx, const float* frames, int n_frames, int W, int hop, const float* window_dev, long start,
                     long length, float* y);
/* spectral.py:193-232 spline_eval (+ step_function when apply_step): out = step(a + f (b + f (c + f d))) with the
 * interval picked like torch.bucketize; knots [n_knots], coef [4][n_knots - 1] (a, b, c, d rows), all device f32. */
int maua_spline_step(maua_ctx* ctx, const float* x, long n, const float* knots, const float* coef, int n_knots, float h,
                     float alpha, int apply_step, float* out);
/* pitch.py:27-87 piptrack on the frame-major magnitude [n_frames][n_bins]; frame_max [n_frames] = max over bins,
 * freqs [n_bins] = the bin frequencies the band test compares (the caller's linspace), bin_hz = sr / n_fft;
 * pitch / mag [n_frames][n_bins] receive the interpolated frequency (Hz) and magnitude at accepted peaks, 0 elsewhere. */
int maua_piptrack(maua_ctx* ctx, const float* mag_frames_bins, int n_frames, int n_bins, const float* frame_max,
                  float threshold, const float* freqs, float bin_hz, float fmin, float fmax, float* pitch, float* mag);
/* replaces processing.py:53-56 normalize (eps = 1e-8) and signal.py:27-38 normalize (eps = 0):
 * y = (x - min) / ((max - min) + eps) over all n elements. */
int maua_normalize(maua_ctx* ctx, const float* x, long n, float eps, float* y);
/* replaces features/audio.py:31-37 rms: out[f] = sqrt(mean(frame_f^2)), centred reflect-padded frames. */
int maua_rms(maua_ctx* ctx, const float* y, int n_samples, int frame_length, int hop, int n_frames, float* out);
/* replaces signal.py:108-157 / processing.py:11-49 gaussian_filter: depthwise correlation along axis 0 of
 * x [T][C] with taps [2*radius+1] (built by the caller exactly as the reference builds its kernel), padding
 * min(radius,T) samples with `mode` (maua_pad_mode) and the rest by replication.  x and y must not alias. */
int maua_gaussian_filter1d(maua_ctx* ctx, const float* x, const float* taps, int radius, int T, long C, int mode,
                           float* y);
/* exact order statistics of the non-NaN (and, if mask != NULL, mask[i] != 0) elements of x[n]:
 *  mode 0: midpoint quantile with float32 q — replaces efficient_quantile.cpp:86-206 as __init__.py:6-7 calls it;
 *  mode 1: k-th smallest value, k 1-based — replaces signal.py:41-52 percentile's kthvalue;
 *  mode 2: torch.quantile(x, q) (linear interpolation, float32 rank arithmetic) — latent.py:37.
 * out3 (device) = {result, x_(lo), x_(hi)}; ranks2 (device, may be NULL) = {lo, hi} 0-based, -1 when empty. */
int maua_order_stat(maua_ctx* ctx, const float* x, const uint8_t* mask, long n, int mode, float q, long k,
                    float* out3, long long* ranks2);
/* ---- further audio features (selfsupervised/features/audio.py) ----
 * c [M][N] = a [M][K] x b [N][K]^T (f32).  Replaces rosa/spectral.py:35-56 dct (as a cosine-basis product, used by
 * audio.py:65-70 mfcc) and the phi @ chroma product of audio.py:50-62 tonnetz. */
int maua_matmul_nt(maua_ctx* ctx, const float* a, const float* b, float* c, int M, int N, int K);
/* mapping network (inference/stylegan2.py:116-192) around its matmul_nt + bias_act layers: the prologue normalize_2nd_moment
 * (ops.py:142-143: y = x / sqrt(mean(x^2, dim 1) + eps), x [P][D]) and the epilogue w.unsqueeze(1).repeat(1, n, 1)
 * (:183; out [P][n][D]) - so that the once-per-clip mapper needs no kernel outside this library. */
int maua_normalize_2nd_moment(maua_ctx* ctx, const float* x, int P, int D, float eps, float* y);
int maua_repeat_rows(maua_ctx* ctx, const float* x, int P, int D, int n, float* out);
/* replaces audio.py:118-126 spectral_flatness on the frame-major complex STFT spec [n_frames][n_bins]:
 * out[f] = exp(mean(log(max(amin, |z|^power)))) / mean(max(amin, |z|^power)). */
int maua_spectral_flatness(maua_ctx* ctx, const float* spec, int n_frames, int n_bins, float amin, float power,
                           float* out);
/* the per-band part of audio.py:76-115 spectral_contrast: for every frame sort |spec[f][lo..hi)| ascending
 * (torch.sort :107) and return the mean of the first k (valley, :109) and of the last k (peak, :110); hi - lo <= 1024.
 * The band edges / k follow the reference's float32 linspace comparisons and are computed by the caller. */
int maua_band_sorted_means(maua_ctx* ctx, const float* spec, int n_frames, int n_bins, int lo, int hi, int k,
                           float* valley, float* peak);
/* out2 (device) = {min(x), max(x)} over n elements (processing.py:134-136). */
int maua_minmax(maua_ctx* ctx, const float* x, long n, float* out2);
/* replaces processing.py:133-139 emphasize on xn = (x - min) / max(x - min) (maua_normalize with eps 0):
 * y = xn * (1 + tanh(strength * (xn - q))) * (max - min) + min; minmax_dev = {min, max} of x, q_dev = the quantile
 * of xn (maua_order_stat mode 2), both device scalars. */
int maua_emphasize(maua_ctx* ctx, const float* xn, long n, const float* minmax_dev, const float* q_dev, float strength,
                   float* y);
/* signal.py:84-105 compress / expand before their normalise: y = x * ratio where x > threshold (invert: x < threshold) */
int maua_threshold_scale(maua_ctx* ctx, const float* x, long n, float threshold, float ratio, int invert, float* y);
/* latent.py:46-51 on same-shape operands: mode 0 eerp a^(1-t) * b^t, mode 1 copeerp a^t (1 - b^t) / (1 - a^t + b^t) */
int maua_eerp(maua_ctx* ctx, const float* a, const float* b, const float* t, long n, int mode, float* y);
/* signal.py:69-76: mask[i] = x[i] > x[i+1] && x[i] > x[i-1] with neighbours clamped to the ends. */
int maua_peak_mask(maua_ctx* ctx, const float* x, int n, uint8_t* mask);
/* y = min(max(x, lo), hi + hi_add); lo = lo_dev[0] if lo_dev else lo_const; hi = hi_dev[0] (device scalars,
 * e.g. from maua_order_stat).  processing.py:59-62 standardize, signal.py:78 percentile_clip. */
int maua_clamp(maua_ctx* ctx, const float* x, const float* lo_dev, const float* hi_dev, float lo_const, float hi_add,
               long n, float* y);

/* ---- latent schedules (once per clip); tensors are [T][C] f32 with C = num_ws * w_dim ------------------- */
/* replaces latent.py:83-92 spline_loops / selfsupervised/latent.py:7-13 spline_loop_latents (the spline itself:
 * un-vendored torchcubicspline = natural cubic spline).  t_knots_host [n] and t_eval_host [T] are HOST f64 arrays
 * (strictly increasing knots); y [n][C] and out [T][C] are device f32.  Synchronises once (uploads the grids). */
int maua_spline_natural(maua_ctx* ctx, const double* t_knots_host, int n, const float* y, long C,
                        const double* t_eval_host, int T, float* out);
/* replaces latent.py:12-18 single_weighted: out[t] = a[t]*(1-env[t]) + b[t]*env[t]; a/b row strides in elements
 * (0 broadcasts one [C] row over time). */
int maua_latent_blend(maua_ctx* ctx, const float* a, long a_tstride, const float* b, long b_tstride, const float* env,
                      int T, long C, float* out);
/* replaces latent.py:21-31 multi_weighted: out[t] = sum_a (env[t][a]/sum_a' env[t][a']) * latents[a % n]. */
int maua_weighted_sum(maua_ctx* ctx, const float* env, const float* latents, int T, int A, int n, long C, float* out);
/* latent.py:38-40: idx = int64(round_half_even(x * scale)) — the onset-bin assignment (bit-exact). */
int maua_scale_round_index(maua_ctx* ctx, const float* x, float scale, long n, long long* idx);
/* latent.py:41: out[t] = src[idx[t]] (rows of C floats). */
int maua_gather_rows(maua_ctx* ctx, const float* src, const long long* idx, int n_rows, int T, long C, float* out);
/* replaces signal.py:5-24 resample / F.interpolate(mode="linear", align_corners=False) along axis 0. */
int maua_resample_linear(maua_ctx* ctx, const float* x, int n, long C, int size, float* out);
/* replaces latent.py:54-65 slerp: y [n_seg+1][L][D], t [k] -> out [k][n_seg][L][D] (unit-normalised, Q9). */
int maua_slerp(maua_ctx* ctx, const float* y, const float* t, int k, int n_seg, int L, int D, float* out);
/* replaces selfsupervised/latent.py:57-78: merge `sequence` into layers [l0,l1) of latents [T][L][D] in place;
 * mode 0 average, 1 modulate by mod[t], 2 overwrite. */
int maua_latent_merge(maua_ctx* ctx, float* latents, const float* sequence, const float* mod, int mode, int T, int L,
                      int D, int l0, int l1);

/* ---- per-batch noise (selfsupervised patch) ----------------------------------------------------- */
/* replaces selfsupervised/noise.py:42-53 Loop.forward(i, b): planes [3,h,w] f32 (the module's randn buffer),
 * idx [T] f32 (= linspace(0, 2*pi*n_loops, T)), frames i0..i0+B-1 ->
 * out[b] = sin(cos(idx[i0+b] + n0) / (sigma/50) + n1) * n2, divided by its per-frame RMS + eps(f32).  out [B,h,w]. */
int maua_noise_loop(maua_ctx* ctx, const float* planes, const float* idx, int i0, int B, int h, int w, float sigma,
                    float* out);
/* the same for n Loop modules (one per synthesis layer) in two launches: host arrays of n device pointers / sizes;
 * out[l] receives [B, h[l], w[l]].  What selfsupervised/sample.py:93-95 does per batch with 17 module calls. */
int maua_noise_loop_batch(maua_ctx* ctx, int n, const float* const* planes, const float* const* idx, const int* h,
                          const int* w, const float* sigma, int i0, int B, float* const* out);
/* replaces noise.py:11-24 Blend.forward (noise2 != NULL: sum_m noise[m]*mod[b,m] + sum_m noise2[m]*(1-mod[b,m]))
 * and :27-39 Multiply.forward (noise2 == NULL).  noise/noise2 [M,h,w], mod [B,M] (rows i..i+B of the modulator). */
/* The same maps UN-NORMALISED in one pass + the per-(layer, sample) factor 1 / (rms + eps) that noise.py:52 divides by:
 * scales [n][B] f32 (device).  Hand the maps to the synthesis call as usual and the factors through maua_synth_set_noise_scale:
 * the consuming convolution epilogues multiply them into their noise strength, so the maps are written once and sin(cos(.)) is
 * evaluated once per value (the normalised form needs two passes).  Same reference lines as maua_noise_loop_batch. */
int maua_noise_loop_batch_raw(maua_ctx* ctx, int n, const float* const* planes, const float* const* idx, const int* h,
                              const int* w, const float* sigma, int i0, int B, float* const* out, float* scales);

int maua_noise_mix(maua_ctx* ctx, const float* noise, const float* noise2, const float* mod, int M, int B, int h,
                   int w, float* out);
/* replaces noise.py:56-63 Average (mode 0: (x+y)/2), :66-75 Modulate (mode 1: x*mod[b] + y*(1-mod[b]), mod [B]),
 * :78-86 ScaleBias (mode 2: scale*x + bias).  x, y, out [B,h,w]. */
int maua_noise_combine(maua_ctx* ctx, const float* x, const float* y, const float* mod, int mode, float scale,
                       float bias, int B, int h, int w, float* out);

/* ---- N4 (first slice): RealESRGAN x4 generator, the per-frame up-scaler of "StyleGAN2 render -> RealESRGAN 4x" ----
 * replaces maua/super/image/models/realesrgan.py:22-49 (basicsr RRDBNet(3, 3, num_feat, num_block, num_grow_ch, scale 4)
 * run by RealESRGANer.enhance: [0,1] image -> network -> clamp(0,1) -> round(x * 255)) and the per-frame loop of
 * maua/super/video/frame_by_frame.py:22-33.  basicsr / realesrgan are un-vendored: published architecture. */
int maua_rrdb_create(maua_ctx* ctx, int num_feat, int num_block, int num_grow_ch, int dtype, maua_rrdbnet** out);
void maua_rrdb_destroy(maua_rrdbnet* net);
/* name: a key of basicsr's RRDBNet state dict (conv_first.weight, body.3.rdb2.conv4.bias, conv_last.weight, ...);
 * weights are [Co][Ci][3][3], biases [Co], f32 on the host. */
int maua_rrdb_load(maua_rrdbnet* net, const char* name, const float* host_data, size_t count);
/* img: device f32 [B][3][H][W] in [0,1].  out_nchw: device f32 [B][3][4H][4W] clamped to [0,1] (or NULL);
 * out_rgb8: device u8 [B][4H][4W][3] = round(clamp * 255) (or NULL). */
int maua_rrdb_forward(maua_rrdbnet* net, const float* img_nchw, int B, int H, int W, float* out_nchw, uint8_t* out_rgb8);
/* the RRDB forward without the [0,1] clamp on out_nchw (RealESRGANer clamps AFTER its tile stitching / pre-pad crop; the
 * u8 output is always clamped).  clamp01 = 1 is maua_rrdb_forward. */
/* RealESRGANer.enhance (realesrgan utils; as super/video/frame_by_frame.py:22-33 applies it per frame) for a batch of device-resident
 * u8 frames [B][h][w][3] in one call: x / 255, reflect pre_pad on the right / bottom, the network, clamp, round(255 y), crop ->
 * out_rgb8 [B][4h][4w][3].  No separate convert / pad / crop passes (round 6). */
int maua_rrdb_enhance_u8(maua_rrdbnet* net, const uint8_t* frames, int B, int h, int w, int pre_pad, uint8_t* out_rgb8);
int maua_rrdb_forward_ex(maua_rrdbnet* net, const float* img_nchw, int B, int H, int W, int clamp01, float* out_nchw,
                         uint8_t* out_rgb8);

/* ---- N4: the "xsx4-animevideo" model of realesrgan.py:34-35: realesrgan's SRVGGNetCompact(3, 3, num_feat, num_conv,
 * upscale, act_type) - convolutions + PReLU, PixelShuffle, + nearest-upsampled input (un-vendored: published architecture).
 * act_type: 0 prelu, 1 relu, 2 leakyrelu(0.1).  Parameter names: body.<2k>.weight / .bias, body.<2k+1>.weight (PReLU). */
int maua_srvgg_create(maua_ctx* ctx, int num_feat, int num_conv, int upscale, int act_type, int dtype, maua_srvgg** out);
void maua_srvgg_destroy(maua_srvgg* net);
int maua_srvgg_load(maua_srvgg* net, const char* name, const float* host_data, size_t count);
/* img device f32 [B][3][H][W]; out_nchw device f32 [B][3][sH][sW] (clamped to [0,1] when clamp01) or NULL; out_rgb8 device u8
 * [B][sH][sW][3] = round(clamp * 255) or NULL */
int maua_srvgg_forward(maua_srvgg* net, const float* img_nchw, int B, int H, int W, int clamp01, float* out_nchw,
                       uint8_t* out_rgb8);

/* ---- N4 (second half): guided-diffusion UNet + DDIM, BASELINE configs[3] "guided-diffusion 256x256, 100-step DDIM" ------
 * replaces the network maua/diffusion/processors/guided.py:164-209 create_models builds (guided_diffusion.script_util.
 * create_model_and_diffusion: UNetModel with num_channels 256, num_res_blocks 2, attention_resolutions "32, 16, 8",
 * num_head_channels 64, learn_sigma, resblock_updown, use_scale_shift_norm) and, with maua_ddim_step, the
 * diffusion.ddim_sample(model, x, t, cond_fn=...) call that guided.py:277-339 GuidedDiffusion.forward loops over.  The
 * guided_diffusion submodule is EMPTY in the reference checkout: the published architecture / sampler are restated, parity
 * unpinned.  channel_mult: n_mult multipliers of model_channels per level; attention_ds: the down-sampling rates
 * (image_size // resolution) at which blocks carry attention; only the flag set of guided.py:171-190 is implemented
 * (use_scale_shift_norm, resblock_updown, legacy attention order, no class conditioning). */
int maua_unet_create(maua_ctx* ctx, int image_size, int in_channels, int model_channels, int out_channels, int num_res_blocks,
                     const float* channel_mult, int n_mult, const int* attention_ds, int n_attn, int num_head_channels,
                     int dtype, maua_unet** out);
void maua_unet_destroy(maua_unet* net);
/* number of named parameters the network expects (every key of UNetModel.state_dict() + "timestep_embedding.freqs") */
int maua_unet_param_count(maua_unet* net, long* count);
/* name: a key of guided_diffusion's UNetModel.state_dict() (input_blocks.4.0.in_layers.2.weight, middle_block.1.qkv.weight,
 * out.2.bias, time_embed.0.weight, ...), f32 on the host: 3x3 weights [Co][Ci][3][3], 1x1 weights [N][K][1(,1)], linear
 * weights [N][K], GroupNorm scale / shift [C].  Optional "timestep_embedding.freqs" [model_channels / 2]: the float32
 * frequency table of nn.py timestep_embedding as the host computes it (otherwise computed on the device). */
int maua_unet_load(maua_unet* net, const char* name, const float* host_data, size_t count);
/* "route": 0 per-shape routing of the 3x3 convolutions (default), 1 generic kernel only, 2 no split-K gather GEMM;
 * "psum_off": 1 = GroupNorm statistics always by their own pass over the tensor (default 0: where the LDS-direct convolution
 * produced the tensor, from the piece sums its epilogue left) */
int maua_unet_set_option(maua_unet* net, const char* key, int value);
/* UNetModel.forward(x, timesteps): x device f32 [B][in_channels][H][W], timesteps device f32 [B] (the value the wrapped
 * model of respace.py passes: original timestep index, rescaled to 0..1000), out device f32 [B][out_channels][H][W].
 * H, W: multiples of 2^(levels - 1). */
int maua_unet_forward(maua_unet* net, const float* x, const float* timesteps, int B, int H, int W, float* out);
/* Guidance speed "regular" (guided.py:214-218, 250-272): the loss gradient is taken THROUGH the diffusion UNet -
 * `torch.autograd.grad(img, x, img_grad)` with img a function of p_mean_variance(model, x, t)["pred_xstart"].  The library's
 * form of that autograd call: maua_unet_forward_keep = maua_unet_forward that leaves what the input gradient reads on the
 * network's arena (GroupNorm inputs and statistics, qkv, the attention rows' log-sum-exp), then
 * maua_unet_vjp: g_x [B][in_channels][H][W] = (d out / d x)^T g_out, g_out [B][out_channels][H][W] (device f32).
 * Back to back on one network, same shape; option "vjp" = 1 (maua_unet_set_option) BEFORE maua_unet_load, which then also
 * prepares every weight's transposed copy (3x3: transposed + spatially flipped - the gradient is the same MFMA convolution). */
int maua_unet_forward_keep(maua_unet* net, const float* x, const float* timesteps, int B, int H, int W, float* out);
int maua_unet_vjp(maua_unet* net, const float* g_out, int B, int H, int W, float* g_x);
/* gaussian_diffusion.py ddim_sample for an epsilon model, clip_denoised False (guided.py:303-306): x [B][C][HW], model_out
 * [B][Cm][HW] (first C channels = eps), cond_grad = cond_fn(x, t) [B][C][HW] or NULL (condition_score), noise or NULL
 * (eta 0), coef device f32 [B][8] = {sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, sqrt(1 - alphas_cumprod),
 * sqrt(alphas_cumprod_prev), sqrt(1 - alphas_cumprod_prev - sigma^2), sigma * (t != 0), 0, 0} at each sample's t.
 * sample (may alias x) and pred_xstart (or NULL): [B][C][HW]. */
int maua_ddim_step(maua_ctx* ctx, const float* x, const float* model_out, const float* cond_grad, const float* noise,
                   const float* coef, int B, int C, int Cm, long HW, float* sample, float* pred_xstart);
/* gaussian_diffusion.py p_sample (guided.py:302-303, sampler "p"): ancestral step of an epsilon model with learned-range
 * variance (model_out [B][2 C][H][W]), clip_denoised False; cond_grad (or NULL): condition_mean.  noise [B][C][H][W].
 * coef: device f32 [B][8] = {sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1,
 * posterior_mean_coef2, posterior_log_variance_clipped, log(betas), t != 0, 0}. */
int maua_p_sample_step(maua_ctx* ctx, const float* x, const float* model_out, const float* cond_grad, const float* noise,
                       const float* coef, int B, int C, long HW, float* sample, float* pred_xstart);
/* plms_sample (guided.py:308-311, sampler "plms"; the sampler of the guided-diffusion fork the reference vendors as an
 * empty submodule: published algorithm - pseudo linear multistep, Liu et al. 2022): one model evaluation -> eps (after the
 * optional condition_score), its pred_xstart, the unconditioned pred_xstart; coef = maua_ddim_step's table. */
int maua_plms_eps(maua_ctx* ctx, const float* x, const float* model_out, const float* cond_grad, const float* coef, int B,
                  int C, int Cm, long HW, float* eps, float* pred_xstart, float* pred_xstart_orig);
/* ... and the update: eps' = (sum_i weights[i] eps_list[i]) / divisor (n_eps <= 4; host arrays of device pointers / floats), pred' from
 * eps', sample = (pred' sqrt(ac_prev) + sqrt(1 - ac_prev) eps') (t != 0) + pred_xstart (t == 0).  coef: device f32 [B][8] =
 * {sqrt_recip_ac, sqrt_recipm1_ac, sqrt(ac_prev), sqrt(1 - ac_prev), t != 0, 0, 0, 0}. */
int maua_plms_update(maua_ctx* ctx, const float* x, const float* const* eps_list, const float* weights, int n_eps,
                     float divisor, const float* pred_xstart, const float* coef, int B, long chw, float* sample);
/* out[b] = ab[b][0] * x[b] + ab[b][1] * y[b] over rows of `row` floats: q_sample (guided.py:331, gaussian_diffusion.py) */
int maua_axpby_rows(maua_ctx* ctx, const float* x, const float* y, const float* ab, int B, long row, float* out);
/* 1 when the last maua_ddim_sample_loop(use_graph = 1) replayed a captured hipGraph, 0 when it ran launch by launch */
int maua_unet_graph_active(maua_unet* net, int* active);

/* ---- secondary diffusion model: the network behind the reference's DEFAULT guidance speed ("fast")
 * Replaces maua/diffusion/processors/guided.py:68-143 (SecondaryDiffusionImageNet2.forward -> DiffusionOutput(v, pred, eps)) and the
 * torch.autograd.grad(img, x, img_grad) of GradientGuidedConditioning.forward (:236-272), which differentiates through it: the library
 * has no autograd, so the vector-Jacobian product is evaluated on the transposed network (csrc/secondary.hip).
 * Convolutions are numbered 0..23 in execution order (maua_amd/diffusion.py maps the reference's state-dict keys).
 * forward: x [B][3][H][W] f32, t [B] f32 (the cosine-schedule time in [0, 1]); H, W multiples of 32; any of v / pred / eps may be NULL.
 * vjp: g_v = d loss / d v  ->  g_x = (d v / d x)^T g_v, using the activations of the LAST forward (same B, H, W).  Device pointers. */
typedef struct maua_secondary maua_secondary;
int maua_secondary_create(maua_ctx* ctx, int dtype, maua_secondary** out);
void maua_secondary_destroy(maua_secondary* net);
int maua_secondary_conv_shape(int index, int* ci, int* co);
/* what: 0 = weight [Co][Ci][3][3] (host), 1 = bias [Co], 2 = timestep_embed.weight [8] (index ignored) */
int maua_secondary_load(maua_secondary* net, int index, int what, const float* host_data, size_t count);
int maua_secondary_forward(maua_secondary* net, const float* x, const float* t, int B, int H, int W, float* v_out, float* pred_out,
                           float* eps_out);
int maua_secondary_vjp(maua_secondary* net, const float* g_v, int B, int H, int W, float* g_x);
/* operator-level forms of the UNet's building blocks, NHWC tensors in `dtype` (guided_diffusion/unet.py, nn.py):
 * QKVAttentionLegacy.forward - qkv [B][T][3 * heads * head_ch] with channel = head * 3 ch + {q | k | v} * ch + c (what
 * `qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)` sees) -> out [B][T][heads * head_ch]; head_ch 32 or 64 */
int maua_attention_legacy(maua_ctx* ctx, const void* qkv, void* out, int B, int T, int heads, int head_ch, int dtype);
/* its input gradient (autograd through QKVAttentionLegacy.forward): d_out [B][T][heads * head_ch] -> d_qkv like qkv */
int maua_attention_legacy_vjp(maua_ctx* ctx, const void* qkv, const void* d_out, void* d_qkv, int B, int T, int heads, int head_ch,
                              int dtype);
/* conv_nd(1, K, N, 1) / a linear layer over rows (AttentionBlock.qkv / proj_out + residual, ResBlock.skip_connection):
 * c[M][N] = a[M][K] x w[N][K]^T + bias[N] (+ res[M][N]); K % 32 (bf16) / 16 (f32) == 0, N % 32 == 0 */
int maua_linear_nt(maua_ctx* ctx, const void* a, const void* w, const float* bias, const void* res, void* c, long M, int N,
                   int K, int dtype);
/* GroupNorm32(32, C)(x) [* (1 + scale) + shift with scale_shift [B][2C] (ResBlock use_scale_shift_norm)] [-> SiLU];
 * statistics in float64, eps 1e-5.  x, y [B][H][W][C] */
int maua_group_norm_nhwc(maua_ctx* ctx, const void* x, const float* gamma, const float* beta, const float* scale_shift,
                         int silu, int B, int H, int W, int C, int dtype, void* y);
/* its input gradient, with the ResBlocks' resampling behind the activation (resample 0 none, 1 avg_pool2d 2, 2 nearest x2:
 * h_upd(in_rest(x)), unet.py ResBlock._forward): dy [B][Ho][Wo][C] -> dx [B][H][W][C]; dres (optional, shaped like dy): the
 * gradient of x_upd(x), the same resampling of the raw input, added */
int maua_group_norm_nhwc_vjp(maua_ctx* ctx, const void* x, const float* gamma, const float* beta, const float* scale_shift, int silu,
                             int resample, const void* dy, const void* dres, int B, int H, int W, int C, int dtype, void* dx);
/* the unconditioned sampling loop of guided.py:333-337 inside the library: n_steps x (forward + DDIM update) on x in place;
 * model_t host f32 [n_steps], coef host f32 [n_steps][8]; use_graph: capture the loop in one hipGraph and replay it. */
int maua_ddim_sample_loop(maua_unet* net, float* x, int B, int H, int W, const float* model_t, const float* coef, int n_steps,
                          int use_graph, float* pred_xstart);
/* the GUIDED sampling loop - configs[3] as BASELINE states it - inside the library, one hipGraph per shape.  Replaces the per-step
 * Python of maua/diffusion/processors/guided.py:302-311, 333-337 (diffusion.ddim_sample(..., cond_fn=self.conditioning)) with
 * cond_fn = GradientGuidedConditioning.forward at the reference's default speed "fast" (:236-272: secondary model forward, img =
 * pred * sigma + x * (1 - sigma), the grad modules' d loss / d img, torch.autograd.grad back through the secondary model) and an
 * image-MSE grad module (d loss / d img = (img - target) * mse_k; a result holding a NaN counts as zeros, :262-265).  Per step:
 * out = unet(x, t); pred = secondary(x, cos_t).pred; img; g; grad = c0 g + c1 (dv/dx)^T g; x, pred_xstart = ddim_step(x, out, grad).
 * model_t / coef as maua_ddim_sample_loop; guide: host f32 [n_steps][5] = {cos_t, sigma, 1 - sigma, -(sigma a_c + 1 - sigma),
 * sigma s_c} (the host evaluates them like :249-252, :266-268 do); target: device f32 [B][3][H][W] (target_bstride = 3 H W) or one
 * image for every sample (target_bstride = 0).  Both networks must have been created on the same context.
 * sec == NULL: speed "regular" (guided.py:214-218, 250-252) - img is built from THIS network's pred_xstart and the gradient goes back
 * through it: per step out = forward_keep(x, t); pred = ra x - rm eps (eps = the first in_channels of out); img; g; grad = c0 g +
 * c1 (d eps / d x)^T g (maua_unet_vjp); DDIM update.  guide[s] = {-, sigma, 1 - sigma, -(sigma ra + 1 - sigma), sigma rm}; option
 * "vjp" = 1 before the weights are loaded.  One hipGraph as well. */
int maua_ddim_guided_loop(maua_unet* net, maua_secondary* sec, float* x, int B, int H, int W, const float* model_t, const float* coef,
                          const float* guide, int n_steps, const float* target, long target_bstride, float mse_k, int use_graph,
                          float* pred_xstart);
/* 1 when the last maua_ddim_guided_loop(use_graph = 1) replayed a captured hipGraph, 0 when it ran launch by launch */
int maua_unet_guided_graph_active(maua_unet* net, int* active);
/* that grad module as an operator: out = (img - target) * k over B rows of `row` floats, zeros if the result holds a NaN
 * (guided.py:256-265 with an image-MSE loss); target_bstride = row or 0 */
int maua_mse_guide_grad(maua_ctx* ctx, const float* img, const float* target, long target_bstride, float k, int B, long row,
                        float* out);

/* ---- text-prompt guidance: CLIPGrads (configs[3]: "audio-onset-switched text prompts") ------------------------------------
 * Replaces maua/grad.py:96-165: per sampler step, `cutout_batches` x { MauaCutouts of the image estimate (maua/ops/cutouts.py:8-50),
 * Normalize, clip_model.encode_image (OpenAI CLIP VisionTransformer: un-vendored pip dependency, published architecture restated,
 * parity unpinned), spherical_dist_loss to the target embeddings (maua/loss.py:22-25), weights, mean over cutouts,
 * torch.autograd.grad(loss.sum() * scale, img) / cutout_batches }, clamp_gradient.  The gradient is evaluated on the transposed
 * network inside the library (csrc/clip.hip, cutouts.hip, gemm_dma.hip).  The text tower is NOT here: target embeddings are handed
 * in (TextPrompt embeddings computed once, off the loop - set_targets :117-143); image (style) targets can be embedded with
 * maua_clip_encode_image.  dtype: MAUA_BF16 (the reference runs the perceptor in fp16) or MAUA_F32 (exact products, parity mode).
 * Parameter names: CLIP's state-dict keys below "visual." (conv1.weight, class_embedding, positional_embedding, ln_pre.*,
 * transformer.resblocks.<i>.{attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.*, ln_1.*, mlp.c_fc.*, mlp.c_proj.*, ln_2.*},
 * ln_post.*, proj), host float32 in the checkpoint's layout. */
typedef struct maua_clip maua_clip;
int maua_clip_create(maua_ctx* ctx, int input_resolution, int patch_size, int width, int layers, int heads, int output_dim, int dtype,
                     maua_clip** out);
void maua_clip_destroy(maua_clip* net);
int maua_clip_load(maua_clip* net, const char* name, const float* host_data, size_t count);
/* VisionTransformer.forward (clip/model.py): images device f32 [N][3][R][R], already normalised -> embeds device f32 [N][E];
 * keep != 0 keeps the activations for maua_clip_encode_image_vjp: d_embeds [N][E] -> d_images [N][3][R][R] */
int maua_clip_encode_image(maua_clip* net, const float* images, int N, int keep, float* embeds);
int maua_clip_encode_image_vjp(maua_clip* net, const float* d_embeds, int N, float* d_images);
/* the buffers CLIPGrads.set_targets registers (:134-143): S sets of P target embeddings [S][P][E] (host or device; normalised here
 * as spherical_dist_loss does) with prompt weights [S][P] (already divided by |sum|, :139-143); sel: host [B] - the set each sample
 * of a batch is guided towards (prompts switched per frame by the clip's onsets) - or NULL: set 0 for every sample */
int maua_clip_set_targets(maua_clip* net, const float* targets, const float* weights, int S, int P, const int* sel, int B);
/* CLIPGrads.forward (:145-159): img device f32 [B][3][H][W] in [-1, 1]; rects HOST int [batches][cutn][3] = (size, top, left) of
 * every cutout of every cutout batch (the host draws them like MauaCutouts.forward does); grad device f32 [B][3][H][W];
 * clamp_gradient <= 0: none; a result holding a NaN is returned as zeros (guided.py:262-265).
 * mult: NULL, or HOST float [batches][cutn] - cutout i stands for mult[i] IDENTICAL cutouts of the reference's list (its first
 * cutn // 4 cutouts of a square image are the same rectangle, cutouts.py:16-27): one pass through the tower carries their weight; the
 * mean over cutouts then divides by the sum of a batch's multiplicities.  Same gradient, fewer images. */
int maua_clip_guide_grad(maua_clip* net, const float* img, int B, int H, int W, const int* rects, const float* mult, int cutn, int batches,
                         float scale, float clamp_gradient, float* grad);
/* sum_p w_p dist_p of the first `count` cutout images of the last pass through the tower (cutout-major; device f32 [count]) */
int maua_clip_last_image_losses(maua_clip* net, int count, float* out);
/* the guided loop with THIS grad module: the following maua_ddim_guided_loop calls on `net` (speed "fast": a secondary model, or
 * "regular") evaluate CLIPGrads on the image estimate instead of the image-MSE module (their target / mse_k arguments are ignored);
 * rects: HOST int [n_steps][batches][cutn][3], one draw per step and cutout batch (mult: NULL or [n_steps][batches][cutn], as above),
 * read at every call of the loop.  clip == NULL:
 * back to the image-MSE module.  The whole step - UNet, secondary model, cutouts, image tower forward and backward, DDIM update -
 * stays one hipGraph. */
int maua_unet_set_clip_guide(maua_unet* net, maua_clip* clip, const int* rects, const float* mult, int n_steps, int cutn, int batches,
                             float scale, float clamp_gradient);
/* operator-level pieces.  maua_cutouts: random_cutouts (cutouts.py:8-38) on HOST rectangles [n_cut][3] applied to img * mul + add,
 * resized to cut_size^2 by the `resize_right` algorithm (cubic, antialiased, zero padding), Normalize(mean3, std3) -> out device f32
 * [n_cut * B][3][cut_size][cut_size] (cutout-major, like torch.cat); maua_cutouts_vjp: d_out -> d_img [B][3][H][W] */
int maua_cutouts(maua_ctx* ctx, const float* img, int B, int H, int W, const int* rects, int n_cut, int cut_size, float mul, float add,
                 const float* mean3, const float* std3, float* out);
int maua_cutouts_vjp(maua_ctx* ctx, const float* d_out, int B, int H, int W, const int* rects, int n_cut, int cut_size, float mul,
                     const float* std3, float* d_img);
/* nn.LayerNorm over the last dimension of x [rows][C] in dtype (float32 statistics, eps 1e-5; clip/model.py LayerNorm); stats:
 * optional device f32 [rows][2] = (mean, rstd), what maua_layer_norm_vjp needs: dx = d LayerNorm / d x applied to dy (+ add) */
int maua_layer_norm(maua_ctx* ctx, const void* x, const float* gamma, const float* beta, long rows, int C, int dtype, void* y, float* stats);
int maua_layer_norm_vjp(maua_ctx* ctx, const void* x, const float* stats, const float* gamma, const void* dy, const void* add, long rows,
                        int C, int dtype, void* dx);

/* ---- image-prompt grad modules: ColorMatchGrads, VGGGrads, LPIPSGrads (the other members of get_diffusion_model's list,
 * maua/diffusion/image.py:92-97) ------------------------------------------------------------------------------------------
 * ColorMatchGrads (maua/grad.py:27-70): hue histogram of clamp((img + 1) / 2) under kornia's rgb_to_hsv (restated; radians clamped
 * to [0, 1] as the reference does), weighted by sqrt(sat * val), normalised to sum 1 - differentiable_histogram's 255 masked passes
 * as one pass with 64-bit fixed-point sums (bit-identical from run to run).  img device f32 [B][3][H][W] in [-1, 1].
 *   maua_colormatch_hist : hist device f32 [B][nbins]                                   (histogram(), set_targets :65-69)
 *   maua_colormatch_grad : grad = d (scale * mse_loss(hist, target)) / d img in closed form (forward :67-70); target device f32
 *                          [nbins] (target_per_sample == 0) or [B][nbins]; loss: optional device f32 [B], the samples' shares of the
 *                          loss (their sum is the reference's scalar) */
int maua_colormatch_hist(maua_ctx* ctx, const float* img, int B, int H, int W, int nbins, int sat_weighting, float* hist);
int maua_colormatch_grad(maua_ctx* ctx, const float* img, int B, int H, int W, int nbins, int sat_weighting, const float* target,
                         int target_per_sample, float scale, float* grad, float* loss);
/* the conditioning's sum over its grad modules (guided.py:258-266 `if torch.isnan(sub).any(): sub = zeros; img_grad += sub`) without
 * the host round trip: acc = (first ? 0 : acc) + (sub holds a NaN ? 0 : sub) over n device floats */
int maua_grad_accumulate(maua_ctx* ctx, const float* sub, float* acc, long n, int first);
/* VGG perceptors (csrc/perceptor.hip).  plan: n_ops entries, > 0 = Conv2d(3x3, pad 1) to that many channels (multiples of 64) + ReLU,
 * 0 = MaxPool2d(2) - torchvision's vgg19 / vgg16 `features` cut after the last tap; replicate_first: the first convolution pads by
 * replication (vgg_kbc.py:40); the network sees ((img * in_mul + in_add) - mean) / std (VGGGrads: img.add(1).div(2) + ImageNet
 * Normalize, vgg_kbc.py:33; LPIPS: its ScalingLayer).  dtype MAUA_F32 (exact products, parity mode) or MAUA_BF16.  Weights: torch
 * layout [Co][Ci][3][3] / [Co] per convolution, in plan order (torchvision keys "<features index>.weight / .bias"); the published
 * networks are un-vendored (torchvision, lpips: parity unpinned), random-init in tests and bench. */
typedef struct maua_vgg maua_vgg;
int maua_vgg_create(maua_ctx* ctx, int dtype, const int* plan, int n_ops, int replicate_first, float in_mul, float in_add,
                    const float* mean3, const float* std3, maua_vgg** out);
void maua_vgg_destroy(maua_vgg* net);
int maua_vgg_conv_count(maua_vgg* net);
int maua_vgg_conv_shape(maua_vgg* net, int index, int* ci, int* co);
int maua_vgg_load(maua_vgg* net, int conv_index, int what /* 0 weight, 1 bias */, const float* host_data, size_t count);
/* forward of img device f32 [B][3][H][W]; every activation is kept for the calls below (plan entry index `op`):
 * maua_vgg_features -> planar f32 [B][C][h][w]; maua_vgg_gram -> Gram matrices [B][C][C] (loss.py:57-80 gram_matrix per image:
 * Perceptor.get_target_embeddings, perceptors/__init__.py:44-76); maua_vgg_lpips_features -> unit-normalised features [B][h w][C] */
int maua_vgg_forward(maua_vgg* net, const float* img, int B, int H, int W);
int maua_vgg_features(maua_vgg* net, int op, float* out);
int maua_vgg_gram(maua_vgg* net, int op, float* out);
int maua_vgg_lpips_features(maua_vgg* net, int op, float* out);
/* VGGGrads.forward (maua/grad.py:90-93): grad = d sum_taps strength * feature_loss(gram(tap), target) / d img, per image
 * (feature_loss = scaled_mse_loss / numel, loss.py:33-54); taps HOST int [n_taps] (plan entries), targets HOST array of n_taps DEVICE
 * pointers to f32 [C][C] (target_bstride NULL / 0) or [B][C][C] (target_bstride[k] = C * C); loss optional device f32 [B] */
int maua_vgg_style_grad(maua_vgg* net, const float* img, int B, int H, int W, const int* taps, int n_taps, const float* const* targets,
                        const long* target_bstride, float strength, float* grad, float* loss);
/* LPIPSGrads.forward (maua/grad.py:189-193) at the network's own size: grad = d (scale * sum_b lpips(img_b, target)) / d img;
 * targets: HOST array of DEVICE pointers to the target's unit-normalised tap features (maua_vgg_lpips_features: [hw][C], shared, or
 * [B][hw][C] with target_bstride[k] = hw * C), lins: DEVICE pointers to the lin layers' weights [C]; dist optional device f32 [B] */
int maua_vgg_lpips_grad(maua_vgg* net, const float* img, int B, int H, int W, const int* taps, int n_taps, const float* const* targets,
                        const long* target_bstride, const float* const* lins, float scale, float* grad, float* dist);

/* grad modules as library objects, so that the guided loop can evaluate a LIST of them per step (guided.py:258-266 sums the modules'
 * gradients; get_diffusion_model hands the sampler up to four, maua/diffusion/image.py:92-97).  kind: MAUA_GUIDE_STYLE = VGGGrads
 * (arguments of maua_vgg_style_grad: net, taps, targets = Gram matrices, target_bstride, scale = strength), MAUA_GUIDE_LPIPS =
 * LPIPSGrads (arguments of maua_vgg_lpips_grad, lins), MAUA_GUIDE_COLORMATCH = ColorMatchGrads (targets[0] = the target histogram
 * [nbins] - target_bstride[0] == nbins: one per sample -, nbins, sat_weighting, scale; net / taps / lins unused).  The target tensors
 * stay the caller's: device pointers that must outlive the guide; updating them IN PLACE re-targets the guide, also inside an already
 * captured loop.  maua_guide_grad: the module as an operator, grad = d loss / d img of img device f32 [B][3][H][W]. */
#define MAUA_GUIDE_STYLE 0
#define MAUA_GUIDE_LPIPS 1
#define MAUA_GUIDE_COLORMATCH 2
typedef struct maua_guide maua_guide;
int maua_guide_create(maua_ctx* ctx, int kind, maua_vgg* net, const int* taps, int n_taps, const float* const* targets,
                      const long* target_bstride, const float* const* lins, float scale, int nbins, int sat_weighting, maua_guide** out);
void maua_guide_destroy(maua_guide* guide);
int maua_guide_grad(maua_guide* guide, const float* img, int B, int H, int W, float* grad);
/* the guided loop with THESE grad modules: the following maua_ddim_guided_loop calls on `net` evaluate every guide of the list on the
 * step's image estimate and sum the results (a module whose gradient holds a NaN is skipped, guided.py:262-265) - after CLIPGrads when
 * maua_unet_set_clip_guide is active as well, instead of the image-MSE module otherwise (its target / mse_k arguments are then
 * ignored).  n_guides == 0: back to the default.  The whole step stays inside the one captured hipGraph. */
int maua_unet_set_guides(maua_unet* net, maua_guide* const* guides, int n_guides);

/* ---- build-owned counter RNG: Philox4x32-10, identical on every device / rank and in the oracle twin (oracle/rng.py,
 * pinned to the published known-answer vectors).  No reference counterpart: the reference's random-init generator and noise planes
 * come from torch's host generator (inference/stylegan2.py:216-227, selfsupervised/noise.py:42-53); the benchmark's synthetic
 * network and planes are drawn on the device instead (a clip's set-up then costs kernels, not 32 M host draws + their upload).
 * Element i of (seed, stream): counter {i / 4, stream}, key seed, word i % 4; offset = index of out[0] in the stream (any value:
 * a tensor may be filled in pieces).  normal: Box-Muller on word pairs, u = ((x >> 9) + 0.5) 2^-23; out = mean + stdev z. */
int maua_philox_u32(maua_ctx* ctx, unsigned long long seed, unsigned long long stream, unsigned long long offset, uint32_t* out, long n);
int maua_philox_normal(maua_ctx* ctx, unsigned long long seed, unsigned long long stream, unsigned long long offset, float* out, long n,
                       float mean, float stdev);
/* The benchmark clip's waveform on the device (0.3 sin(2 pi 220 t) + 0.2 (u - 0.5) [(2t mod 1) < 0.05] + 0.01 n; u / n =
 * streams 0 / 1 of `seed`): out device f32 [n].  No reference counterpart (synthetic input); host twin oracle/rng.py clip_audio. */
int maua_philox_clip_audio(maua_ctx* ctx, unsigned long long seed, long n, double sample_rate, float* out);

/* ---- multi-GPU: the one exchange step of the frame-sharded render ------------------------------------
 * One process per GPU; frames are sharded by contiguous range (no data-path collective).  maua_gather_frames moves every
 * rank's packed u8 shard to `root`, ordered by rank, as grouped RCCL point-to-point transfers on the context's stream
 * (replaces the single-process frame list of maua/audiovisual/generate.py:57-98; process layout as
 * maua/super/image/bulk.py:31-109).  Bootstrap: rank 0 calls maua_comm_unique_id and hands the 128 bytes to the other
 * ranks out of band (torch.distributed broadcast, a file, MPI ...); every rank then calls maua_comm_init.  RCCL is bound
 * at first use (dlopen): single-GPU hosts never load it. */
typedef struct { char internal[128]; } maua_comm_id;   /* = ncclUniqueId */
typedef struct maua_comm maua_comm;
int maua_comm_unique_id(maua_comm_id* id);
int maua_comm_init(maua_ctx* ctx, const maua_comm_id* id, int rank, int world, maua_comm** out);
/* bytes_per_rank [world]: shard sizes (host array); send: this rank's shard (device); recv: root only, sum of the sizes. */
int maua_gather_frames(maua_comm* comm, const uint8_t* send, const long* bytes_per_rank, uint8_t* recv, int root);
/* one round of the STREAMED gather: rank r's piece (bytes_per_rank[r] bytes; 0 = not in this round) lands at byte
 * offsets_per_rank[r] of the root's clip buffer recv_base - the k-th finished chunk of every rank, sent on a side stream
 * while chunk k + 1 renders, so the root's ingress hides behind the render.  The root's own piece is not moved. */
int maua_gather_frames_at(maua_comm* comm, const uint8_t* send, const long* bytes_per_rank, uint8_t* recv_base,
                          const long* offsets_per_rank, int root);
/* the stream this communicator's transfers are enqueued on (default: the context's stream).  The streamed gather gives it a
 * side stream that waits on an event of the render stream, so that a round travels while the next chunk renders
 * (use_ctx_stream != 0: back to the context's stream). */
int maua_comm_set_stream(maua_comm* comm, void* stream, int use_ctx_stream);
/* the communicator's own rank count / rank (ncclCommCount, ncclCommUserRank): what a result line quotes as its RCCL world */
int maua_comm_count(maua_comm* comm, int* nranks, int* rank);
int maua_comm_destroy(maua_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* MAUA_HIP_H */

// Internal launcher prototypes shared between the translation units of libmaua_hip.so.
#pragma once
#include "common.h"

namespace maua {

// ---- ops_kernels.hip
int launch_pack_rgb8(hipStream_t stream, const float* img, uint8_t* out, int B, int H, int W);
template <typename TI, typename TO>
int launch_nchw_to_nhwc(hipStream_t s, const void* x, void* y, int N, int C, int HW, int Cp);
template <typename TI, typename TO>
int launch_nhwc_to_nchw(hipStream_t s, const void* x, void* y, int N, int C, int HW, int Cp);

// ---- modconv.hip
// Arguments of one fused modulated-convolution launch (3x3, up in {1,2}); all pointers device.
struct ConvArgs {
  const void* x;        // NHWC [B][H][W][Ci] (T)
  long x_bstride;       // elements between samples (0: broadcast, e.g. the learned const input)
  const void* w;        // [phases][9][Co][Ci] (T), phases = up*up
  const float* s;       // [B][Ci] styles (input scale)
  const float* d;       // [B][Co] demodulation coefficients or NULL
  const float* noise;   // [B|1][Ho][Wo] or NULL
  long noise_bstride;   // elements (0: broadcast)
  float noise_strength;
  const float* noise_scale;  // [B] or NULL: per-sample factor on the noise (un-normalised Loop maps: 1 / (rms + eps), noise.hip raw mode)
  const float* bias;    // [Co] or NULL
  void* y;              // NHWC [B][Ho][Wo][Co] (T)
  int B, H, W, Ci, Co, up;
  int act;
  float alpha, gain, clamp;
  // optional fused toRGB + upsampled skip on a conv1 layer whose channels fit one N tile (bf16, up == 1, Co == 128:
  // modconv_rgb_fusable); rgb_out == NULL disables.  Same meaning as in HiresArgs.
  const float* rgb_wmod;  // [B][3][Co]
  const float* rgb_bias;  // [3]
  const float* rgb_prev;  // [B][3][H/2][W/2] or NULL
  float* rgb_out;         // [B][3][H][W]
  float rgb_clamp;
  float fir[16];
  int variant;            // launch_tconv2: TCONV_EDGES_ONLY restricts the launch to the thin regions (0 = everything)
  // channel-sliced operands (modconv3x3_kernel only; 0 = dense): elements between consecutive pixels of x / y, first
  // output channel inside a y pixel (dense-block buffers of the RRDB network, super.hip)
  int x_pstride, y_pstride, y_coff;
  long y_bstride;         // elements between samples of y (0 = Ho * Wo * Co)
  const float* out_scale; // modconv_dma: [B][Co] or NULL - the stored features are multiplied by the NEXT layer's styles
  void* y_scaled;         // modconv_dma, with out_scale: y keeps the plain features (a separate toRGB pass reads them) and the
                          // scaled copy goes here, dense [B][H][W][Co] - instead of a premod pass over y afterwards
  const void* res;        // optional residual added after activation / gain / clamp: NHWC, res_pstride elements per pixel
  int res_pstride;
  long res_bstride;
  // optional second residual applied to the value as it would have been stored (rounded to the network dtype):
  // y = res_gain * y + res2 - the RRDB-level "out * 0.2 + x" folded into the last dense block's conv5 (super.hip)
  const void* res2;
  int res2_pstride;
  long res2_bstride;
  float res_gain;
  // modconv_dma: input channels actually read (0 = Ci).  A K dimension padded with zero weights up to the kernel's chunk pair
  // (Ci = 128 for a 96-channel layer) need not fetch the padding: only ceil(Ci_read / 32) chunks are loaded, at least 2.
  int Ci_read;
  // modconv_dma: x is a half-size tensor [B][H/2][W/2] that the layer sees up-sampled x2 by pixel repetition
  // (F.interpolate(scale_factor=2, mode="nearest") in front of RRDBNet's conv_up1 / conv_up2): the up-sampled tensor is never
  // written, the halo loads address the source pixel.  x_bstride is the SOURCE's sample stride.
  int x_up2;
  // modconv_dma (32-channel tile): write channels 0..2 of the output as an image instead of storing y - planar f32
  // [B][3][H][W] (clamped to [0, 1] when img_clamp) and / or u8 HWC = round_half_even(clamp(v, 0, 1) * 255); y is not written
  float* img_f32;
  uint8_t* img_u8;
  int img_clamp;
  int img_h, img_w;       // the u8 frame's own size (0 = H x W): only the top-left img_h x img_w window is written, densely - an up-scaler
                          // whose input was reflect-padded (RealESRGANer.pre_pad) crops its output in the store instead of in a copy
  const float* prelu;     // optional per-channel negative slopes [Co] (PReLU: replaces act / alpha; SRVGGNetCompact, super.hip)
  // modconv_dma (wide tiles) only: optional side output for a GroupNorm that follows (unet.hip) - per (sample, 8 x 32-pixel tile)
  // row and 8-channel piece the sum and the sum of squares of the STORED values: psum[b][tile][Co / 8][16] floats
  // ([0..8) sums, [8..16) sums of squares of the piece's channels).  Saves the statistics pass its read of the tensor.
  float* psum;
};
int launch_modconv3x3(hipStream_t stream, int dtype, const ConvArgs& a);
bool modconv_rgb_fusable(int dtype, int Ci, int Co, int up, int H, int W);

// modconv_dma.hip: up = 1, bf16, input already multiplied by the styles (x * s); both operands by LDS-direct loads
bool dma_conv_supported(int dtype, int Ci, int Co, int up, int H, int W);
bool dma_rgb_fusable(int Co);
bool dma_conv_narrow_supported(int dtype, int Ci, int Co, int H, int W);  // 32 / 64 output channels (plain convs: x_pstride / y_pstride / y_coff / res honoured)
int launch_modconv_dma(hipStream_t stream, const ConvArgs& a, int dtype = MAUA_BF16);
int dma_psum_rows(const ConvArgs& a);   // rows per sample of ConvArgs.psum for such a launch
int launch_premod_nhwc(hipStream_t stream, const void* x, long x_bstride, const float* s, void* y, int B, long HW, int Ci,
                       int dtype = MAUA_BF16);

// lowest-resolution layers (modconv_lowres.hip, <= 8x8 input pixels per sample): the GEMM over all samples at once,
// split-K + deterministic reduce/epilogue.  xm / ws: workspaces of at least lowres_workspace() bytes.
bool lowres_supported(int dtype, int Ci, int Co, int up, int H, int W);
void lowres_workspace(int dtype, int B, int H, int W, int Ci, int Co, int up, size_t* xm_bytes, size_t* ws_bytes);
int launch_modconv_lowres(hipStream_t stream, int dtype, const ConvArgs& a, void* xm, float* ws);

// ... the same gather GEMM for plain 3x3 convolutions (a.s == NULL, dense input) on grids of <= 1024 pixels; honours
// bias / act / gain / res / y_pstride / y_coff.  ws: gather_conv_workspace() bytes.
bool gather_conv_supported(int dtype, int Ci, int Co, int H, int W);
size_t gather_conv_workspace(int dtype, int B, int H, int W, int Ci, int Co);
int launch_conv_gather(hipStream_t stream, int dtype, const ConvArgs& a, float* ws);

// high-resolution specialisation (modconv_hires.hip): weights stationary in registers, persistent tile walk,
// optional fused toRGB + skip on conv1 layers
struct HiresArgs {
  const void* x;        // NHWC bf16 [B][H][W][Ci]
  const void* w;        // prepared weights [9][phases][Co][Ci] bf16
  const float* s;       // [B][Ci]
  const float* d;       // [B][Co] or NULL
  const float* noise;   // [B|1][H*up][W*up] or NULL
  long noise_bstride;
  float noise_strength;
  const float* noise_scale;  // [B] or NULL: per-sample factor on the noise (as in ConvArgs)
  const float* bias;    // [Co] or NULL
  void* y;              // NHWC bf16 [B][H*up][W*up][Co]; NULL with rgb_out: features are not stored
  int B, H, W, Ci, Co, up, act;
  float alpha, gain, clamp;
  // fused toRGB (conv1 only; rgb_out == NULL disables)
  const float* rgb_wmod;  // [B][3][Co]
  const float* rgb_bias;  // [3]
  const float* rgb_prev;  // [B][3][H/2][W/2] or NULL
  float* rgb_out;         // [B][3][H][W]
  float rgb_clamp;
  float fir[16];
  uint8_t* rgb8_out;      // optional: the final frame packed to u8 HWC in the same epilogue (last block only)
  int rgb_skip_f32;       // with rgb8_out: do not store the f32 image (nobody reads it)
};
bool hires_supported(int dtype, int Ci, int Co, int up, int H, int W);
int launch_modconv_hires(hipStream_t stream, const HiresArgs& a, int dtype = MAUA_BF16);
// modconv_upwalk.hip: the 64 -> 32 channel up-layer in half-folded form (horizontal FIR in the weights, vertical FIR on
// the accumulators of a row walk); a.w = weights from launch_prep_upwalk_weights ([3][2][3][Co][Ci] bf16)
bool upwalk_supported(int dtype, int Ci, int Co, int up, int H, int W);
size_t upwalk_weight_elems(int Co, int Ci);
int launch_upwalk(hipStream_t stream, const HiresArgs& a, int dtype = MAUA_BF16);
int launch_prep_upwalk_weights(hipStream_t stream, const float* w, void* wt, int Co, int Ci, int flip, int dtype = MAUA_BF16);
// ... and the whole block (that up-layer, the 3x3 conv1 behind it, toRGB + skip, optional u8 pack) in one walk: the
// block's features never reach HBM.  up = conv0's arguments (w from launch_prep_upwalk_weights, y unused), c1 = conv1's
// (w from launch_prep_weights, rgb_* set, y unused)
bool upwalk_fused_supported(int dtype, int Ci, int Cm, int H, int W);
// force_segs > 0: that many row segments instead of the cost model's choice; narrow_ok: walk a last strip of <= 32 columns as two
// half-height sub-items (both: results are identical by construction, tests compare them)
int launch_upwalk_fused(hipStream_t stream, const HiresArgs& up, const HiresArgs& c1, int force_segs = 0, int narrow_ok = 1,
                        int dtype = MAUA_BF16);

// weight preparation: f32 [Co][Ci][k][k] -> T [phases][k*k][Cop][Cip] (+ Wsq f32 [Co][Ci] = sum_k W^2)
int launch_prep_weights(hipStream_t stream, int dtype, const float* w, void* wt, float* wsq, int Co, int Ci, int k,
                        int up, int flip, int Cop, int Cip);
// MAUA_F32_SPLIT: a float32 weight buffer prepared by launch_prep_weights(MAUA_F32, ...) -> [hi x 8 | lo x 8] bf16 per 8 floats (one 32-byte
// group per lane half of modconv.hip's k-step), in place; n_floats % 8 == 0
int launch_f32_split_inplace(hipStream_t stream, void* w, long n_floats);

// modconv_tconv.hip: up-layer as the minimal stride-2 transposed convolution; writes the raw tensor
// t [B][2H+1][2W+1][Co] (uses x, x_bstride, w (from launch_prep_tconv_weights), s, y, B, H, W, Ci, Co of ConvArgs)
int launch_tconv2(hipStream_t stream, int dtype, const ConvArgs& a);
constexpr int TCONV_EDGES_ONLY = 100;  // ConvArgs.variant: launch_tconv2 covers only the last row / column of positions
// modconv_tconv_dma.hip: the main H x W block on LDS-direct loads; x already multiplied by the styles (bf16)
bool tconv_dma_supported(int dtype, int Ci, int Co, int H, int W);
int launch_tconv_dma(hipStream_t stream, const ConvArgs& a, int dtype = MAUA_BF16);
int launch_tconv_edges(hipStream_t stream, const ConvArgs& a, int dtype = MAUA_BF16);   // the thin last row / column of positions (bf16, pre-modulated x)
int launch_prep_tconv_weights(hipStream_t stream, int dtype, const float* w, void* wt, int Co, int Ci, int flip);

// second half of the minimal up-layer: out = act(d * FIR4x4(t) + noise + bias) (ops.py:225 upfirdn2d pad 1 gain 4,
// then :184-185 noise and bias_act :65-84), t [B][2H+1][2W+1][Co] -> y [B][2H][2W][Co], both NHWC in dtype
struct UpfirArgs {
  const void* t;
  void* y;
  const float* d;      // [B][Co] or NULL
  const float* noise;  // [B|1][2H][2W] or NULL
  long noise_bstride;
  float noise_strength;
  const float* noise_scale;  // [B] or NULL: per-sample factor on the noise (as in ConvArgs)
  const float* bias;   // [Co] or NULL
  const float* out_scale;  // [B][Co] or NULL: the output is multiplied by the NEXT layer's styles (modconv_dma.hip)
  int B, H, W, Co;     // H, W = INPUT grid of the layer (output is 2H x 2W)
  int act;
  float alpha, gain, clamp;
};
int launch_upfir_epilogue(hipStream_t stream, int dtype, const UpfirArgs& a);
// modconv_tconv_fir.hip: both halves in one kernel, t stays in LDS (bf16; x already multiplied by the styles); the output is
// bit-identical to launch_tconv_dma (+ edges) followed by launch_upfir_epilogue
bool tconv_fir_supported(int dtype, int Ci, int Co, int H, int W);
int launch_tconv_fir(hipStream_t stream, const ConvArgs& a, const UpfirArgs& u, int dtype = MAUA_BF16);
size_t prepped_weight_elems(int k, int up, int Cop, int Cip);

// resize.hip: bicubic / pad / crop of NHWC features (network dtype) or planar images; optional per-channel noise
struct ResizeArgs {
  const void* x;
  long x_bstride;   // elements between samples (0 = broadcast one sample)
  void* y;
  int B, H, W, C, oh, ow;
  int mode;         // 0 bicubic, 1 pad / crop
  int align;        // bicubic: align_corners (source = o * (in-1)/(out-1)) instead of half-pixel centres
  int pl, pt;       // left / top offset of the input inside the output (negative = crop)
  int how;          // maua_pad_mode
  float value;
  const float* noise;  // [C][oh][ow] or NULL
};
int launch_resize2d(hipStream_t stream, int dtype, bool nhwc, const ResizeArgs& a);
int launch_skip_add(hipStream_t stream, const float* y, const float* prev, float* out, int B, int H, int W,
                    const float* fir16);
// bilinear, reflection-padded affine warp of NHWC features; minv [B][6] maps output pixels to source pixels
int launch_warp_affine_nhwc(hipStream_t stream, int dtype, const void* x, void* y, const float* minv, int B, int H, int W,
                            int C);

// styles / demod / toRGB pre-modulation for a list of layers in one launch
struct StyleLayer {
  const float* affine_w;  // [Cin][w_dim]
  const float* affine_b;  // [Cin]
  const float* wsq;       // [Co][Cin] or NULL (no demod)
  const float* wrgb;      // [3][Cin] or NULL
  float* s;               // out [B][Cs]   (Cs >= Cin, padding written as 0)
  float* d;               // out [B][Cd]   or NULL
  float* wmod;            // out [B][3][Cin] or NULL
  int w_index;            // which of the num_ws vectors feeds this layer
  int Cin, Co, Cs, Cd;
  float scale;            // 1 (conv) or 1/sqrt(Cin) (toRGB)
};
// f16_prenorm: the styles of the demodulating layers divided by their per-sample maximum before the demodulation coefficients
// are formed (ops.py:161-165; MAUA_F16 networks)
int launch_styles(hipStream_t stream, const StyleLayer* layers_dev, int n_layers, const float* ws, int num_ws,
                  int w_dim, int B, int max_channels, int f16_prenorm = 0);
// ops.py:161-165 on plain tensors: out[co] = w[co] / (max |w[co]| * sqrt(Ci kk)); s[b][0 .. Cin) /= max |s[b]|
int launch_f16_prenorm_weights(hipStream_t stream, const float* w, float* out, int Co, int Ci, int kk);
int launch_f16_prenorm_styles(hipStream_t stream, float* s, int B, int Cs, int Cin);

// toRGB (1x1 modconv, no demod, + bias, clamp) + FIR-upsampled skip + add -> f32 planar image
struct RgbArgs {
  const void* x;      // NHWC [B][H][W][C] (T)
  const float* wmod;  // [B][3][C]
  const float* bias;  // [3]
  const float* prev;  // [B][3][H/2][W/2] or NULL
  float* out;         // [B][3][H][W]
  int B, H, W, C;
  float clamp;
  float fir[16];      // 4x4 filter incl. gain 4 (upsample2d, ops.py:117-133)
};
int launch_torgb(hipStream_t stream, int dtype, const RgbArgs& a);

// ---- gemm.hip: C[M][N] = A[M][K] x W[N][K]^T (+ bias[N]) (+ res[M][N]); A's columns may come from two tensors (K0 from
// a0, K1 from a1: a channel concatenation that is never materialised).  T = network dtype; K0, K1 multiples of 64 bytes,
// N % 32 == 0.  The 1x1 convolutions of the diffusion UNet (NHWC rows = pixels).
struct GemmArgs {
  const void* a0; long lda0; int K0;
  const void* a1; long lda1; int K1;
  const void* w;        // T [N][K0 + K1]
  const float* bias;    // [N] or NULL
  const void* res;      // T [M][ldr] or NULL
  long ldr;
  void* c;              // T [M][ldc] (f32 when c_f32)
  long ldc;
  long M;
  int N;
  int c_f32;
  // gemm_dma.hip only (prefer_dma: large plain GEMMs of the CLIP image tower, clip.hip; bf16, N % 128 == 0, K % 64 == 0):
  int prefer_dma;       // route to the LDS-direct kernel when the shape allows it
  int epi;              // 0; 1: c2 <- QuickGELU(c) as well; 2: c <- c * QuickGELU'(aux)
  void* c2; long ldc2;
  const void* aux; long ldaux;
  // register-staged kernels only (gemm.hip): `batch` independent products in one launch (grid z) - a0 / w / c of product b at
  // + b * a_bstride / w_bstride / c_bstride ELEMENTS (0 batch = 1 product; a1, res, bias unsupported with batch > 1).  The style head's
  // per-image d loss / d F = F_b (dG_b + dG_b^T) (perceptor.hip)
  int batch;
  long a_bstride, w_bstride, c_bstride;
  // set by launch_gemm_nt (128 x 128 kernel): > 0 = the launch is one-dimensional and block L computes N tile (L / 8) % remap_nt of
  // M tile ((L / 8) / remap_nt) * 8 + L % 8 - the N tiles of one M tile run back to back on ONE XCD (block b runs on XCD b % 8), so the
  // second and later reads of its A rows hit that XCD's L2 instead of HBM
  int remap_nt;
};
int launch_gemm_nt(hipStream_t stream, int dtype, const GemmArgs& g);
// gemm_dma.hip: 256 x 128 tiles on LDS-direct loads (see there); launch_gemm_nt routes when g.prefer_dma
bool gemm_dma_supported(int dtype, const GemmArgs& g);
int launch_gemm_dma(hipStream_t stream, const GemmArgs& g);

// ---- attention.hip: softmax(Q K^T / sqrt(D)) V per (sample, head); qkv [B][T][ld_qkv] with head-major [q | k | v] channel
// layout (guided-diffusion's QKVAttentionLegacy), out [B][T][ld_out] with channel = head * D + d
struct AttnArgs {
  const void* qkv;
  void* out;
  int B, T, heads, D;
  long ld_qkv, ld_out;
  float scale;          // 1 / sqrt(D)
  float* lse;           // optional [B][heads][T]: each row's log-sum-exp of the scaled scores (what the input gradient needs)
};
bool attention_supported(int head_ch);
int launch_attention(hipStream_t stream, int dtype, const AttnArgs& a);

// attention_vjp.hip: the input gradient of launch_attention.  d_out [B][T][ld_out] -> d_qkv [B][T][ld_qkv]; lse = what the
// forward left in AttnArgs.lse, out = its result; workspace: B * heads * T floats (the rows' sum d_out . out)
struct AttnVjpArgs {
  const void* qkv;
  const void* out;
  const void* d_out;
  const float* lse;
  void* d_qkv;
  float* delta;
  int B, T, heads, D;
  long ld_qkv, ld_out;
  float scale;
};
int launch_attention_vjp(hipStream_t stream, int dtype, const AttnVjpArgs& a);

// groupnorm_vjp.hip: the input gradient of the UNet's GroupNorm (+ scale-shift) (+ SiLU) (+ resample) pass over [x0 | x1]
// ([B][H][W][C0 | C1], dense NHWC in the network dtype).  dy, dres: [B][Ho][Wo][C0 + C1] at the forward's output size (dres: the
// gradient of the resampled raw input - the ResBlock's x_upd - or of an identity skip; optional); add0 / add1: optional gradients
// already known for x0 / x1 (may alias dx0 / dx1); stats: the forward's [B][32][2] (mean, rstd).
struct GnVjpArgs {
  const void* x0; int C0;
  const void* x1; int C1;
  const float* stats;
  const float* gamma;
  const float* beta;
  const float* ss;      // [B][ss_ld]: scale at [c], shift at [C + c]; or NULL
  long ss_ld;
  int silu, mode;
  const void* dy;
  const void* dres;
  const void* add0;
  const void* add1;
  void* dx0;
  void* dx1;
  int B, H, W;
};
size_t group_norm_vjp_workspace(int B, int C, long HW, int esize);
int launch_group_norm_vjp(hipStream_t stream, int dtype, const GnVjpArgs& a, void* workspace);

// cutouts.hip: random cutouts resized to the perceptor's input (maua/ops/cutouts.py:8-50 as CLIPGrads calls it) and their gradient.
// rects: DEVICE [n_cut][3] (size, top, left); tables: cutouts_table_bytes() of device scratch filled by launch_cutout_tables;
// out / d_out: planar f32 [n_cut * B][3][cs][cs] (patch == 0) or patch rows in dtype [n_cut * B * (cs / patch)^2][3 * patch^2];
// th: cutouts_th_bytes() of scratch; grad: [B][3][H][W] f32
// a rectangle's size entry carries two flags in its high bits (DangoCutouts, cutouts.py:171-199): the cutout is converted to
// 3-channel luma (torchvision Grayscale(3): 0.2989 r + 0.587 g + 0.114 b) / mirrored horizontally (TF.hflip)
constexpr int CUT_GREY = 1 << 30, CUT_FLIP = 1 << 29, CUT_SIZE_MASK = (1 << 24) - 1;
struct CutoutPlan {
  const float* img;
  const int* rects;        // [n_cut][3] (size | flags, top, left)
  int B, H, W, n_cut, cs;
  float mul, add;          // affine applied to the image first ((img + 1) / 2: 0.5, 0.5)
  float mean[3], std[3];   // Normalize applied to the cutouts
  int patch;
};
size_t cutouts_table_bytes(int n_cut, int cs);
size_t cutouts_th_bytes(int n_cut, int B, int cs, int smax);
int launch_cutout_tables(hipStream_t stream, const CutoutPlan& p, void* tables);
int launch_cutouts_forward(hipStream_t stream, int dtype, const CutoutPlan& p, void* tables, void* out);
int launch_cutouts_vjp(hipStream_t stream, int dtype, const CutoutPlan& p, void* tables, const void* d_out, float* th, float* grad,
                       int accumulate);

// colormatch.hip: ColorMatchGrads.forward on caller-owned workspaces (nothing allocated: capturable), the conditioning's NaN-screened sum
size_t colormatch_fix_bytes(int B, int nbins);
int colormatch_grad_into(hipStream_t st, const float* img, int B, int H, int W, int nbins, int sat_weighting, const float* target,
                         int target_per_sample, float scale, unsigned long long* fix, float* gr, float* grad, float* loss);
int screened_accumulate(hipStream_t st, const float* sub, float* acc, long n, int first, int* flag);
// perceptor.hip
maua_ctx* vgg_ctx(maua_vgg* n);
unsigned long long vgg_epoch(maua_vgg* n);   // generation of the network's device buffers (a captured graph compares it before reuse)
// guides.hip: a grad module of the guided loop (maua_guide_*): evaluates d loss / d img into `out` on the context's stream;
// guide_prepare allocates for a batch shape (never inside a capture)
int guide_prepare(maua_guide* g, int B, int H, int W);
int guide_eval(maua_guide* g, const float* img, int B, int H, int W, float* out);
maua_ctx* guide_ctx(maua_guide* g);
unsigned long long guide_uid(maua_guide* g);
unsigned long long guide_epoch(maua_guide* g);   // generation of the device buffers a guide's launches point into

// secondary.hip: the context a secondary diffusion model was created on
maua_ctx* secondary_ctx(maua_secondary* n);
// (uid, epoch) of a secondary model's device buffers: whoever caches pointers into them (a captured graph) compares both before reuse
void secondary_stamp(maua_secondary* n, unsigned long long* uid, unsigned long long* epoch);

// clip.hip: pieces the captured guided loop (unet.hip) drives.  clip_prepare_guide allocates (never inside a capture);
// clip_guide_grad = CLIPGrads.forward on DEVICE rectangles [batches][cutn][3]
maua_ctx* clip_ctx(maua_clip* n);
void clip_stamp(maua_clip* n, unsigned long long* uid, unsigned long long* epoch);
int clip_group_size(maua_clip* n, int B, int cutn);
int clip_prepare_guide(maua_clip* n, int B, int H, int W, int n_cut_group);
int clip_guide_grad(maua_clip* n, const float* img, int B, int H, int W, const int* rects_dev, const float* mult_dev, int cutn, int cutn_total,
                    int batches, float scale, float clamp_gradient, float* grad);

}  // namespace maua

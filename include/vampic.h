/*
 * libvampic — C ABI of the MI355X-native hot path of the variance-aware-masking
 * progressive image codec (reference: das-ankur/Efficient-PIC-with-Variance-Aware-Masking).
 *
 * The reference has no FFI: its boundary is the Python model object
 * (src/models/pic.py, src/models/rem_pic.py).  Each entry point below replaces the
 * torch operator pattern cited next to it; the Python package
 * `efficient-pic-with-variance-aware-masking_amd` binds them with ctypes and keeps the
 * reference's module/attribute surface on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are NHWC fp32: element (b,y,x,c) of a tensor with pixel stride
 *     `ld` lives at ptr[((b*H + y)*W + x)*ld + c]  (ld >= C lets a tensor be a channel
 *     window of a wider buffer, which is how torch.cat is avoided);
 *   - the caller owns every buffer; kernels are enqueued on `stream` (a hipStream_t
 *     passed as void*), never allocate, never synchronise;
 *   - return value 0 = ok, negative = VAM_E*; vam_last_error() gives the message of
 *     the last failure on the calling thread.
 */
#ifndef VAMPIC_H
#define VAMPIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAM_OK 0
#define VAM_EINVAL (-1)   /* bad argument / unsupported shape */
#define VAM_EHIP (-2)     /* a HIP runtime call failed        */
#define VAM_ENOGPU (-3)   /* no gfx950 device visible          */

const char* vam_last_error(void);
int vam_version(void);
/* 0 if a HIP device is usable; fills name (<=128 bytes) and CU count. */
int vam_device_info(char* name128, int* cu_count);

/* ------------------------------------------------------------------ convolution */

/* activation applied to (acc + bias [+ pre]) */
enum vam_act {
  VAM_ACT_NONE = 0,
  VAM_ACT_GELU = 1,      /* exact erf GELU, nn.GELU() default  (pic.py:86)          */
  VAM_ACT_LEAKY = 2,     /* LeakyReLU(0.01)                    (layers/rem.py:41)   */
  VAM_ACT_HALF_TANH = 3, /* 0.5*tanh(v)                        (pic.py:550,637)     */
  VAM_ACT_SIGMOID = 4,   /*                                    (layers/layers.py:72)*/
  VAM_ACT_CLAMP01 = 5,   /* clamp_(0,1)                        (pic.py:558,651)     */
  VAM_ACT_RSQRT = 6,     /* GDN   (layers/gdn.py:72)                                */
  VAM_ACT_SQRT = 7,      /* IGDN  (layers/gdn.py:70)                                */
  VAM_ACT_DOUBLE = 8     /* 2 v: GDN backward, dx = dy dy/dx|norm + x * 2 (gamma'^T dL/dnorm) in one launch
                            (autograd of layers/gdn.py:62-75; exact: a power-of-two factor)  */
};

enum vam_conv_flags {
  VAM_CONV_SQUARE_IN = 1,   /* operand is x*x (GDN norm pool, gdn.py:68)                 */
  VAM_CONV_PS2 = 2,         /* phase-major output channels scattered like PixelShuffle(2):
                               n = phase*Cq + c -> pixel (2y+phase/2, 2x+phase%2), chan c  */
  VAM_CONV_OUT_NCHW = 4,    /* store the result NCHW (model edge, x_hat)                  */
  VAM_CONV_IN_BF3 = 8,      /* every input segment holds bf16x3 planes ("P3": [pixel][8-channel group][plane 3][8 bf16],
                               48 bytes per group; seg.ld counts GROUPS per pixel, seg.C channels) written by a launch
                               with VAM_CONV_OUT_BF3 — the split-operand kernel then stages them by plain copies     */
  VAM_CONV_OUT_BF3 = 16,    /* write the result as P3 planes (ldo counts groups per pixel) instead of fp32 NHWC; for
                               tensors whose only consumer is another convolution (inside conv stacks)               */
  /* bf16-storage mode (BASELINE configs[2] "bf16": activations of the large feature maps stored as bf16, fp32
   * accumulation; never used by the fp32 configurations) */
  VAM_CONV_W_BF16 = 32,     /* wpack comes from vam_pack_conv_weights_bf16: bf16 x bf16 products, one MFMA per block    */
  VAM_CONV_IN_BF16 = 64,    /* every input segment is a bf16 NHWC tensor (seg.ld counts bf16 elements, multiple of 8)   */
  VAM_CONV_OUT_BF16 = 128,  /* store the result as bf16 NHWC (ldo counts bf16 elements)                                 */
  VAM_CONV_AUX_BF16 = 256,  /* pre / mul / post / post2 are bf16 NHWC tensors (their ld counts bf16 elements)           */
  /* training (taped forward / backward plans): activations fused into the producing launch */
  VAM_CONV_MUL_GELU_GRAD = 512 /* the mul operand holds the PRE-ACTIVATION z of a GELU and the result is multiplied by
                               gelu'(z) instead: out = ... + gelu'(mul) * act(...).  A data-gradient launch then delivers
                               dL/dz of the GELU in front of the layer directly (autograd of nn.GELU, pic.py:86)     */
};

#define VAM_MAX_SEG 4

typedef struct vam_seg {
  const float* ptr; /* first channel of this segment at pixel (0,0,0) */
  int32_t C;        /* channels taken from it                          */
  int32_t ld;       /* pixel stride of the underlying buffer (floats)  */
} vam_seg;

typedef struct vam_aux {
  const float* ptr; /* NULL = unused; indexed like the output (pixel*ld + channel) */
  int32_t ld;
  int32_t pad_;
} vam_aux;

/*
 * One convolution problem:  out = post2 + post + mul * act(conv(cat(seg...)) + bias + pre)
 * (and, when preact is set, preact = conv(cat(seg...)) + bias + pre: the taped training forward keeps a GELU's input)
 * Replaces nn.Conv2d / nn.ConvTranspose2d (one sub-pixel phase per problem) /
 * nn.Linear and the element-wise ops the reference applies around them
 * (layers/layers.py:5-86, layers/gdn.py:62-75, layers/rem.py:52-66,130-141,
 *  models/pic.py:528-551,598-641).
 */
typedef struct vam_conv {
  vam_seg seg[VAM_MAX_SEG];
  int32_t n_seg;
  int32_t B, H, W;          /* input extent                                           */
  int32_t kh, kw;           /* taps                                                   */
  int32_t stride;           /* 1 or 2                                                 */
  int32_t pad_y, pad_x;     /* input y = oy*stride - pad_y + ty                       */
  int32_t Ho, Wo;           /* extent of the grid of output positions of THIS problem */
  int32_t N;                /* output channels (packed order)                         */
  const float* wpack;       /* from vam_pack_conv_weights (same kh,kw,Cin,N,BK)       */
  const float* bias;        /* N floats (packed order) or NULL                        */
  float* out;
  int32_t ldo;              /* output pixel stride (floats)                           */
  int32_t Hf, Wf;           /* full output extent                                     */
  int32_t osy, osx, ooy, oox; /* output pixel = (oy*osy+ooy, ox*osx+oox) (ignored w/ PS2) */
  int32_t Cq;               /* PS2: channels per phase (N == 4*Cq)                    */
  int32_t act;              /* enum vam_act                                           */
  int32_t flags;            /* enum vam_conv_flags                                    */
  vam_aux pre, mul, post, post2;
  const int32_t* in_amax[VAM_MAX_SEG]; /* fp16x2 mode (vam_conv_set_mode(3)): per input segment a device cell holding the bits
                               of an upper bound of max |x| of that segment (vam_absmax, or the out_amax cell of the launches
                               that wrote it); the launch scales by the largest.  Ignored in the other modes.            */
  int32_t* out_amax;        /* any mode, optional: the epilogue folds max |stored value| into this device cell (integer
                               atomicMax on the float's bits; zero the cell before the first producer of a step)     */
  vam_aux preact;           /* optional SECOND output (ptr is written): the value the activation is applied to, fp32 NHWC,
                               indexed like the output (pixel*ld + channel)                                           */
} vam_conv;

/* sizeof(vam_conv) as compiled into the library (binding layout guard). */
size_t vam_conv_struct_size(void);
/* Size in floats of the packed weight buffer for (kh,kw,Cin,N). */
size_t vam_conv_wpack_floats(int kh, int kw, int cin, int n);

enum vam_pack_mode {
  VAM_PACK_CONV = 0,     /* src OIHW [N][Cin][kh][kw]   (nn.Conv2d, nn.Linear with kh=kw=1) */
  VAM_PACK_DECONV5S2 = 1,/* src IOHW [Cin][Cout][5][5] (layers/layers.py:14-22); builds the
                            sub-pixel phase `phase` (0..3 = py*2+px) as a (kh,kw) in
                            {3,2}x{3,2} correlation, or with phase = -1 the merged
                            3x3 / N = 4*Cout phase-major form used with VAM_CONV_PS2   */
  VAM_PACK_PS2 = 2,      /* src OIHW with O = Cq*4 in PixelShuffle order (c*4+i*2+j) ->
                            packed n = (i*2+j)*Cq + c   (layers/layers.py:82-86)          */
  VAM_PACK_GDN = 3,      /* src gamma [N][Cin] in reparametrised storage -> max(g,2^-18)^2-2^-36
                            (layers/gdn.py:52-66, compressai NonNegativeParametrizer)     */
  VAM_PACK_CONV_DGRAD = 4,/* weights of the DATA-GRADIENT conv of a stride-1 nn.Conv2d: src is the forward
                            OIHW tensor [Cout][Cin][kh][kw]; call with cin = Cout, n = Cin (taps flipped,
                            channel roles swapped) — what autograd's conv backward-data computes          */
  VAM_PACK_GDN_T = 5     /* VAM_PACK_GDN with the reparametrised gamma TRANSPOSED: the 1x1 problem that carries
                            dL/dnorm back to the squared inputs (GDN / IGDN backward)                       */
};
/* Device-side repack of a weight tensor into the kernel's [tap][k-chunk][n][k] layout. */
int vam_pack_conv_weights(const float* src, float* dst, int mode, int phase,
                          int kh, int kw, int cin, int n, void* stream);
/* bf16-storage mode: weights rounded to nearest-even bf16, [tap][32-channel chunk][N up to 32][32 bf16]. */
size_t vam_conv_wpack_bf16_bytes(int kh, int kw, int cin, int n);
int vam_pack_conv_weights_bf16(const float* src, void* dst, int mode, int phase, int kh, int kw, int cin, int n,
                               void* stream);
/* Many repacks as ONE launch (training refreshes every trained layer's packed weights after every optimiser step).
 * bias != 0: a vam_pack_bias job (kh, kw, cin, phase ignored); else a vam_pack_conv_weights job. */
#define VAM_MAX_PACK_GROUP 32
typedef struct vam_pack_job {
  const float* src;
  void* dst;
  int32_t bias;
  int32_t mode, phase, kh, kw, cin, n;
  int32_t pad_;
} vam_pack_job;
int vam_pack_group(const vam_pack_job* jobs, int n_jobs, void* stream);
/* bias helpers: PS2 permutation / GDN beta reparam (max(b, sqrt(1e-6+2^-36))^2 - 2^-36) /
 * merged-deconv replication (4x). mode as above. */
int vam_pack_bias(const float* src, float* dst, int mode, int n, void* stream);

/* max |x| over up to VAM_MAX_SEG NHWC windows of n_pix pixels each, folded into *cell like vam_conv.out_amax. */
int vam_absmax(const vam_seg* segs, int n_seg, long n_pix, int32_t* cell, void* stream);

/* Launch up to VAM_MAX_GROUP independent problems as ONE grid (grouped launch:
 * e.g. the mean and scale stacks of one slice, or the four phases of a deconv). */
#define VAM_MAX_GROUP 8
int vam_conv_group(const vam_conv* problems, int n_problems, void* stream);
/* Tuning hook: force the tile (BM in {64,128}, BN in {32,...,224}, BK in {16,32}); 0 = automatic. */
int vam_conv_force_tile(int bm, int bn, int bk);
/* Test / measurement hook.  fp32 NHWC outputs normally leave the kernel straight from the accumulator registers
 * ("direct" epilogue); 1 sends every problem through the LDS-staged epilogue instead (the one bf16 / plane / NCHW
 * outputs always take), 0 = automatic, -1 = follow the environment variable VAMPIC_EPILOGUE=staged.  Both epilogues
 * perform the same operations per element: results are bit-identical. */
int vam_conv_force_epilogue(int staged);
/* Arithmetic of the convolution kernel.  1 (default): every fp32 operand is split exactly into three bf16 terms and
 * the product is formed from six exact bf16 x bf16 partial products on the bf16 matrix pipe, fp32 accumulation
 * (error vs float64 no larger than the fp32 fma chain's); 0: fp32 operands on the fp32 matrix pipe;
 * 3 (opt-in, round 3 prototype): every fp32 operand scaled by a power of two (one per launch input, from vam_conv.in_amax;
 * one per output channel of the weights, made at pack time) and split into two fp16 terms, three products on the fp16
 * matrix pipe, fp32 accumulation — 22 of 24 significand bits per operand, below fp32 accumulation noise.
 * The mode fixes the packed-weight layout: choose it (or the environment variable VAMPIC_CONV=f32|bf16x3|f16x2) before
 * the first vam_conv_wpack_floats / vam_pack_conv_weights call and do not change it while packed weights exist. */
int vam_conv_set_mode(int mode);
int vam_conv_get_mode(void);
/* Diagnostics: the (BM, BN, BK) the tile heuristic picked for the most recent vam_conv_group launch. */
int vam_conv_last_tile(int* bm, int* bn, int* bk);

/* ------------------------------------------------------------------ fused residual unit */
/*
 * One ResidualUnit (reference layers/layers.py:30-48) as ONE launch:
 *   out = GELU( x + conv1x1(GELU(conv3x3(GELU(conv1x1(x))))) )
 * with both C/2-channel intermediates kept in LDS (csrc/resunit.hip).  The weights are the three convolutions'
 * ordinary packed weights (vam_pack_conv_weights, VAM_PACK_CONV, split-operand mode) and biases; results are
 * bit-identical to three vam_conv_group launches.  Built for C = 192 (the 64x64-position attention blocks of
 * g_a / g_s); vam_resunit_supported says whether a shape has a fused kernel — callers fall back to three
 * vam_conv_group launches (the same arithmetic) where it does not.
 */
typedef struct vam_resunit {
  const float* x;           /* input and identity, NHWC fp32 */
  float* out;               /* NHWC fp32, must not alias x   */
  int32_t ldx, ldo;         /* pixel strides (floats)        */
  int32_t B, H, W, C;
  const float* w1; const float* b1;   /* conv[0]: 1x1 C   -> C/2 */
  const float* w2; const float* b2;   /* conv[2]: 3x3 C/2 -> C/2 */
  const float* w3; const float* b3;   /* conv[4]: 1x1 C/2 -> C   */
  int32_t flags;            /* 0, or VAM_RESUNIT_BF16: bf16-storage mode (BASELINE configs[2]) — x / out are bf16 NHWC
                               tensors (ldx / ldo count bf16 elements, multiples of 8), w1..w3 come from
                               vam_pack_conv_weights_bf16, biases stay fp32; bit-identical to three vam_conv_group launches
                               with VAM_CONV_W_BF16 | IN_BF16 | OUT_BF16 [| AUX_BF16] */
  int32_t pad_;
} vam_resunit;
#define VAM_RESUNIT_BF16 1
size_t vam_resunit_struct_size(void);
int vam_resunit_supported(int C, int H, int W);
int vam_resunit_group(const vam_resunit* problems, int n_problems, void* stream);
/* Measurement hook: 1 (default) = weight slabs by LDS-DMA through a three-slot ring, 0 = register-staged (the A/B
 * arm), -1 = follow the environment variable VAMPIC_RU_DMA.  Both are bit-identical. */
int vam_resunit_set_dma(int mode);
/* Measurement hook: device buffer receiving 8 cycle-counter stamps per workgroup at the kernel's phase boundaries
 * (scratch/ru_phases.py); NULL (default) = off. */
int vam_resunit_set_debug(void* stamps);

/* ------------------------------------------------------------------ model edges */
/* x NCHW [B,3,H,W] -> space-to-depth NHWC [B,H/2,W/2,16] (12 real channels (py,px,c), 4 zero)
 * so that conv5x5 s2 (3->N) becomes a 3x3 s1 MFMA problem (layers/layers.py:5-12, builder.py:44). */
int vam_s2d_input(const float* x_nchw, float* out, int B, int H, int W, void* stream);
int vam_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int ld_dst, void* stream);
int vam_nhwc_to_nchw(const float* src, int ld_src, float* dst, int B, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------ window attention */
/* Swin block core (layers/win_attention.py:84-115,153-207): qkv is [B,H,W,3C] (q|k|v,
 * heads contiguous inside each), out[b,y,x,:] = softmax(q*scale k^T + bias + shiftmask) v
 * written at the un-shifted position; roll/partition/reverse are folded into addressing.
 * table: relative_position_bias_table [(2ws-1)^2][heads].  ws in {4,8}. */
int vam_win_attention(const float* qkv, int ld_qkv, float* out, int ld_out, const float* table,
                      int B, int H, int W, int C, int heads, int ws, int shift, void* stream);
/* 8 x 8 windows run on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products), forward and backward;
 * vam_attn_set_mfma(0) selects the FMA kernels instead (A/B measurements, equivalence test), -1 = follow the
 * environment variable VAMPIC_ATTN_MFMA.  vam_attn_mfma() returns the mode in force. */
int vam_attn_set_mfma(int mode);
int vam_attn_mfma(void);

/* ------------------------------------------------------------------ variance mask */
/* ChannelMask.forward "point-based-std" (layers/channel_mask.py:132-151) for n_seg
 * independent segments.  Segment s covers, for every pixel p < n_pix, the C channels at
 * sigma + s_b*batch_stride ... ; element (p,c) at sigma[seg_off(s) + p*ld + c].
 *   seg s = b*n_slice + j  ->  seg_off = b*batch_stride + j*slice_stride
 * thr_out[s] receives the fp32 threshold (NaN if the segment holds a NaN, +inf/-inf
 * conventions for pr==0 / pr>=10: mask all 0 / all 1).  mask_out has the layout of sigma
 * (its own ld / strides) and receives 0.0f / 1.0f. pr is the reference's `pr` (0..10+). */
int vam_variance_mask(const float* sigma, int ld, long batch_stride, long slice_stride,
                      int n_batch, int n_slice, int n_pix, int C, double pr,
                      float* mask_out, int ld_mask, long mask_batch_stride, long mask_slice_stride,
                      float* thr_out, void* stream);

/* ------------------------------------------------------------------ Gaussian conditional */
/* Fused slice tail (models/pic.py:545-546,625-629; entropy_models.py:620-652).
 *  base  (mask == NULL):  v = round(y-mu);           lik = L(|(v+mu)-mu|, sigma);        yhat = v+mu
 *  prog  (mask != NULL):  r = y - ybase_raw; v = round(r-mu); in = (r-mu)*m;
 *                         lik = L(|round(in)|, sigma*m);   yhat = v*m + mu
 * Each tensor is a [n_pix, C] channel window with its own pixel stride. y2 (may be NULL)
 * is subtracted from y first (delta_encode, pic.py:583-584).
 * log2sum (may be NULL): per-batch-item sum of log2(lik) accumulated atomically
 * (pix_per_item pixels per item) — cleared by the caller.
 * sym (may be NULL): int32 quantised symbols (v, or v*m) for the entropy coder. */
int vam_gauss_tail(const float* y, int ld_y, const float* y2, int ld_y2,
                   const float* mu, int ld_mu, const float* sigma, int ld_sigma,
                   const float* mask, int ld_mask,
                   float* yhat, int ld_yhat, float* lik, int ld_lik,
                   int32_t* sym, int ld_sym, double* log2sum, int pix_per_item,
                   long n_pix, int C, void* stream);

/* GaussianConditional.build_indexes (entropy_models.py:654-659): idx = 63 - #{i<63: max(s,.11) <= T_i}
 * table: 64 floats (device). mask (may be NULL) multiplies sigma first (pic.py:809). */
int vam_build_indexes(const float* sigma, int ld_sigma, const float* mask, int ld_mask,
                      const float* table, int n_table, int32_t* idx, int ld_idx,
                      long n_pix, int C, void* stream);

/* EntropyBottleneck eval forward (entropy_models.py:403-436,449-492) on z NHWC [n_pix, C]:
 * zhat = round(z-med)+med ; sym (may be NULL) = round(z-med) ; lik = |sigmoid(s*upper)-sigmoid(s*lower)| clamped at 1e-9.
 * params: the 15 tensors _matrix0.._4,_bias0.._4,_factor0.._3 and quantiles as stored in the
 * state_dict, concatenated per tensor (see INTEGRATION.md for the order), C channels. */
int vam_eb_forward(const float* z, int ld_z, const float* params, int C,
                   float* zhat, int ld_zhat, float* lik, int ld_lik, int32_t* sym, int ld_sym,
                   double* log2sum, int pix_per_item, long n_pix, void* stream);
/* Training variant: z_hat is still round(z-med)+med (pic.py:282-284) but the likelihood is evaluated at
 * z + noise (additive-uniform proxy, entropy_models.py:132-138,471-473).  noise == NULL: identical to
 * vam_eb_forward. */
int vam_eb_forward_noise(const float* z, int ld_z, const float* params, int C, float* zhat, int ld_zhat,
                         float* lik, int ld_lik, int32_t* sym, int ld_sym, double* log2sum, int pix_per_item,
                         long n_pix, const float* noise, int ld_noise, void* stream);
/* EntropyBottleneck.loss (entropy_models.py:398-401; models/base.py:22-29 aux_loss): loss[0] (device double) =
 * sum_{c,k} |logits_cumulative(quantiles[c,k]) - target[k]| with the density network held constant (stop_gradient),
 * dquantiles[c*3+k] = d loss / d quantiles[c,k].  params as for vam_eb_forward; target3_host: 3 HOST floats. */
int vam_eb_aux_loss(const float* params, int C, const float* target3_host, double* loss, float* dquantiles, void* stream);
/* EntropyModel.dequantize (entropy_models.py:161-168): out = float(sym) + mu (mu may be NULL). */
int vam_dequantize(const int32_t* sym, int ld_sym, const float* mu, int ld_mu, float* out, int ld_out,
                   long n_pix, int C, void* stream);

/* out[p, c] = a[p, c] + b[p, c]   (channel windows) — mu_total = mu + yhat_base (pic.py:603) */
int vam_add(const float* a, int ld_a, const float* b, int ld_b, float* out, int ld_out,
            long n_pix, int C, void* stream);
/* Zero-fill KERNEL on the stream (ptr and bytes multiples of 4): accumulators are cleared inside the captured graph, where
 * every node is a kernel node */
int vam_memset_zero(void* ptr, size_t bytes, void* stream);
/* sum((a-b)^2) accumulated in double into acc[0] (PSNR, utility/functions.py:172-174) */
int vam_sqdiff_sum(const float* a, const float* b, long n, double* acc, void* stream);

/* ------------------------------------------------------------------ MS-SSIM pieces (utility/functions.py:176-177) */
/* One SSIM level over `planes` contiguous HxW planes (NCHW): 11-tap Gaussian window `win11` applied to x, y, x^2, y^2, xy
 * at the (H-10)x(W-10) valid positions; ADDS the per-plane sums of the ssim map and of the cs map to ssim_sum / cs_sum
 * (device doubles, one per plane; zero them first).  c1 = (0.01 L)^2, c2 = (0.03 L)^2. */
int vam_ssim_level(const float* x, const float* y, int planes, int H, int W, const float* win11, float c1, float c2,
                   double* ssim_sum, double* cs_sum, void* stream);
/* F.avg_pool2d(kernel 2, stride 2, padding (pad_h, pad_w) in {0,1}, padded zeros counted): out is
 * [planes][(H+2ph-2)/2+1][(W+2pw-2)/2+1]. */
int vam_avgpool2(const float* x, float* out, int planes, int H, int W, int pad_h, int pad_w, void* stream);

/* ------------------------------------------------------------------ REM fine-tune backward (configs[4]) */
/* Weight (and bias) gradient of a stride-1, pad k/2 convolution (autograd's conv backward-weight for
 * layers/rem.py:40-49):  dw[n][c_off + c][ty][tx] = sum_p dy[p][n] * x[pix(p)+(ty-k/2, tx-k/2)][c]  in OIHW with
 * `cin_total` input channels; x is a C-channel window (one segment of a concatenated input lands at c_off).
 * db (may be NULL; honoured by the problem with c_off == 0) = sum_p dy[p][n].  Fixed summation order.
 * Up to VAM_MAX_WGRAD_GROUP independent problems run in one launch (the ten slices' REM blocks in lockstep). */
#define VAM_MAX_WGRAD_GROUP 16
typedef struct vam_wgrad {
  const float* x;
  const float* dy;
  float* dw;
  float* db;
  int ld_x, ld_dy;
  int B, H, W;        /* extent of dy (the convolution's output grid)                                          */
  int kh, kw;
  int C, N;
  int cin_total, c_off;
  int stride;         /* 0 or 1: stride 1, pad k/2, x has the extent of dy.  2: a stride-2, pad k/2 convolution (k5: g_a,  */
  int Hx, Wx;         /* k3: h_a) — x is [B, Hx, Wx] (Hx = 2H, Wx = 2W): input pixel = 2*o - k/2 + tap.  Also the   */
                      /* transposed convolutions of g_s: their weight gradient is this with x and dy exchanged      */
  int splits;         /* 0 / 1: one block per weight tile walks all pixels.  > 1 (from vam_conv_wgrad_plan): the      */
  float* workspace;   /* pixels are cut into `splits` ranges whose partial tiles go to this caller-owned buffer and   */
                      /* are added in range order by a second launch (layers with few weight tiles and many pixels)  */
  float slot_share;   /* planning hint (vam_conv_wgrad_plan only): the fraction of the chip this problem can count on  */
                      /* when it shares a grouped launch with others (its share of the group's FLOPs); 0 = alone       */
  int32_t flags;      /* VAM_WGRAD_X_P3: x is a bf16x3 plane tensor (VAM_CONV_OUT_BF3 layout; ld_x counts 8-channel     */
                      /* groups per pixel) — k3 stride-1 problems on grids vam_conv_wgrad_lds_grid() accepts only      */
} vam_wgrad;
#define VAM_WGRAD_X_P3 1
/* 1 when weight gradients on an H x W output grid take the LDS-tiled kernel (which alone reads plane tensors). */
int vam_conv_wgrad_lds_grid(int H, int W);
/* Suggested number of pixel splits for a problem (>= 1) and the workspace it needs (0 bytes when 1). */
int vam_conv_wgrad_plan(const vam_wgrad* problem, size_t* workspace_bytes);
int vam_conv_wgrad_group(const vam_wgrad* problems, int n_problems, void* stream);
int vam_conv_wgrad(const float* x, int ld_x, const float* dy, int ld_dy, int B, int H, int W, int kh, int kw,
                   int C, int N, float* dw_oihw, float* db, int cin_total, int c_off, void* stream);
/* out[n] = sum_p dy[p][n]  (bias gradient): pixel ranges in parallel into `workspace` (vam_colsum_workspace bytes,
 * caller-owned), then added in range order — deterministic. */
size_t vam_colsum_workspace(long n_pix, int N);
int vam_colsum(const float* dy, int ld, long n_pix, int N, float* out, float* workspace, void* stream);
/* dx = dy * (act > 0 ? 1 : 0.01): LeakyReLU(0.01) backward from its OUTPUT (same sign as its input) */
int vam_leaky_bwd(const float* act, int ld_a, const float* dy, int ld_dy, float* dx, int ld_dx, long n_pix, int C, void* stream);
int vam_mul(const float* a, int ld_a, const float* b, int ld_b, float* out, int ld_out, long n_pix, int C, void* stream);
/* Training-mode Gaussian likelihood (entropy_models.py:132-138,620-652; pic.py:437-442): v = ((y-y2)-mu)*m + noise,
 * s = max(sigma*m, .11), lik = max(Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), 1e-9).  grad_lik == NULL: forward (writes lik).
 * grad_lik != NULL: backward (writes dmu, dsigma = grad_lik * dlik/d{mu,sigma}, with compressai's LowerBound
 * gradient rule on both bounds).  y2 / mask may be NULL. */
int vam_gauss_train(const float* y, int ld_y, const float* y2, int ld_y2, const float* mu, int ld_mu,
                    const float* sigma, int ld_sigma, const float* mask, int ld_mask, const float* noise, int ld_noise,
                    const float* grad_lik, int ld_glik, float* lik, int ld_lik, float* dmu, int ld_dmu,
                    float* dsigma, int ld_dsigma, long n_pix, int C, void* stream);

/* ------------------------------------------------------------------ transform backward (refine_gs) */
/* Element-wise derivatives of the synthesis / analysis transforms (csrc/train_gs.hip).  Every tensor is a channel
 * window [n_pix, C] with its own pixel stride. */
enum vam_ew_op {
  VAM_EW_GELU_FWD = 0,      /* out0 = gelu(in0)                                          (layers/layers.py:36-40) */
  VAM_EW_GELU_BWD = 1,      /* in0 = pre-activation, in1 = dy: out0 = dy * gelu'(in0)                                 */
  VAM_EW_GATE_BWD = 2,      /* a*sigmoid(b)+x (layers.py:72-74): in0 = a, in1 = b, in2 = dout: out0 = da, out1 = db    */
  VAM_EW_GDN_APPLY = 3,     /* in0 = x, in1 = norm: out0 = x*sqrt(norm) (flag 1, IGDN) or x*rsqrt(norm) (gdn.py:70-72) */
  VAM_EW_GDN_BWD_PREP = 4,  /* in0 = x, in1 = norm, in2 = dy: out0 = dL/dnorm, out1 = dy * dy/dx at fixed norm, out2 = x^2 */
  VAM_EW_GDN_BWD_FIN = 5,   /* in0 = out1 above, in1 = x, in2 = gamma^T dL/dnorm: out0 = in0 + 2 x in2                  */
  VAM_EW_CLAMP_BWD = 6,     /* in0 = clamp_(v,0,1), in1 = dL/dout: out0 = dL/dv                      (pic.py:558,651) */
  VAM_EW_AXPY = 7,          /* out0 = in0 + coef * in1                                                                 */
  VAM_EW_GATE_FWD = 8,      /* in0 = a, in1 = b, in2 = x: out0 = a*sigmoid(b) + x                                      */
  VAM_EW_REPARAM_BWD = 9,   /* NonNegativeParametrizer backward: in0 = stored parameter, in1 = dL/dvalue, coef = bound */
  VAM_EW_HTANH_FWD = 10,    /* LRP tail (pic.py:635-641): in0 = z, in1 = quantised residual, in2 = base: (0.5 tanh z + in1) + in2 */
  VAM_EW_HTANH_BWD = 11,    /* in0 = z, in1 = dy: out0 = dy * 0.5 (1 - tanh(z)^2)                                         */
  VAM_EW_MASK_SPLIT = 12    /* straight-through rounding under the variance mask (pic.py:443: ste_round(r - mu) * m + mu):
                               in0 = dL/d(result), in1 = m: out0 = in0 * m (to r), out1 = in0 * (1 - m) (to mu)           */
};
typedef struct vam_ew {
  vam_aux in[4];
  vam_aux out[3];
  long n_pix;
  int32_t C;
  int32_t flag;
  float coef;
  int32_t pad_;
} vam_ew;
/* The last two layers of a slice stack — conv3x3(128 -> 64) + GELU, conv3x3(64 -> 32) with the epilogue
 * out = post2 + post + act(conv + bias), act = VAM_ACT_NONE or VAM_ACT_HALF_TANH — as ONE launch per VAM_MAX_TAIL_GROUP stacks
 * (reference: the tails of the Sequentials cc_mean_transforms[_prog] / cc_scale_transforms[_prog] / lrp_transforms[_prog],
 * models/pic.py:86-121, 550, 635-641).  x = the 128-channel input as bf16x3 planes (what a conv launch with VAM_CONV_OUT_BF3
 * writes; x_groups = 8-channel groups per pixel of that buffer: 16, the tensor is not a window of a wider one), w4 / b4 / w5 / b5 = the packed weights and bias of the two layers
 * as vam_conv takes them; out fp32 NHWC with row pitch ld_out.  Same bits as the two vam_conv_group launches.  Built for latents
 * 16 columns wide (W == 16, H a multiple of 4): the caller falls back to vam_conv_group otherwise. */
#define VAM_MAX_TAIL_GROUP 8
typedef struct vam_stack_tail {
  const void* x;
  const void* w4;
  const float* b4;
  const void* w5;
  const float* b5;
  float* out;
  vam_aux post, post2;
  int32_t B, H, W;
  int32_t x_groups;
  int32_t ld_out;
  int32_t act;
} vam_stack_tail;
int vam_stack_tail_group(const vam_stack_tail* probs, int n, void* stream);
int vam_train_elementwise(int op, const vam_ew* e, void* stream);
/* n <= VAM_MAX_EW_GROUP independent VAM_EW_AXPY updates (out0 = in0 + coef * in1, each job with its own windows, extent and
 * coefficient) in ONE launch: the backward of `torch.cat` in front of a slice stack (pic.py:407-408, 452-453: the first layer's
 * input gradient is added, channel range by channel range, into the accumulators of the concatenated tensors).  Windows WRITTEN
 * by different jobs of one call must not overlap. */
#define VAM_MAX_EW_GROUP 8
int vam_train_axpy_group(const vam_ew* jobs, int n, void* stream);
/* Backward of vam_win_attention: dqkv [B,H,W,3C] (dq | dk | dv, same layout as qkv) from dout = dL/d(attention output),
 * and the relative-position-bias gradient WRITTEN to dtable [(2ws-1)^2][heads].  Deterministic: every (window, head group)
 * block writes its partial table into `workspace` (vam_win_attention_bwd_workspace bytes, caller-owned) and a second
 * launch adds the rows in a fixed order — no float atomics in global memory. */
size_t vam_win_attention_bwd_workspace(int B, int H, int W, int heads, int ws);
int vam_win_attention_bwd(const float* qkv, int ld_qkv, const float* dout, int ld_do, float* dqkv, int ld_dq,
                          const float* table, float* dtable, float* workspace, int B, int H, int W, int C, int heads, int ws,
                          int shift, void* stream);

/* ------------------------------------------------------------------ first-stage training (configs[3]) */
/* Backward of the training-mode entropy bottleneck (entropy_models.py:403-436,449-492 with quantize "noise"): for
 * lik = max(|sigmoid(s u) - sigmoid(s l)|, 1e-9), u / l = logits_cumulative(z + noise +/- 0.5), s = -sign(l + u) detached:
 * dz = grad_lik * dlik/dz (written, not accumulated) and dparams (layout of `params`, see vam_eb_forward; the quantiles'
 * entries are zero: the noise likelihood does not read them) = sum over pixels, fixed order.  LowerBound rule on 1e-9. */
int vam_eb_train_bwd(const float* z, int ld_z, const float* noise, int ld_noise, const float* params, int C,
                     const float* grad_lik, int ld_glik, float* dz, int ld_dz, float* dparams, long n_pix, void* stream);
/* Gradient of PixelShuffle(2) (layers/layers.py:82-86): dst[b,y,x,c*4+i*2+j] = src[b,2y+i,2x+j,c]; H, W = extent of dst. */
int vam_ps2_unshuffle(const float* src, int ld_src, float* dst, int ld_dst, int B, int H, int W, int Cq, void* stream);
/* dst[b,2y,2x,:] = src[b,y,x,:], zeros elsewhere (H, W = extent of src): with the VAM_PACK_CONV_DGRAD problem on dst this is
 * the data gradient of a k3 / stride-2 / pad-1 convolution (h_a, models/builder.py:72-82). */
int vam_upsample2_zero(const float* src, int ld_src, float* dst, int ld_dst, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------ bitstream (HOST pointers) */
/* compressai `_CXX.pmf_to_quantized_cdf` (reference entropy_models.py:61-64): n probabilities ->
 * n+1 cumulative counts at `precision` bits, every symbol given a non-zero frequency. */
int vam_pmf_to_quantized_cdf(const float* pmf_host, int n, int precision, int32_t* cdf_out_host);
/* compressai `ans.RansEncoder.encode_with_indexes` / `RansDecoder.decode_with_indexes`
 * (reference entropy_models.py:231-239,280-290): symbols[i] coded with table indexes[i];
 * cdfs is [n_cdfs][cdf_stride], cdf_sizes[k] = valid entries of table k (pmf length + 2), offsets[k]
 * = symbol value of entry 0; out-of-range symbols use 4-bit bypass chunks.  16-bit precision.
 * encode returns the stream length in bytes (<0 on error). */
long vam_rans_encode(const int32_t* symbols_host, const int32_t* indexes_host, long n, const int32_t* cdfs_host,
                     int cdf_stride, const int32_t* cdf_sizes_host, const int32_t* offsets_host, int n_cdfs,
                     uint8_t* out_host, long out_capacity);
int vam_rans_decode(const uint8_t* in_host, long n_bytes, const int32_t* indexes_host, long n,
                    const int32_t* cdfs_host, int cdf_stride, const int32_t* cdf_sizes_host,
                    const int32_t* offsets_host, int n_cdfs, int32_t* symbols_out_host);

/* ------------------------------------------------------------------ graphs / timing */
int vam_graph_begin(void* stream);
int vam_graph_end(void* stream, void** graph_exec_out);
int vam_graph_launch(void* graph_exec, void* stream);
/* The caller guarantees that no launch of graph_exec is in flight and that the calling thread is not capturing. */
int vam_graph_destroy(void* graph_exec);

/* Per-kernel-family timing with HIP events on the launch stream (bench.py roofline leg).
 * While enabled every launch is bracketed by an event pair; vam_prof_read synchronises the
 * events and returns accumulated milliseconds / launch count / algorithmic flops & bytes. */
enum vam_family { VAM_FAM_CONV = 0, VAM_FAM_ATTN = 1, VAM_FAM_MASK = 2, VAM_FAM_TAIL = 3,
                  VAM_FAM_MISC = 4, VAM_FAM_COUNT = 5 };
int vam_prof_enable(int on);
int vam_prof_reset(void);
int vam_prof_read(int family, double* ms, long* launches, double* flops, double* bytes);
/* Convolution launches are also summed per caller-defined class (which part of the model the launch belongs to:
 * g_a, g_s, hyperprior, stack heads, slice chain ... — the host plan sets the class before each launch while the
 * profiler is on), so that bench.py can report the roofline of the g_a/g_s conv stack on its own. */
#define VAM_PROF_CLASSES 16
int vam_prof_set_class(int cls);
int vam_prof_read_class(int cls, double* ms, long* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* VAMPIC_H */

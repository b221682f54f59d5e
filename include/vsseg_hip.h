/* vsseg_hip.h — C ABI of libvsseg_hip.so: the MI355X (gfx950) hot path of the VS_Seg 2.5D attention U-Net.
 *
 * The reference (KCL-BMEIS/VS_Seg) has no FFI of its own: its seam is the Python object protocol between
 * `VSparams` and the model/loss objects.  This library sits *behind* that protocol; every entry
 * point below replaces the torch/ATen/cuDNN work the reference dispatches from the cited lines.  Plain pointers and
 * sizes only, no torch types; the caller owns every buffer; all calls are asynchronous on `stream` (a hipStream_t)
 * and return 0 or a negative VSSEG_E* code (text via vsseg_last_error()).
 *
 * Data layout: activations are channels-last [N][X][Y][Z][C] ("NDHWC", identical to torch.channels_last_3d strides
 * of a [N,C,X,Y,Z] tensor) described by vsseg_tensor; `pitch` lets a tensor be a channel slice of a wider buffer, and
 * `ptr2/csplit` let it be the concatenation of two dense tensors, so that the skip-connection concat (MONAI
 * SkipConnection, ref:params/networks/nets/unet2d5_spvPA.py:89) is never materialised by a copy.
 */
#ifndef VSSEG_HIP_H
#define VSSEG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VSSEG_F32 0
#define VSSEG_BF16 1

#define VSSEG_OK 0
#define VSSEG_EINVAL (-1)
#define VSSEG_ELAUNCH (-2)

#define VSSEG_ACT_NONE 0
#define VSSEG_ACT_PRELU 1
#define VSSEG_ACT_RELU 2
#define VSSEG_ACT_SIGMOID 3

#define VSSEG_RES_NONE 0
#define VSSEG_RES_ADD 1      /* out = f(acc) + res                      (ResidualUnit add, ref:.../convolutions.py:252-255) */
#define VSSEG_RES_RELUMASK 2 /* out = acc * (res > 0)                   (backward of the attention ReLU)                   */
#define VSSEG_RES_GATE 3     /* out = acc + res * (1 + gate[voxel])     (backward of AttentionBlock2 `att*x + x`, ref:.../attentionblock.py:43-47, fused
                              *  into the data gradient of the attention branch's first convolution: both flow into d(x))                          */

#define VSSEG_RES_IN1 5      /* out = f(acc) + in1[voxel] * in1_w[c] + in1_b[c]: a 1x1x1 convolution of a ONE-channel tensor added behind the activation (the first
                              *  ResidualUnit's residual convolution of the network input, ref:.../convolutions.py:241-255 with in_channels = 1): marching kernel only */

#define VSSEG_SEED_INDIRECT 0x80000000u /* OR-ed into a dropout `salt`: the `seed` argument is then the DEVICE ADDRESS of the 64-bit seed */

#define VSSEG_MAX_TAPS 27
#define VSSEG_STAT_SHARDS 256

typedef struct {
  void* ptr;     /* first element of the view (channel offset already applied) */
  int32_t dtype; /* VSSEG_F32 | VSSEG_BF16 */
  int32_t c;     /* channels in the view */
  int32_t pitch; /* elements between consecutive voxels (>= c, or >= max(csplit, c - csplit) for a two-part tensor) */
  int32_t n, x, y, z;
  /* Two-part tensors (the skip-connection concat, ref:params/networks/nets/unet2d5_spvPA.py:89 / MONAI SkipConnection):
   * channels [0, csplit) live at ptr, channels [csplit, c) at ptr2, both with the same pitch — the concat is the pair of
   * its (dense) operands and is never materialised.  ptr2 == NULL: ordinary tensor.  csplit must be a multiple of 16.
   * Accepted by vsseg_igemm (in, out, res), vsseg_wgrad (h) and vsseg_att_apply_fwd/bwd (x, dx); every other entry point
   * rejects it with VSSEG_EINVAL. */
  void* ptr2;
  int32_t csplit;
  int32_t reserved;  /* 0, or VSSEG_ZERO_PADDED on a destination: its channels c .. pitch-1 are zero padding the call may rewrite with zeros */
} vsseg_tensor;
#define VSSEG_ZERO_PADDED 1

/* One implicit-GEMM launch over an output lattice q in [0,q): out[q*os+oo][n] = epi( sum_t sum_c in[q*is+off_t][c] * W[t][c][n] ).
 * Covers Conv3d forward, every parity class of ConvTranspose3d forward, and both data-gradients
 * (ref:params/networks/blocks/convolutions.py:114-146 and their autograd at ref:params/VSparams.py:461). */
typedef struct {
  vsseg_tensor in, out;
  int32_t q[3];            /* lattice extent */
  int32_t is[3], os[3], oo[3];
  int32_t ntaps;
  int32_t tap_off[VSSEG_MAX_TAPS][3]; /* input offset of each tap relative to q*is */
  int32_t tile[3];         /* output-lattice tile per workgroup; product must be 64*mtw */
  int32_t mtw;             /* 16-voxel M-tiles per wave: 1, 2 or 4 */
  int32_t nt;              /* 16-channel output tiles per workgroup (1..6) */
  int32_t nsplit;          /* output-channel splits (grid.y); packed weights hold nsplit*nt tiles */
  int32_t ck;              /* input channels staged per chunk (multiple of 8, divides padded Cin) */
  int32_t nchunks;
  int32_t ksteps;          /* K-steps (4 groups of 8 channels) per chunk */
  int32_t depth;           /* LDS-DMA prefetch distance in stages (1..3; 0 = 1): the halo ring holds depth+1 buffers.  -1: no prefetch, one buffer (half the LDS, more resident workgroups) */
                           /* negative depths below -1 select a specialised kernel with the same contract (outside its domain: VSSEG_EINVAL, never a fallback): -2 / -4 streaming (sconv.hip), -3 compute (cconv.hip),
                            * -5 / -6 marching (mconv.hip), -7 deep levels (dconv.hip), -8 the level 2 <-> 3 transition kernel (tconv.hip: class_split = 8 parity classes of a 3x3x3 stride-(2,2,2) transposed
                            * convolution / strided data gradient, 48 or 64 input and 48 output channels, tile 4x8x8 with mtw = 16, nt = 3, ck = in.c, bf16, plain / statistics / eval affine / accumulate epilogue),
                            * -9 the gathering marching kernel (gconv.hip: is = (2, 2, 1) 3x3x1 launches that read the fine level and write the coarse one — strided convolutions 16 -> 16 / 32 -> 32, data
                            * gradients of the transposed convolutions 32 -> 16 / 48 -> 32 —, tile = (x steps, 64 * mtw / tz rows, tz), ck = in.c in {16, 32}, one-part bf16 tensors, the same four epilogues) */
  const void* wpack;       /* [nsplit][nchunks][ksteps][nt][64 lanes][8] in the compute dtype (== in.dtype) */
  /* epilogue: v = acc + bias; stats(v); v = v*scale+shift; v = act(v); residual; accumulate; store */
  const float* bias;       /* [cout] or NULL */
  const float* bias2;      /* second bias added to the first (merged 1x1x1 residual convolution) or NULL */
  const float* scale;      /* [cout] or NULL (eval-mode BatchNorm folded) */
  const float* shift;
  const float* alpha;      /* device pointer to the PReLU slope (1 element) */
  int32_t act;
  int32_t res_mode;
  vsseg_tensor res;
  int32_t accumulate;      /* out += value (gradient accumulation at fan-out points) */
  double* stats;           /* [VSSEG_STAT_SHARDS][2][cout_padded] sum / sum-of-squares of v, or NULL */
  int32_t stats_stride;    /* cout_padded */
  /* z-folded launches (a convolution without taps along z, with 1 real input or output channel, run on tensors whose 8
   * z-neighbours are reinterpreted as 8 channels; the packed weights are block-diagonal): output channel c of the launch is
   * real channel c % cout_mod for bias / bias2 / scale / shift / stats.  0: off. */
  int32_t cout_mod;
  const float* gate;       /* VSSEG_RES_GATE: fp32 attention map [N][X][Y][Z] of the output tensor, or NULL */
  const float* in_gate;    /* fp32 attention map [N][X][Y][Z] of the INPUT tensor, or NULL: input voxel v is multiplied by (1 + in_gate[v]) on load
                            * (AttentionBlock2 folded into the convolution that reads its output; marching kernel, depth -5, only) */
  /* All output-parity classes of a strided transposed convolution / data gradient in ONE launch of the general kernel (class_split = number of
   * classes, 2..8; 0 = off): `nsplit` = class_split, workgroup row s of the grid computes class s — ALL output channels (out.c = nt*16) at output
   * voxels q*os + class_oo[s] — from the class's own taps class_tap[s][0..class_ntaps[s]) (indices into tap_off, which lists the union of the classes'
   * input offsets; oo must be 0).  wpack is [class][nchunks][ksteps][nt][64][8] with ksteps = the largest class's; a class runs only its own K-steps.
   * One read of the halo table, one launch and one weight staging per class instead of 8 launches of 15-40 us on levels 3-5. */
  int32_t class_split;
  int32_t class_oo[8][3];
  int32_t class_ntaps[8];
  int32_t class_tap[8][8];
  /* A 1x1x1 convolution of the SAME input riding along (the ResidualUnit's residual convolution, ref:params/networks/blocks/convolutions.py:241-255; marching kernel,
   * depth -5 / -6, plain or statistics epilogue only): res_tiles more 16-channel output tiles that run only the K-steps of the centre tap; the input is read once
   * for both convolutions.  res_out.ptr != NULL: their values (+ bias_res) are stored there (bf16) — training, where the add sits behind the BatchNorm pass;
   * res_out.ptr == NULL: they are added to the main tiles behind their activation — eval: out = act(bn(conv(x))) + residual(x).  0: off. */
  int32_t res_tiles;
  const void* wpack_res;   /* [K-steps of the centre tap][res_tiles][64 lanes][8] in the compute dtype (planner.residual_tile_pack_map) */
  const float* bias_res;   /* [res_out.c] or NULL */
  vsseg_tensor res_out;
  const void* in1;         /* VSSEG_RES_IN1: one-channel tensor [N][X][Y][Z] in the compute dtype (bf16) */
  const float* in1_w;      /* [cout] weights and */
  const float* in1_b;      /* [cout] bias of the 1 -> cout convolution */
} vsseg_igemm_desc;

/* Weight gradient: dW[t][cP][cH] += sum_q P[q][cP] * H[q*hs + off_t][cH]  (fp32 atomics into the flat grad buffer).
 * Conv3d: P = dY, H = X.  ConvTranspose3d: P = X, H = dY. */
typedef struct {
  vsseg_tensor p, h;
  int32_t q[3];            /* lattice = spatial extent of P */
  int32_t hs[3];
  int32_t ntaps;
  int32_t tap_off[VSSEG_MAX_TAPS][3];
  int32_t tap_widx[VSSEG_MAX_TAPS]; /* flat index of the tap inside the weight's kernel dims */
  int32_t tile[3];         /* product must be a multiple of 32 */
  int32_t ntp;             /* 16-channel tiles of P */
  float* dw;               /* destination weight-gradient tensor (pre-zeroed or holding earlier contributions) */
  int64_t stride_p, stride_h, stride_tap; /* element strides of dw along cP, cH, tap */
  int32_t cp_valid, ch_valid;
  int32_t persistent_blocks;
  float* scratch;          /* partial-sum slabs: >= ceil(ch_valid/16) * ntaps * ntp*16 * 16 floats per workgroup */
  int64_t scratch_elems;   /* capacity in floats; the number of persistent workgroups is clamped to what fits */
  int32_t single_buffer;   /* 1: no prefetch, one LDS tile buffer (half the LDS, more resident workgroups); 0: double-buffered */
  int32_t hgroup;          /* 16-channel chunks of H one workgroup multiplies with the P tile it fetched (1..4; 0 = 1): P is read ceil(chunks/hgroup) times.
                            * Clamped by the library to a divisor of the chunk count that fits registers / LDS. */
  float* dbias_p;          /* optional: dbias_p[cP] += sum_q P[q][cP] (bias gradient of a convolution without BatchNorm, P = dY) or NULL */
  int32_t march;           /* 1: the marching kernel (csrc/mwgrad.hip; stride-1 3x3x1 bf16 only, outside its domain is an error): tile = (x steps per
                            * workgroup, rows per workgroup, z slices per workgroup in {2, 4, 8}); persistent_blocks / single_buffer / hgroup unused */
  const float* h_gate;     /* march = 1 only: fp32 attention map [N][X][Y][Z] of H, or NULL: H[v] is multiplied by (1 + h_gate[v]) on load (the gated
                            * tensor `att.repeat(C) * x + x` of AttentionBlock2 as the convolution input, never materialised) */
} vsseg_wgrad_desc;

/* Fused backward of one stride-1 3x3x1 Convolution block, Conv3d -> BatchNorm3d(train) -> Dropout -> PReLU (ref:params/networks/blocks/convolutions.py:114-156,
 * differentiated by loss.backward() at ref:params/VSparams.py:461): the second BatchNorm-backward pass (what vsseg_bn_act_bwd_apply computes) is applied ON LOAD,
 * and the data gradient and the weight gradient of the convolution are formed from the same LDS planes in ONE marching launch (csrc/mbwd.hip) — the
 * gradient of the convolution output is never written to HBM.  Call after vsseg_bn_act_bwd_reduce + vsseg_bn_act_bwd_finalize of the layer.  bf16 only;
 * outside its instantiated shapes the call fails with VSSEG_EINVAL (no fallback). */
typedef struct {
  vsseg_tensor y, dout;    /* convolution output before BatchNorm, gradient of the block output: [N][X][Y][Z][cout] */
  vsseg_tensor x;          /* convolution input [N][X][Y][Z][cin] */
  vsseg_tensor dx;         /* OUT: data gradient of x (overwritten) */
  const float *mean, *invstd, *gamma, *scale, *shift, *alpha; /* as for vsseg_bn_act_bwd_apply */
  const float *mean_dz, *mean_dzx;
  float p_drop;
  const uint8_t* keep;     /* keep-mask bytes written by vsseg_bn_act_fwd (required when p_drop > 0) */
  const void* wpack;       /* packed weights of the data gradient: [ksteps][nt][64 lanes][8] bf16 of the marching conv_dgrad plan (K = 9 * cout -> N = cin) */
  float* dw;               /* IN/OUT: weight gradient [cout][cin][3][3][1] fp32, += */
  int32_t tile[3];         /* (x steps per workgroup, rows per workgroup, z slices per workgroup in {2, 4, 8}); rows * z a multiple of 64 */
  float* scratch;          /* partial-sum slabs: (cout/16) * (9, or 10 with a residual convolution) * cin * 16 floats per workgroup */
  int64_t scratch_elems;
  /* Optional: the ResidualUnit's 1x1x1 residual convolution of the SAME input x (ref:params/networks/blocks/convolutions.py:241-255) rides along: dx also gets
   * Wr' dres, dw_res[cout][cin] += sum_q dres[q] x[q].  dres.ptr == NULL: no residual convolution.  dres may be the tensor `dout` itself (single-subunit units). */
  vsseg_tensor dres;       /* gradient of the residual convolution's output [N][X][Y][Z][cout] */
  const void* wpack_res;   /* packed weights of its data gradient: [ksteps][nt][64][8] bf16 of the conv_dgrad plan of the 1x1x1 convolution (K = cout -> N = cin, one chunk) */
  float* dw_res;           /* IN/OUT: its weight gradient [cout][cin] fp32, += */
  const float* x_gate;     /* optional fp32 attention map [N][X][Y][Z] of x: x[v] is multiplied by (1 + x_gate[v]) on load (AttentionBlock2 in front of the unit, never
                            * materialised; x may then be the two-part concat); dx is the gradient of the GATED tensor.  64 input channels with dres == dout only. */
} vsseg_conv_bwd_desc;
int vsseg_conv_bwd_fused(const vsseg_conv_bwd_desc* d, void* stream);

/* Two consecutive stride-1 3x3x1 convolutions of an INFERENCE forward as one launch (csrc/chain.hip; bf16):
 *   h   = act_a(scale_a * (conv_a(in) + bias_a) + shift_a)                         conv + eval-mode BatchNorm (folded) + PReLU, ref:params/networks/blocks/convolutions.py:114-146
 *   out = act_b(scale_b * (conv_b(h) + bias_b) + shift_b) [+ in * in1_w + in1_b]
 * h (cmid channels) lives in LDS only.  The pairs of the sliding-window predictor (ref:params/VSparams.py:553-567): the first ResidualUnit of the encoder, 1 -> 16 -> 16 with the
 * 1x1x1 residual convolution of the one-channel input added behind the second activation (ref:params/networks/blocks/convolutions.py:241-255), and the attention block of the
 * finest decoder level, 32 -> 16 -> 1 + sigmoid (ref:params/networks/blocks/attentionblock.py:20-41).  Results are bit-identical to the two vsseg_igemm marching launches
 * (depth -5) with the same packed weights.  With a BatchNorm in stage A it is not applicable in training (the BatchNorm needs the
 * statistics of all of h first). */
typedef struct {
  vsseg_tensor in;         /* bf16: a multiple of 8 channels (16-byte aligned voxel rows; may be two-part) or a COMPACT one-channel tensor (c = pitch = 1) standing for one zero-extended group */
  vsseg_tensor out;        /* same extent; bf16 with a multiple of 4 channels (<= 32), or 1..3 channels bf16 / fp32 (the attention map: c = 1, fp32) */
  int32_t cmid;            /* channels of h (16 or 32) */
  const void* wpack_a;     /* [K-steps of 9 taps x in channels][cmid / 16][64 lanes][8] bf16: the packed weights of conv_a's marching plan (planner.pack_map, nt = cmid / 16) */
  const float *bias_a, *scale_a, *shift_a; /* [cmid]; scale_a / shift_a NULL: 1 / 0 */
  const float* alpha_a;    /* device pointer to the PReLU slope of stage A */
  int32_t act_a;           /* VSSEG_ACT_NONE | VSSEG_ACT_PRELU | VSSEG_ACT_RELU */
  const void* wpack_b;     /* [K-steps of 9 taps x cmid][ceil(out.c / 16)][64 lanes][8] bf16 */
  const float *bias_b, *scale_b, *shift_b; /* [out.c] or NULL */
  const float* alpha_b;
  int32_t act_b;           /* VSSEG_ACT_* */
  const float *in1_w, *in1_b; /* [out.c] each or both NULL: + in[voxel] * in1_w[c] + in1_b[c] behind act_b (compact one-channel input only) */
  int32_t res_tiles;       /* 0, or ceil(out.c / 16): + bf16(residual(in) + bias_res) behind act_b, residual = a 1x1x1 convolution of `in` (the ResidualUnit's residual convolution,
                            * ref:params/networks/blocks/convolutions.py:241-255; ordinary multi-channel input only) */
  const void* wpack_res;   /* [K-steps of conv_a's centre tap][res_tiles][64 lanes][8] bf16 (planner.residual_tile_pack_map) */
  const float* bias_res;   /* [out.c] or NULL */
  int32_t tz;              /* plan: z voxels per workgroup column (1, 2, 4 or 8; in.z a multiple) */
  int32_t mtw;             /* ... 16-voxel M-tiles per wave: a workgroup owns ALL rows, in.y == waves * mtw * 16 / tz */
  int32_t lx;              /* ... x positions per workgroup (a segment re-fetches 4 input planes and recomputes 2 planes of h) */
  int32_t waves;           /* ... waves per workgroup: 4, 8 or 16 */
  int32_t lead;            /* ... iterations between the fetch of an input plane and its use: 1 (several small workgroups per CU) or 3 (one large one; always 3 for a compact input) */
} vsseg_chain_desc;
int vsseg_conv_chain(const vsseg_chain_desc* d, void* stream);
int vsseg_conv_chain_lds_bytes(const vsseg_chain_desc* d); /* LDS bytes of the launch, or VSSEG_EINVAL (vsseg_last_error: why the descriptor is outside the kernel's domain) */

const char* vsseg_last_error(void);
/* Forking a second stream without a marker packet on the first (ABI version 7).  Between vsseg_fork_arm(ev) and vsseg_fork_disarm() every kernel this library launches on the
   calling thread carries `ev` as the stop event of its own dispatch (a later kernel replaces an earlier one); vsseg_fork_disarm returns how many kernels carried it.
   vsseg_stream_wait_event(stream, ev) = hipStreamWaitEvent.  The training backward forks its weight-gradient stream this way: a hipEventRecord on the main stream delays the main
   stream's next kernel by ~5 us, 42 times per step. */
void* vsseg_fork_event_create(void);
int vsseg_fork_event_destroy(void* ev);
int vsseg_fork_arm(void* ev);
int vsseg_fork_disarm(void);
int vsseg_stream_wait_event(void* stream, void* ev);
int vsseg_version(void); /* 7: + vsseg_dice_pred_bwd_to, vsseg_dice_level_sums, vsseg_dice_tail_sums, vsseg_dice_att_bwd_levels, vsseg_fork_*, vsseg_stream_wait_event; 6: + launch plans with depth -8 (transition kernel) and -9 (gathering marching kernel); 5: vsseg_wgrad march = 2 (compute weight-gradient kernel), + vsseg_conv_to1, vsseg_conv_chain_desc without h_out (the training variant measured no gain and was deleted); 4: + vsseg_conv_chain; 3: the BatchNorm-block-on-load fields (in_bn_*, keep_*, x_bn_*: round-4 experiment, measured a loss, deleted) left the descriptors, depth -7 plans;
                            * 2: fixed-point accumulators documented + vsseg_fx_status; 1: the buffers below were described as plain doubles */

/* ---- Accumulator buffers are 64-bit FIXED-POINT integers, not doubles ------------------------------------------------------------------
 * Every `double*` accumulator of this ABI — vsseg_igemm_desc.stats and vsseg_bn_finalize's `stats`, the `sums` / `alpha_acc` of
 * vsseg_bn_act_bwd_reduce / _finalize, the `pred_sums` / `att_sums` of vsseg_dice_* — is declared `double*` for its size and alignment only:
 * each 8-byte slot holds a two's-complement int64 = round(value * scale), added to with integer atomics so that the total does not depend on
 * the order in which workgroups finish (a training step is run-to-run bit-identical).  Zero bytes are the value 0, so callers keep zero-filling
 * the buffers (vsseg_memset_zero) exactly as before; callers that READ or PRE-FILL a slot must use the scale of its family:
 *     VSSEG_FX_STAT_SCALE  2^20  BatchNorm forward statistics (sum, sum of squares)         per-workgroup partial |v| < 4.3e9
 *     VSSEG_FX_GRAD_SCALE  2^44  BatchNorm / PReLU backward sums (sums, alpha_acc)           per-workgroup partial |v| < 256
 *     VSSEG_FX_DICE_SCALE  2^32  Dice sums (pred_sums, att_sums)                              per-workgroup partial |v| < 1.0e6
 * (vsseg_hard_dice_counts keeps REAL doubles: its sums are integers, exact in fp64 in any order.)
 * A partial sum outside its range, or a NaN / Inf one, is clamped and sets a sticky per-process device flag; while the flag is set the decoding
 * kernels (vsseg_bn_finalize, vsseg_bn_act_bwd_finalize, vsseg_dice_finalize) return NaN, so a diverging run or an out-of-range loss scale
 * surfaces as NaN in the loss / statistics / gradients instead of as wrapped integers.  vsseg_fx_status reads (and optionally clears) the flag. */
typedef double vsseg_fx_acc; /* one accumulator slot (reinterpret as int64_t) */
#define VSSEG_FX_STAT_SCALE 1048576.0
#define VSSEG_FX_GRAD_SCALE 17592186044416.0
#define VSSEG_FX_DICE_SCALE 4294967296.0
static inline vsseg_fx_acc vsseg_fx_encode(double value, double scale) {
  union { int64_t i; double d; } u;
  double s = value * scale;
  u.i = (int64_t)(s < 0 ? s - 0.5 : s + 0.5);
  return u.d;
}
static inline double vsseg_fx_decode(vsseg_fx_acc slot, double scale) {
  union { int64_t i; double d; } u;
  u.d = slot;
  return (double)u.i / scale;
}
/* Returns 1 if a fixed-point partial sum was non-finite or out of range since the last reset, 0 if not, < 0 on error.  Synchronises `stream`.
 * reset != 0 clears the flag (after the caller has handled the diverged step). */
int vsseg_fx_status(int32_t reset, void* stream);

int vsseg_igemm(const vsseg_igemm_desc* d, void* stream);
int vsseg_igemm_lds_bytes(const vsseg_igemm_desc* d);
int vsseg_wgrad(const vsseg_wgrad_desc* d, void* stream);
/* Weight gradient of a 3x3x1 (k3 = 3) or 1x1x1 (k3 = 1) stride-1 convolution with ONE input or ONE output channel, as a bandwidth reduction
 * (no MFMA, no zero-extended operand):  dw[c*stride_c + tap] += sum_v t[v][c] * s[v + sign*off_tap],  zero padding.
 * t: the C-channel tensor (C = 8, 16, 32 or 64; y extent a multiple of 4), s: the one-channel field [N][X][Y][Z] in t's dtype.
 * Conv3d 1 -> C: t = dY, s = X, sign = +1.  Conv3d C -> 1: t = X, s = dY (its one real channel, compact), sign = -1.  stride_c = taps. */
int vsseg_wgrad_narrow(vsseg_tensor t, const void* s, int32_t k3, int32_t sign, float* dw, int64_t stride_c,
                       float* dbias /* sign = -1 only, or NULL: *dbias += sum_v s[v], the bias gradient of the C -> 1 convolution */,
                       float* scratch /* partial-sum slabs: >= k3*k3*C + 1 floats per workgroup */, int64_t scratch_elems, void* stream);
/* vsseg_wgrad_narrow for the 1 -> C, 3x3x1 convolution of a Convolution block (ref:params/networks/blocks/convolutions.py:114-156) with T = d(conv output) formed ON
 * LOAD from y (the convolution output), dout (the gradient of the block output) and the forward's keep-mask bytes — the second BatchNorm-backward pass, bit-identical
 * to vsseg_bn_act_bwd_apply, which is then not launched and d(conv output) is never written (the network input needs no data gradient).  bf16 only. */
int vsseg_wgrad_narrow_bn(vsseg_tensor y, vsseg_tensor dout, const uint8_t* keep, const float* mean, const float* invstd, const float* gamma, const float* scale, const float* shift, const float* alpha,
                          const float* mean_dz, const float* mean_dzx, float p_drop, const void* s, float* dw, int64_t stride_c, float* scratch, int64_t scratch_elems, void* stream);

/* Forward of a stride-1 3x3x1 convolution with ONE output channel (+ bias, + sigmoid): the second convolution of AttentionBlock1 on the two finest levels
 * (ref:params/networks/blocks/attentionblock.py:20-35: conv2 = Convolution(C/2 -> 1, conv_only) followed by Sigmoid).  A bandwidth kernel on the vector ALUs
 * (csrc/nconv.hip): every input voxel is read once, the output voxel's nine taps are partial sums exchanged between neighbouring threads.  in: one-part bf16,
 * 16 or 32 channels, y extent 16 .. 256 (a power of two; a workgroup owns all rows), z a multiple of 512 / y.  w: the fp32 master weights [1][C][3][3][1]
 * (rounded to bf16 inside, as the packed weights of the MFMA launches are).  out: dense one-channel fp32 / bf16 tensor of the input's extent.  lx: x planes per
 * workgroup (<= 0: all).  Outside this domain the call fails with VSSEG_EINVAL (no fallback: the caller lowers vsseg_igemm instead). */
int vsseg_conv_to1(vsseg_tensor in, const float* w, const float* bias /* [1] or NULL */, int32_t act /* VSSEG_ACT_NONE | VSSEG_ACT_SIGMOID */, vsseg_tensor out, int32_t lx, void* stream);

/* dst[i] = map[i] >= 0 ? cast(src[map[i]]) : 0 — (re)packs the fp32 master weights into MFMA fragment order. */
int vsseg_gather_cast(const float* src, const int32_t* map, const int32_t* map2 /* optional second addend, or NULL */, void* dst, int64_t n, int32_t dst_dtype, void* stream);
/* A 1x1x1 residual convolution merged into the centre tap of the k-tap convolution it is added to (last_conv_only ResidualUnit,
 * ref:params/networks/blocks/convolutions.py:217-255): its gradients are the centre-tap slice / the bias gradient of the merged conv. */
int vsseg_merge_residual_grads(const float* dw, const float* db, float* dwr, float* dbr, int32_t cout, int32_t cin, int32_t ktaps, int32_t centre, void* stream);

/* dst[n][x][y][z][0..cpad) <- src window (crop + zero-pad outside + zero-extend channels + cast). src is [N][SX][SY][SZ] f32, 1 channel.
 * Used for the network input (ref:params/VSparams.py:456) and for sliding-window crops (MONAI sliding_window_inference step 6). */
int vsseg_stage_input(const float* src, int32_t n, const int32_t sdims[3], const int32_t origin[3], vsseg_tensor dst, void* stream);

/* BatchNorm statistics: reduce shards -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; running-stat update
 * (torch BatchNorm3d training semantics, momentum 0.1, unbiased running var; ref:.../convolutions.py:152). */
int vsseg_bn_finalize(const double* stats, int32_t stride, int32_t c, double count, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, int64_t* num_batches, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale for every BN layer at once (flat param addressing). */
int vsseg_bn_fold_eval(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift, int32_t c, void* stream);

/* out = PReLU(dropout(y*scale+shift)) [+ res]   (ref:.../convolutions.py:148-156; residual add :252-255).
 * Dropout: keep-mask from Philox4x32-10(seed, salt, element index); p = 0 disables.  salt | VSSEG_SEED_INDIRECT: `seed` holds the
 * device address of the seed (every dropout entry point), so that a launch list with fixed arguments can be replayed with a new seed.
 * keep_out (or NULL): [voxels * c/8] bytes — the forward stores the keep-mask of every 8-channel group there, and a backward pass given the
 * same buffer as keep_in reads it instead of running Philox again (the three BN kernels are VALU-bound on the generator, not on HBM;
 * NULL regenerates the bit-identical mask from (seed, salt, element index)). */
int vsseg_bn_act_fwd(vsseg_tensor y, const float* scale, const float* shift, const float* alpha, float p_drop, uint64_t seed, uint32_t salt,
                     vsseg_tensor res, int32_t has_res, vsseg_tensor out, uint8_t* keep_out, void* stream);
/* Same with the residual computed on the fly as the 1x1x1 convolution of a ONE-channel tensor: res[v][c] = x1[v]*res_w[c] + res_b[c]
 * (first encoder ResidualUnit, in_channels = 1, ref:params/networks/blocks/convolutions.py:241-255); x1 is [N][X][Y][Z] in y's dtype. */
int vsseg_bn_act_fwd_res1(vsseg_tensor y, const float* scale, const float* shift, const float* alpha, float p_drop, uint64_t seed, uint32_t salt,
                          const void* x1, const float* res_w, const float* res_b, vsseg_tensor out, uint8_t* keep_out, void* stream);
/* Backward, pass 1: sums[shard][0][c] += dz, [1][c] += dz*xhat, [2][c] += dout, alpha_acc[shard] += dA*d(d<0). */
int vsseg_bn_act_bwd_reduce(vsseg_tensor y, vsseg_tensor dout, const float* mean, const float* invstd, const float* gamma, const float* beta,
                            const float* scale, const float* shift, /* the forward's folded affine: the PReLU/dropout branch is re-decided on the SAME fp32 value */
                            const float* alpha, float p_drop, uint64_t seed, uint32_t salt, double* sums, int32_t stride, double* alpha_acc, const uint8_t* keep_in, void* stream);
/* finalize: dgamma, dbeta, dalpha (+= into flat grads) and the two per-channel means used by pass 2 */
int vsseg_bn_act_bwd_finalize(const double* sums, int32_t stride, const double* alpha_acc, int32_t c, double count, float* dgamma, float* dbeta, float* dalpha,
                              float* mean_dz, float* mean_dzx, float* dres_bias /* += sum(dout) or NULL */, void* stream);
/* pass 2: dy = gamma*invstd*(dz - mean_dz - xhat*mean_dzx) */
int vsseg_bn_act_bwd_apply(vsseg_tensor y, vsseg_tensor dout, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* scale, const float* shift, const float* alpha, float p_drop, uint64_t seed, uint32_t salt, const float* mean_dz, const float* mean_dzx, vsseg_tensor dy,
                           const uint8_t* keep_in, void* stream);
/* debug/parity: write the keep-mask (0/1 as f32, [voxel][c]) the forward used */
int vsseg_dropout_mask(float* mask, int64_t nvox, int32_t c, float p_drop, uint64_t seed, uint32_t salt, void* stream);

/* AttentionBlock2: out = x * (1 + att)   (ref:params/networks/blocks/attentionblock.py:43-47); att is f32 [voxel]. */
int vsseg_att_apply_fwd(vsseg_tensor x, const float* att, vsseg_tensor out, void* stream);
/* dx (+)= dout*(1+att);  dpre[voxel][0] = (sum_c dout*x + datt_ext) * att*(1-att)  (sigmoid backward), channels 1..7 zero. */
/* accumulate_dx: 0 dx = dout*(1+att); 1 dx += ...; 2 dx is not written (its contribution is fused into a VSSEG_RES_GATE igemm launch). */
int vsseg_att_apply_bwd(vsseg_tensor x, const float* att, vsseg_tensor dout, const float* datt_ext, vsseg_tensor dx, int32_t accumulate_dx, vsseg_tensor dpre,
                        float* dbias /* += sum(dpre): bias gradient of the sigmoid convolution, or NULL */,
                        void* dpre1 /* optional compact [N,X,Y,Z] copy of channel 0 of dpre (x's dtype), or NULL */, void* stream);

/* generic helpers */
int vsseg_store_u64(uint64_t* dst, uint64_t value, void* stream);                 /* *dst = value by a one-thread kernel (the per-step dropout seed of a replayed launch list) */
int vsseg_memset_zero(void* dst, int64_t bytes, void* stream);                    /* hipMemsetAsync(dst, 0, bytes) on the caller's stream */
int vsseg_copy_bytes(const void* src, void* dst, int64_t bytes, void* stream);    /* device-to-device hipMemcpyAsync */
int vsseg_channel_sum(vsseg_tensor t, float* out /* [c], += */, void* stream);
int vsseg_add_inplace(vsseg_tensor dst, vsseg_tensor src, void* stream); /* dst += src */
int vsseg_copy_cast(vsseg_tensor src, vsseg_tensor dst, void* stream);  /* dst[v][0..c) = src[v][0..c) (fp32 / bf16 either side); fp32 [v][2] -> VSSEG_ZERO_PADDED bf16 rows of 8: one full-row store per voxel */

/* Dice_spvPA (ref:params/losses/dice_spvPA.py:238-297).  sums layout: see loss.hip. */
int vsseg_maxpool_label(const float* src, int32_t n, const int32_t sdims[3], const int32_t ratio[3], float* dst, void* stream);
int vsseg_dice_pred_sums(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, double* sums /* [n][2][3] */, void* stream);
int vsseg_dice_att_sums(const float* att, const float* label, int32_t n, int64_t nvox, double* sums /* [n][3] */, void* stream);
int vsseg_dice_finalize(const double* pred_sums, const double* att_sums, int32_t n, int32_t nlevels, float* loss, float* coef /* [n][2][2] + [levels][n][2] */, void* stream);
int vsseg_dice_pred_bwd(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, const float* coef, const float* gscale, float* dlogits, void* stream);
int vsseg_dice_att_bwd(const float* label, int32_t n, int64_t nvox, const float* coef, float inv_levels, const float* gscale, float* datt, void* stream);
/* The same sums with fewer passes and launches (the fused train step; the values feed the same vsseg_dice_finalize):
   vsseg_dice_level_sums — one pass over a level whose NEXT level pools (2, 2, 1) (the two finest levels of the 2.5D network): att_sums[n][3] of `att` against `label`,
   pooled = MaxPool3d((2,2,1))(label) (NULL: not wanted) and, when `logits` is given ([n][dims][2], the finest level), pred_sums[n][2][3].  dims must be made of
   2 x 2 x 4 blocks, operands 16-byte aligned.
   vsseg_dice_tail_sums — the remaining (coarse) levels in one launch: level i's label = MaxPool3d(sdims / dims[i])(src) is written to label[i] (NULL allowed when
   dims[i] == sdims: the level is src itself) and sums[i][n][3] are those of att[i] against it. */
#define VSSEG_DICE_MAX_LEVELS 8
typedef struct {
  int32_t n, nlevels;
  const float* src;   /* [n][sdims] */
  int32_t sdims[3];
  const float* att[VSSEG_DICE_MAX_LEVELS];
  float* label[VSSEG_DICE_MAX_LEVELS];
  int32_t dims[VSSEG_DICE_MAX_LEVELS][3];
  double* sums;       /* [nlevels][n][3], fixed-point (see below) */
} vsseg_dice_tail_desc;
int vsseg_dice_level_sums(const float* logits, const float* att, const float* label, int32_t n, const int32_t dims[3], int32_t hardness, double* pred_sums, double* att_sums, float* pooled, void* stream);
int vsseg_dice_tail_sums(const vsseg_dice_tail_desc* d, void* stream);
/* vsseg_dice_att_bwd of several levels in one launch: datt[i][b][v] = coef[i][b][0] * label[i][b][v] + coef[i][b][1] (times *gscale when given) */
typedef struct {
  int32_t n, nlevels;
  const float* label[VSSEG_DICE_MAX_LEVELS];
  float* datt[VSSEG_DICE_MAX_LEVELS];
  const float* coef[VSSEG_DICE_MAX_LEVELS]; /* this level's [n][2] slice of vsseg_dice_finalize's coef */
  int64_t nvox[VSSEG_DICE_MAX_LEVELS];
  const float* gscale;                      /* NULL = 1 */
} vsseg_dice_bwd_levels_desc;
int vsseg_dice_att_bwd_levels(const vsseg_dice_bwd_levels_desc* d, void* stream);
/* vsseg_dice_pred_bwd writing the gradient in a layout the training plan stages it in (the fused train step of vs_seg_amd.parallel.DataParallelTrainer: no fp32
   gradient tensor, no cast pass): dst = 2 channels of `n * nvox` voxels, fp32 or bf16 (round-to-nearest-even, as vsseg_copy_cast), pitch 2, or pitch 8 =
   VSSEG_ZERO_PADDED rows whose channels 2..7 are written as zeros (bf16) / left as they are (fp32).  gscale may be NULL (= 1). */
int vsseg_dice_pred_bwd_to(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, const float* coef, const float* gscale, vsseg_tensor dst, void* stream);

/* torch.optim.Adam(lr, weight_decay) over the flat parameter buffer (ref:params/VSparams.py:388-391,462). */
int vsseg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2, float gscale, void* stream);

/* ---- data side: the reference's MONAI transform chain on volumes cached in HBM -------------------- */
/* One crop of one cached volume: RandFlipd(spatial_axis=0) then RandSpatialCropd / SpatialPadd's zero padding
 * (ref:params/VSparams.py:208-222).  origin is in flipped + padded coordinates and may be negative. */
typedef struct {
  const float* src;     /* [sx][sy][sz] fp32, z contiguous (the reference's X,Y,Z array order) */
  int32_t sdims[3];
  int32_t origin[3];
  int32_t flip_x;
} vsseg_crop_job;
/* dst[j][rx][ry][rz] for j < njobs; `jobs` is a DEVICE array of vsseg_crop_job structs, image and label of every batch element. */
int vsseg_crop_flip(const void* jobs, int32_t njobs, float* dst, const int32_t roi[3], void* stream);
/* NormalizeIntensityd (ref:params/VSparams.py:213): y = (x - mean) / std over all n voxels (population std; std == 0: no
 * division).  acc2 = 2 doubles of device scratch (sum, sum of squares; left filled for inspection). */
int vsseg_normalize_intensity(const float* x, float* y, int64_t n, double* acc2, void* stream);

/* Sliding-window blend (MONAI sliding_window_inference steps 6-7; call site ref:params/VSparams.py:568-574). */
/* (seg == NULL: only cnt += w — the weight map of a window geometry, which does not depend on the data; cnt == NULL: only out += w * seg) */
int vsseg_swi_accumulate(const float* seg /* [rx][ry][rz][c] */, const float* imap /* [rx][ry][rz] */, const int32_t roi[3], const int32_t start[3], int32_t c,
                         float* out /* [PX][PY][PZ][c] */, float* cnt /* [PX][PY][PZ] */, const int32_t pdims[3], void* stream);
int vsseg_swi_finalize(const float* out, const float* cnt, const int32_t pdims[3], const int32_t pad_before[3], const int32_t dims[3], int32_t c, float* dst /* [X][Y][Z][c] */, void* stream);
/* hard Dice of argmax vs label (ref:params/VSparams.py:393-408): counts[0]=|P∩G|, [1]=|P|, [2]=|G| */
int vsseg_hard_dice_counts(const float* logits, int32_t pitch, const float* label, int64_t nvox, double* counts, void* stream);
int vsseg_argmax2(const float* logits, int32_t pitch, int64_t nvox, uint8_t* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif

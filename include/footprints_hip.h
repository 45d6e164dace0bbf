/*
 * footprints_hip.h -- flat C ABI of libfootprints_hip.so (gfx950 / MI355X).
 *
 * The reference (nianticlabs/footprints) has no FFI: its hot path is implicit
 * ATen/cuDNN ops behind nn.Modules.  Each entry point below replaces the ATen op
 * family named in its comment (reference file:line = where the op is issued).
 * Conventions:
 *   - raw device pointers (tensor.data_ptr()), explicit int32 dims, fp32 only;
 *   - activations NHWC [N][H][W][C]; network input image and the four network
 *     outputs NCHW (the reference's layout at the nn.Module boundary);
 *   - the caller (PyTorch) owns every buffer, workspace included; the library
 *     never allocates or frees device memory and keeps no pointer after return;
 *   - every function returns 0 on success, else a negative FP_E* code or a
 *     positive hipError_t; fp_last_error_string() describes the last failure of
 *     the calling thread; nothing throws or aborts across the ABI;
 *   - kernels are asynchronous on the hipStream_t passed as the last argument
 *     (torch.cuda.current_stream().cuda_stream), so calls are graph-capturable.
 */
#ifndef FOOTPRINTS_HIP_H
#define FOOTPRINTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fp_stream_t; /* hipStream_t */

#define FP_OK 0
#define FP_EINVAL (-1)     /* bad descriptor / unsupported shape */
#define FP_EWORKSPACE (-2) /* workspace too small */

/* ---- implicit-GEMM convolution family ---------------------------------- */

/* how the GEMM A-operand (one row per output pixel, one K-slice per tap) is gathered */
enum fp_gather {
  FP_GATHER_FWD_ZERO = 0,        /* conv fwd, zero padding, any stride  (torchvision resnet convs; network.py:38-44) */
  FP_GATHER_FWD_REFLECT = 1,     /* conv fwd, ReflectionPad2d(1), stride 1 (network.py:115,126,132) */
  FP_GATHER_FWD_REFLECT_UP2 = 2, /* same, input = cat[nearest_x2(src0), src1] never materialised (network.py:154-155,98) */
  FP_GATHER_DGRAD_ZERO = 3,      /* data-gradient of FWD_ZERO (gather form, stride parity skipped) */
  FP_GATHER_DGRAD_REFLECT = 4,   /* data-gradient of FWD_REFLECT: halo gradients folded back onto rows/cols 1 and H-2 */
  FP_GATHER_STEM = 5             /* 7x7/2 pad 3 on the NCHW image with (x-0.45)/0.225 folded into the load (network.py:50) */
};

enum fp_act { FP_ACT_NONE = 0, FP_ACT_ELU = 1, FP_ACT_RELU = 2 };

/* epilogue: v = acc + bias[n] + addend[m][n]*(addend_mask>0) ; v *= actgrad(actsrc[m][n]) ; v = act(v) ; y = v (+ y) */
#define FP_EPI_BIAS 1u
#define FP_EPI_ADDEND 2u
#define FP_EPI_ADDEND_MASK 4u  /* addend is multiplied by (addend_mask[m][n] > 0) */
#define FP_EPI_ACTGRAD_ELU 8u  /* v *= (s > 0 ? 1 : s + 1), s = actsrc = saved ELU OUTPUT (nn.ELU(inplace=True), network.py:118) */
#define FP_EPI_ACTGRAD_RELU 16u /* v *= (actsrc > 0) */
#define FP_EPI_ACCUM 32u       /* y += v (second decoder / second consumer accumulating into the same gradient) */
#define FP_EPI_BF16X2 64u      /* fp_conv3x3_bf3 forward only, opt-in INFERENCE mode: operands rounded to two bf16 terms (16 significant
                                  bits), three MFMA products instead of six; not exact -- never set by the training path */

typedef struct fp_conv_desc {
  int32_t N;          /* batch */
  int32_t OH, OW;     /* spatial domain of the GEMM rows (fwd: conv output; dgrad: conv input) */
  int32_t IH, IW;     /* spatial dims of the gathered tensor's virtual domain (fwd: conv input, hi-res for UP2; dgrad: dZ) */
  int32_t C0, C1;     /* K channels per tap taken from src0 / src1 (C1 = 0 unless UP2 concat); multiples of 4 */
  int32_t Nout;       /* GEMM N (fwd: Cout; dgrad: Cin) */
  int32_t KH, KW, stride, pad;
  int32_t gather;     /* enum fp_gather */
  int32_t act;        /* enum fp_act */
  uint32_t epi;       /* FP_EPI_* */
} fp_conv_desc;

/* Optional side outputs of ONE launch (round 6; replaces the per-thread fp_amax_out_next / fp_bn_stats_out_next / fp_bn_bwd_out_next sinks of
 * rounds 3-5: the library holds no state between calls besides the calling thread's last error string).  Passed as `const fp_aux* aux` right
 * before the stream argument of the entry points that can produce them; NULL = none; fields that an entry point cannot serve are ignored.
 * The struct is read during the call only.
 *   amax_out   amax slot (see "fp16-pair operands" below; zeroed by the caller) that also receives max |output| of the launch:
 *              fp_bn_apply, fp_bn_bwd / fp_bn_bwd_partials (their dz), fp_maxpool_fwd, fp_up2_fold_bwd, fp_head_dgrad,
 *              fp_conv_up2_phase_fwd_bf3, fp_conv_stem_hp.
 *   bn_part    BatchNorm partials out of a convolution's epilogue -- fp_conv3x3_bf3 / fp_conv3x3_hp, fp_conv_igemm_bf3 / _hp (split grids: from
 *              their reduce launch), fp_conv_stem_hp, fp_conv_igemm (stem gather):
 *                bnb_z == NULL: a plain forward launch (no bias / addend / activation) in front of a train-mode BatchNorm writes Welford partials
 *                  bn_part[pixel tile][Nout] = (count, mean, M2) of what it stores -> fp_bn_train_stats_partials;
 *                bnb_z != NULL: a data gradient whose stored output IS g = dy * (relu_out > 0) of a train-mode BatchNorm (mask applied by its own
 *                  FP_EPI_ACTGRAD_RELU epilogue, no FP_EPI_ACCUM) writes bn_part[pixel tile][Nout] = (sum g, sum g * xhat), xhat = (bnb_z - bnb_mean)
 *                  * bnb_invstd of THAT BatchNorm -> fp_bn_bwd_partials.
 *              *bn_nblk_out (host memory) = number of pixel tiles written, set before the call returns; 0 = nothing emitted (launch form cannot
 *              emit, bn_capacity_floats too small): the caller then runs the BatchNorm's own reduction pass. */
typedef struct fp_aux {
  uint32_t* amax_out;
  float* bn_part;
  int64_t bn_capacity_floats;
  int32_t* bn_nblk_out;
  const float* bnb_z;
  const float* bnb_mean;
  const float* bnb_invstd;
} fp_aux;

/* Y[m][n] = epilogue( sum_{tap,k} A[m][tap,k] * Wp[tap][k][n] ).
 * Wp is the packed weight produced by fp_pack_conv_weight (fwd) / fp_pack_conv_weight_dgrad.
 * Replaces aten::convolution (+reflection_pad2d, upsample_nearest2d, cat, elu_) and the
 * data-gradient half of aten::convolution_backward (+elu_backward, relu mask, residual add).
 * Small problems (few output tiles) are split along K across workgroups when `workspace` (>=
 * fp_conv_igemm_workspace(d) bytes; may be NULL = never split) is given; partials are summed in a fixed order. */
int64_t fp_conv_igemm_workspace(const fp_conv_desc* d);
int fp_conv_igemm(const fp_conv_desc* d, const float* src0, const float* src1, const float* wpacked,
                  const float* bias, const float* addend, const float* addend_mask, const float* actsrc,
                  float* y, void* workspace, int64_t workspace_bytes, const fp_aux* aux, fp_stream_t stream);

/* Weight gradient: dW (OIHW [Nout][C0+C1][KH][KW]) = sum_m A[m][tap,k] * dZ[m][n]; A gathered as the
 * matching forward conv (desc.gather is a FWD_* / STEM mode, desc.OH/OW = conv output dims).
 * Deterministic two-stage reduction through `workspace` (>= fp_conv_wgrad_workspace(d) bytes).
 * accumulate != 0 => dW += result.  Replaces the weight half of aten::convolution_backward. */
int64_t fp_conv_wgrad_workspace(const fp_conv_desc* d);
/* fp_conv_wgrad_slice: same, but the C0+C1 input channels of `d` are the slice [k_begin, k_begin+C0+C1) of a wider
 * gradient dw_oihw[Nout][kc_total][KH][KW] (the skip half of a concat conv). */
int fp_conv_wgrad_slice(const fp_conv_desc* d, const float* src0, const float* src1, const float* dz, float* dw_oihw,
                        int32_t kc_total, int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes,
                        fp_stream_t stream);
int fp_conv_wgrad(const fp_conv_desc* d, const float* src0, const float* src1, const float* dz,
                  float* dw_oihw, int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream);

/* One repacking job of fp_pack_weights_batched: every convolution's packed copies are refreshed by ONE launch after the
 * optimizer step (the table lives in device memory and is built once: parameter and packed buffers never move). */
enum { FP_PACK_FWD = 0, FP_PACK_DGRAD = 1, FP_PACK_STEM = 2, FP_PACK_UP2_FWD = 3, FP_PACK_UP2_DGRAD = 4,
       FP_PACK_FWD_BF3 = 5, FP_PACK_DGRAD_BF3 = 6, FP_PACK_UP2_FWD_BF3 = 7, FP_PACK_UP2_DGRAD_BF3 = 8, /* bf16x3 split planes, see fp_conv3x3_bf3 */
       FP_PACK_FWD_HP = 9, FP_PACK_DGRAD_HP = 10, FP_PACK_UP2_FWD_HP = 11, FP_PACK_UP2_DGRAD_HP = 12, /* scaled fp16 pairs, see fp_conv3x3_hp */
       FP_PACK_STEM_HP = 13 /* the 7x7x3 stem as fp16 pairs, K = 7 rows x 24 (21 taps + 3 zeros): [11 K-steps][2 planes][64][16], see fp_conv_stem_hp */ };
typedef struct fp_pack_job {
  const float* w;  /* [Cout][Cin][KH][KW] */
  float* wp;       /* packed destination */
  int32_t Cout, Cin, KH, KW;
  int32_t kind;             /* FP_PACK_* */
  int32_t c_begin, c_count; /* input-channel slice (whole tensor: 0, Cin) */
  int32_t block_begin, block_count; /* this job's contiguous range of workgroups in the batched launch */
  uint32_t* amax;           /* *_HP kinds: the weight tensor's amax slot (FP_AMAX_ELEMS uint32, shared by all jobs of the tensor) */
} fp_pack_job;
int32_t fp_pack_job_blocks(int32_t kind, int32_t Cout, int32_t KH, int32_t KW, int32_t c_count);
int fp_pack_weights_batched(const fp_pack_job* jobs_dev, const int32_t* blk2job_dev, int32_t nblocks, fp_stream_t stream);
/* the same launch, persistent: at most `max_wgs` workgroups (> 0) walk the nblocks virtual blocks -- a repack on a side stream must not take
 * the wave slots of the chain it runs beside (Engine.refresh_packed; 0 = one workgroup per virtual block) */
int fp_pack_weights_batched_capped(const fp_pack_job* jobs_dev, const int32_t* blk2job_dev, int32_t nblocks, int32_t max_wgs,
                                   fp_stream_t stream);

/* packed-weight sizes (floats) and packers; w_oihw is the torch Conv2d.weight [Cout][Cin][KH][KW] */
int64_t fp_packed_weight_elems(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad, int32_t stem);
int fp_pack_conv_weight(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                        int32_t stem, fp_stream_t stream);
int fp_pack_conv_weight_dgrad(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                              fp_stream_t stream);

/* ---- 3x3 stride-1 convolution with exactly split operands (conv3x3_tile_bf3.hip).
 * Every fp32 operand is split into three bf16 terms whose sum is exact (8+8+8 significant bits); six of the nine bf16 x bf16
 * products (all those >= 2^-16 of the leading one) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The result is
 * closer to the exact dot product than the fp32 MFMA path (the dropped terms are <= 2^-24 relative, below fp32 rounding) and
 * runs at up to 2.7x its MFMA rate.  Same operation, arguments and epilogue flags as fp_conv_igemm for
 * FP_GATHER_{FWD_ZERO, FWD_REFLECT, DGRAD_ZERO, DGRAD_REFLECT} (C1 = 0, src1 = NULL) and FP_GATHER_FWD_REFLECT_UP2 (src = the
 * half-resolution tensor with C0 % 16 == 0 channels, src1 = the skip tensor with C1 channels), for the shapes fp_conv3x3_bf3_supported accepts
 * (3x3 / stride 1 / pad 1; 8x16- or 6x20-pixel tiles with <= 25 % padding; small grids are split along the input
 * channels into raw partials in `workspace`, summed in a fixed order before the epilogue); weights packed by
 * fp_pack_conv_weight_bf3 (fp_packed_weight_elems_bf3 floats of storage) or FP_PACK_{FWD,DGRAD}_BF3 jobs. */
int fp_conv3x3_bf3_supported(const fp_conv_desc* d);
int64_t fp_conv3x3_bf3_workspace(const fp_conv_desc* d);   /* split-K scratch for small grids (0 = none) */
int fp_conv3x3_bf3(const fp_conv_desc* d, const float* src, const float* src1, const void* wpacked_bf3, const float* bias,
                   const float* addend, const float* addend_mask, const float* actsrc, float* y, void* workspace,
                   int64_t workspace_bytes, const fp_aux* aux, fp_stream_t stream);
int64_t fp_packed_weight_elems_bf3(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad);
int fp_pack_conv_weight_bf3(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                            int32_t for_dgrad, fp_stream_t stream);

/* ---- fp16-pair operands ("hp"): x * 2^k = h + m, h = fp16(x * 2^k), m = fp16(x * 2^k - h) -- 22 significant bits after a per-tensor
 * power-of-two scaling taken from the tensor's largest magnitude; three fp16 products (hh + hm + mh; mm is <= 2^-22 of a product and left out; each exact in the fp32
 * accumulator of v_mfma_f32_32x32x16_f16) instead of the six of the exact split and two operand planes instead of three.
 * An "amax slot" is FP_AMAX_SLOTS = 16 uint32 (FP_AMAX_STRIDE apart) holding float bit patterns of |x| (combined with max): zero it (fp_zero_u32), then
 * either reduce a tensor into it (fp_amax_f32) or let the kernel that produces the tensor publish into it (`amax_out`). */
#define FP_AMAX_SLOTS 16                 /* sub-slots per amax slot */
#ifndef FP_AMAX_STRIDE
#define FP_AMAX_STRIDE 32                /* uint32 elements between consecutive sub-slots: one 128-byte line each */
#endif
#define FP_AMAX_ELEMS (FP_AMAX_SLOTS * FP_AMAX_STRIDE)     /* uint32 elements of storage per slot */
int32_t fp_amax_slot_elems(void);        /* = FP_AMAX_ELEMS of the loaded build */
int fp_zero_u32(uint32_t* p, int64_t n, fp_stream_t stream);
int fp_amax_f32(const float* x, int64_t n, uint32_t* slot, fp_stream_t stream);          /* x 16-byte aligned; slot zeroed by the caller */
int fp_weight_amax(const float* w, int64_t n, uint32_t* amax_slot, fp_stream_t stream);  /* zero + reduce */
int64_t fp_packed_weight_elems_hp(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad);   /* floats of storage */
int fp_pack_conv_weight_hp(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad,
                           uint32_t* amax_slot, int32_t amax_ready, fp_stream_t stream);
int fp_pack_weights_amax(const fp_pack_job* jobs_dev, const int32_t* blk2job_dev, int32_t nblocks, fp_stream_t stream);
/* fp_conv3x3_bf3 with fp16-pair operands: same shapes, gathers and epilogue flags; weights from fp_pack_conv_weight_hp /
 * FP_PACK_{FWD,DGRAD}_HP jobs together with the slot they were scaled by (`amax_w`); `amax_src` (and `amax_src1` for the skip tensor
 * of FP_GATHER_FWD_REFLECT_UP2) = max |x| of the source tensor(s); `amax_out` (optional, zeroed by the caller) receives max |y|. */
int fp_conv3x3_hp(const fp_conv_desc* d, const float* src, const float* src1, const void* wpacked_hp, const float* bias,
                  const float* addend, const float* addend_mask, const float* actsrc, float* y, void* workspace,
                  int64_t workspace_bytes, const uint32_t* amax_src, const uint32_t* amax_src1, const uint32_t* amax_w,
                  uint32_t* amax_out, const fp_aux* aux, fp_stream_t stream);

/* nearest-x2 phase kernels (conv_up2_phase.hip) with fp16-pair operands: weights from FP_PACK_UP2_FWD_HP / FP_PACK_UP2_DGRAD_HP jobs */
int fp_conv_up2_phase_fwd_hp(const float* low, const void* wphase_hp, const float* bias, const float* addend, float* y, int32_t N,
                             int32_t h, int32_t w, int32_t C0, int32_t Nout, int32_t act, const uint32_t* amax_low,
                             const uint32_t* amax_w, uint32_t* amax_out, fp_stream_t stream);
int fp_conv_up2_phase_dgrad_hp(const float* dz, const void* wpacked_hp, float* ext, int32_t N, int32_t h, int32_t w, int32_t Cout,
                               int32_t C0, const uint32_t* amax_dz, const uint32_t* amax_w, fp_stream_t stream);

/* weight gradients with fp16-pair operands: fp_conv_wgrad_bf3 / fp_conv_up2_phase_wgrad_bf3 with the amax slots of their two tensors */
int fp_conv_wgrad_hp(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total,
                     int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes, const uint32_t* amax_x,
                     const uint32_t* amax_dz, fp_stream_t stream);
int fp_conv_up2_phase_wgrad_hp(const float* low, const float* dz, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w,
                               int32_t C0, int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                               int64_t workspace_bytes, const uint32_t* amax_low, const uint32_t* amax_dz, fp_stream_t stream);

/* Weight gradient with the same exactly split operands (wgrad3x3_bf3.hip): 3x3 / stride 1 / pad 1, FWD_ZERO, FWD_REFLECT or
 * FWD_REFLECT_UP2 gather (then x is the half-resolution tensor [N][OH/2][OW/2][C0]: the upsampled half of a concat conv), C1 = 0, C0 and Nout multiples of 32; fp_conv_wgrad_bf3_workspace returns -1 for anything else (use fp_conv_wgrad).
 * dw_oihw is [Nout][kc_total][3][3]; the C0 input channels of `d` are its slice [k_begin, k_begin + C0).
 * db (optional, [Nout]): the bias gradient = column sums of dz, produced from the dz tiles the kernel stages anyway
 * (replaces a separate fp_colsum pass over dz); (+)= like dw_oihw. */
int64_t fp_conv_wgrad_bf3_workspace(const fp_conv_desc* d);
int fp_conv_wgrad_bf3(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total,
                      int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream);

/* ---- nearest-x2 phase decomposition (reference footprints/network.py:98,126-134,154: upsample -> [cat skip] ->
 * ReflectionPad2d(1) -> Conv2d 3x3).  A 3x3 conv over the x2-upsampled `low` equals four 2x2 convs (one per output
 * parity phase) over `low` itself with row/column-collapsed weights and replicate padding: 2.25x fewer MACs and no
 * upsampled temporary.  fp_pack_up2_weight collapses input channels [c_begin, c_begin+c_count) of w_oihw
 * [Cout][Cin][3][3] into wp[phase 4][tap 4][ceil(c_count/16)][Cout][16]; fp_pack_conv_weight_slice packs the remaining
 * (skip) channels in the ordinary fp_pack_conv_weight layout.  fp_conv_up2_phase_fwd:
 *   y[n][2y+dy][2x+dx][:] = act( sum_taps Wc . low + bias + addend )      (addend may alias y: skip-half partial sums) */
int64_t fp_up2_packed_weight_elems(int32_t Cout, int32_t c_count);
int fp_pack_up2_weight(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                       fp_stream_t stream);
int fp_pack_conv_weight_slice(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin,
                              int32_t c_count, fp_stream_t stream);
int fp_conv_up2_phase_fwd(const float* low, const float* wphase, const float* bias, const float* addend, float* y,
                          int32_t N, int32_t h, int32_t w, int32_t C0, int32_t Nout, int32_t act, fp_stream_t stream);
/* bf16x3-split variant (see fp_conv3x3_bf3): same operation, weights from fp_pack_up2_weight_bf3 / FP_PACK_UP2_FWD_BF3
 * (fp_up2_packed_weight_elems(Cout, c_count) * 3 / 2 floats of storage). */
int fp_pack_up2_weight_bf3(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                           fp_stream_t stream);
int fp_conv_up2_phase_fwd_bf3(const float* low, const void* wphase_bf3, const float* bias, const float* addend, float* y,
                              int32_t N, int32_t h, int32_t w, int32_t C0, int32_t Nout, int32_t act, const fp_aux* aux,
                              fp_stream_t stream);
/* Backward (the upsample_nearest2d_backward + reflection_pad2d_backward + convolution_backward(data) chain): d(low) is a
 * 4x4 stride-2 convolution over dZ.  Pack with fp_pack_up2_weight_dgrad (fp_up2_packed_weight_elems(c_count, Cout)
 * floats), run fp_conv_igemm{FWD_ZERO, K=4, stride 2, pad 3, IH=2h, OH=h+2} into ext[N][h+2][w+2][c_count], then
 * fp_up2_fold_bwd folds the replicate-padding border back:  dlow = (fold(ext) + addend) * ELU'(ylow_elu)  (both optional).
 * The skip half of a concat conv takes the ordinary DGRAD_REFLECT path with fp_pack_conv_weight_dgrad_slice weights. */
int fp_pack_up2_weight_dgrad(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                             fp_stream_t stream);
int fp_pack_conv_weight_dgrad_slice(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin,
                                    int32_t c_count, fp_stream_t stream);
/* bf16x3 variant of the 4x4 stride-2 data-gradient convolution: writes the same ext[N][h+2][w+2][C0] as fp_conv_igemm would;
 * weights from fp_pack_up2_weight_dgrad_bf3 / FP_PACK_UP2_DGRAD_BF3 (fp_up2_packed_weight_elems(c_count, Cout) * 3 / 2 floats). */
int fp_pack_up2_weight_dgrad_bf3(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t c_begin,
                                 int32_t c_count, fp_stream_t stream);
int fp_conv_up2_phase_dgrad_bf3(const float* dz, const void* wpacked_bf3, float* ext, int32_t N, int32_t h, int32_t w,
                                int32_t Cout, int32_t C0, fp_stream_t stream);
int fp_up2_fold_bwd(const float* ext, int32_t N, int32_t h, int32_t w, int32_t C, const float* addend,
                    const float* ylow_elu, float* dlow, const fp_aux* aux, fp_stream_t stream);
/* Weight gradient of the upsampled half: dw_oihw[:, k_begin:k_begin+C0] (+)= un-collapse of the 16 per-phase products
 * (dw_oihw is [Nout][kc_total][3][3]).  fp_conv_up2_phase_wgrad_workspace returns -1 for shapes it does not take
 * (C0, Nout multiples of 32; <= 30 % padded 2x16 chunks) -- use fp_conv_wgrad{FWD_REFLECT_UP2} there.  The skip half is
 * fp_conv_wgrad_slice{FWD_REFLECT} into dw_oihw[:, C0:]. */
int64_t fp_conv_up2_phase_wgrad_workspace(int32_t N, int32_t h, int32_t w, int32_t C0, int32_t Nout);
int fp_conv_up2_phase_wgrad(const float* low, const float* dz, float* dw_oihw, int32_t N, int32_t h, int32_t w, int32_t C0,
                            int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                            int64_t workspace_bytes, fp_stream_t stream);
/* same contract and workspace; fp32 operands split exactly into three bf16 terms, six bf16 MFMA products (error below the fp32 MFMA's);
 * db (optional, [Nout]) (+)= the bias gradient = column sums of dz, from the dz tiles the kernel stages anyway */
int fp_conv_up2_phase_wgrad_bf3(const float* low, const float* dz, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w, int32_t C0,
                                int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                                int64_t workspace_bytes, fp_stream_t stream);

/* column sums: out[c] (+)= sum_m x[m][c]  -- conv bias gradient (weight half of convolution_backward) */
int64_t fp_colsum_workspace(int64_t M, int32_t C);
int fp_colsum(const float* x, int64_t M, int32_t C, float* out, int accumulate, void* workspace,
              int64_t workspace_bytes, fp_stream_t stream);

/* backward of cat[nearest_x2(low), skip] feeding a conv: dxv is the conv's input gradient at hi-res
 * [N][2h][2w][C0+C1]; dlow[N][h][w][C0] = (sum2x2(dxv[..:C0]) + addend) * elu'(ylow) ; dskip (+)= dxv[..C0:].
 * Replaces upsample_nearest2d_backward + cat backward + elu_backward (network.py:154-155,98). */
int fp_up2cat_bwd(const float* dxv, int32_t N, int32_t h, int32_t w, int32_t C0, int32_t C1,
                  const float* addend, const float* ylow_elu, float* dlow, float* dskip, int accumulate_skip,
                  fp_stream_t stream);

/* ---- 2-channel output heads (OutConvBlock, network.py:161-183) --------- */
/* low[N][h][w][2] = [sigmoid](reflect-pad 3x3 conv Cin->2 + bias); w_oihw [2][Cin][3][3] */
int fp_head_fwd(const float* x, const float* w_oihw, const float* bias, float* low, int32_t N, int32_t h, int32_t w,
                int32_t Cin, int32_t apply_sigmoid, fp_stream_t stream);
/* out[N][out_channels][H][W] channels c0,c0+1 = bilinear_xS(low), align_corners=False (S=1: copy) */
int fp_head_upsample(const float* low, float* out_nchw, int32_t N, int32_t h, int32_t w, int32_t scale,
                     int32_t out_channels, int32_t c0, fp_stream_t stream);
/* dzlow[N][h][w][2] = bilinear^T(dout[:, c0:c0+2]) [* s(1-s), s = low when apply_sigmoid] */
int fp_head_upsample_bwd(const float* dout_nchw, const float* low, float* dzlow, int32_t N, int32_t h, int32_t w,
                         int32_t scale, int32_t out_channels, int32_t c0, int32_t apply_sigmoid, fp_stream_t stream);
/* dx[N][h][w][Cin] = conv^T(dzlow) with the reflection halo folded back (plain store);
 * elu_src != NULL: dx *= elu'(elu_src) (the head is the only consumer of an ELU output: outconv4, network.py:78-80) */
int fp_head_dgrad(const float* dzlow, const float* w_oihw, const float* elu_src, float* dx, int32_t N, int32_t h, int32_t w,
                  int32_t Cin, const fp_aux* aux, fp_stream_t stream);
/* dw_oihw[2][Cin][3][3], db[2] (+)= ... ; deterministic two-stage */
int64_t fp_head_wgrad_workspace(int32_t N, int32_t h, int32_t w, int32_t Cin);
int fp_head_wgrad(const float* x, const float* dzlow, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w,
                  int32_t Cin, int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream);

/* The stem (7x7 / stride 2 / pad 3 on the NCHW image, (x - 0.45) / 0.225 applied before the zero padding; network.py:48-52) with fp16-pair
 * operands: the image patch of an 8 x 16 output tile is normalised, scaled by 2^12 (|x| <= 2.45) and split into its two fp16 planes once in
 * LDS; weights from an FP_PACK_STEM_HP job with the slot they were scaled by.  Same desc / epilogue rules as fp_conv_igemm's STEM gather
 * (bias flag, act); honors the statistics and amax sinks.  fp_conv_stem_hp_supported: Nout == 64, IH == 2 OH, IW == 2 OW. */
int fp_conv_stem_hp_supported(const fp_conv_desc* d);
int fp_conv_stem_hp(const fp_conv_desc* d, const float* img_nchw, const void* wpacked_hp, const float* bias, float* y,
                    const uint32_t* amax_w, const fp_aux* aux, fp_stream_t stream);

/* ... and the stem's weight gradient likewise (dW [64][3][7][7] (+)= ...; desc / workspace of fp_conv_wgrad on the STEM gather): the image
 * patch and the dZ tile are split into fp16 pairs in LDS, fragments by gfx950's transposing LDS read; `amax_dz` = amax slot of dz */
int fp_conv_stem_wgrad_hp(const fp_conv_desc* d, const float* img_nchw, const float* dz, float* dw_oihw, int accumulate, void* workspace,
                          int64_t workspace_bytes, const uint32_t* amax_dz, fp_stream_t stream);

/* ---- BatchNorm (train-mode batch statistics; torchvision BN in the encoder) ---- */
/* stats over z[M][C]: save_mean, save_invstd, fused scale = gamma*invstd, shift = beta - mean*scale;
 * running stats updated in place with momentum (unbiased var), num_batches_tracked += 1 (int64).
 * Replaces aten::native_batch_norm (training=True). workspace >= fp_bn_workspace(M, C). */
int64_t fp_bn_workspace(int64_t M, int32_t C);
int fp_bn_train_stats(const float* z, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      float* save_mean, float* save_invstd, float* scale, float* shift, void* workspace,
                      int64_t workspace_bytes, fp_stream_t stream);
/* fp_bn_train_stats's second launch on Welford partials that the producing convolution's epilogue wrote (fp_aux.bn_part, bnb_z == NULL): same
 * outputs, the activation is not read for its statistics (torchvision BatchNorm2d behind footprints/network.py:38-44). */
int fp_bn_train_stats_partials(const float* part, int32_t nblk, int32_t C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* save_mean, float* save_invstd, float* scale, float* shift, fp_stream_t stream);
/* The backward counterpart (round 4): fp_bn_bwd's combination + apply launches on partial sums (sum g, sum g * xhat) that a data gradient's
 * epilogue wrote (fp_aux.bn_part with bnb_z / bnb_mean / bnb_invstd): fp_bn_bwd's reduction pass over (dy, relu_out, z) and its launch are gone
 * (the backward of torchvision BatchNorm2d behind footprints/network.py:38-44; coef = 2 C floats of scratch). */
int fp_bn_bwd_partials(const float* g, const float* z, const float* save_mean, const float* save_invstd, const float* gamma, float* dz,
                       float* dgamma, float* dbeta, int accumulate, int64_t M, int32_t C, const float* part, int32_t nblk, float* coef,
                       const fp_aux* aux, fp_stream_t stream);
/* eval mode: scale/shift from running statistics */
int fp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int32_t C, float* scale, float* shift, fp_stream_t stream);
/* out[r][:] = w[r][:] * scale[r]: folds the eval-mode scale into the preceding conv's OIHW weights (the shift becomes the
 * conv's bias), so inference runs conv + bias + ReLU (+ residual) in one launch -- SURVEY.md section 8(f) N1. */
int fp_scale_rows(const float* w, const float* scale, float* out, int64_t rows, int64_t inner, fp_stream_t stream);
/* y = [relu](z*scale + shift [+ residual]) */
/* fp16-pair operands for the flattened implicit GEMM (round 3): the 3x3 stride-2 and 1x1 convolutions of torchvision's BasicBlocks
 * (footprints/network.py:38-44) and their data gradients -- everything fp_conv3x3_hp does not take -- on the fp16 matrix path instead of
 * the fp32 one.  Same operation / epilogue flags / split-K workspace as fp_conv_igemm for zero-padding gathers of one source tensor
 * (FP_GATHER_FWD_ZERO, FP_GATHER_DGRAD_ZERO; C1 = 0); weights from FP_PACK_FWD_HP / FP_PACK_DGRAD_HP jobs (any kernel size) with the
 * slot `amax_w` they were scaled by; `amax_src` holds max |src| (fp_amax_f32 or a producer's publication). */
int fp_conv_igemm_hp_supported(const fp_conv_desc* d);
int fp_conv_igemm_hp(const fp_conv_desc* d, const float* src, const void* wpacked_hp, const float* bias, const float* addend,
                     const float* addend_mask, const float* actsrc, float* y, void* workspace, int64_t workspace_bytes,
                     const uint32_t* amax_src, const uint32_t* amax_w, const fp_aux* aux, fp_stream_t stream);
/* ... and with EXACTLY split bf16x3 operands (round 5): the default operand format's path for the same convolutions -- x = h + m + l in
 * bf16, six MFMA products, no operand bit of fp32 dropped, no amax slots.  Weights from FP_PACK_FWD_BF3 / FP_PACK_DGRAD_BF3 jobs (any
 * kernel size; fp_packed_weight_elems_bf3 floats).  Shapes: fp_conv_igemm_hp_supported.  Replaces aten::convolution /
 * convolution_backward(data) of the stride-2 BasicBlock convs and the 1x1 downsample convs (footprints/network.py:38-44). */
int fp_conv_igemm_bf3(const fp_conv_desc* d, const float* src, const void* wpacked_bf3, const float* bias, const float* addend,
                      const float* addend_mask, const float* actsrc, float* y, void* workspace, int64_t workspace_bytes, const fp_aux* aux,
                      fp_stream_t stream);

int fp_bn_apply(const float* z, const float* scale, const float* shift, const float* residual, float* y, int64_t M,
                int32_t C, int32_t relu, const fp_aux* aux, fp_stream_t stream);
/* backward: g = dy * (relu_out > 0 if relu_out) ; dgamma (+)= sum g*xhat ; dbeta (+)= sum g ;
 * dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) ; g_out (optional) = g. */
int fp_bn_bwd(const float* dy, const float* relu_out, const float* z, const float* save_mean, const float* save_invstd,
              const float* gamma, float* dz, float* g_out, float* dgamma, float* dbeta, int accumulate, int64_t M,
              int32_t C, void* workspace, int64_t workspace_bytes, const fp_aux* aux, fp_stream_t stream);

/* ---- test-set inference output: [B][4][H][W] fp32 predictions -> float16 with sigmoid on channels 0, 1
 * (reference evaluation/inference.py:105-108, datasets/inference_dataset.py:35-38) ---- */
int fp_pack_pred_fp16(const float* pred_nchw, void* out_half, int32_t B, int32_t H, int32_t W, fp_stream_t stream);

/* ---- test-set metrics (reference evaluation/evaluate_model.py:50-99 evaluate_depth / evaluate_mask, :160-177 call sites).
 * pred: float32 or float16 (pred_is_half) images, image b at pred + b * pred_stride elements; gt: float32 [B][pixels];
 * region: optional uint8 [B][pixels] (pixels with 0 are skipped); invert = 1 evaluates (1 - gt, 1 - pred) like the footprint score.
 * counts: int64 [B][4] = n_true, tp, fp, fn.   sums: float64 [B][5] = n, n(thresh < 1.25), sum sq, sum abs_rel, sum sq_rel over gt > 0. */
int fp_eval_mask_counts(const void* pred, int32_t pred_is_half, const float* gt, const uint8_t* region, int32_t invert, int32_t B,
                        int64_t pixels, int64_t pred_stride, int64_t* counts, fp_stream_t stream);
int fp_eval_depth_sums(const void* pred_disp, int32_t pred_is_half, const float* gt, int32_t B, int64_t pixels, int64_t pred_stride,
                       double min_depth, double max_depth, double clip_min, double clip_max, double* sums, fp_stream_t stream);

/* ---- maxpool 3x3 stride 2 pad 1 (encoder.maxpool, network.py:41) ---------- */
int fp_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C,
                   const fp_aux* aux, fp_stream_t stream);
int fp_maxpool_bwd(const float* dy, const uint8_t* argmax, float* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                   int accumulate, fp_stream_t stream);

/* ---- fused multi-head loss, forward + backward (training/losses.py:31-152) ---- */
/* preds[4]: '1/8','1/4','1/2','1/1' each [B][4][H][W]; targets [B][H][W];
 * losses_out[21] in the order of LossManager's dict (5 per scale: visible_ground, all_ground, depth,
 * ground_depth, loss; then total 'loss'); dpreds[4] = d(total loss)/d(pred) (may be NULL => forward only). */
int64_t fp_loss_workspace(int32_t B, int32_t H, int32_t W);
int fp_loss_fwd_bwd(const float* const preds[4], const float* visible_ground, const float* all_ground,
                    const float* depth, const float* ground_depth, const float* moving_object_mask,
                    const float* depth_mask, float min_depth, float max_depth, float prior_weight,
                    float* const dpreds[4], float* losses_out, int32_t B, int32_t H, int32_t W, void* workspace,
                    int64_t workspace_bytes, fp_stream_t stream);

/* ---- fused Adam over a flat parameter buffer (torch.optim.Adam, model_manager.py:27) ---- */
/* hyper-parameters are doubles (python floats) like torch's; gradients are multiplied by grad_scale (1/world under DP) */
int fp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                 double beta2, double eps, int32_t step, double grad_scale, fp_stream_t stream);
/* Graph-replay variant: the seven per-step scalars (fp_adam_hyper fills them on the host exactly as fp_adam_step derives
 * them) live in device memory, so a captured launch stays valid across steps; n must be a multiple of 4. */
int fp_adam_hyper(double lr, double beta1, double beta2, double eps, int32_t step, double grad_scale, float* hyper7_host);
int fp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                     const float* hyper7_dev, fp_stream_t stream);

/* ---- recorded launch plans: replay a whole training step (~740 launches on five streams) from C ------------------------------ */
/* While a plan records (thread local) every kernel launch of this library and every fp_event_record / fp_event_wait is executed AND
 * appended to the plan with a private copy of its arguments; fp_plan_replay re-issues nodes [begin, end) (end < 0: to the last node)
 * with the same streams.  Valid as long as every pointer argument stays valid (static arena, workspaces and tables).
 * fp_plan_mark returns the current node count (stage boundaries for a replay in pieces, e.g. around gradient all-reduces). */
void* fp_plan_begin(void);
int32_t fp_plan_mark(void* plan);
int32_t fp_plan_end(void* plan);
int fp_plan_replay(void* plan, int32_t begin, int32_t end);
void fp_plan_destroy(void* plan);
/* stream-to-stream ordering used by the engine (instead of framework events, so that a recording sees it): record returns an event
 * id >= 0 (valid for fp_event_wait until ~4000 further records, or -- inside a recording -- until the recording ends) */
int64_t fp_event_record(fp_stream_t stream);
int fp_event_wait(fp_stream_t stream, int64_t event_id);

/* ---- device-side data path (footprints/datasets/footprint_dataset.py:55-65,73-85; kitti_dataset.py:66-112; matterport_dataset.py:69-97) ---- */
/* one per sample; fp_aug_params_bytes() == sizeof(fp_aug_params) */
typedef struct fp_aug_params {
  int32_t flip;        /* horizontal flip of the image and every label map (footprint_dataset.py:73-75,84-85) */
  int32_t n_ops;       /* 0 = no colour jitter, 4 = torchvision ColorJitter's four ops */
  int32_t ops[4];      /* application order (random.shuffle in ColorJitter.get_params): 0 brightness, 1 contrast, 2 saturation, 3 hue */
  float factor[4];     /* indexed by op id: ImageEnhance factors (float, as libImaging's Image.blend takes them) */
  int32_t hue_shift;   /* np.uint8(hue_factor * 255) of torchvision's adjust_hue */
  int32_t pad;
} fp_aug_params;
int32_t fp_aug_params_bytes(void);
/* images_hwc uint8 [B][H][W][3] (as PIL delivers them after the resize, not flipped) -> out_nchw float [B][3][H][W] in [0,1]:
 * flip, ColorJitter (bit-exact with Pillow 12's ImageEnhance / convert arithmetic), ToTensor.  luma_sums: B uint64 scratch
 * (the contrast op needs the mean grey level of the image as it stands before that op: one integer reduction pass). */
int fp_assemble_images(const uint8_t* images_hwc, const void* aug_params, uint64_t* luma_sums, float* out_nchw, int32_t B, int32_t H,
                       int32_t W, fp_stream_t stream);
/* label maps [B][H][W] (float32 or float64 as they leave the resize, not flipped) -> the six float32 maps of the batch schema.
 * dataset 0 = KITTI (aux = resized pixel disparity before the -1.25; fxb = focal * baseline; moving may be NULL when !use_moving),
 * 1 = Matterport (aux = raw 16-bit depth, depth_scaling = metres per unit).  Arithmetic in float64 like numpy in the reference. */
int fp_assemble_labels(const void* visible_ground, const void* ground_depth, const void* depth_mask, const void* aux, const void* moving,
                       int32_t is_double, const void* aug_params, float* o_visible_ground, float* o_depth, float* o_ground_depth,
                       float* o_moving, float* o_depth_mask, float* o_all_ground, int32_t B, int32_t H, int32_t W, int32_t dataset,
                       int32_t no_depth_mask, int32_t project_down_baseline, int32_t use_moving, double threshold, double fxb,
                       double depth_scaling, fp_stream_t stream);

/* ---- pyramid pooling of the ground-segmentation network (footprints/preprocessing/segmentation/network.py:174-207) ---- */
/* nn.AdaptiveAvgPool2d(P) (network.py:180,188): y[N][P][P][C] = window means of x[N][H][W][C]; windows floor(i*H/P) .. ceil((i+1)*H/P) */
int fp_adaptive_avgpool_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, fp_stream_t stream);
/* its gradient: dx[N][H][W][C] (+)= sum over the windows containing the pixel of dy / window area */
int fp_adaptive_avgpool_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, int accumulate,
                            fp_stream_t stream);
/* F.interpolate(size=(H,W), mode='bilinear', align_corners=True) (network.py:189) of src[N][P][P][C], written into channels
 * [c_off, c_off + C) of dst[N][H][W][dstC]: the torch.cat of network.py:207 is this channel offset */
int fp_bilinear_ac_fwd(const float* src, float* dst, int32_t N, int32_t P, int32_t C, int32_t H, int32_t W, int32_t dstC, int32_t c_off,
                       fp_stream_t stream);
/* its transpose: dsrc[N][P][P][C] = sum of weight * ddst[N][H][W][c_off + c] (plain store) */
int fp_bilinear_ac_bwd(const float* ddst, float* dsrc, int32_t N, int32_t P, int32_t C, int32_t H, int32_t W, int32_t dstC, int32_t c_off,
                       fp_stream_t stream);
/* dst[M][dst_off + c] (+)= src[M][src_off + c], c < C: channel-slice copy / add (the identity branch of the concatenation and its gradient) */
int fp_copy_channels(const float* src, float* dst, int64_t M, int32_t C, int32_t srcC, int32_t src_off, int32_t dstC, int32_t dst_off,
                     int accumulate, fp_stream_t stream);

/* ---- misc ---- */
int fp_nchw_to_nhwc(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, fp_stream_t stream);
int fp_nhwc_to_nchw(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, fp_stream_t stream);
int fp_fill(float* x, int64_t n, float value, fp_stream_t stream);

/* ---- loss of the ground-segmentation trainer (footprints/preprocessing/segmentation/train.py:184-193, evaluation.py:39-58) ----------
 * Forward and gradient in one call: preds[s] = logit map of scale s ([B][h_s][w_s], batch stride bstrides[s] elements -- a channel slice
 * of a wider tensor is fine), up-sized bilinearly (align_corners = False) to H x W, per-image masked BCE-with-logits means, averaged over
 * the four scales and the batch.  losses: 5 B + 1 floats = per image the four per-scale means [s * B + b], their average [4 B + b], and
 * the batch mean [5 B]; dpreds (nullable): d losses[5 B] / d preds[s], same addressing.  workspace >= fp_seg_loss_workspace(B, H, W). */
int64_t fp_seg_loss_workspace(int32_t B, int32_t H, int32_t W);
int fp_seg_loss_fwd_bwd(const float* const* preds, float* const* dpreds, const int32_t* hs, const int32_t* ws, const int64_t* bstrides,
                        const float* ground_mask, const float* loss_mask, int32_t B, int32_t H, int32_t W, float* losses, void* workspace,
                        int64_t workspace_bytes, fp_stream_t stream);

/* ---- data-parallel gradient exchange: RCCL over xGMI (the reference is single-GPU, README.md:128,136) -- */
/* One process per GPU.  Rank 0 obtains a unique id (fp_comm_unique_id_bytes() bytes, ncclUniqueId) and hands it to every rank by
 * any host transport; every rank then calls fp_comm_init on its own device (collective: returns when all `world` ranks called it).
 * fp_comm_allreduce_async sums `count` floats in place across the ranks, asynchronously on `stream` -- the caller orders that
 * stream behind the kernels that write the buffer (fp_event_record / fp_event_wait) -- and, while a launch plan records, is also
 * appended to the plan (fp_plan_replay re-issues it).  fp_comm_wait makes `consumer` wait for everything queued on `comm_stream`
 * so far.  RCCL is resolved at run time (dlopen); a box without librccl.so gets FP_EINVAL from these calls and nothing else of
 * the library is affected.  Errors: -100 - ncclResult_t. */
int32_t fp_comm_unique_id_bytes(void);
int fp_comm_unique_id(void* id_out, int32_t cap);
int fp_comm_init(const void* id_bytes, int32_t rank, int32_t world, void** comm_out);
int32_t fp_comm_version(void); /* ncclGetVersion, or -1 */
int32_t fp_comm_count(void* comm); /* ranks of the communicator as RCCL counts them (ncclCommCount) */
int fp_comm_allreduce_async(void* comm, float* buf, int64_t count, fp_stream_t stream);
int fp_comm_broadcast(void* comm, float* buf, int64_t count, int32_t root, fp_stream_t stream);
int fp_comm_wait(void* comm, fp_stream_t comm_stream, fp_stream_t consumer);
int fp_comm_destroy(void* comm);

/* ---- per-kernel timing: HIP events on the launch stream around every kernel launch of the library (bench.py roofline leg) ------ */
/* fp_ktime_begin starts collecting (eager launches and plan replays alike); fp_ktime_end stops, synchronises the device and
 * returns the number of distinct kernel symbols seen; fp_ktime_row(i) gives a symbol's demangled name, launch count and summed
 * event-to-event milliseconds -- the per-kernel quantity of `rocprofv3 --kernel-trace --stats`.  Single host thread. */
int fp_ktime_begin(void);
int32_t fp_ktime_end(void);
int fp_ktime_row(int32_t i, char* name, int32_t name_cap, int64_t* launches, double* total_ms);

/* measurement aid (bench.py --sustain): (shader cycles, constant-rate ticks) of each of the 8 XCDs into out16[xcc * 2 + {0, 1}] (device memory,
 * 16 uint64); the difference of two probes on one stream gives the shader clock the chip sustained in between: d cycles / d ticks x
 * fp_wall_clock_khz().  fp_wall_clock_khz: hipDeviceAttributeWallClockRate of the current device, -1 on error. */
int fp_clock_probe(uint64_t* out16, fp_stream_t stream);
int fp_wall_clock_khz(void);
/* Matrix-pipe probe (round 6; bench.py's roofline leg): ONE launch of 768 workgroups x 4 waves x iters x 24 v_mfma_f32_32x32x16_bf16 and nothing
 * else -- mode 0 constant operands, mode 1 pseudo-random bf16 operands (the operand buses toggle like on real activations).  out: 768 * 256
 * floats of scratch; clk2 (optional): {shader cycles, constant-rate ticks} of workgroup 0 across the launch (clock = cycles / ticks x
 * fp_wall_clock_khz).  fp_mfma_probe_flop(iters) = the FLOPs of one launch.  What it showed: 2.46 PFLOP/s at 2.39 GHz on constant data,
 * 1.85 PFLOP/s at 1.81 GHz on random data (profiles/round6_mfma_sustained_clock.txt) -- the chip's dense bf16 peak is a constant-data figure. */
int fp_mfma_probe(float* out, uint64_t* clk2, int32_t iters, int32_t mode, fp_stream_t stream);
double fp_mfma_probe_flop(int32_t iters);
int fp_version(void);
const char* fp_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* FOOTPRINTS_HIP_H */

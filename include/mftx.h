/*
 * mftx.h -- C ABI of libmftx.so: MI355X (gfx950) kernels for the MFT hot path.
 *
 * Drop-in boundary.  The reference has no FFI for this
 * path -- it calls torch ops from Python -- so each entry point replaces a
 * Python-level function of the reference, cited below, and is what a
 * maintainer would bind from Python with ctypes.  The one
 * native precedent is the optional pybind op
 * MFT/RAFT/alt_cuda_corr/correlation.cpp:19-54 (contiguous device tensors in,
 * RuntimeError on bad input); the conventions here follow it.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm
 *    allocations); nothing is allocated, freed or retained past the call,
 *    except by mftx_raft_create, which keeps the weight POINTERS it is given;
 *  - all tensors are contiguous fp32 unless stated otherwise;
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only
 *    enqueue work, they never synchronise;
 *  - return 0 on success, a negative MFTX_E_* for argument errors, a positive
 *    hipError_t for runtime errors; mftx_last_error_string() describes the last
 *    failure on the calling thread;
 *  - re-entrant; one host thread per device.
 *
 * Layouts ("pixel-major" = NHWC): a map with C channels over P images of h x w
 * cells is stored [P*h*w][ld] with ld >= C floats per cell.
 */
#ifndef MFTX_H
#define MFTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFTX_VERSION 400

#define MFTX_E_ARG (-1)      /* null pointer / non-positive size / unsupported shape */
#define MFTX_E_ALIGN (-2)    /* pointer or leading dimension not 16-byte aligned */
#define MFTX_E_WORKSPACE (-3) /* workspace too small */
#define MFTX_E_STATE (-4)    /* bad handle */

#define MFTX_MAX_CANDIDATES 16

int mftx_version(void);
const char *mftx_last_error_string(void);

/* Optional per-kernel timing (HIP events on the launch stream; bench.py's
 * roofline leg).  Categories: 0 corr volume GEMM, 1 pyramid pooling, 2 lookup,
 * 3 conv implicit GEMM, 4 convf1, 5 glue, 6 convex upsample, 7 chain/select,
 * 8 small-N conv, 9 encoder instance-norm passes, 10 lookup fused into convc1,
 * 11 convf1 + convf2 fused (the flow branch), 12 the encoders' conv GEMMs (booked apart from 3), 13 a SepConvGRU pass as one
 * kernel (mftx_gru_half).
 * work[] = algorithmic flops (0, 3, 4, 8, 11, 12, 13) or bytes (1, 2, 6, 7, 9, 10) booked per launch. */
#define MFTX_PROFILE_CATEGORIES 14
int mftx_profile_begin(void);
int mftx_profile_end(double *ms, double *work, long long *count, int n);

/* ---- a4 + a5: all-pairs correlation volume and its pyramid ---------------
 * Replaces CorrBlock.corr + CorrBlock.__init__ (MFT/RAFT/core/corr.py:14-28,
 * 53-69).  f1, f2: pixel-major feature maps [P][h*w][C] (C % 32 == 0).
 * Level l holds, per pair and query cell i, the h_l x w_l map (h_l = h >> l, w_l = w >> l, floor) of
 * lvl0[p][i][j] = sum_c f1[p][i][c] * f2[p][j][c] / sqrt(C) (fp32 MFMA) resp. the 2x2 means of the level
 * below over the TARGET dims, all four written by ONE launch (the GEMM's epilogue pools).  Storage per
 * query cell, stride[l] floats (mftx_corr_pyramid_layout):
 *   levels 0, 1: 8 x 4-float blocks (x fastest, one 128-byte line each), block grid hb_l x wb_l, element
 *                (y, x) at ((y >> 2) * wb_l + (x >> 3)) * 32 + (y & 3) * 8 + (x & 7); cells of the padded
 *                block grid outside the level are unspecified (level 0: zero);
 *   levels 2, 3: row-major [h_l][w_l], stride rounded up to a multiple of 4 floats.
 * lvl_l must hold P*h*w*stride[l] floats, 16-byte aligned. */
int mftx_corr_pyramid(const float *f1, const float *f2, int P, int C, int h, int w,
                      float *lvl0, float *lvl1, float *lvl2, float *lvl3, void *stream);
/* The same in split-fp16 arithmetic (MFTX_ARITH_SPLIT: three fp16 MFMAs per product, fp32 accumulation; what the
 * refinement engine runs by default).  f2_scratch: P*h*w*C floats, 32-byte aligned -- receives f2 in split form. */
int mftx_corr_pyramid_split(const float *f1, const float *f2, int P, int C, int h, int w,
                            float *lvl0, float *lvl1, float *lvl2, float *lvl3, float *f2_scratch, void *stream);
/* stride[4]: floats per query cell of each level; block_grid[4] = {hb_0, wb_0, hb_1, wb_1}. */
int mftx_corr_pyramid_layout(int h, int w, long long *stride, int *block_grid);

/* ---- a6: multi-scale 9x9 correlation lookup --------------------------------
 * Replaces CorrBlock.__call__ + bilinear_sampler (core/corr.py:30-51,
 * core/utils/utils.py:98-112).  coords: [P*h*w][2] (x, y) interleaved.
 * out: pixel-major [P*h*w][ld_out], channel l*81 + a*9 + b = level l sampled at
 * (x/2^l + a - 4, y/2^l + b - 4), bilinear, zeros outside. r must be 4.
 * lvl0..3 in the layout mftx_corr_pyramid writes. */
int mftx_corr_lookup(const float *lvl0, const float *lvl1, const float *lvl2, const float *lvl3,
                     const float *coords, int P, int h, int w, int r,
                     float *out, int ld_out, void *stream);

/* ---- a6 + first layer of a7, fused: lookup -> convc1 without materialising the 324 features ----------------------
 * Replaces CorrBlock.__call__ (core/corr.py:30-51) TOGETHER WITH its consumer, the first layer of the motion encoder
 * cor = relu(convc1(corr)) (core/update.py:152-153), in the split-fp16 arithmetic (MFTX_ARITH_SPLIT):
 *   out[m][n] = relu(sum_k lookup(m)[k] * W[n][k] + bias[n]),  n < 256,
 * with lookup(m) exactly what mftx_corr_lookup writes for cell m -- it stays in the CU's LDS.
 * wfused: MFTX_LOOKUP_CONVC1_WEIGHT_BYTES bytes (16-byte aligned) filled by mftx_pack_lookup_convc1_weights from
 * convc1's weight in the mftx_conv2d packing, wpk = [256][ld_w] fp32 with ld_w >= 324 (channel l*81 + a*9 + b).
 * out: pixel-major [P*h*w][ld_out] (ld_out >= 256; a multiple of 8 and 32-byte aligned rows when out_split), fp32 or
 * -- out_split != 0 -- the split form of mftx_conv_desc.out_split.  bias: 256 floats, 16-byte aligned. */
#define MFTX_LOOKUP_CONVC1_WEIGHT_BYTES 393216
int mftx_pack_lookup_convc1_weights(const float *wpk, int ld_w, void *wfused, void *stream);
int mftx_corr_lookup_convc1(const float *lvl0, const float *lvl1, const float *lvl2, const float *lvl3,
                            const float *coords, int P, int h, int w, const void *wfused, const float *bias,
                            float *out, int ld_out, int out_split, void *stream);

/* ---- the motion encoder's flow branch, fused: convf1 -> convf2 without materialising the 128 features --------------
 * Replaces flo = relu(convf1(flow)); flo = relu(convf2(flo)) (core/update.py:154-156; 7 x 7, 2 -> 128 and 3 x 3,
 * 128 -> 64, both zero padded) on flow = coords - grid, in the split-fp16 arithmetic (MFTX_ARITH_SPLIT): the 128-channel
 * features of a tile of 8 x 16 cells (and its one-cell halo) stay in the CU's LDS.
 * coords: [P][h*w][2] (x, y) as mftx_corr_lookup.  wflow: MFTX_FLOW_BRANCH_WEIGHT_BYTES bytes (16-byte aligned) filled by
 * mftx_pack_flow_branch_weights from w98 = convf1's weight as [98 = (ky, kx, c)][128] fp32 and w2pk = convf2's weight in the
 * mftx_conv2d packing [>= 64 rows][9 taps][128] fp32.  b1: 128 floats, b2: 64 floats (16-byte aligned).
 * out: the 64 output channels of cell m in SPLIT form (mftx_conv_desc.out_split) at out + m * ld_out floats (ld_out >= 64, a
 * multiple of 8, 32-byte aligned rows).  hx (optional, may be NULL): rows of ld_hx >= 384 floats in split form; the flow
 * itself is written to channels 382, 383 (the tail of the GRU input, core/update.py:160).
 * |flow| must stay below MFTX_SPLIT_LIMIT px (beyond it the outputs of the cells that see it are NaN). */
#define MFTX_FLOW_BRANCH_WEIGHT_BYTES 352256
int mftx_pack_flow_branch_weights(const float *w98, const float *w2pk, void *wflow, void *stream);
int mftx_flow_branch(const float *coords, int P, int h, int w, const void *wflow, const float *b1, const float *b2,
                     float *out, int ld_out, float *hx, int ld_hx, void *stream);

/* ---- a17: on-demand correlation lookup (no stored volume) ------------------------------------------------
 * Replaces AlternateCorrBlock + the optional CUDA op alt_cuda_corr (MFT/RAFT/core/corr.py:72-100,
 * alt_cuda_corr/correlation_kernel.cu:18-119; raft_params.alternate_corr).  mftx_fmap_pyramid average-pools the
 * SECOND feature map three times (pixel-major [P][h_l*w_l][C] per level, 2x2 means, floor sizes);
 * mftx_corr_lookup_ondemand evaluates <f1[cell], f2_l[tap]> / sqrt(C) for the 10x10 taps of every level and blends
 * them exactly like mftx_corr_lookup: same output layout, equal up to fp32 rounding (pooling the features equals
 * pooling the volume over the target dims).  C must be 256, r must be 4. */
int mftx_fmap_pyramid(const float *f2, int P, int C, int h, int w, float *lvl1, float *lvl2, float *lvl3, void *stream);
int mftx_corr_lookup_ondemand(const float *f1, const float *f2l0, const float *f2l1, const float *f2l2, const float *f2l3,
                              const float *coords, int P, int C, int h, int w, int r,
                              float *out, int ld_out, void *stream);

/* ---- a7-a9, a11: one convolution as an fp32-MFMA implicit GEMM -------------
 * Replaces the nn.Conv2d calls of core/update.py (zero "same" padding, bias,
 * optional activation).  Input = up to two pixel-major segments concatenated on
 * the channel axis (c0 % 32 == 0 when c1 > 0).  wpk: weights packed
 * [n_pad][kh*kw][cin_pad] (cin_pad = round_up(c0+c1, 32), n_pad =
 * round_up(N, 128), zero filled), tap index = ky*kw + kx.
 * out[m][n] = out_scale * act(conv + bias + addend[m][n]).  act: 0 none, 1 relu,
 * 2 sigmoid, 3 tanh.  addend (optional, pixel-major [M][ld_addend]) carries a
 * pre-computed partial convolution: the engine uses it to evaluate the
 * iteration-invariant "inp" third of the GRU gate convolutions once per pair. */
typedef struct mftx_conv_desc {
    const float *a0; int lda0; int c0;
    const float *a1; int lda1; int c1;
    const float *wpk; const float *bias;
    float *out; int ldo;
    int P, h, w;         /* M = P*h*w output cells */
    int N;               /* output channels */
    int kh, kw;
    int act;
    float out_scale;
    const float *addend; int ld_addend;
    /* strided / unpadded convolutions (encoders, core/extractor.py): all 0 = stride 1 on the output grid with
     * "same" zero padding.  stride: 1..4; hin, win: input grid per image (0 = h, w); pad_y, pad_x: 0 = k/2,
     * -1 = no padding, else explicit; residual_mode 1: out = relu(act(conv + bias) + addend). */
    int stride, hin, win, pad_y, pad_x, residual_mode;
    /* arithmetic of the products: MFTX_ARITH_F32 = fp32 MFMA (exact fp32 products); MFTX_ARITH_SPLIT = every fp32
     * operand as two fp16 halves, three fp16 MFMAs per product with fp32 accumulation (error of a product
     * <= ~2^-23 relative: the same grade as fp32) -- wpk must then be the output of
     * mftx_split_weights, and all operands below 65504 in magnitude. */
    int arith;
    /* MFTX_ARITH_SPLIT only.  a_split: the A operand(s) are already stored in split form -- every 8 consecutive channels
     * of a row as 32 bytes [hi x 8 | lo x 8] of fp16 (same size and row stride as fp32; c0, c1, lda0, lda1 multiples of 8,
     * rows 32-byte aligned) -- as written by a producer with out_split = 1: the kernel then spends no instruction on
     * splitting.  out_split: write the output in that form (ldo a multiple of 8). */
    int a_split, out_split;
} mftx_conv_desc;
#define MFTX_ARITH_F32 0
#define MFTX_ARITH_SPLIT 1
int mftx_conv2d(const mftx_conv_desc *d, void *stream);
/* The same with the workgroup tile shape forced (tests and micro-benchmarks; -1 = the library's choice, as mftx_conv2d).
 * Every shape gives the same bits.  fp32 MFMA: 0 128x128, 1 128x64, 2 64x64, 3 128x32, 4 64x128, 5 32x32 of 16x16 MFMAs;
 * split arithmetic: 0 128x128 (4 waves), 6 128x128 (8 waves), 9 64x64, 10 128x256, 14 128x192 (13, 15: measurement-only
 * shapes, built with -DMFTX_EXPERIMENTAL_TILES only).  A shape that does not exist for the arithmetic falls back to the
 * arithmetic's default small tile. */
int mftx_conv2d_tile(const mftx_conv_desc *d, int tile, void *stream);

/* ---- a8-a10, tile-resident form: the update block's convolutions whose input tile fits a CU's LDS ------------------
 * The same layer as mftx_conv2d with arith = MFTX_ARITH_SPLIT, a_split = 1 -- a 3 x 3 convolution over 128 channels or a
 * 1 x 5 / 5 x 1 one over 128 or 256 (two segments of 128: c0 = c1 = 128), N = 128 or 256, stride 1, "same" padding, act
 * none or relu, optional bias and pre-activation addend -- by another kernel (csrc/tile_conv.hip): the
 * input of an output tile of 128 cells is loaded into LDS ONCE, the weights stream from L2 into registers; no LDS ring,
 * no barrier in the K loop.  Results agree with mftx_conv2d to fp32 rounding of the K sum (not bit for bit).
 * wtile: N * taps * cin * 4 bytes (16-byte aligned) filled by mftx_pack_tile_conv_weights from the layer's weight in the
 * mftx_conv2d packing wpk = [>= N rows][taps][cin_pad] fp32.  d->wpk is ignored.
 * Round 6: 3 x 3 over 256 channels (c0 = 256, one segment; the motion encoder's convc2 and conv, core/update.py:152-160) -- the
 * tile goes through LDS in TWO channel passes of 128 with the accumulators kept; N = 192, or an even N <= 128 (conv's 126:
 * channels >= N of the 128-wide output rows are left untouched); relu, split-form output, bias required.  Pack with
 * mftx_pack_tile_conv_weights(wpk, 192 | 128, 9, 256, cin_pad, ...). */
int mftx_pack_tile_conv_weights(const float *wpk, int N, int taps, int cin, int cin_pad, void *wtile, void *stream);
int mftx_tile_conv2d(const mftx_conv_desc *d, const void *wtile, void *stream);
/* The occlusion and uncertainty heads (core/update.py:177-214, 17-75: per head conv3x3 712 -> 128, relu, conv3x3 128 -> 2 | 1; the
 * two first layers stacked into one 712 -> 256 layer, the two second layers into one block-diagonal 256 -> 3 layer) as ONE
 * tile-resident kernel + a stencil sum: the 712-channel input tile goes through LDS in five channel passes of 144 with the
 * accumulators kept, relu(. + b1) stays in LDS and is multiplied there with the second layers as a [256 x 27] matrix (T[m][3 tap
 * + o]); out[m][o] = b2[o] + sum_tap T[m + offset(tap)][3 tap + o] (zero padding), o = 0, 1: occlusion logits, 2: log-variance.
 * a_split: the heads' input in split form, 712 channels at a_split + m * lda floats (the concatenation of core/update.py:197);
 * w1pk: the stacked first layers in the mftx_conv2d packing [>= 256 rows][9][cin_pad >= 712]; w2pk: the second layers
 * [>= 3 rows][9][256]; wtile: MFTX_OU_HEADS_WTILE_BYTES, wproj: MFTX_FLOW_HEAD_WEIGHT_BYTES, both 16-byte aligned; b1: 256 floats,
 * b2: 3; T: [P*h*w][27] floats of scratch; out: [P*h*w][ld_out >= 3].  Agrees with mftx_conv2d + the small-N kernel to fp32
 * rounding of the K sums (the channel passes reorder them). */
#define MFTX_OU_HEADS_WTILE_BYTES 6635520
int mftx_pack_ou_heads_weights(const float *w1pk, int cin_pad, const float *w2pk, void *wtile, void *wproj, void *stream);
int mftx_ou_heads(const float *a_split, int lda, int P, int h, int w, const void *wtile, const float *b1, const void *wproj, const float *b2,
                  float *T, float *out, int ld_out, void *stream);
/* One pass of the SepConvGRU (core/update.py:108-123; pass 0: the 1 x 5 convolutions, pass 1: the 5 x 1 ones) as ONE kernel:
 *     z | r = sigmoid(conv_zr([h | motion]) + pre_zr),  q = tanh(conv_q([r * h | motion]) + pre_q),  h <- (1 - z) h + z q
 * with the tile's [h | motion] loaded into LDS once and r * h formed there in place (csrc/tile_conv.hip: gru_half_kernel).  The
 * context features' share of every gate sum and the biases arrive as pre-activation addends (they do not change over the
 * iterations: core/raft.py:146-149).  h_in / motion: split form, 128 channels each; wzr / wq: mftx_pack_tile_conv_weights
 * streams of the gates' columns for [h | motion] (N = 256 with z in channels 0..127, r in 128..255; N = 128), cin = 256, 5 taps;
 * pre_zr [M][256], pre_q [M][128], z [M][128] (scratch), hf_in [M][128] (h in fp32); hf_out / h_out: the new h in fp32 and in split
 * form -- DIFFERENT buffers than hf_in / h_in: a tile's halo cells are its neighbours' outputs, and with more tiles than CUs a late
 * workgroup would otherwise read a halo that is already updated.  Same bits as the two mftx_tile_conv2d-style launches it
 * replaces. */
int mftx_gru_half(const float *h_in, int ld_hin, const float *motion, int ld_mo, const void *wzr, const void *wq, const float *pre_zr,
                  const float *pre_q, float *z, const float *hf_in, float *hf_out, float *h_out, int ld_hout, int P, int h, int w, int pass,
                  void *stream);
/* 1 when the tile-resident kernels fill the current device for a batch of P pairs of h x w cells: whole rounds of 128-cell tiles
 * at least 5/8 full, or at least half a round of 32-cell tiles (the kernels come with 128, 64 or 32 cells per tile, picked per launch:
 * the same bits).  The tile-resident and the ring-buffered kernels differ by fp32 rounding of the K sums, so a caller that needs a
 * pair's result to be independent of the batch it is computed in (a tracker whose batches ramp up, ranks of a sharded job) asks
 * ONCE, for its nominal batch, and pins MFTX_RAFT_OPT_TILE_CONV to 2 or 0 (mft_amd/raft.py does). */
int mftx_tile_conv_fills_chip(int P, int h, int w);
/* The flow head (core/update.py:6-14: delta = conv2(relu(conv1(h))), 3 x 3, 128 -> 256 -> 2) without materialising the 256
 * hidden channels: the tile-resident kernel keeps relu(conv1) of a tile in LDS and multiplies it there with conv2's filter
 * as a [256 x 18] matrix -- T[m][2 tap + o], the partial product of cell m for tap and output o -- and a second small
 * kernel adds the nine shifted terms: delta[m][o] = b2[o] + sum_tap T[m + offset(tap)][2 tap + o] (zero padding), and, when
 * coords is given, coords[m][o] += delta[m][o] (core/raft.py:184).
 * hsplit: h in split form, 128 channels at hsplit + m * ld_h floats; wtile: conv1's weight from mftx_pack_tile_conv_weights
 * (N = 256, 9 taps, cin = 128); wproj: 32768 bytes filled by mftx_pack_flow_head_weights from conv2's weight in the
 * mftx_conv2d packing [>= 2 rows][9][256]; b1: 256 floats, b2: 2; T: [P*h*w][18] floats of scratch; delta: [P*h*w][2]. */
#define MFTX_FLOW_HEAD_WEIGHT_BYTES 32768
int mftx_pack_flow_head_weights(const float *w2pk, void *wproj, void *stream);
int mftx_flow_head(const float *hsplit, int ld_h, int P, int h, int w, const void *wtile, const float *b1, const void *wproj,
                   const float *b2, float *T, float *delta, float *coords, void *stream);
/* packed fp32 weights (n_floats of them, a multiple of 8) -> the split form MFTX_ARITH_SPLIT streams: same size,
 * every 8 consecutive floats replaced by their 8 fp16 high halves and 8 fp16 low halves (x 2048) */
int mftx_split_weights(const float *wpk, void *out, long long n_floats, void *stream);
/* Range guard of MFTX_ARITH_SPLIT.  The high half of an operand is fp16(x): finite for |x| < 65520 (documented limit:
 * 65504, the largest fp16).  Beyond it hi = +-inf, the residual is -+inf and every product the operand takes part in
 * is NaN -- an out-of-range operand surfaces as NaN in the outputs of that cell's receptive field, never as a finite wrong
 * value (tests/test_gpu_kernels.py::test_split_arith_out_of_range_is_nan).  *count (device, caller-zeroed) is incremented
 * by the number of elements with !(|x| < limit), NaN included; limit = INFINITY counts the non-finite ones.  The Python
 * plugin checks every weight tensor with it at load (falls back to MFTX_ARITH_F32 with a logged reason) and, with
 * raft_params.check_finite, the outputs of every refinement (raises). */
#define MFTX_SPLIT_LIMIT 65504.0f
int mftx_count_not_below(const float *x, long long n, float limit, unsigned *count, void *stream);

/* ---- a2, a7-a12: the whole RAFT refinement loop ----------------------------
 * Replaces RAFT.forward from the correlation volume on (core/raft.py:141-226)
 * plus RAFTWrapper's post-processing (MFT/raft.py:57-62), for P image pairs at
 * once.  mftx_raft_create keeps pointers to packed weights (see
 * mft_amd/raft.py:pack_weights for the order). */
typedef struct mftx_raft mftx_raft;
#define MFTX_RAFT_NUM_WEIGHTS 34
int mftx_raft_create(const float *const *weights, int n_weights, mftx_raft **out);
void mftx_raft_destroy(mftx_raft *r);
size_t mftx_raft_workspace_bytes(int P, int h, int w);
/* on != 0: mftx_raft_refine uses the on-demand correlation (a17) instead of the stored pyramid -- no N x N volume in
 * the workspace (29.8 GB per 7-pair 1080p frame), 4-5x more time per lookup at 512x512, about even at 1080p.
 * mftx_raft_workspace_bytes_for gives the workspace size for the handle's current mode. */
int mftx_raft_set_ondemand(mftx_raft *r, int on);
/* Arithmetic of the update block's / OU heads' matrix products (mftx_conv_desc.arith).  split: MFTX_RAFT_NUM_WEIGHTS
 * pointers, the mftx_split_weights form of every weight that feeds a GEMM layer (NULL in the other slots: biases, the
 * 7x7 flow conv and the two N <= 3 output layers stay fp32) -> MFTX_ARITH_SPLIT; split = NULL -> back to
 * MFTX_ARITH_F32 (the state after mftx_raft_create).  The pointers are kept, not copied.  mftx_raft_arith: current mode. */
int mftx_raft_set_split_weights(mftx_raft *r, const void *const *split, int n);
int mftx_raft_arith(const mftx_raft *r);
/* wfused (mftx_pack_lookup_convc1_weights of the engine's convc1 weight; the pointer is kept): with the split arithmetic
 * and the stored pyramid, every iteration then runs lookup + convc1 as the one fused kernel above; the 324 features are
 * materialised on the last iteration only, for the occlusion / uncertainty heads (core/raft.py:199-206).  NULL: off. */
int mftx_raft_set_lookup_fused(mftx_raft *r, const void *wfused);
/* wflow (mftx_pack_flow_branch_weights of the engine's convf1 / convf2 weights; the pointer is kept): with the split
 * arithmetic every iteration then runs the motion encoder's flow branch as the one fused kernel above, in order on the
 * call's stream (no side stream).  NULL: off. */
int mftx_raft_set_flow_fused(mftx_raft *r, const void *wflow);
/* tile: MFTX_RAFT_NUM_WEIGHTS pointers, the mftx_pack_tile_conv_weights form of the weights of the layers that have a
 * tile-resident kernel -- the GRU gates' eight slots (per-iteration and context parts), the first layers of the flow head
 * and of the mask head -- NULL elsewhere (and NULL for a layer to keep on mftx_conv2d's kernel); with the split arithmetic
 * those layers then run on csrc/tile_conv.hip.  The pointers are kept.  tile = NULL: off. */
int mftx_raft_set_tile_weights(mftx_raft *r, const void *const *tile, int n);
/* wproj (mftx_pack_flow_head_weights of the engine's flow_head.conv2 weight; kept): with the flow head's first layer on the
 * tile-resident kernel, both layers run as mftx_flow_head does.  NULL: off. */
int mftx_raft_set_flow_head(mftx_raft *r, const void *wproj);
/* the weight streams of mftx_ou_heads (mftx_pack_ou_heads_weights) for the occlusion + uncertainty heads of the handle; NULL, NULL: off */
int mftx_raft_set_ou_heads(mftx_raft *r, const void *wtile, const void *wproj);
/* Debug payload of RAFT.forward(vis_debug=True) (core/raft.py:159-176, 255-257): trace = (iters + 1) x [P*h*w][2] floats
 * (device, kept) receives coords1 as every iteration finds it and, last, as the final iteration leaves it; NULL: off.  The
 * cost-volume pyramid of the same call stays in the workspace (mftx_raft_workspace_layout_for: lvl0..3). */
int mftx_raft_set_coords_trace(mftx_raft *r, float *trace);
/* Per-handle scheduling options (no global state; defaults are the measured best):
 *   MFTX_RAFT_OPT_FORK      -1 default (flow branch of the motion encoder on a side stream with the split arithmetic), 0 never, 1 always
 *   MFTX_RAFT_OPT_PRESPLIT   1 default (GEMM inputs kept in split form in the workspace), 0 fp32 activations split in registers
 *   MFTX_RAFT_OPT_GROUP      1 default (fp32 MFMA: lookup + convf1 and convc2 + convf2 as grouped launches), 0 one launch per layer
 *   MFTX_RAFT_OPT_FUSE_LOOKUP 1 default (use the fused lookup + convc1 kernel when its weights are set), 0 keep them apart
 *   MFTX_RAFT_OPT_GRAPH      1 default (the launch sequence between the first and the last kernel of mftx_raft_refine -- it
 *                            touches the workspace only -- is captured per (shape, workspace, stream) on its second use and
 *                            replayed as a hipGraph from then on: same kernels, same bits, ~170 launches less host work),
 *                            0 plain launches
 *   MFTX_RAFT_OPT_FUSE_FLOW  1 default (use the fused convf1 + convf2 kernel when its weights are set), 0 keep them apart
 *   MFTX_RAFT_OPT_TILE_CONV  1 default (layers with tile-resident weights set run on that kernel when its tiles of 128 cells come in
 *                            rounds of the chip that are at least 5/8 full -- e.g. 5 to 7 pairs of 512 x 512, 1080p), 2 always,
 *                            0 all on mftx_conv2d's
 *   MFTX_RAFT_OPT_FUSE_HEAD  1 default (both layers of the flow head as mftx_flow_head when its weights are set; with the fused flow
 *                            branch, an iteration's coordinate update is applied by the next iteration's flow-branch kernel instead
 *                            of by a launch of its own), 2 the same with the update always applied by its own kernel, 0 two layers
 *   MFTX_RAFT_OPT_TILE_VOLUME 1 default (split arithmetic: the correlation volume by the tile-resident kernel, csrc/volume_tile.hip),
 *                            0 the ring-buffered GEMM
 *   MFTX_RAFT_OPT_FUSE_GRU   1 default (where the tile-resident layers run: each SepConvGRU pass as ONE kernel, mftx_gru_half), 0 z | r
 *                            gates and candidate + blend as two tile-resident launches.  Same bits either way.
 *   MFTX_RAFT_OPT_TILE_CELLS 0 default (cells per tile of the tile-resident kernels by how the tiles fill the chip: 128, else 64, else
 *                            32), or 128 / 64 / 32 forced.  Same bits whatever the value: a smaller tile is fewer MFMA row tiles per wave.
 *   MFTX_RAFT_OPT_FUSE_OU    1 default (where the tile-resident layers run and mftx_raft_set_ou_heads has been called: the occlusion and
 *                            uncertainty heads as mftx_ou_heads), 0 the 712 -> 256 GEMM on mftx_conv2d + the small-N kernel
 *   MFTX_RAFT_OPT_TILE_CONV2P 1 default (where the tile-resident layers run and their streams are set: convc2 and conv of the motion
 *                            encoder -- 3 x 3 over 256 channels -- tile-resident in two channel passes, csrc/tile_conv.hip
 *                            tile_conv2p_kernel), 0 on mftx_conv2d's ring-buffered kernel.  fp32 rounding of the K sums apart. */
#define MFTX_RAFT_OPT_FORK 0
#define MFTX_RAFT_OPT_PRESPLIT 1
#define MFTX_RAFT_OPT_GROUP 2
#define MFTX_RAFT_OPT_FUSE_LOOKUP 3
#define MFTX_RAFT_OPT_GRAPH 4
#define MFTX_RAFT_OPT_FUSE_FLOW 5
#define MFTX_RAFT_OPT_TILE_CONV 6
#define MFTX_RAFT_OPT_FUSE_HEAD 7
#define MFTX_RAFT_OPT_TILE_VOLUME 8
#define MFTX_RAFT_OPT_FUSE_GRU 9
#define MFTX_RAFT_OPT_TILE_CELLS 10
#define MFTX_RAFT_OPT_FUSE_OU 11
#define MFTX_RAFT_OPT_TILE_CONV2P 12
int mftx_raft_set_option(mftx_raft *r, int option, int value);
/* A device-resident counter (4 bytes, zeroed by the caller) that the last kernel of every mftx_raft_refine* call increments
 * by the number of output pixels with a non-finite flow / occlusion / sigma; null switches it off.  The reference has no
 * such check (MFT/MFT.py:96-107 stores whatever the network returned): with the split arithmetic an activation beyond
 * the fp16 range surfaces as NaN by design (DESIGN.md "Range of the split arithmetic"), and a tracker must not chain
 * through it unnoticed.  No host synchronisation: the caller reads the counter whenever it synchronises anyway. */
int mftx_raft_set_nonfinite_counter(mftx_raft *r, unsigned *counter);
/* Drop every captured graph of the handle (after draining the streams they were launched on).  Graphs are keyed by the
 * workspace address: call it before freeing or replacing a workspace the handle has been run with. */
int mftx_raft_clear_graphs(mftx_raft *r);
/* graphs captured / graph launches so far (tests, bench) */
int mftx_raft_graph_stats(const mftx_raft *r, unsigned long long *captures, unsigned long long *replays);
size_t mftx_raft_workspace_bytes_for(const mftx_raft *r, int P, int h, int w);
/* Byte offsets (19 of them) of the workspace regions lvl0..3, coords1, corr,
 * cor1, corflo, flo1, hx, z, rh, fh, delta, mask, ouin, ouh, ou, flow_lr: after
 * mftx_raft_refine they hold the intermediates of the last iteration (tests). */
int mftx_raft_workspace_layout(int P, int h, int w, size_t *offsets, int n);
/* The same for a handle's current mode.  With MFTX_ARITH_SPLIT the regions that feed GEMMs -- cor1, corflo, flo1, hx,
 * rh, ouin -- hold their values in SPLIT form (mftx_conv_desc.a_split: every 8 channels as
 * [hi x 8 | lo x 8] fp16, value = hi + lo / 2048); the others (corr, z, fh, delta, mask, ouh, ou, coords1, flow_lr) are fp32. */
int mftx_raft_workspace_layout_for(const mftx_raft *r, int P, int h, int w, size_t *offsets, int n);
/* fmap1/fmap2: [P][h*w][256]; net, inp: [P][h*w][128] (tanh / relu already
 * applied).  flow_init (optional, may be NULL): [P*h*w][2] initial flow at 1/8
 * resolution, added to the start coordinates (core/raft.py:153-154).  Outputs are planar and UNPADDED: flow [P][2][H0][W0], occl
 * [P][1][H0][W0] (softmax channel 1), sigma [P][1][H0][W0] (sqrt(exp(u))), with
 * H0 = 8h - pad_top - pad_bottom etc.  packed (optional, may be NULL): the same four values interleaved per
 * pixel, [P][H0][W0][4] = (flow x, flow y, occl, sigma), the right-operand format of mftx_chain_select_packed;
 * with packed given, flow / occl / sigma may all three be NULL (the tracker's hot path needs only packed).
 * flow_lr (optional) [P*h*w][2]. */
int mftx_raft_refine(mftx_raft *r, int P, int h, int w, int iters,
                     const float *fmap1, const float *fmap2, const float *net, const float *inp,
                     const float *flow_init,
                     int pad_left, int pad_right, int pad_top, int pad_bottom,
                     float *flow, float *occl, float *sigma, float *packed, float *flow_lr,
                     void *workspace, size_t workspace_bytes, void *stream);
/* The same with the pairs' maps where they lie: fmap1 / fmap2 / net / inp are arrays of P pointers (host memory), pair b's maps
 * [h*w][256] / [h*w][128] at fmap1[b] etc. -- the P left frames of a tracker step are P cached tensors and its right frame ONE:
 * no gather copy into batch tensors, and a second map that all pairs share (fmap2[b] all equal) is split once.  P <= 16;
 * needs the split arithmetic with the stored, tile-resident correlation volume (otherwise MFTX_E_STATE: use mftx_raft_refine).
 * Same kernels on the same values as mftx_raft_refine on the stacked maps: same bits. */
int mftx_raft_refine_gather(mftx_raft *r, int P, int h, int w, int iters, const float *const *fmap1,
                            const float *const *fmap2, const float *const *net, const float *const *inp,
                            const float *flow_init, int pad_left, int pad_right, int pad_top, int pad_bottom,
                            float *flow, float *occl, float *sigma, float *packed, float *flow_lr_out,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- a3: feature / context encoder (BasicEncoder, core/extractor.py:118-195) ---------
 * Replaces fnet / cnet of RAFT.forward (core/raft.py:122-149) incl. RAFTWrapper's
 * pre-processing (MFT/raft.py:41-48): BGR->RGB, replicate pad to /8, 2x/255-1.
 * weights: 16 (instance_norm = 1, fnet) or 17 (cnet, batch norm folded, head split into
 * tanh / relu halves) x (packed weight, bias), order in mft_amd/ops.py:pack_encoder_weights.
 * img: uint8 [H0][W0][3] BGR (device).  fnet: out0 = [h*w][256]; cnet: out0 = net [h*w][128],
 * out1 = inp [h*w][128]; h = ceil(H0/8), w = ceil(W0/8). */
typedef struct mftx_encoder mftx_encoder;
int mftx_encoder_create(const float *const *weights, int n_weights, int instance_norm, mftx_encoder **out);
/* Arithmetic of the encoder's convolutions (mftx_conv_desc.arith): split = the mftx_split_weights form of the
 * n conv weights, in the order of mftx_encoder_create's (weight, bias) pairs -> MFTX_ARITH_SPLIT; NULL -> fp32 MFMA.
 * The pointers are kept, not copied. */
int mftx_encoder_set_split_weights(mftx_encoder *e, const void *const *split, int n);
/* on (default 1): the layers between the pre-processing kernel and the head -- workspace only -- are replayed as a hipGraph
 * per (image size, workspace, stream) from their third use on (as MFTX_RAFT_OPT_GRAPH); 0: plain launches. */
int mftx_encoder_set_graph(mftx_encoder *e, int on);
void mftx_encoder_destroy(mftx_encoder *e);
size_t mftx_encoder_workspace_bytes(int H0, int W0);
int mftx_encoder_forward(mftx_encoder *e, const uint8_t *img, int H0, int W0, float *out0, float *out1,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- a10 + a12: convex 8x upsampling + post-processing ---------------------
 * Replaces RAFT.upsample_flow (core/raft.py:83-94) for the three heads and
 * MFT/raft.py:57-62.  flow_lr [M][2], ou [M][ld_ou] (occl logit0, logit1,
 * log-variance), mask pixel-major [M][576] (channel k*64 + sy*8 + sx).
 * packed (optional): [P][H0][W0][4] interleaved copy of the outputs (see mftx_raft_refine). */
int mftx_convex_upsample(const float *flow_lr, const float *ou, int ld_ou, const float *mask,
                         int P, int h, int w,
                         int pad_left, int pad_right, int pad_top, int pad_bottom,
                         float *flow, float *occl, float *sigma, float *packed, void *stream);

/* ---- a14: chain_results -----------------------------------------------------
 * Replaces chain_results (MFT/MFT.py:233-239) = FlowOUTrackingResult.chain +
 * 2x warp_backward (MFT/results.py:87-136).  Planar [2|1][H][W] inputs. */
int mftx_chain(const float *flowL, const float *occlL, const float *sigmaL,
               const float *flowR, const float *occlR, const float *sigmaR,
               int H, int W, float *flowO, float *occlO, float *sigmaO, void *stream);

/* ---- a14 (part): FlowOUTrackingResult.warp_backward (MFT/results.py:116-136)
 * out[c] = bilinear sample of img[c] ([C][H][W]) at pixel grid + flow ([2][H][W]),
 * zeros outside, same normalise / un-normalise round trip as the reference. */
int mftx_warp_backward(const float *flow, const float *img, int C, int H, int W, float *out, void *stream);

/* ---- a15: per-pixel best-chain selection -----------------------------------
 * Replaces MFT/MFT.py:112-143 + invalid_mask (MFT/results.py:250-265).
 * Candidates must be ordered [inf, 1, 2, ...]; first arg-max of -sigma with
 * occl > thr scored -inf; occl := 1 where the selected flow leaves the image.
 * chosen (optional) [H][W] int8 candidate index. */
int mftx_select(int K, const float *const *flow, const float *const *occl, const float *const *sigma,
                float thr, int H, int W,
                float *flowO, float *occlO, float *sigmaO, int8_t *chosen, void *stream);

/* ---- a14 + a15 fused: chain every candidate and select in one pass --------- */
int mftx_chain_select(int K,
                      const float *const *flowL, const float *const *occlL, const float *const *sigmaL,
                      const float *const *flowR, const float *const *occlR, const float *const *sigmaR,
                      float thr, int H, int W,
                      float *flowO, float *occlO, float *sigmaO, int8_t *chosen, void *stream);

/* The same with the right operands (left_id -> current flows, fresh from mftx_raft_refine) in the packed
 * per-pixel format [H][W][4] = (flow x, flow y, occl, sigma): each bilinear tap of the chain is one 16-byte
 * gather.  packedR[k] must be 16-byte aligned.  Bitwise equal to mftx_chain_select. */
int mftx_chain_select_packed(int K,
                             const float *const *flowL, const float *const *occlL, const float *const *sigmaL,
                             const float *const *packedR,
                             float thr, int H, int W,
                             float *flowO, float *occlO, float *sigmaO, int8_t *chosen, void *stream);

/* ---- 8f-2: flow-cache codec (".flowouX16" entries) -----------------------------
 * Replaces compress_channel / decompress_channel of write_flowou_X16 / read_flowou_X16
 * (MFT/utils/io.py:495-512, 548-551): per-channel min/max, uint16 quantisation with
 * round-half-even, and its inverse, in float32 exactly as numpy evaluates them.
 *   x [n] float32 (device), q [n] uint16 (device), lohi [2] float32 (device: min, max)
 *   workspace: mftx_quantize_workspace_bytes() bytes of device memory. */
size_t mftx_quantize_workspace_bytes(void);
int mftx_quantize_u16(const float *x, long long n, uint16_t *q, float *lohi,
                      void *workspace, size_t workspace_bytes, void *stream);
int mftx_dequantize_u16(const uint16_t *q, long long n, float lo, float hi, float *x, void *stream);

/* ---- 8f-4: frame / result transport without copy queues -------------------------------------------------------------------
 * A copy KERNEL: src -> dst, n bytes, both 16-byte aligned; either may be PINNED HOST memory (hipHostMalloc / torch
 * pin_memory: mapped into the device's address space), which a kernel reads and writes over PCIe directly.  Unlike
 * hipMemcpyAsync it is ordered like any kernel of `stream` and uses no SDMA queue -- uploads and downloads enqueued this
 * way cannot serialise behind each other (MFT/utils/io.py:566-615 reads frames on the host, MFT/MFT.py:145-148 hands
 * results back to it: every frame crosses PCIe both ways).  16 bytes per lane, coalesced: ~25 GB/s either way.  (Kernels that
 * read host memory in narrow pieces crawl -- the encoders take a device copy of the frame, uploaded with this.) */
int mftx_copy_bytes(const void *src, void *dst, long long n, void *stream);

/* HOST helper of the same codec: reconstruct the scanlines of an inflated, non-interlaced PNG
 * IDAT stream in place (what cv2.imdecode does inside read_flowou_X16, MFT/utils/io.py:541-543).
 * rows: height x (1 + row_bytes) bytes in, height x row_bytes bytes out (compacted to the front);
 * bpp: bytes per pixel. */
int mftx_png_unfilter(uint8_t *rows, int height, int row_bytes, int bpp);

#ifdef __cplusplus
}
#endif
#endif /* MFTX_H */

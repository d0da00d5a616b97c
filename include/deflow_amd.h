/* deflow_amd.h -- C ABI of the MI355X-native DeFlow hot path (libdeflow_amd.so).
 *
 * The reference has no C/FFI plugin interface: its boundary is the Python class contract of
 * `model=deflow` ([REF deflow.py:20-113]) plus two native CUDA ops it builds from
 * `assets/cuda/mmcv` ([REF README.md:38]: dynamic voxelize + dynamic scatter).  This header is
 * what a maintainer binds instead of those ops and instead of the torch/cuDNN calls made by
 * the model blocks; INTEGRATION.md shows the ctypes stub.  Every entry point
 *   - takes raw DEVICE pointers, explicit sizes and a hipStream_t (as void*),
 *   - never allocates, never synchronises, is safe to call from any host thread per stream,
 *   - returns 0 on success, a negative DF_E_* code for a rejected argument, or a positive
 *     hipError_t from the launch.
 * All activations are NHWC ("pixel-major"): element (n,y,x,c) of an image set lives at
 *   ptr + (n % grp_size) * img_stride + (n / grp_size) * grp_off + (y * w + x) * ld + c.
 * A plain [N,H,W,C] tensor has grp_size = N, grp_off = 0, ld = C, img_stride = H*W*C.
 * The (grp_size, grp_off) pair lets the two point clouds' feature maps share one channel-
 * concatenated buffer ([REF deflow.py:93] torch.cat((pc0_img, pc1_img), dim=1)) with no copy.
 */
#ifndef DEFLOW_AMD_H
#define DEFLOW_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DF_OK 0
#define DF_E_SHAPE -1    /* unsupported / inconsistent shape */
#define DF_E_ALIGN -2    /* pointer or stride not 16-byte aligned where required */
#define DF_E_ARG -3      /* bad enum / null pointer */
#define DF_E_WORKSPACE -4

typedef struct df_img {
  void* ptr;
  int32_t n, h, w, c; /* images, rows, cols, channels addressed */
  int32_t ld;         /* elements between consecutive pixels */
  int32_t grp_size;   /* images per group */
  int64_t img_stride; /* elements between images inside a group */
  int64_t grp_off;    /* element offset of group g */
  int32_t elt;        /* element type: 0 = float32 (every entry point), 1 = bfloat16 (bf16-STORAGE training, round 3: only the
                         entry points that say so accept it -- df_conv2d_w16 x / y, df_conv2d_mp y, df_bn_gelu_*_t,
                         df_conv2d_wgrad_bf16; all others return DF_E_ARG), 2 = pre-split fp16x2 "h2" image (round 4: see "PRE-SPLIT
                         tensors" below; only the *_h2p / *_yh2 / *_h2 entry points).  ld / img_stride / grp_off stay in ELEMENTS */
  int32_t reserved;   /* 0 */
} df_img;

int df_version(void);

/* ---------------------------------------------------------------- pillarise (A2-A4) ----
 * Replaces mmcv Voxelization(max_num_points=-1) + DynamicScatter + DynamicPillarFeatureNet +
 * PointPillarsScatter as used by DynamicEmbedder ([REF deflow.py:27-30,82-83]).           */
typedef struct df_pillar_geom {
  float vx, vy, vz;          /* voxel size (fp32, as mmcv's tensors hold it)              */
  float minx, miny, minz;    /* point_cloud_range[:3]                                      */
  float offx, offy, offz;    /* v/2 + min computed in double then rounded (feature net)    */
  int32_t gx, gy, gz;        /* grid = round((max-min)/v); gz must be 1 (pillars)          */
} df_pillar_geom;

/* ---- second generation (csrc/pillar_bands.hip): the path DynamicEmbedder takes.  The grid is cut into bands of R rows
 * (R * gx <= 2048 cells, df_pillar2_rows_per_band); points are bucketed by (sample, band) with a stable counting sort over
 * 1024-point tiles (hist -> scan -> scatter) and each band's workgroup finishes the sort by cell in LDS, runs the feature
 * net and writes its slice of the canvas -- zeros included, every byte once; no separate zero-fill, no library sort.
 * S = number of cloud samples in this call (B, or 2B when both clouds are pillarised together), NB = ceil(gy / R),
 * nblk = ceil(N / df_pillar2_tile()).  Sorted layout: sample s owns sorted positions
 * [sum counts[0..s), + counts[s]), ascending cell key, input order inside a cell; positions past the last valid point are
 * not written. */
int df_pillar2_rows_per_band(int H, int W);   /* 0 if the grid is not supported (W > 2048) */
int df_pillar2_tile(void);
/* hist [S, NB+1, nblk] i32: points of each tile per band; column NB = valid points of the tile */
int df_pillar2_hist(const float* pts, int S, int N, df_pillar_geom g, int rows_per_band, int32_t* hist, void* stream);
/* off (shape of hist): exclusive scan over the tiles of every (sample, column); tot [S, NB+1]; counts [S] = valid points */
int df_pillar2_scan(const int32_t* hist, int S, int ncol, int nblk, int32_t* off, int32_t* tot, int32_t* counts,
                    void* stream);
/* order-preserving compaction (points_c [S,N,3] f32, coords_c [S,N,3] i32 (z,y,x), idx_c [S,N] i64, offs_c [S,N,3] f32, cpos [S*N] i32 = compact position of each original point or -1; rows >= counts[s] untouched) + bucketed key [S*N] u32 / flat index [S*N] u32 / xyz [S*N,3] */
int df_pillar2_scatter(const float* pts, int S, int N, df_pillar_geom g, int rows_per_band, const int32_t* off,
                       const int32_t* tot, float* points_c, int32_t* coords_c, int64_t* idx_c, float* offs_c,
                       int32_t* cpos, uint32_t* bkey, uint32_t* bidx, float* bpts, int32_t* bucket0 /* [S, NB] out */,
                       void* stream);
/* flags: 1 = sort (inputs are the bucketed arrays; writes key_sorted / idx_sorted / pts_sorted), 2 = BatchNorm1d batch
 * statistics partials [S, NB, 32, 2] (finalise with df_pfn_bn_finalize, nblk_stat = NB), 4 = canvas (writes ALL of
 * out [S, gy, gx, 32]).  Valid: 1|4 (inference), 1|2 then 4 (training; the second call takes the SORTED arrays as inputs).
 * cell_rng: optional dense [S*gy*gx, 2] table of sorted [start, end) per cell (written completely). */
int df_pillar2_band(const uint32_t* in_key, const uint32_t* in_idx, const float* in_pts, const int32_t* tot,
                    const int32_t* bucket0 /* from df_pillar2_scatter */, int S,
                    df_pillar_geom g, int rows_per_band, int flags, const float* w_pfn, const float* bn_ss,
                    int bn_sample_stride, int mode, df_img out, uint32_t* key_sorted, uint32_t* idx_sorted,
                    float* pts_sorted, int32_t* cell_rng, float* stats_partial, void* stream);
/* round 5, PERSISTENT canvas: the same canvas-writing call (flags & 4) on a buffer that is zero wherever the previous call with the
 * same (S, grid, rows_per_band, occ) left no pillar -- initially all zero, occ all zero.  Only the occupied cells and the cells that
 * were occupied last time and are empty now are written (the dense form streams 128 B of zeros into every empty cell: 87 % of the
 * stage's bytes); occ [S][bands][64] u32 (one bit per cell of a band) is read and rewritten in place by the band's workgroup.
 * amax_out (optional device scalar, ZERO before the call; several calls into one canvas may share it): receives the maximum of the
 * canvas values written (they are >= 0: the feature net ends in a ReLU) -- the bound an fp16x2 consumer of the canvas scales by.
 * Replaces the zero-initialised canvas of PointPillarsScatter [REF deflow.py:82-83 -> embedder] kept across calls. */
int df_pillar2_band_sp(const uint32_t* in_key, const uint32_t* in_idx, const float* in_pts, const int32_t* tot,
                       const int32_t* bucket0, int S, df_pillar_geom g, int rows_per_band, int flags, const float* w_pfn,
                       const float* bn_ss, int bn_sample_stride, int mode, df_img out, uint32_t* key_sorted,
                       uint32_t* idx_sorted, float* pts_sorted, int32_t* cell_rng, float* stats_partial, uint32_t* occ,
                       float* amax_out, void* stream);

/* (the first-generation forward entry points df_pillar_keys / _scan / _compact / _sort / _gather_sorted / _cells, df_pfn_stats
 * and df_pfn_canvas -- a library radix sort and separate kernels over a zero-filled canvas -- were retired in round 3: the band
 * pipeline above is the only pillariser; its tests compare it with the oracle directly.)
 * finalize (training): per-sample scale/shift/mean/invstd of the feature net's BatchNorm1d from the band kernel's statistics
 * partials (flag 2), sequential running-stat update (one update per sample, as the reference calls feature_net once per sample).
 * counts [B] i32 valid points; bn_ss [B,4,32] f32 = scale, shift, mean, invstd. */
int df_pfn_bn_finalize(float* partial /* clobbered: slot 0 of every sample is reused as scratch */, int B, int nblk_stat, const int32_t* counts, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* bn_ss, void* stream);
/* round 4: the same, also leaving an a-priori bound of max |canvas| (from the statistics, the feature net's weights w_pfn [32][9]
 * and the geometry) in *canvas_bound (integer atomic max: zero-initialise it, or pass one slot for both clouds) */
int df_pfn_bn_finalize2(float* partial, int B, int nblk_stat, const int32_t* counts, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, float* bn_ss, const float* w_pfn, df_pillar_geom g,
                        float* canvas_bound, void* stream);
/* Stable counting sort of caller-supplied cell keys (the stand-alone decoder-head call takes arbitrary voxel_coords
 * [REF decoder.py:185-199], not the pillariser's): idx_sorted [n] u32 = indices of the keys < ncells grouped by key, ascending
 * index inside a group (entries past the number of such keys are not written); cell_rng [ncells, 2] i32 = [start, end) per key,
 * written completely.  Keys >= ncells are dropped.  ws: df_cell_sort_ws_bytes(ncells).  In-tree replacement of the rocPRIM radix
 * sort of rounds 1-2 (hist / offsets / scatter with integer atomics, then each group put in ascending order: deterministic). */
int64_t df_cell_sort_ws_bytes(int64_t ncells);
int df_cell_sort(const uint32_t* key, int64_t n, int64_t ncells, uint32_t* idx_sorted, int32_t* cell_rng, void* ws, void* stream);
/* backward of the feature net over the sorted runs (all pfn kernels iterate over occupied pillars = heads of the sorted key
 * runs of sample b = sorted positions [sum counts[0..b), +counts[b])); bn_sample_stride = 128 (per-sample stats) or 0 (shared:
 * eval); mode (0 'avg': every point of a pillar gets g / count; 1 'max': per channel the pillar's
 * first maximal point gets g, mmcv's traceback rule): pass A partial sums [B,nblk_stat,32,2] of (g_hat, g_hat*xhat); finalize -> dgamma, dbeta,
 * coef [B,2,32] = (S1/M_b, S2/M_b); pass B dW partials [B*nblk_stat,32,9] (sum with df_colsum_finalize). */
int df_pfn_bwd_stats(const float* pts_sorted, const int32_t* cell_rng, const uint32_t* key_sorted,
    const int32_t* counts, int B, df_pillar_geom g,
                     const float* w_pfn, const float* bn_ss, int bn_sample_stride, int mode, df_img gout,
                     float* partial, int nblk_stat, void* stream);
int df_pfn_bwd_finalize(const float* partial, int B, int nblk_stat, const int32_t* counts, float* dgamma,
                        float* dbeta, int accumulate, float* coef, void* stream);
int df_pfn_bwd_weights(const float* pts_sorted, const int32_t* cell_rng, const uint32_t* key_sorted,
    const int32_t* counts, int B, df_pillar_geom g,
                       const float* w_pfn, const float* bn_ss, int bn_sample_stride, int mode, const float* coef,
                       df_img gout, float* dw_partial, int nblk_stat, void* stream);

/* (autograd of backbone(pc0_img, pc1_img) [REF deflow.py:87-88] w.r.t. its inputs, which only DynamicEmbedder's pillar
 * features [REF deflow.py:82-83] consume)
 * Gradient of the pillar canvas [B,H,W,32] of one cloud (cloud 0 = pc0 = channels 0..31 of the network input, cloud 1 =
 * pc1) evaluated ONLY at that cloud's occupied cells -- the only cells df_pfn_bwd_* read.  Adds (accumulate != 0) or
 * writes the two conv consumers of the canvas in FastFlow3DUNet: the first encoder conv (3x3, stride 2, 32 -> 64:
 * dy1 [2B,H/2,W/2,64] with image = cloud * B + b, w1 [64,3,3,32]) and the decoder's 1x1 skip conv on the 64-channel
 * input (dskip [B,H,W,64], w3 [64,64]).  Replaces two dense data-gradient convolutions.  Unoccupied cells are untouched. */
int df_pillar_input_grad(const uint32_t* key_sorted, const int32_t* counts, int B, int H, int W, int cloud,
                         const float* dy1, const float* w1, df_img dskip, const float* w3, df_img dcanvas,
                         int accumulate, int nblk, void* stream);

/* (autograd of the integer gather after_pseudoimage[:, y, x] [REF decoder.py:165-168]: the backbone output is read, and
 * its gradient is non-zero, only at the voxel coordinates of pc0's points)
 * Weight (and bias) gradient of a 3x3 stride-1 64 -> 64 conv whose OUTPUT gradient dy [B,H,W,64] is exactly zero outside
 * the occupied cells of a pillarised cloud (the UNet's last conv: dy comes from the decoder's gather backward): sums only
 * over those cells.  ws [nblk*B][64][9][64] / bias_ws [nblk*B][64] partials; finish with
 * df_conv2d_wgrad_reduce(ws, nblk*B, 64, 9, 64, ...) and df_colsum_finalize(bias_ws, nblk*B, 64, 1, ...). */
int df_sparse_wgrad3x3(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, df_img dy, float* ws,
                       float* bias_ws, int nblk, void* stream);
/* ... with both operands as two bf16 planes (16 significant bits; three v_mfma_f32_16x16x32_bf16 per tile and 32 pixels -- the product of
 * the GRU kernels) instead of the fp32 matrix pipe; x / dy rows 16-byte aligned.  Same arguments and partial layout. */
int df_sparse_wgrad3x3_x2(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, df_img dy, float* ws,
                          float* bias_ws, int nblk, void* stream);

/* The same conv's forward where only those cells of the OUTPUT are consumed (the `after` image is read by the decoder's
 * gather alone): y[p] = bias + conv3x3(x)[p] for the occupied cells p of the pillarised cloud; other cells of y are not
 * written.  x, y [B,H,W,64]; w [64,3,3,64]. */
int df_sparse_conv3x3(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, const float* w,
                      const float* bias, df_img y, int nblk, void* stream);
/* ... on the 16-bit matrix pipe as an fp32-accurate fp16x2 product (three v_mfma_f32_16x16x32_f16 per 32-deep k step, as the dense
 * 3x3 layers: df_conv2d_h2): w2 = the [hi | lo] fp16 planes of w scaled by df_h2_scale(*w_amax) (df_split_h2 / df_weight_prep),
 * x_amax = a bound of max |x| (device scalars).  Same cells, same result class as df_sparse_conv3x3. */
int df_sparse_conv3x3_h2(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, const void* w2,
                         const float* x_amax, const float* w_amax, const float* bias, df_img y, int nblk, void* stream);
/* ... with bf16 operands (the bf16-operand training mode: x and w rounded to bf16 on the way to the matrix pipe, fp32 accumulation). */
int df_sparse_conv3x3_bf16(const uint32_t* key_sorted, const int32_t* counts, int B, df_img x, const float* w,
                           const float* bias, df_img y, int nblk, void* stream);

/* Weight gradient of the first encoder conv (3x3, stride 2, pad 1, 32 -> 64; dy1 [2B,H/2,W/2,64], image = cloud*B + b)
 * summed over the occupied cells of one cloud's canvas [B,H,W,32] only.  ws [nblk*B][64][9][32] partials; finish with
 * df_conv2d_wgrad_reduce(ws, nblk*B, 64, 9, 32, ..., accumulate = cloud). */
int df_sparse_in_wgrad(const uint32_t* key_sorted, const int32_t* counts, int B, int H, int W, int cloud,
                       const float* dy1, df_img canvas, float* ws, int nblk, void* stream);

/* ------------------------------------------------------------- BEV convolutions (A5) ---
 * Replaces torch.nn.Conv2d / BatchNorm2d / GELU / interpolate inside FastFlow3DUNet and
 * ConvWithNorms ([REF decoder.py:202-220]).  fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM. */

#define DF_CONV_FWD 0   /* y[p] = sum_k x[p*stride + k - pad] w[.,k,.]                 */
#define DF_CONV_DGRAD 1 /* y[p] = sum_k x[(p + pad - k)/stride] w[.,k,.] if divisible  */
#define DF_EPI_BIAS 0   /* y = acc (+ bias)                                            */
#define DF_EPI_STATS 1  /* + per-tile per-channel (sum, sumsq) partials for BatchNorm  */
#define DF_EPI_BN_GELU 2 /* y = gelu((acc + bias) * scale[c] + shift[c]) (eval mode)   */

/* w [Cout, k*k, Cin] f32 (OHWI).  bias/scale/shift may be NULL when unused.
 * stats_partial [tiles_m, Cout, 2] with tile_m = df_conv2d_tile_m(rows per stat group, Cout) rows per tile.
 * accumulate != 0: y += result (used to sum gradients arriving from two consumers). */
int df_conv2d(df_img x, const float* w, const float* bias, df_img y, int ksize, int stride, int pad,
              int mode, int epi, const float* scale, const float* shift, float* stats_partial,
              int accumulate, void* stream);
/* Mixed-precision form (training with BASELINE configs[4]'s "bf16 MFMA"): mfma_bf16 != 0 rounds both MFMA operands to bf16
 * as they leave LDS (v_cvt_pk_bf16_f32, round to nearest even) and multiplies on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation; every tensor stays fp32 in memory, epilogues unchanged.  mfma_bf16 == 0 is df_conv2d. */
int df_conv2d_mp(df_img x, const float* w, const float* bias, df_img y, int ksize, int stride, int pad,
                 int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                 int accumulate, int mfma_bf16, void* stream);
int df_conv2d_tile_m(int64_t rows_per_stat_group, int cout); /* row-tile the launcher will pick for DF_EPI_STATS */
/* The bf16-operand conv with PRE-CAST weights: w16 = df_cast_bf16 of the [Cout,kh,kw,Cin] weights (one cast per optimizer
 * step).  Activations in and out stay fp32; the workgroup converts its input halo once on the way into LDS, tiles are bf16 in
 * LDS, every MFMA operand is one 16-byte LDS read.  Exists for the haloed 3x3 stride-1 form only (forward and data gradient,
 * W % 128 == 0, Cin % 32 == 0, Cout % 64 == 0): df_conv2d_w16_ok() == 1 says so for a call, other shapes get DF_E_SHAPE and use
 * df_conv2d_mp.  Results are those of df_conv2d_mp(mfma_bf16 = 1) up to fp32 summation order.
 * [REF decoder.py:202-220 under torch.autocast(bfloat16)] */
int df_conv2d_w16(df_img x, const void* w16, const float* bias, df_img y, int ksize, int stride, int pad, int mode, int epi,
                  const float* scale, const float* shift, float* stats_partial, int accumulate, void* stream);
/* fp32-ACCURATE 3x3 stride-1 convolution on the bf16 matrix pipe (round 3, "bf16x3"): every fp32 operand is the exact sum of
 * three bf16 values, six of the nine bf16 x bf16 products (exact in the fp32 accumulator) reproduce the fp32 product to
 * <= 2^-23 relative -- the same accuracy class as v_mfma_f32_32x32x2_f32 at 16 / 6 of its rate.  w3 = df_split_bf16x3 of
 * the [Cout,3,3,Cin] weights (3 * Cout * 9 * Cin bf16); x, y fp32; arguments / epilogues as df_conv2d.  Replaces the fp32
 * Conv2d of ConvWithNorms / UpsampleSkip [REF decoder.py:205,213] forward and data gradient.  df_conv2d_x3_ok: 1 if the form
 * exists for the call (3x3, stride 1, W % 128 == 0, full 128-row tiles, DMA-addressable tensors; DF_CONV_X3=0 disables). */
int df_conv2d_x3(df_img x, const void* w3, const float* bias, df_img y, int ksize, int stride, int pad, int mode, int epi,
                 const float* scale, const float* shift, float* stats_partial, int accumulate, void* stream);
int df_conv2d_x3_ok(df_img x, df_img y, int ksize, int stride, int mode, int epi);
int df_split_bf16x3(const float* w, void* out3, int64_t n, void* stream);
/* The same convolution through TWO fp16 planes per operand ("fp16x2", round 3; conv_halo_x3_kernel<.., NP = 2>): with power-of-
 * two scales s taken from upper bounds of max|x| and max|w| (device scalars, df_absmax) an fp32 value is x s = hi + lo / 2048 with
 * hi = fp16(x s), lo = fp16((x s - hi) 2048) -- 22 significant bits -- and x w = [hi hi' + (hi lo' + lo hi') / 2048] / (s s') to
 * 3 x 2^-22 relative: THREE fp16 MFMAs (v_mfma_f32_32x32x16_f16, two accumulators) instead of six, still far inside the rounding
 * of the fp32 accumulation over K = 9 Cin terms; elements more than 2^29 below the tensor's maximum lose relative (not absolute)
 * accuracy.  w2 = df_split_h2 of the weights with w_amax = df_absmax of them; shapes: df_conv2d_x3_ok.
 * df_absmax: *amax (a float, ZERO-initialised by the caller) <- max(*amax, max |x|) by an integer atomic max of the bit patterns
 * (exact, order-independent); a bound taken over a larger tensor that contains the operand is as good. */
int df_absmax(df_img x, float* amax, void* stream);
int df_split_h2(const float* w, const float* amax, void* out2, int64_t n, void* stream);
/* y_amax (optional, zero-initialised): the epilogue also leaves max |y| there -- the next fp16x2 kernel's bound, for free.
 * df_conv2d_amax: df_conv2d (the fp32-MFMA kernels: 1x1, stride 2) with the same by-product. */
int df_conv2d_h2(df_img x, const void* w2, const float* x_amax, const float* w_amax, const float* bias, df_img y, int ksize,
                 int stride, int pad, int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                 int accumulate, float* y_amax, void* stream);
int df_conv2d_amax(df_img x, const float* w, const float* bias, df_img y, int ksize, int stride, int pad, int mode, int epi,
                   const float* scale, const float* shift, float* stats_partial, int accumulate, float* y_amax, void* stream);
/* fp16x2 for the convolutions without a haloed form (1x1, stride 2: conv_dma_kernel<.., H2>): the fp32 weights as they are,
 * fragments split in registers; bounds as for df_conv2d_h2 */
int df_conv2d_h2f(df_img x, const float* w, const float* x_amax, const float* w_amax, const float* bias, df_img y, int ksize,
                  int stride, int pad, int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                  int accumulate, float* y_amax, void* stream);
/* ... with the weights also handed over PRE-SPLIT (w2: the [hi | lo] fp16 planes of w s_w from df_split_h2 / df_weight_prep): the
 * weight tiles go to LDS as planes by DMA and no wave splits a weight fragment (conv_dma_kernel<.., H2, BP>).  y.elt = 2: y_bound
 * defines the output planes (as df_conv2d_yh2), else y_amax (optional) receives max |y| (as df_conv2d_h2f).  [REF decoder.py:205-211] */
int df_conv2d_h2f_wp(df_img x, const float* w, const void* w2, const float* x_amax, const float* w_amax, const float* bias, df_img y,
                     const float* y_bound, int ksize, int stride, int pad, int mode, int epi, const float* scale, const float* shift,
                     float* stats_partial, int accumulate, float* y_amax, void* stream);
/* ... for the 1x1 skip convolution of an UpsampleSkip block (forward, bias epilogue; y = the SECOND half of the pre-split concatenation:
 * channels [t.c, 2 t.c) of pixels whose first t.c channels precede y.ptr) that ALSO writes the first half -- the bilinear x2 of
 * t [N, H/2, W/2, C] (fp32, F.interpolate semantics), scaled by *y_bound (>= max |t| too): one kernel writes whole pixels of the
 * concatenation.  Replaces df_upsample2x_h2 + df_conv2d_h2f_wp [REF deflow.py:32 FastFlow3DUNet's UpsampleSkip].  DF_E_SHAPE where
 * the 8-wave DMA kernels with pre-split weights do not cover the call. */
int df_conv2d_h2f_wp_up(df_img x, const float* w, const void* w2, const float* x_amax, const float* w_amax, const float* bias,
                        df_img y, const float* y_bound, df_img t, int align_corners, void* stream);
int df_conv2d_w16_ok(df_img x, df_img y, int ksize, int stride, int mode, int epi);
/* tile variant the launcher picks, as BM * 1000 + BN (for profiling tools) */
int df_conv2d_variant(int64_t rows, int64_t rows_per_stat_group, int cout, int epi);
int df_conv2d_last_dma(void); /* 1 if the previous df_conv2d call launched the LDS-DMA kernel (profiling tools) */
/* BatchNorm2d training statistics from the partials. groups = stat groups (the shared encoder is
 * applied to pc0 then pc1: two calls => two groups, running stats updated in call order).
 * bn_ss [groups,4,C] = scale, shift, mean, invstd.  Optional two-stage reduction for layers with many tiles:
 * scratch [groups, splits, 2, C] doubles and splits > 1 (NULL / <= 1: single stage). */
int df_bn_finalize(const float* partial, int tiles_per_group, int groups, int C, int64_t count_per_group,
                   const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, float* bn_ss, double* scratch, int splits, void* stream);
/* z = gelu(y * scale + shift) ; y plain [n,h,w,C]; imgs_per_group images share one stat group */
int df_bn_gelu_apply(const float* y, const float* bn_ss, int imgs_per_group, df_img z, void* stream);
/* backward of z = gelu(bn(y)): pass 1 partial sums [nblk, C, 2] of (dyh, dyh*xhat) */
int df_bn_gelu_bwd_reduce(df_img dz, const float* y, const float* bn_ss, int imgs_per_group,
                          float* partial, int nblk, void* stream);
/* finalize: dgamma/dbeta += over groups; coef [groups,2,C] = (S1/N, S2/N). partial rows are per group contiguous */
int df_bn_bwd_finalize(const float* partial, int nblk_per_group, int groups, int C, int64_t count_per_group,
                       float* dgamma, float* dbeta, float* coef, void* stream);
/* pass 2: dy = scale * (dyh - c1 - xhat * c2) (plain [n,h,w,C]); dbias partial [nblk, C] */
int df_bn_gelu_bwd_apply(df_img dz, const float* y, const float* bn_ss, const float* coef, int imgs_per_group,
                         float* dy, float* dbias_partial, int nblk, void* stream);
/* generic column sums: out[c] = sum over partial rows (used for conv bias grads etc.) */
/* bf16-STORAGE training (round 3): the same three passes with typed tensors.  *_elt / df_img.elt: 0 = float32, 1 = bfloat16;
 * arithmetic in fp32 registers either way.  df_bn_gelu_bwd_apply_t with dy_elt = 1 takes the bias-gradient column sums from
 * the ROUNDED dy (what the weight- and data-gradient kernels read).  Replaces the BatchNorm2d + GELU of ConvWithNorms and its
 * backward [REF decoder.py:202-220] when activations are kept in bfloat16 (Lightning precision="bf16-mixed" keeps them so). */
/* z_amax / dy_amax (optional, ZERO-initialised float): the pass also leaves max |z| resp. max |dy| there (integer atomic max of the
 * bit patterns, as df_absmax) for the fp16x2 convolution kernels that read the tensor next. */
int df_bn_gelu_apply_t(const void* y, int y_elt, const float* bn_ss, int imgs_per_group, df_img z, float* z_amax, void* stream);
int df_bn_gelu_bwd_reduce_t(df_img dz, const void* y, int y_elt, const float* bn_ss, int imgs_per_group, float* partial, int nblk,
                            void* stream);
int df_bn_gelu_bwd_apply_t(df_img dz, const void* y, int y_elt, const float* bn_ss, const float* coef, int imgs_per_group, void* dy,
                           int dy_elt, float* dbias_partial, int nblk, float* dy_amax, void* stream);
int df_colsum_partial(df_img x, float* partial, int nblk, void* stream);
int df_colsum_finalize(const float* partial, int nblk, int C, int nvals, float* out, int accumulate, void* stream);
/* first stage for very many partial rows: out[g][total] = sum of row group g (rows split evenly into `groups`);
 * finish with df_colsum_finalize(out, groups, ...) */
int df_colsum_stage(const float* partial, int nblk, int total, int groups, float* out, void* stream);
/* weight layout helpers: wt[ci][k][co] = w[co][k][ci] */
int df_weight_transpose(const float* w, float* wt, int cout, int taps, int cin, void* stream);
/* dW[co][k][ci] (row stride ldw elements between co rows... taps*cin when dense) = sum_p dy[p][co] x[p*s+k-pad][ci].
 * ws: splits * Cout * taps * Cin floats, splits = df_conv2d_wgrad_splits(...). */
int df_conv2d_wgrad_splits(df_img x, df_img dy, int ksize, int stride);
/* row_counts (optional, 1x1 with h == 1 only): pixel p is summed only if p % rows_per_seg < row_counts[p / rows_per_seg]
 * (padded per-sample point rows of the decoder).
 * bias_ws (optional) [splits, Cout]: per-split column sums of dy = the conv's bias gradient, produced from the dy tiles
 * the kernel stages anyway (sum its rows with df_colsum_finalize). */
int df_conv2d_wgrad(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits,
                    const int32_t* row_counts, int rows_per_seg, float* bias_ws, void* stream);
int df_conv2d_wgrad_mp(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits,
                       const int32_t* row_counts, int rows_per_seg, float* bias_ws, int mfma_bf16, void* stream);
int df_conv2d_wgrad_reduce(const float* ws, int splits, int cout, int taps, int cin, float* dw, int64_t ld_co,
                           int accumulate, void* stream);
/* the same reduction and the bias partials' (bias_ws [splits, Cout] -> db [Cout], summed in double) in ONE launch:
 * what a ConvWithNorms layer's weight + bias gradient [REF decoder.py:205,213] needs after its split-K pass */
int df_conv2d_wgrad_reduce_bias(const float* ws, int splits, int cout, int taps, int cin, float* dw, int64_t ld_co,
                                int accumulate, const float* bias_ws, float* db, void* stream);
/* fp32-ACCURATE 3x3 stride-1 weight gradient on the bf16 matrix pipe (round 3, "bf16x3", the twin of df_conv2d_x3): fp32 x and
 * dy, each staged element split into three bf16 planes, six exact products per operand pair, transposing LDS reads, fp32
 * accumulation and split-K partials.  splits / ws / bias_ws / df_conv2d_wgrad_reduce as df_conv2d_wgrad_mp.  _ok: 1 if the form
 * exists for the call (3x3, stride 1, W % 32 == 0, DMA-addressable tensors).  [REF decoder.py:205,213] weight gradient. */
int df_conv2d_wgrad_x3(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits, float* bias_ws, void* stream);
int df_conv2d_wgrad_x3_ok(df_img x, df_img dy, int ksize, int stride);
/* its fp16x2 form (wgrad3_x3_kernel<2>): x_amax / dy_amax as for df_conv2d_h2 */
int df_conv2d_wgrad_h2(df_img x, df_img dy, const float* x_amax, const float* dy_amax, int ksize, int stride, int pad, float* ws,
                       int splits, float* bias_ws, void* stream);
/* 1x1 weight gradient of fp32 x and dy with fp16x2 products (round 5, wgrad1_h2_kernel): dW[co, ci] = sum_p dy[p, co] x[p, ci] is a
 * row GEMM over the pixels that reads every operand byte once -- each staged element is split in flight into two scaled fp16 planes
 * (x_amax / dy_amax as for df_conv2d_h2), three MFMAs per product, so that the kernel runs at the memory system's pace instead of the
 * fp32 MFMA's.  _ok: 1 if the form takes the call (same geometry, Cin % 32 == 0, Cout % 64 == 0, DMA-addressable tensors;
 * DF_WGRAD1_H2=0: never); _splits: its split-K count; ws [splits][Cout][Cin], bias_ws [splits][Cout] or NULL, then
 * df_conv2d_wgrad_reduce(_bias) as for df_conv2d_wgrad_mp.  x_amax = dy_amax = NULL: the bf16-MFMA training mode's form (ONE bf16
 * plane per operand, one MFMA per product; also for df_conv2d_wgrad_s2_h2).  [REF decoder.py:205,213] weight gradient of the UNet's 1x1 convolutions
 * (the reference's backbone: scripts/network/models/basic/unet.py UpsampleSkip u1 / u3 through torch autograd). */
int df_conv2d_wgrad1_h2_ok(df_img x, df_img dy);
int df_conv2d_wgrad1_h2_splits(df_img x, df_img dy);
int df_conv2d_wgrad1_h2(df_img x, df_img dy, const float* x_amax, const float* dy_amax, float* ws, int splits, float* bias_ws,
                        void* stream);
/* the 3x3 STRIDE-2 (pad 1) weight gradient of fp32 x and dy with fp16x2 products (round 5, wgrad3s2_h2_kernel): the two
 * downsampling layers' form of df_conv2d_wgrad_h2 -- elements split in flight, input columns de-interleaved in LDS so that a tap's
 * 16 pixels are consecutive rows of the transposing-read image.  _ok / _splits / ws [splits][Cout][9][Cin] / bias_ws / reduce as
 * above (DF_WGRAD_S2_H2=0: never).  [REF decoder.py:205,213] weight gradient of the UNet encoder's stride-2 ConvWithNorms. */
int df_conv2d_wgrad_s2_h2_ok(df_img x, df_img dy);
int df_conv2d_wgrad_s2_h2_splits(df_img x, df_img dy);
int df_conv2d_wgrad_s2_h2(df_img x, df_img dy, const float* x_amax, const float* dy_amax, float* ws, int splits, float* bias_ws,
                          void* stream);
/* bf16-STORAGE training (round 3): the 3x3 stride-1 weight gradient of BFLOAT16 x and dy (df_img.elt = 1 on both; W % 32 == 0):
 * bf16 tiles by LDS-DMA into a four-deep ring, fragments by transposing LDS reads (ds_read_b64_tr_b16), fp32 accumulation and
 * fp32 split-K partials.  splits / ws / bias_ws / df_conv2d_wgrad_reduce exactly as df_conv2d_wgrad_mp.  Replaces the weight
 * gradient autograd takes for the Conv2d of ConvWithNorms [REF decoder.py:205,213] under bf16 autocast. */
int df_conv2d_wgrad_bf16(df_img x, df_img dy, int ksize, int stride, int pad, float* ws, int splits, float* bias_ws, void* stream);
/* bf16 inference convolution (BASELINE configs[4], "bf16 MFMA"): x, w bf16 (NHWC / [Cout,kh,kw,Cin]), fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, epilogue DF_EPI_BIAS or DF_EPI_BN_GELU in fp32, output bf16 (out_f32 = 0) or fp32.
 * df_img element counts (c, ld, strides) are in elements of the respective type; Cin, Cout multiples of 64.
 * df_cast_bf16: y[row][c] = bf16(x[row*ldx + c]) for c < cin, 0 for cin <= c < cout (channel padding). */
int df_conv2d_bf16(df_img x, const void* w, const float* bias, df_img y, int ksize, int stride, int pad, int epi,
                   const float* scale, const float* shift, int out_f32, void* stream);
int df_cast_bf16(const float* x, void* y, int64_t rows, int cin, int ldx, int cout, void* stream);
int df_upsample2x_bf16(df_img x, df_img y, int align_corners, void* stream); /* bf16 in / out, fp32 lerp */
/* bilinear x2 (PyTorch F.interpolate semantics, align_corners selectable), forward and backward */
int df_upsample2x(df_img x, df_img y, int align_corners, void* stream);
int df_upsample2x_bwd(df_img dy, df_img dx, int align_corners, void* stream);

/* ---- PRE-SPLIT ("h2") tensors, round 4: the operands of the fp16x2 kernels written split by their PRODUCERS -------------------
 * df_img.elt = 2: the tensor has the geometry of its fp32 form (4 bytes per element: ld / img_stride / grp_off unchanged), but per
 * pixel and 32-channel chunk the 128-byte line holds [32 x fp16 hi | 32 x fp16 lo] with  x s = hi + lo / 2048,  s = the power of
 * two that puts a BOUND of max |x| into [2^14, 2^15).  The bound is a device scalar known BEFORE the producer writes (BatchNorm
 * statistics, weight norms: below), so nothing synchronises; every consumer takes the SAME scalar.  Requirements: c, ld,
 * img_stride, grp_off multiples of 32, 128-byte aligned base.  What this replaces in the reference: nothing -- it is the storage
 * form of the activations / gradients between `ConvWithNorms` / `UpsampleSkip` layers ([REF decoder.py:202-220]) inside this
 * engine; results stay those of the fp32 tensors to 22 significant bits.
 *   df_conv2d_h2p        df_conv2d_h2 with x and / or y as h2 images (x_amax = the bound that defined x; y_bound defines y)
 *   df_conv2d_yh2        the fp32-input kernels (1x1, stride 2; fp32 MFMA or fp16x2-on-fragments when x_amax / w_amax are given)
 *                        writing an h2 output
 *   df_conv2d_wgrad_h2p  3x3 stride-1 weight gradient of h2 x and dy (LDS-DMA ring, no in-kernel split)
 *   df_bn_finalize2 / df_bn_bwd_finalize2   the finalisations, also leaving the bound of the z / dy their apply pass writes next
 *                        (z_bound / dy_bound: zero-initialised slots, integer atomic max): df_bn_gelu_apply_t with z.elt = 2 and
 *                        df_bn_gelu_bwd_apply_t with dy_elt = 2 take that scalar in their z_amax / dy_amax argument (an INPUT then)
 *   df_upsample2x_h2, df_h2_pack / df_h2_unpack, df_rows_l1max, df_h2_bound   helpers (see csrc/elementwise.hip) */
int df_conv2d_h2p(df_img x, const void* w2, const float* x_amax, const float* w_amax, const float* bias, df_img y, const float* y_bound,
                  int ksize, int stride, int pad, int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                  int accumulate, float* y_amax, void* stream);
int df_conv2d_h2p_ok(df_img x, df_img y, int ksize, int stride, int mode, int epi);
/* the 3x3 stride-1 fp16x2 data gradient whose epilogue also leaves the BatchNorm + GELU backward's partial sums of the layer in front
 * (replaces df_bn_gelu_bwd_reduce's pass over dz and y): see csrc/conv.hip */
int df_conv2d_h2p_dgrad_bn(df_img x, const void* w2, const float* x_amax, const float* w_amax, df_img dz, const float* bn_y,
                           const float* bn_ss, float* bwd_partial, float* dz_amax, void* stream);
int df_conv2d_yh2(df_img x, const float* w, const float* x_amax, const float* w_amax, const float* bias, df_img y, const float* y_bound,
                  int ksize, int stride, int pad, int mode, int epi, const float* scale, const float* shift, float* stats_partial,
                  int accumulate, void* stream);
int df_conv2d_wgrad_h2p(df_img x, df_img dy, const float* x_bound, const float* dy_bound, int ksize, int stride, int pad, float* ws,
                        int splits, float* bias_ws, void* stream);
int df_conv2d_wgrad_h2p_ok(df_img x, df_img dy, int ksize, int stride);
int df_conv2d_wgrad_h2p_splits(df_img x, df_img dy);   /* split-K count for df_conv2d_wgrad_h2p (one resident workgroup per CU) */
int df_bn_finalize2(const float* partial, int tiles_per_group, int groups, int C, int64_t count_per_group, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* bn_ss,
                    double* scratch, int splits, const float* y_amax, float* z_bound, void* stream);
int df_bn_bwd_finalize2(const float* partial, int nblk_per_group, int groups, int C, int64_t count_per_group, float* dgamma,
                        float* dbeta, float* coef, const float* bn_ss, const float* dz_amax, const float* y_amax, float* dy_bound,
                        void* stream);
int df_upsample2x_h2(df_img x, df_img y, int align_corners, const float* y_bound, void* stream);
int df_h2_pack(df_img x, const float* bound, df_img y, void* stream);
int df_h2_unpack(df_img x, const float* bound, df_img y, void* stream);
int df_rows_l1max(const float* w, int rows, int row_len, const float* bias, int nbias, float* l1max, float* bmax, void* stream);
int df_h2_bound(float* out, const float* a, const float* l1, const float* b, const float* other, float slack, void* stream);
/* all convolution layers' per-step weight forms in ONE launch: transposed weights, [hi | lo] fp16 planes of both, row L1 norms and
 * max |bias| for the a-priori bounds (replaces ~80 df_weight_transpose / df_split_h2 / df_rows_l1max launches per training step).
 * table: nlayers records {int64 w_off, b_off (elements in `params`; b_off < 0: none), wt_off, w2_off, wt2_off (floats in `out`);
 * int32 cout, taps, cin, split, blk0 (first workgroup of the layer = sum of cout + cin of the layers before), pad};
 * norms [nlayers][3] = (max row L1 of w, of w^T, max |bias|), zero-initialised by the caller; w_amax = max |p| over the arena */
int df_weight_prep(const float* params, const void* table, int nlayers, int total_blocks, const float* w_amax, float* out, float* norms,
                   void* stream);

/* ------------------------------------------------------- point decoder (A6-A10) --------
 * Replaces ConvGRUDecoder / LinearDecoder forward_single ([REF decoder.py:72-199]): integer
 * gather of the 2x64 pillar vectors, offset encoder, num_iters GRU steps, MLP head.        */
typedef struct df_gru_weights {
  const float* w_off; const float* b_off;   /* [64,3], [64]                       */
  const float* w_zr;  const float* b_zr;    /* [256,192] (z rows then r rows), [256] */
  const float* w_q;   const float* b_q;     /* [128,192], [128]                   */
  const float* w_1;   const float* b_1;     /* [32,192], [32]                     */
  const float* w_2;   const float* b_2;     /* [3,32], [3]                        */
} df_gru_weights;
/* before/after: n=B images of 64 channels.  coords [B,N,3] i32 (z,y,x), offs [B,N,3], counts [B] (device).
 * flow [B,N,3] (rows >= counts[b] untouched).  save (training, may be NULL):
 *   [5, T, B*N, 128] = h_in, z, r, q, r*h per iteration, then hT [B*N,128]. */
int df_gru_decoder_fwd(df_img before, df_img after, const int32_t* coords, const float* offs,
                       const int32_t* counts, int B, int N, int num_iters, df_gru_weights wts,
                       float* flow, float* save, void* stream);
/* mixed-precision form (Trainer(dtype="bf16")): mfma_bf16 != 0 rounds the operands of every gate / head GEMM to bf16 on the
 * way into v_mfma_f32_16x16x32_bf16; hidden state, gate math and accumulators stay fp32.  The saved planes h_in, z, r, q, r*h
 * (and the gate-gradient planes the backward puts in their place) are then stored as bf16 -- 128 values in the first 256 bytes
 * of each row's 512-byte slot, offsets as in the fp32 layout -- which halves the dominant HBM stream of the three kernels; hT
 * stays fp32.  df_gru_decoder_fwd_mp, df_gru_decoder_bwd_mp and df_gru_wgrad_mp of one step must get the same mfma_bf16 (zero or
 * not).  mfma_bf16 == 2 (forward and backward): the GEMM weights -- wts.w_zr, wts.w_q, wts.w_1 and wtt.wt_zr, wtt.wt_q, wtt.wt_1
 * -- point at bf16 COPIES of the same [rows, cols] arrays (cast once per optimizer step); their tiles are then bf16 in LDS and
 * every weight fragment is one 16-byte read.  All other fields of the weight structs stay fp32.
 * mfma_bf16 == 3 (forward and backward; round 3, the default of fp32 training): "bf16x2" -- both operands of every gate / head
 * GEMM as two bf16 planes (hi + lo: 16 significant bits), three MFMAs per product; planes saved fp32 exactly as with 0 (the
 * weight-gradient pass gets 0).  The same six weight pointers then hold rows of [hi (cols) | lo (cols)] bfloat16
 * (df_split_bf16x2_rows of the [rows, cols] fp32 arrays, once per optimizer step). */
int df_split_bf16x2_rows(const float* w, void* out, int64_t rows, int ld, void* stream);
int df_gru_decoder_fwd_mp(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts,
                          int B, int N, int num_iters, df_gru_weights wts, float* flow, float* save, int mfma_bf16,
                          void* stream);
typedef struct df_gru_weights_t {          /* transposed copies (df_weight_transpose) for the data-gradient GEMMs */
  const float* wt_zr; /* [192,256] */
  const float* wt_q;  /* [192,128] */
  const float* wt_1;  /* [192,32]  */
} df_gru_weights_t;
/* backward data pass.  Consumes dflow [B,N,3] and the forward's `save`; overwrites save's z, r, q planes with
 * dz_pre, dr_pre, dq_pre (inputs of the weight-gradient GEMMs); writes dh0 [B*N,128], dx [B*N,64], dpre1 [B*N,32],
 * xout [B*N,64] (rows of valid points only), and per-workgroup partial sums of every small gradient:
 * partial [B * ceil(N/64), 772] = d b_z|b_r|b_q (384) | d b_1 (32) | dW_off[64][3] (192) | d b_off (64) | dW_2[3][32] (96) |
 * d b_2 (3) | pad.  `partial` must be zero-filled (workgroups with no valid point do not write); sum its rows with
 * df_colsum_finalize. */
int df_gru_decoder_bwd(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters,
                       df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0, float* dx, float* dpre1,
                       float* xout, float* partial, void* stream);
int df_gru_decoder_bwd_mp(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters,
                          df_gru_weights wts, df_gru_weights_t wtt, float* save, float* dh0, float* dx, float* dpre1,
                          float* xout, float* bias_partial, int mfma_bf16, void* stream);
/* Weight gradients of the three GRU gate convolutions in one streaming pass over the planes df_gru_decoder_fwd saved
 * and df_gru_decoder_bwd overwrote (replaces autograd's six 1x1 conv weight-gradient GEMMs over [REF decoder.py:139-147]):
 * ws[split][384][192] partial tiles, rows 0..127 dW_z, 128..255 dW_r, 256..383 dW_q, columns [h | x]; sum the splits
 * with df_conv2d_wgrad_reduce(ws, splits, 384, 1, 192, ...).  x = the [B*N,64] offset encoding df_gru_decoder_bwd wrote. */
int df_gru_wgrad_splits(void);
/* round 4: the head's first-layer weight gradient dW1 [32][192] = dpre1^T [hT | x] over the valid rows in one streaming pass
 * (partials ws [nsplit][32][192], summed by df_conv2d_wgrad_reduce); bf16x2 products.  [REF decoder.py:151-153,182] differentiated */
int df_gru_head_wgrad(const float* dpre, const float* hT, const float* x, const int32_t* counts, int B, int N, float* ws, int nsplit,
                      void* stream);
int df_gru_wgrad(const float* save, const float* x, const int32_t* counts, int B, int N, int num_iters, float* ws,
                 int nsplit, void* stream);
int df_gru_wgrad_mp(const float* save, const float* x, const int32_t* counts, int B, int N, int num_iters, float* ws,
                    int nsplit, int mfma_bf16, void* stream);

/* ---- round 5: the "lean" ConvGRU decoder (csrc/decoder4.hip) -- the engine's default; the entries above remain for A/B ----
 * Same computation as df_gru_decoder_fwd_mp / _bwd_mp / df_gru_wgrad_mp ([REF decoder.py:123-199] and its derivative), two changes:
 *  (1) x = W_off o + b_off [REF decoder.py:172] is affine in the point's 3 offsets and enters every gate and the head linearly
 *      [REF decoder.py:126-139,151,182], so its contribution is evaluated from a [416][4] table (rows z | r | q | head layer 1:
 *      W[:, 128:] W_off | W[:, 128:] b_off + b) as 3 FMAs per value -- df_gru_xtab builds it from the fp32 parameters once per
 *      optimizer step -- and backwards every x-side product collapses onto [416][4] sums S (plain and offset-weighted column sums of
 *      the gate / head pre-activation gradients) which df_gru_lean_finalize turns into dW[:, 128:], d b, dW_off, d b_off;
 *  (2) the forward saves only the hidden state entering each iteration and h_T, hsave [T + 1][B*N][128] (bf16 modes: planes
 *      0 .. T-1 as bf16 half rows); the backward recomputes z, r, q and writes gplanes [4][T][B*N][128] = dz_pre | dr_pre | dq_pre |
 *      r * h for df_gru_lean_wgrad, which multiplies the 128 h columns only: ws [nsplit][384][128] (reduce with
 *      df_conv2d_wgrad_reduce(ws, nsplit, 384, 1, 128, dW, 192, ...)).  df_gru_lean_head_wgrad: ws [nsplit][32][128] = dpre1^T hT.
 * wts: w_zr, w_q, w_1 (and wtt.*) in the form mfma_bf16 selects, exactly as for the _mp entries; w_2, b_2 fp32; the bias and
 * offset-encoder fields are read by df_gru_xtab / df_gru_lean_finalize only (which take the fp32 struct).
 * partial: [B * ceil(N / 64)][df_gru_lean_partial_width()] zero-filled; its column sums = S [416][4] | dW_2 [3][32] | d b_2 [3] | pad.
 * df_gru_lean_finalize: dW_gates [384][192] and dW1 [32][192] get their columns 128..191; db [416] = d b_z | d b_r | d b_q | d b_1. */
int df_gru_xtab(df_gru_weights wts, float* xtab, void* stream);
int df_gru_lean_partial_width(void);
int df_gru_lean_fwd(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts, int B, int N,
                    int num_iters, df_gru_weights wts, const float* xtab, float* flow, float* hsave, int mfma_bf16, void* stream);
int df_gru_lean_bwd(const float* dflow, const float* offs, const int32_t* counts, int B, int N, int num_iters, df_gru_weights wts,
                    df_gru_weights_t wtt, const float* xtab, const float* hsave, float* gplanes, float* dh0, float* dpre1,
                    float* partial, int mfma_bf16, void* stream);
int df_gru_lean_wgrad(const float* hsave, const float* gplanes, const int32_t* counts, int B, int N, int num_iters, float* ws,
                      int nsplit, int mfma_bf16, void* stream);
int df_gru_lean_head_wgrad(const float* dpre, const float* hT, const int32_t* counts, int B, int N, float* ws, int nsplit,
                           void* stream);
int df_gru_lean_finalize(const float* sums, df_gru_weights wts, float* dW_gates, float* dW1, float* dW_off, float* db_off, float* db,
                         void* stream);
/* gather backward without atomics: every BEV cell sums the dh0 rows of its own pc0 points (cell_rng / idx_sorted /
 * cpos from the pillarise step).  dbefore / dafter (64 ch each) are fully written (zeros for empty cells) or,
 * with accumulate_* != 0, added to.  dbefore.ptr == NULL skips the `before` image. */
int df_gather_bwd(const float* dh0, const uint32_t* idx_sorted, const int32_t* cell_rng, const int32_t* cpos,
                  int B, int N, df_img dbefore, df_img dafter, int accumulate_before, int accumulate_after,
                  int nblk, void* stream);
/* the same, and amax_after (a ZEROED device scalar) receives max |dafter| of what the call writes: the bound the UNet backward's first
 * fp16x2 data gradient scales dv by, without a pass over the image (round 5).  dafter is written, never added to.  Returns DF_E_SHAPE
 * where only the one-cell-at-a-time kernel applies (DF_GATHER_BWD_V1, > 2^29 points): call df_gather_bwd + df_absmax then. */
int df_gather_bwd_m(const float* dh0, const uint32_t* idx_sorted, const int32_t* cell_rng, const int32_t* cpos,
                    int B, int N, df_img dbefore, df_img dafter, int accumulate_before, int nblk, float* amax_after, void* stream);
/* partial[blk][i*nb+j] = sum over valid rows of a[row][i] * (b ? b[row][j] : 1); row r is valid iff
 * r % rows_per_seg < counts[(r / rows_per_seg) % nseg].  na*nb <= 256.  Sum with df_colsum_finalize. */
int df_small_outer(const float* a, int lda, int na, const float* b, int ldb, int nb, const int32_t* counts,
                   int rows_per_seg, int nseg, int64_t rows, float* partial, int nblk, void* stream);
/* ConvGRUDecoder forward with the gate GEMMs and the first head layer on bf16 MFMA (inference, BASELINE configs[4]):
 * w_zr [256,192], w_q [128,192], w_1 [32,192] bf16; everything else (images, biases, offset encoder, last layer, state,
 * gate non-linearities, accumulation) fp32. */
int df_gru_decoder_fwd_bf16(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts,
                            int B, int N, int num_iters, const float* w_off, const float* b_off, const void* w_zr,
                            const float* b_zr, const void* w_q, const float* b_q, const void* w_1, const float* b_1,
                            const float* w_2, const float* b_2, float* flow, void* stream);
/* LinearDecoder ([REF decoder.py:72-120]) forward: w_off [128,3], w_1 [32,256], w_2 [3,32] */
int df_linear_decoder_fwd(df_img before, df_img after, const int32_t* coords, const float* offs,
                          const int32_t* counts, int B, int N, const float* w_off, const float* b_off,
                          const float* w_1, const float* b_1, const float* w_2, const float* b_2,
                          float* flow, void* stream);

/* LinearDecoder backward data pass (recomputes gather + hidden layer).  wt_1 = W1^T [256,32].  Writes
 * vx [B*N,256] = [before|after|offset_enc] rows, dh0 [B*N,128], dxe [B*N,128] (grad of the offset encoding),
 * dpre1 [B*N,32], hid [B*N,32]; weight gradients follow with df_conv2d_wgrad / df_small_outer, image gradients
 * with df_gather_bwd. */
int df_linear_decoder_bwd(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts,
                          const float* dflow, int B, int N, const float* w_off, const float* b_off, const float* w_1,
                          const float* b_1, const float* w_2, const float* wt_1, float* vx, float* dh0, float* dxe,
                          float* dpre1, float* hid, void* stream);

/* ------------------------------------------------------------ ego motion + loss --------*/
/* [REF deflow.py:60-77]: pc0' = pc0 R^T + t ; pose_flow = pc0' - pc0.  T [B,4,4] row-major. */
int df_ego_transform(const float* pc0, const float* T, int B, int N, float* pc0_t, float* pose_flow, void* stream);
/* deflowLoss over padded per-sample rows (rows < counts[b]): partial [B,nblk,3,2] (sum err, count);
 * finalize -> bins [B,3,2], loss[0] = sum_b sum_bins mean; bwd -> dest = gscale * (*gscale_dev) * dloss/dest. */
int df_deflow_loss_fwd(const float* est, const float* gt, const int32_t* counts, int B, int N,
                       float* bins_partial, int nblk, void* stream);
int df_deflow_loss_finalize(const float* bins_partial, int B, int nblk, float* bins, float* loss, void* stream);
int df_deflow_loss_bwd(const float* est, const float* gt, const int32_t* counts, int B, int N, const float* bins,
                       const float* gscale_dev /*nullable*/, float gscale, float* dest, int nblk, void* stream);
/* the ablation losses of the reference's loss_fn switch (round 5; [REF assets/slurm/1_train.sh:53-60], README.md:68): kind 0 = ff3dLoss
 * (weight 0.1 on background points -- class cls[b, idx_c[b, i]] == 0 of compact row i; cls [B, Ncls] int64 -- 1.0 elsewhere), kind 1 =
 * zeroflowLoss (weight clamp(1.8 * 10 |gt| - 0.8, 0.1, 1.0); cls / idx_c unused): sum over samples of the mean over valid rows of
 * w |est - gt|.  fwd -> partial [B, nblk, 2]; finalize -> bins [B, 2] = (sum w err, rows), loss[0]; bwd -> dest as df_deflow_loss_bwd. */
int df_wloss_fwd(const float* est, const float* gt, const int32_t* counts, int B, int N, int kind, const int64_t* cls,
                 const int64_t* idx_c, int Ncls, float* partial, int nblk, void* stream);
int df_wloss_finalize(const float* partial, int B, int nblk, float* bins, float* loss, void* stream);
int df_wloss_bwd(const float* est, const float* gt, const int32_t* counts, int B, int N, int kind, const int64_t* cls,
                 const int64_t* idx_c, int Ncls, const float* bins, const float* gscale_dev /*nullable*/, float gscale, float* dest,
                 int nblk, void* stream);
/* trainer gt: gt[b,i] = flow[b, idx_c[b,i]] - pose_flow[b, idx_c[b,i]] for i < counts[b] */
int df_gather_gt(const float* flow, const float* pose_flow, const int64_t* idx_c, const int32_t* counts,
                 int B, int N, float* gt, int nblk, void* stream);

/* ------------------------------------------------------------------ optimiser (A12) ----
 * torch.optim.Adam (defaults: no amsgrad, no weight decay) over ONE flat fp32 arena holding every
 * parameter; grad/exp_avg/exp_avg_sq are arenas of the same layout.  n % 4 == 0. */
int df_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                 float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/* the same step with the step number read from device memory (*step_dev >= 1, incremented by the caller before the
 * launch): the form that can be captured in a HIP graph and replayed -- nothing about the step is a host-side constant */
int df_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, const int32_t* step_dev, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif

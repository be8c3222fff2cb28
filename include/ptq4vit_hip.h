/*
 * ptq4vit_hip.h -- C ABI of the MI355X (gfx950) PTQ4ViT calibration engine.
 *
 * The reference (hahnyuan/PTQ4ViT) has no FFI: its boundary is a Python nn.Module protocol
 * (SURVEY.md s8-b1).  This header is the boundary the reference's hot path would bind if its
 * `calibration_step2()` bodies were replaced by native code: one entry point per hot class,
 * called from `quant_layers.*.calibration_step2()` through ctypes.
 *
 * Conventions
 *   - plain C symbols, POD descriptors, no torch / C++ types in any signature;
 *   - every pointer named d_* is DEVICE memory owned by the caller (a torch tensor's data_ptr());
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing allocates: scratch comes
 *     from the caller-provided workspace (size from the matching *_workspace_bytes()).  The *_calibrate entry points
 *     read the interval selected by each search pass back to the host (pass memoisation: a pass whose input
 *     interval was already evaluated is skipped) and therefore synchronise `stream` after every pass; setting
 *     desc.reserved bit 1 (or requesting score tables) disables that and makes the call fully asynchronous.
 *     *_quant_forward, p4v_quantize_i8 and p4v_fake_quant never synchronise;
 *   - return value 0 = ok, <0 = error; p4v_last_error() returns a per-thread message;
 *   - safe to call concurrently on different devices / streams (no global mutable state).
 *
 * Data layout (all fp32, row-major, K contiguous unless strides are given):
 *   Linear  x[M][K], weight[N][K], bias[N], out/grad[M][N]        (M = batch*tokens)
 *   MatMul  A[Z][M][K] / B[Z][K][N] given with element strides, out/grad[Z][M][N], Z = batch*heads
 *   Conv2d  x[b][ic][H][W], weight[oc][ic][kh][kw], out/grad[b][oc][fh][fw]
 */
#ifndef PTQ4VIT_HIP_H
#define PTQ4VIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P4V_VERSION 120 /* 0.1.2: + granular entry points (amax_init / search_* / score_argmax_gather) */

/* similarity metrics: reference quant_layers/linear.py:399-424 */
enum p4v_metric {
    P4V_METRIC_L1_NORM = 0,
    P4V_METRIC_L2_NORM = 1,
    P4V_METRIC_LINEAR_WEIGHTED_L2 = 2,
    P4V_METRIC_SQUARE_WEIGHTED_L2 = 3,
    P4V_METRIC_HESSIAN = 4,
    P4V_METRIC_COSINE = 5
};

enum p4v_status {
    P4V_OK = 0,
    P4V_ERR_INVALID = -1,     /* bad descriptor / null pointer              */
    P4V_ERR_UNSUPPORTED = -2, /* configuration not implemented on the GPU   */
    P4V_ERR_WORKSPACE = -3,   /* workspace too small                        */
    P4V_ERR_HIP = -4          /* a HIP runtime call failed                  */
};

int p4v_version(void);
const char* p4v_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Linear: replaces PTQSLBatchingQuantLinear.calibration_step2 (quant_layers/linear.py:536-555)
 * and PostGeluPTQSLBatchingQuantLinear (linear.py:557-642, `twin_postgelu` = 1).
 * ---------------------------------------------------------------------------------------- */
typedef struct p4v_linear_desc {
    int32_t batch;        /* raw_input.shape[0]                                   */
    int32_t tokens;       /* product of the middle dims (1 for the 2-D head case) */
    int32_t in_features;  /* K */
    int32_t out_features; /* N */
    int32_t n_V, n_H, n_a;
    int32_t w_bit, a_bit;
    int32_t metric;       /* enum p4v_metric */
    int32_t eq_n;         /* searched candidates 0..eq_n-1 of an (eq_n+1)-entry table */
    int32_t search_round;
    int32_t twin_postgelu;
    int32_t init_layerwise;
    int32_t has_bias;
    int32_t reserved;     /* bit 0: force the generic fp32-operand path; bit 1: disable pass memoisation;
                             bit 2: p4v_linear_workspace_bytes sizes the workspace for p4v_linear_quant_forward only */
} p4v_linear_desc;

size_t p4v_linear_workspace_bytes(const p4v_linear_desc* desc);

/*
 * d_mult          [eq_n+1] candidate multipliers alpha + i(beta-alpha)/eq_n rounded to fp32 (linear.py:544)
 * d_w_interval    [n_V*n_H] out: calibrated weight intervals (reference shape n_V,1,n_H,1)
 * d_a_interval    [n_a]     out: calibrated activation intervals (reference shape n_a,1)
 * d_scores        optional (may be NULL): [search_round][2][eq_n][n_V] score tables, w-search then a-search
 *                 (a-search uses only column 0) -- the tables the reference feeds to argmax.
 * d_best          optional: [search_round][2][n_V] selected candidate indices (int32)
 */
int p4v_linear_calibrate(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias,
                         const float* d_x, const float* d_out, const float* d_grad, const float* d_mult,
                         float* d_w_interval, float* d_a_interval, float* d_scores, int32_t* d_best,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * MatMul: replaces PTQSLBatchingQuantMatMul.calibration_step2 (quant_layers/matmul.py:565-576)
 * and SoSPTQSLBatchingQuantMatMul (matmul.py:633-644, `sos` = 1: split-of-softmax on A).
 * Head-wise intervals (n_G = heads, matmul.py:411-417); n_V = n_H = 1 (all shipped configs).
 * ---------------------------------------------------------------------------------------- */
typedef struct p4v_matmul_desc {
    int32_t batch, heads;
    int32_t M, K, N;            /* A: (batch,heads,M,K)  B: (batch,heads,K,N) */
    int64_t a_stride[4];        /* element strides of A for (batch, head, m, k) */
    int64_t b_stride[4];        /* element strides of B for (batch, head, k, n) */
    int32_t A_bit, B_bit;
    int32_t metric;
    int32_t eq_n;
    int32_t search_round;
    int32_t sos;
    int32_t init_layerwise;
    int32_t reserved;           /* bit 1: disable pass memoisation; bit 2: workspace query for quant_forward only */
} p4v_matmul_desc;

size_t p4v_matmul_workspace_bytes(const p4v_matmul_desc* desc);

/*
 * d_A_interval [heads] out (for sos: [1] = split/(qmax-1));  d_B_interval [heads] out;
 * d_split      [1] out (sos only, else may be NULL)
 * d_scores     optional: [search_round][2][eq_n][heads]; the sos split search uses rows 0..19, column 0
 */
int p4v_matmul_calibrate(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                         const float* d_grad, const float* d_mult, float* d_A_interval, float* d_B_interval,
                         float* d_split, float* d_scores, int32_t* d_best, void* d_workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conv2d: replaces ChannelwiseBatchingQuantConv2d.calibration_step2 (quant_layers/conv.py:591-603,
 * `channelwise` = 1) and BatchingEasyQuantConv2d (conv.py:429-441, `channelwise` = 0). groups == 1.
 * ---------------------------------------------------------------------------------------- */
typedef struct p4v_conv_desc {
    int32_t batch, in_channels, height, width;
    int32_t out_channels, kernel_h, kernel_w;
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int32_t w_bit, a_bit;       /* a_bit >= 32 disables input quantisation (conv.py:544,600) */
    int32_t metric;
    int32_t eq_n;
    int32_t search_round;
    int32_t channelwise;
    int32_t init_layerwise;
    int32_t has_bias;
    int32_t reserved;
} p4v_conv_desc;

size_t p4v_conv_workspace_bytes(const p4v_conv_desc* desc);

/*
 * d_w_interval [oc] (channelwise) or [1]; d_a_interval [1];
 * d_scores     optional: [search_round][2][eq_n][oc or 1]
 */
int p4v_conv_calibrate(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                       const float* d_out, const float* d_grad, const float* d_mult, float* d_w_interval,
                       float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Granular entry points: ONE part of calibration_step2 per call, for callers that drive the
 * alternation themselves the way the reference's methods are called one by one.  Same kernels as the fused
 * p4v_*_calibrate entry points, so a granular sequence (init, then search_round x {first, second operand}) gives
 * bit-identical intervals; no memoisation, no host synchronisation.  Candidate tables are INPUTS here:
 * [eq_n+1][blocks] fp32, row i = multiplier_i * initial interval (linear.py:544-545); rows 0..eq_n-1 are searched.
 * Workspace: the p4v_*_workspace_bytes of the same descriptor.  d_scores / d_best (optional) receive the ONE table
 * of the call, [eq_n][blocks] / [blocks].
 * ---------------------------------------------------------------------------------------- */

/* PTQSLBatchingQuantLinear._initialize_intervals (linear.py:380-397; post-GELU twin linear.py:576-599):
 * d_w_interval [n_V*n_H], d_a_interval [n_a] out. */
int p4v_amax_init_linear(const p4v_linear_desc* desc, const float* d_weight, const float* d_x, float* d_w_interval,
                         float* d_a_interval, void* d_workspace, size_t workspace_bytes, void* stream);

/* PTQSLBatchingQuantLinear._search_best_w_interval (linear.py:455-495): all n_H column blocks, argmax per V block.
 * d_w_cands [eq_n+1][n_V*n_H]; d_w_interval in/out; d_a_interval in (the current counterpart, linear.py:476). */
int p4v_linear_search_w(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                        const float* d_out, const float* d_grad, const float* d_w_cands, float* d_w_interval,
                        const float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                        size_t workspace_bytes, void* stream);

/* PTQSLBatchingQuantLinear._search_best_a_interval (linear.py:497-533; twin linear.py:609-642 with `twin_postgelu`).
 * d_a_cands [eq_n+1][n_a]; d_w_interval in; d_a_interval in/out. */
int p4v_linear_search_a(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                        const float* d_out, const float* d_grad, const float* d_a_cands, const float* d_w_interval,
                        float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                        void* stream);

/* PTQSLBatchingQuantMatMul._initialize_intervals (matmul.py:419-440): head-wise d_A_interval / d_B_interval [heads].
 * With `sos` the head-wise A interval is computed and dropped (the split search overwrites it, matmul.py:633-644):
 * d_A_interval is left untouched. */
int p4v_amax_init_matmul(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, float* d_A_interval,
                         float* d_B_interval, void* d_workspace, size_t workspace_bytes, void* stream);

/* PTQSLBatchingQuantMatMul._search_best_A_interval (matmul.py:483-522). d_A_cands [eq_n+1][heads]. */
int p4v_matmul_search_A(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                        const float* d_grad, const float* d_A_cands, float* d_A_interval, const float* d_B_interval,
                        float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes, void* stream);

/* SoSPTQSLBatchingQuantMatMul._search_best_A_interval (matmul.py:600-631): the 20 splits 2^-i against the RAW B.
 * d_split [1] out, d_A_interval [1] out (= split/(qmax-1)); d_scores [20][1] optional. */
int p4v_sos_search_split(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                         const float* d_grad, float* d_split, float* d_A_interval, float* d_scores, int32_t* d_best,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* PTQSLBatchingQuantMatMul._search_best_B_interval (matmul.py:524-563); with `sos`, A is the two-range operand
 * (matmul.py:595-598) described by d_split / d_A_interval. d_B_cands [eq_n+1][heads]. */
int p4v_matmul_search_B(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                        const float* d_grad, const float* d_B_cands, const float* d_A_interval, const float* d_split,
                        float* d_B_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                        void* stream);

/* ChannelwiseBatchingQuantConv2d / BatchingEasyQuantConv2d._initialize_intervals (conv.py:482-496 / 312-320). */
int p4v_amax_init_conv(const p4v_conv_desc* desc, const float* d_weight, const float* d_x, float* d_w_interval,
                       float* d_a_interval, void* d_workspace, size_t workspace_bytes, void* stream);

/* ChannelwiseBatchingQuantConv2d._search_best_w_interval (conv.py:526-557): d_w_cands [eq_n+1][oc], argmax per oc. */
int p4v_conv_search_w_channelwise(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                                  const float* d_out, const float* d_grad, const float* d_w_cands, float* d_w_interval,
                                  const float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                                  size_t workspace_bytes, void* stream);

/* BatchingEasyQuantConv2d._search_best_w_interval (conv.py:365-396): d_w_cands [eq_n+1][1]. */
int p4v_conv_search_w_layerwise(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                                const float* d_out, const float* d_grad, const float* d_w_cands, float* d_w_interval,
                                const float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                                size_t workspace_bytes, void* stream);

/* ChannelwiseBatchingQuantConv2d._search_best_a_interval (conv.py:559-589), a_bit < 32 only. d_a_cands [eq_n+1][1]. */
int p4v_conv_search_a(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                      const float* d_out, const float* d_grad, const float* d_a_cands, const float* d_w_interval,
                      float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                      void* stream);

/* The selection every search ends with (linear.py:493-494): best[j] = argmax_i d_scores[i][j] (first index on ties,
 * NaN counts as the maximum, like torch.argmax), d_interval[j] = d_cands[best[j]][j].  d_scores [eq_n][n_blocks],
 * d_cands [>= eq_n][n_blocks], d_best optional [n_blocks]. */
int p4v_score_argmax_gather(const float* d_scores, int32_t eq_n, int32_t n_blocks, const float* d_cands,
                            float* d_interval, int32_t* d_best, void* stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks exported for parity tests (bit-exact integer planes) and for the
 * quant_forward path (reference linear.py:164-169, matmul.py:124-138).
 * ---------------------------------------------------------------------------------------- */

/* q = clamp(rint(x / s), lo, hi) as int8, s = d_scales[row / rows_per_scale]; x is [rows][cols] fp32,
 * q is [rows][cols_padded] int8 with zero padding (cols_padded multiple of 64). */
int p4v_quantize_i8(const float* d_x, int64_t rows, int64_t cols, int64_t cols_padded, const float* d_scales,
                    int64_t rows_per_scale, int32_t lo, int32_t hi, int8_t* d_q, void* stream);

/* y = clamp(rint(x / s), lo, hi) * s  (fake quantisation, fp32 in / fp32 out), same scale indexing. */
int p4v_fake_quant(const float* d_x, int64_t rows, int64_t cols, const float* d_scales, int64_t rows_per_scale,
                   int32_t lo, int32_t hi, float* d_y, void* stream);

/* ------------------------------------------------------------------------------------------
 * quant_forward of a calibrated module as ONE integer GEMM:
 * Linear (reference linear.py:62-67; post-GELU twin linear.py:601-607 with `twin_postgelu`), MatMul (matmul.py:140-145;
 * split-of-softmax matmul.py:595-598 with `sos`).  Operands are quantised to int8 grid planes (the twin's two ranges
 * as two planes), multiplied on the int8 MFMA path and rescaled in the epilogue:
 *     out = s_a * s_w[block] * (k_x . k_w) + bias              (same arithmetic as the candidate sweeps).
 * Intervals are INPUTS here (device pointers, layouts as produced by p4v_*_calibrate); the descriptors' search
 * fields (metric, eq_n, search_round) are ignored.  Workspace: p4v_linear_workspace_bytes / p4v_matmul_workspace_bytes.
 * Same restrictions as the int8 search path: n_H = n_a = 1 runs on int8, other block layouts on the fp32 MFMA path.
 * ---------------------------------------------------------------------------------------- */
int p4v_linear_quant_forward(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                             const float* d_w_interval, const float* d_a_interval, float* d_out, void* d_workspace,
                             size_t workspace_bytes, void* stream);

/* d_A_interval: [heads] (sos: [1]); d_B_interval: [heads]; d_split: [1] (sos only); d_out: [batch*heads][M][N] */
int p4v_matmul_quant_forward(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_A_interval,
                             const float* d_B_interval, const float* d_split, float* d_out, void* d_workspace,
                             size_t workspace_bytes, void* stream);

/* Timing hook used by bench.py: when non-NULL, the named events bracket every launch of the dominant
 * sweep kernel on `stream` so its duration can be measured live with HIP events. */
typedef struct p4v_kernel_stats {
    double sweep_i8_ms;     /* accumulated duration of k_sweep<int8> launches  */
    int64_t sweep_i8_launches;
    double sweep_i8_macs;   /* integer MACs issued by those launches (padded tiles included) */
    double sweep_f32_ms;
    int64_t sweep_f32_launches;
    double sweep_f32_macs;
    double sweep_i8_alg_macs;  /* MACs of the reference GEMMs those launches stand for (unpadded, one plane) */
    double sweep_f32_alg_macs;
    double sweep6_ms;          /* the register-stationary sweep k_sweep6 alone (also included in sweep_i8_*) */
    int64_t sweep6_launches;
    double sweep6_macs;
    double sweep6_alg_macs;
    int64_t memo_hits;      /* search passes skipped because their input interval had already been evaluated */
    int64_t memo_misses;    /* search passes executed (with memoisation enabled)                            */
} p4v_kernel_stats;

/* Enable (1) / disable (0) per-launch HIP-event timing of the sweep kernels (adds a sync per launch). */
int p4v_stats_enable(int enable);
int p4v_stats_reset(void);
int p4v_stats_get(p4v_kernel_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* PTQ4VIT_HIP_H */

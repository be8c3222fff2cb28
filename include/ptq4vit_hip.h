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
 *     Exact candidate pruning (difference metrics; on by default, off when score tables are requested or with
 *     desc.reserved bit 3): candidates that provably cannot be the argmax -- their score on a slice of the samples is
 *     already below the complete score of another candidate -- are not swept over the remaining samples; the selected
 *     intervals are bit-identical with and without it.
 *     *_quant_forward, p4v_quantize_i8 and p4v_fake_quant never synchronise;
 *   - return value 0 = ok, <0 = error; p4v_last_error() returns a per-thread message;
 *   - safe to call concurrently on different devices / streams: the only state outside the call is per calling
 *     thread (error string, the optional launch timing of p4v_stats_*); the process-wide A/B words of
 *     ptq4vit_hip_debug.h exist for measurements only and are never written in production.
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

#define P4V_VERSION 140 /* 0.1.4: + p4v_calibrate_group (grouped launches over the modules of a network), p4v_launch_counters */

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
                             bit 2: p4v_linear_workspace_bytes sizes the workspace for p4v_linear_quant_forward only;
                             bit 3: disable exact candidate pruning (every candidate is swept over every sample) */
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
    int32_t reserved;           /* bit 1: disable pass memoisation; bit 2: workspace query for quant_forward only;
                                   bit 3: disable exact candidate pruning */
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
    int32_t reserved;           /* bit 3: disable the exact candidate pruning */
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
 * p4v_calibrate_group: calibration_step2() of SEVERAL modules in one call.
 *
 * Replaces the loop `for name, module in ...: module.calibration_step2()` of the reference's calibrator
 * (utils/quant_calib.py:371-372; with sequential=False the modules are independent, quant_calib.py:316-372).  Each member is
 * exactly one p4v_linear_calibrate / p4v_matmul_calibrate / p4v_conv_calibrate call (same descriptor, same pointers, its own
 * workspace of p4v_*_workspace_bytes(desc), no score tables); the members search in lock step and every kernel launch of the
 * same kind is issued ONCE for all members that are at that point (one k_finish / k_pack / k_sweep6 ... over the concatenated
 * grids instead of one per module).  The results are bit-identical to the members' single calls: a grouped kernel runs each
 * member's own code on its own parameters.  One stream; the call returns when every member's operations have been issued
 * (like the single calls, it synchronises the stream wherever a member needs a result on the host).
 *
 * kind / pointers:
 *   P4V_JOB_LINEAR  desc = p4v_linear_desc   in = {weight, bias, x, raw_out, raw_grad}   out = {w_interval, a_interval, NULL}
 *   P4V_JOB_MATMUL  desc = p4v_matmul_desc   in = {A, B, raw_out, raw_grad, NULL}        out = {A_interval, B_interval, split}
 *   P4V_JOB_CONV    desc = p4v_conv_desc     in = {weight, bias, x, raw_out, raw_grad}   out = {w_interval, a_interval, NULL}
 * `status` receives the member's own status; the call returns the first non-zero one (p4v_last_error() has its message).
 * ---------------------------------------------------------------------------------------- */
enum p4v_job_kind { P4V_JOB_LINEAR = 0, P4V_JOB_MATMUL = 1, P4V_JOB_CONV = 2 };
typedef struct p4v_group_job {
    int32_t kind;
    int32_t status;
    const void* desc;
    const float* in[5];
    const float* mult;          /* [eq_n + 1] candidate multipliers, as d_mult of the single calls */
    float* out[3];
    void* workspace;
    size_t workspace_bytes;
} p4v_group_job;
/*
 * `inputs_ready_event` (hipEvent_t, may be NULL): the members' CAPTURED tensors (x / raw_out / raw_grad, A / B) are complete once
 * this event has fired -- the caller recorded it behind its capture passes, which may still be running on other streams.  The
 * call then starts at once: what needs no captured tensor (weight abs-max, candidate tables, the 100 candidate planes of every
 * Linear's weights) is issued first, and `stream` waits for the event exactly once, when every member has reached the point
 * where it reads a captured tensor.  NULL: the tensors are ready (stream-ordered before the call, as for the single calls).
 */
int p4v_calibrate_group(p4v_group_job* jobs, int32_t n_jobs, void* stream, void* inputs_ready_event);

/* PROCESS-WIDE launch counters since the last reset: out4[0] kernel launches the calibration path asked for (one module at a
 * time these are the launches made), out4[1] kernel launches issued to the GPU (grouped ones count once), out4[2] issue
 * rounds of p4v_calibrate_group (one stream synchronisation each), out4[3] p4v_calibrate_group calls.  `out4` may be NULL. */
int p4v_launch_counters(int64_t* out4, int reset);

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

/* ONE int8 operand plane exactly as the candidate sweeps consume it (the pack kernel of the search, one candidate):
 *   P4V_PLANE_SYM     clamp(rint(x / s), lo, hi)                      linear.py:167; post-GELU twin linear.py:605-606 with
 *                     (lo, hi) = (0, q-1) for the positive and (-q, 0) for the negative range (s = const_scale when
 *                     d_scales is NULL: the fixed 0.16997.../q of linear.py:574)
 *   P4V_PLANE_SOS_HI  clamp(rint(clamp(x, split, 1) * (q-1)), 0, q-1)               matmul.py:596, split = d_scales[0]
 *   P4V_PLANE_SOS_LO  clamp(rint(clamp(x, 0, split) / (split / (q-1))), 0, q-1)     matmul.py:597
 *   P4V_PLANE_TWIN    clamp(rint(x / s), 0, hi) + clamp(rint(x / const_scale), lo, 0): both ranges of the post-GELU twin
 *                     (linear.py:605-606) in one plane -- their supports are disjoint -- as the large-K weight search streams
 *                     them (k_sweep7 splits the bytes by sign again); s = d_scales[0], lo < 0 < hi
 * q = qmax = 2^(bit-1).  Output layout as p4v_quantize_i8. */
enum p4v_plane_mode { P4V_PLANE_SYM = 1, P4V_PLANE_SOS_HI = 2, P4V_PLANE_SOS_LO = 3, P4V_PLANE_TWIN = 4 };
typedef struct p4v_plane_desc {
    int64_t rows, cols, cols_padded;
    int64_t rows_per_scale;   /* P4V_PLANE_SYM with d_scales: scale index = row / rows_per_scale */
    int32_t mode, lo, hi, qmax;
    float const_scale;
    int32_t reserved;
} p4v_plane_desc;
int p4v_pack_plane_i8(const p4v_plane_desc* desc, const float* d_x, const float* d_scales, int8_t* d_q, void* stream);

/* y = clamp(rint(x / s), lo, hi) * s  (fake quantisation, fp32 in / fp32 out), same scale indexing. */
int p4v_fake_quant(const float* d_x, int64_t rows, int64_t cols, const float* d_scales, int64_t rows_per_scale,
                   int32_t lo, int32_t hi, float* d_y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Integer export of a calibrated operand -- the data formats of the reference's utils/integer.py:8-110
 * (SURVEY.md s8 row f-3).  The source is a logical 4-D fp32 tensor dims[0..3] read through element strides
 * (non-contiguous views are read in place; a stride of 0 repeats the source along that dimension, which is how a
 * weight is exported once per V-block interval, integer.py:16); the destination is contiguous, 1 byte per element
 * (4 for P4V_EXPORT_SYM_F32).  Scale of region r for element (i0..i3): d_scale_r[sum_k (i_k / scale_r_div[k]) *
 * scale_r_stride[k]] -- the block geometry of the (n_G, n_V, n_H) padding view (integer.py:28-43).
 *   P4V_EXPORT_SYM_I8   int8   clamp(rint(x / s1), lo1, hi1)                               integer.py:16,75,36
 *   P4V_EXPORT_SYM_F32  fp32   the same value as a float grid index                         integer.py:36
 *   P4V_EXPORT_GELU_U8  uint8  (clamp(rint(x / s1), 0, hi1) + 128) + |clamp(rint(x / s2), lo2, 0)|   integer.py:63-70
 *                              (s2 = scale2_const when d_scale2 is NULL)
 *   P4V_EXPORT_SOS_U8   uint8  (clamp(rint(clamp(x, s1, 1) * (q-1)), 0, q-1) + 128)
 *                              + clamp(rint(clamp(x, 0, s1) / s2), 0, q-1),  s1 = split, s2 = A_interval; the uint8 sum
 *                              wraps modulo 256 exactly like the reference's                 integer.py:88-94
 * Never synchronises. */
enum p4v_export_mode { P4V_EXPORT_SYM_I8 = 0, P4V_EXPORT_SYM_F32 = 1, P4V_EXPORT_GELU_U8 = 2, P4V_EXPORT_SOS_U8 = 3 };
typedef struct p4v_export_desc {
    int32_t dims[4];
    int64_t src_stride[4];
    int64_t scale1_stride[4]; int32_t scale1_div[4];
    int64_t scale2_stride[4]; int32_t scale2_div[4];
    float scale2_const;
    int32_t mode, lo1, hi1, lo2, hi2, qmax;
    int32_t reserved;
} p4v_export_desc;
int p4v_export_quantize(const p4v_export_desc* desc, const float* d_src, const float* d_scale1, const float* d_scale2,
                        void* d_dst, void* stream);

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

/* Capture helper (reference utils/quant_calib.py:343-354 concatenates the hooked per-sub-batch tensors with torch.cat):
 * ONE launch copies n device buffers into slice `index` of their destinations.  d_table is a DEVICE array of n triples
 * {src pointer, dst base pointer, bytes}; dst = dst base + index * bytes; bytes must be a multiple of 4; max_bytes = the largest
 * entry (sizes the grid).  Never synchronises. */
int p4v_multi_copy(const int64_t* d_table, int32_t n, int64_t index, int64_t max_bytes, void* stream);

/* Launch timing used by bench.py's roofline: while enabled on the CALLING THREAD, every sweep launch that thread
 * enqueues is bracketed by a HIP event pair on its stream; p4v_stats_get waits for those events and returns the
 * thread's totals.  Other threads / streams are not affected. */
typedef struct p4v_kernel_stats {
    double sweep_i8_ms;     /* accumulated duration of every int8 sweep launch (k_sweep2/2g/4/5/6/7, generic int8) */
    int64_t sweep_i8_launches;
    double sweep_i8_macs;   /* integer MACs issued by those launches (padded tiles, second twin plane included) */
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
    double sweep7_ms;          /* the large-K sweep k_sweep7 alone (also included in sweep_i8_*) */
    int64_t sweep7_launches;
    double sweep7_macs;
    double sweep7_alg_macs;
    double sweep7_twin_ms;     /* the twin (two-plane) launches of k_sweep7 alone (also included in sweep7_*) */
    int64_t sweep7_twin_launches;
    double event_overhead_ms;  /* duration of an EMPTY event pair on an idle stream, measured by p4v_stats_enable(1) and already
                                  subtracted from every launch duration above and in p4v_stats_launches */
} p4v_kernel_stats;

/* One record per sweep launch the calling thread enqueued while timing was enabled, in launch order (= the order of the
 * same kernels in a rocprofv3 kernel trace of a single-stream run): which kernel family, which stage of a pruned search pass,
 * the grid, the measured duration and the work.  bench.py derives `roofline` (per kernel and per stage) from these. */
typedef struct p4v_launch_record {
    int32_t kind;     /* 2 k_sweep6 | 3 k_sweep7 | 4 k_sweep7 twin | 5 k_sweep4/5 | 6 k_sweep9 | 7 k_sweep8 | 8 k_sweep2g | 9 k_sweep2 |
                         0 generic int8 k_sweep | 1 generic fp32 k_sweep | 11 k_sos_split (fp32) | 12 k_bound (stage B1) |
                         13 k_slice_b / 14 k_slice_a (stage A of a MatMul B / A search) */
    int32_t stage;    /* 0 full sweep (pass not pruned) | 1 stage A (all candidates, sample slice) | 2 stage B1 (the bound) |
                         3 stage B2 (survivors, all samples) | 4 stage A2 (survivors of a loose first slice on the larger second one) */
    int32_t grid_x, grid_z;   /* a grouped launch (p4v_calibrate_group): grid_x = its flat grid (all members' blocks, each member padded to a
                                 multiple of 8), grid_z = the number of members; ops / bytes are the members' sums */
    double ms;        /* HIP events around the launch on its stream */
    double ops;       /* 2 x integer / fp MACs issued (padded tiles, both twin planes), candidates outside a device-side range excluded */
    double alg_ops;   /* 2 x MACs of the reference GEMMs the launch stands for (unpadded, one plane); 0: empty candidate range */
    double alg_bytes; /* compulsory bytes of the search pass the launch belongs to (both operands in fp32 as captured +
                         raw_out + the metric weight, each read once); a sweep split over two launches books its share of tiles */
} p4v_launch_record;
/* Copies min(capacity, *count) records of the calling thread since the last p4v_stats_reset; `out` may be NULL to query the count. */
int p4v_stats_launches(p4v_launch_record* out, int64_t capacity, int64_t* count);

int p4v_stats_enable(int enable);   /* 1 / 0: launch timing on the calling thread */
int p4v_stats_reset(void);
int p4v_stats_get(p4v_kernel_stats* out);

/* Exact candidate pruning (csrc/p4v_api.hip::run_pass_pruned), PROCESS-WIDE counters since the last reset -- whatever thread
 * or stream ran the search passes: out4[0] passes run in three stages (slice / bound / survivors), out4[1] those whose
 * survivor stage was empty, out4[2] eligible passes that kept the full sweep (slice too large, metric weight spread over the
 * samples, unsupported layout), out4[3] passes that are never pruned (score tables requested, cosine, fp32 operand planes,
 * desc.reserved bit 3).  Tests and bench.py use them to assert WHICH path produced a result.  `out4` may be NULL. */
int p4v_prune_counters(int64_t* out4, int reset);

/* The A/B switches and test-only entry points (p4v_debug_*) are declared in ptq4vit_hip_debug.h: not part of the drop-in
 * boundary, never called in production. */

#ifdef __cplusplus
}
#endif
#endif /* PTQ4VIT_HIP_H */

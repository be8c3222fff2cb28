/* ptq4vit_hip_debug.h -- measurement / test-only entry points of libptq4vit_hip.so.
 *
 * NOT part of the drop-in boundary (include/ptq4vit_hip.h): nothing here replaces a reference interface, production code
 * never calls it, and it is the only PROCESS-WIDE mutable state of the library (SURVEY.md s8-b3 asks for none beyond the
 * per-thread error string on the boundary itself).  The scripts under tools/, bench.py --tune / --variant and the kernel-vs-kernel
 * agreement tests use it.
 */
#ifndef PTQ4VIT_HIP_DEBUG_H
#define PTQ4VIT_HIP_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A/B switches for measurements and kernel-vs-kernel agreement tests; never needed in production (default 0).
 * `variant` disables individual kernel paths (bit list in csrc/p4v_api.hip), `force_generic` routes every int8 sweep
 * through the generic kernel.  Process-wide, relaxed atomics: set them while no call is in flight. */
int p4v_debug_set_variant(int variant, int force_generic);
/* Overrides of launch heuristics: key 0 / 1 / 2 / 3 = candidate groups of k_sweep6 / k_sweep2 / k_sweep2g / k_sweep7
 * (0 = cost model), key 4 = print the launch plans to stderr, key 5 = workgroup order of k_sweep7 + 1, key 6 = k_sweep6 prologue
 * of the cost model (0.1 us), keys 9-15 = slice sizes / tiers / thresholds of the pruned passes, key 12 = path switches for A/B runs (list in
 * csrc/p4v_api.hip: e.g. 8 read-backs by copy, 9 no per-score-block ranges, 11 the round-4 quantiser). */
int p4v_debug_set_tuning(int key, int value);
/* The row selection of the exact pruning alone (k_topk_rows; csrc/p4v_api.hip::slice_fill runs it on the per-sample metric
 * weight): for each of `segs` segments of `n` fp32 masses, d_mass [segs][n], the segment-local indices of the k heaviest
 * entries in ASCENDING index order, d_idx [segs][k]; among equal masses the lowest indices are taken; negative masses
 * count as the lightest.  Exposed for the tests: a repeated or missing row would make the slice's partial sums an
 * invalid bound.  1 <= k <= n. */
int p4v_debug_topk_rows(const float* d_mass, int segs, int n, int k, int32_t* d_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTQ4VIT_HIP_DEBUG_H */

/*
 * cfnerf_debug.h - TEST HOOKS, exported by a SEPARATE test library (tests/csrc/cfnerf_testhooks.hip ->
 * cf-nerf_amd/build/libcfnerf_testhooks.so), not by the product: libcfnerf_hip.so exports include/cfnerf.h and nothing else
 * (round 4 shipped these six symbols inside the product library).
 *
 * Not part of the drop-in boundary: nothing in cf-nerf_amd/ (the product's host side) loads that library; the hooks let tests/
 * read internal state that the ABI deliberately hides - the packed operand layout, the weight-gradient plan (host-side planners,
 * header-only code shared with the product) and the activation stash of the last CFNERF_F_STASH forward (a copy out of the
 * workspace of a handle the PRODUCT library created).  tests/test_abi_cpu.py holds the product library's dynamic symbol table to
 * exactly include/cfnerf.h and the test library's to exactly this header.
 */
#ifndef CFNERF_DEBUG_H
#define CFNERF_DEBUG_H

#include "cfnerf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* floats of the packed (MFMA-fragment-ordered) operand buffer of a configuration; < 0: rejected configuration */
CFNERF_API int64_t cfnerf_debug_packed_floats(const cfnerf_cfg* cfg);
/* pack flat parameters on the HOST with the index map the device pack kernel uses (cfnerf_layout.h: pack_map) */
CFNERF_API int cfnerf_debug_pack_host(const cfnerf_cfg* cfg, const float* flat_host, float* packed_host);
/* operand-table entry by name ("trunk", "skipseg", "ha", ... "bt_trunk"): out[4] = {w_off, b_off, kc, nt} */
CFNERF_API int cfnerf_debug_operand(const cfnerf_cfg* cfg, const char* name, int index, uint32_t* out);
/* copy a buffer of the last STASH forward / its backward ("h", "g_h", "feat", "v", "theta", "enc", "at", ...) into dst (device);
 * returns the float count, -count if max_floats is too small, -1 if there is no such buffer */
CFNERF_API int64_t cfnerf_debug_copy_stash(cfnerf_model* m, const char* name, int layer, float* dst, int64_t max_floats, cfnerf_stream s);
/* 1 if the last STASH forward wrote its trunk streams ("h", and its backward "g_h", "g_feat") in the Q4 layout (csrc/cfnerf_device.h:
 * whole 64-point tiles, fp32 mode), 0 if row-major, -1 without a stash; tests/util_hip.py::q4_to_rows undoes the layout */
CFNERF_API int cfnerf_debug_stash_q4(cfnerf_model* m);
/* the weight-gradient tiles of a configuration at P points for a stash layout (q4 as in cfnerf_debug_dw_blocks): 20 int32 per tile
 * {is_big, n0, k0, N, K, gk, wk, nseg, seg_row[4], dst_ld, dst_col, row_f, late, lay, 0, 0, 0} + 4 destination segments each */
CFNERF_API int cfnerf_debug_dw_plan(const cfnerf_cfg* cfg, int64_t P, int q4, int32_t* tiles_out, uint32_t* segdst_out, int max_tiles);
/* the blocks of that plan for a point count and CU count: 5 int64 per block {kind, tile, split, pb, pe}, per tile its split count,
 * per parameter tensor the slot count of the reduction and (seg_early, may be NULL) whether every job that feeds it belongs to the big launch
 * in EITHER stash layout.  q4: the layout of the wide streams the plan is made for (1 / 0 as the product decides it: fp32 mode and whole
 * tiles - S % 64 == 0 in ray mode, P % 64 == 0 in points mode; -1 = the points-mode rule applied to P) */
CFNERF_API int cfnerf_debug_dw_blocks(const cfnerf_cfg* cfg, int64_t P, int n_cu, int q4, int64_t* blocks_out, int max_blocks, int32_t* tile_nsplit,
                                      int32_t* seg_nsplit, int32_t* seg_early, int max_segs);

#ifdef __cplusplus
}
#endif
#endif

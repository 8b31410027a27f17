/* duckhts_amd_debug.h -- test and measurement hooks of libduckhts_amd.so.
 *
 * NOT part of the drop-in boundary (that is include/duckhts_extension.h) nor of the scan ABI (include/duckhts_amd.h): nothing in the
 * reference corresponds to these.  They exist so that the tests can look at intermediate results of the device path (the phase-A
 * scratch of a BGZF block, the encoder's BCF2 records of a VCF text batch, a block table built over a prefix) and so that tools/dbg/
 * can time one kernel alone.  They are declared here because the library exports exactly what include/ declares (tests/test_abi.py).
 */
#ifndef DUCKHTS_AMD_DEBUG_H
#define DUCKHTS_AMD_DEBUG_H
#include "duckhts_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* hipMalloc calls the device pool could not serve since the process started (DHTS_TRACE reports them) */
void dhts_debug_malloc_stats(uint64_t *calls, uint64_t *bytes, double *seconds);
/* tests: pretend only the first nbytes of the resident file have arrived (a file that is still being staged); returns the block count */
int64_t dhts_debug_index_prefix(dhts_ctx *, uint64_t nbytes);
/* tests: the BCF2 records the device encoder made of the last VCF text batch (bytes and record offsets) */
int64_t dhts_debug_vcf_records(dhts_ctx *, uint8_t *dst, uint64_t cap, uint32_t *rec_off, int64_t nrec);
/* tools/dbg: phase A (kernel 0: lane per block, 1: wave per block) / phase B alone over blocks [b0, b0 + nb); ms per launch */
double dhts_debug_time_huff(dhts_ctx *, int64_t b0, int64_t nb, int reps);
double dhts_debug_time_lz(dhts_ctx *, int64_t b0, int64_t nb, int reps);
int dhts_debug_huff_run(dhts_ctx *, int64_t b0, int64_t nb, int kernel);
/* tests: metadata, literal bytes and tokens phase A left for scratch slot s (the wave and the lane kernel must agree word for word) */
int dhts_debug_scratch_get(dhts_ctx *, int64_t s, uint32_t *meta4, uint8_t *lit, uint32_t *tok);
int dhts_debug_meta(dhts_ctx *, int64_t s, uint32_t *out4);
/* (diagnostic builds only: -DDHTS_DIAG dhts_debug_diag, -DHW_DIAG dhts_debug_hw_diag, -DTR_DIAG dhts_debug_tr_diag: per-phase cycle counters) */
int dhts_debug_diag(dhts_ctx *, unsigned long long *out8);
int dhts_debug_hw_diag(dhts_ctx *, unsigned long long *out16, int reset);
int dhts_debug_tr_diag(dhts_ctx *, unsigned long long *out16, int reset);

#ifdef __cplusplus
}
#endif
#endif

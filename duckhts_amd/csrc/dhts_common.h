// dhts_common.h -- shared device/host declarations for the MI355X scan path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DHTS_WAVE 64

// ---- BGZF block table (device, SoA) ---------------------------------------
// coff[i]  : byte offset of block i's 18-byte header in the compressed buffer
// clen[i]  : total block length BSIZE+1 (header 18 + deflate payload + trailer 8)
// isize[i] : ISIZE from the trailer (claimed inflated length); claims above 65536 are stored as 65537 (never met: isize_placed)
// uoff[i]  : exclusive prefix sum of isize[] = offset of block i in the inflated stream (u64)
struct BgzfTable {
    const uint64_t *coff;
    const uint32_t *clen;
    const uint32_t *isize;
    const uint64_t *uoff;
    int64_t n;
};

// ---- inflate phase A -> phase B scratch ------------------------------------
// Per stream (BGZF block) phase A emits
//   lit bytes  : every literal / stored byte in stream order (<= 65536)
//   tokens u32 : one per LZ77 match: [31:23] literal run before the match (0..510),
//                [22:15] len-3, [14:0] dist-1; run==511 = "511 literals, no match".
#define DHTS_LIT_STRIDE 65536u        /* bytes per stream  */
#define DHTS_TOK_STRIDE 22528u        /* tokens per stream: 65536/3 + 65536/511 + slack */
#define DHTS_TOK_PURE 511u

struct InflateMeta {           // per stream, written by phase A
    uint32_t ntok;
    uint32_t nlit;
    uint32_t outlen;           // inflated bytes implied by the token stream
    int32_t status;            // 0 ok, <0 malformed deflate stream
};

// per-block status codes (match oracle/dhts_oracle.c orc_bgzf_inflate_all)
#define DHTS_BLK_OK 0
#define DHTS_BLK_ERR_INFLATE (-3)
#define DHTS_BLK_ERR_CRC (-4)
#define DHTS_BLK_ERR_ISIZE (-5)
#define DHTS_BLK_ERR_SCRATCH (-6)      /* internal: the packed phase-A scratch was too small for this block (the host repeats the range with full-size room) */

static inline const char *dhts_hip_err(hipError_t e) { return hipGetErrorString(e); }

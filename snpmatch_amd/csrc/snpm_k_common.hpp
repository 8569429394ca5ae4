// snpm_k_common.hpp -- build switches, tile / epoch constants and vector types shared by every kernel family.
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>


namespace snpm {

#ifndef SNPM_Q4_BITIDX
#define SNPM_Q4_BITIDX 0                // k_fast_packed_q4: 1 = table index with the calls' low bits in bits 0-3 (fewer LDS bank conflicts, 8 more VALU per 64 comparisons)
#endif
#ifndef SNPM_Q4_SWZ
#define SNPM_Q4_SWZ 1                   // k_fast_packed_q4: table index bit 1 ^= low bit of the fourth row's call, bit 3 ^= high bits of rows 3, 4 (level 2): the
#endif                                  // frequent entries (all calls ref / alt) land in 16 distinct LDS bank pairs instead of 8; 0 = the plain field index
#ifndef SNPM_Q4_PHASES
#define SNPM_Q4_PHASES 1                // k_fast_packed_q4: a last wave with <= 32 lanes inside the panel splits its lanes over row groups (see the kernel)
#endif
#ifndef SNPM_Q4_PROTO_NOLOAD
#define SNPM_Q4_PROTO_NOLOAD 0
#endif
#ifndef SNPM_Q4_PROTO_ONE_BARRIER
#define SNPM_Q4_PROTO_ONE_BARRIER 0
#endif
#ifndef SNPM_Q4_PROTO_QUAD
#define SNPM_Q4_PROTO_QUAD 0            // 1: timing experiment only (results are wrong): k_fast_packed_q4 without its 4 x 16 transpose
#endif
#ifndef SNPM_Q4_MIN_WAVES
#define SNPM_Q4_MIN_WAVES 4
#endif
#ifndef SNPM_FAST_G
#define SNPM_FAST_G 4                   // rows per prefetch group of the int8 fast pass (two groups in flight)
#endif
#ifndef SNPM_FAST_G_GATHER
#define SNPM_FAST_G_GATHER SNPM_FAST_G  // the same for the gathered-row instantiations (row lists of samples, windows, batches)
#endif
#ifndef SNPM_FAST_MIN_WAVES
#define SNPM_FAST_MIN_WAVES 6
#endif
#ifndef SNPM_STRICT_BATCH
#define SNPM_STRICT_BATCH 8             // rows per prefetch batch of k_strict4 (two batches in flight)
#endif
#ifndef SNPM_STRICT_EXEC
#define SNPM_STRICT_EXEC 1              // k_strict4 (int8 panels) adds under EXEC masks (v_cmpx); 0: selects 1.0 / 0.0 multipliers (8 % slower)
#endif
constexpr int WAVE = 64;
constexpr int TILE_ROWS = 128;          // rows per LDS LUT tile (4 KiB); also the SWAR counter flush period (<= 255)
#ifndef SNPM_LONG_TILE_ROWS
#define SNPM_LONG_TILE_ROWS 248
#endif
// Long scans of the int8 fast pass (>= 2M rows) walk tiles of 248 rows: half as many barriers / LUT-tile loads per row
// (round 3, profiles/r03c_ab_tile_rows.txt: 10 000 x 20M 0.819 -> 0.828 of HBM peak, 12 500 x 16M 0.779 -> 0.796, 8192 x 24M
// 0.805 -> 0.811, 1252 / 2500 x 50M unchanged); short queries, batches and windows keep 128 (more tiles = more parts to spread).
constexpr int LONG_TILE_ROWS = SNPM_LONG_TILE_ROWS;
static_assert(LONG_TILE_ROWS % 8 == 0 && LONG_TILE_ROWS <= 255 && TILE_ROWS <= 255, "two prefetch groups per iteration; byte counters of missing calls are flushed once per tile");
constexpr int LUT_ROW_BYTES = 32;       // 4 x fp64
constexpr int MAX_WAVES_PER_BLOCK = 8;
constexpr int EPOCH_TILES = 64;         // k_fast writes its partial sums out (and restarts them) every 64 of its tiles
constexpr int REDUCE_GROUP = 64;        // k_reduce_groups adds this many partials sequentially per group
constexpr int PREFETCH_PAD_ROWS = 32;   // rows the fast pass may read (never score) past the last row of a part

// Layout descriptor of a 2-bit packed panel (snpm_panel::desc; 0 for int8 panels), passed to every kernel that reads or writes one:
//   bit 0        the panel is packed (4 accessions per byte)
//   bits 1..7    0: every row is one run of `pitch` bytes (row-major);  t + 1: SPLIT layout -- the first `pitch` bytes of a row (a
//                multiple of 256, possibly 0) lie in the main matrix at `db` with rows `pitch` bytes apart, the remaining bytes of
//                the row (its ragged tail, <= 128 of them) in a second matrix with rows of 2^t bytes
//   bits 8..     where that tail matrix starts: (desc >> 8) * 256 bytes behind `db`
// A 1135-accession panel (284 B per row) is then 256 + 32 = 288 B per row instead of a 512-B pitch: a dense scan fetches the
// bytes it uses (round 3: 388 B per 284-B row), and panels of <= 512 accessions get a 32 / 64 / 128-B pitch.  The column blocks of
// the packed kernels are 256 B wide, so a wave (and a 256-thread block of k_strict4) lies entirely in one of the two matrices.
__host__ __device__ __forceinline__ int64_t pk_tail_pitch(int64_t desc)
{
    const int t = (int)((desc >> 1) & 0x7F);
    return t ? (int64_t)1 << (t - 1) : 0;
}
__host__ __device__ __forceinline__ int64_t pk_tail_off(int64_t desc) { return (desc >> 8) * 256; }
// offset from `db` of byte b (0 <= b < pitch + tail pitch) of packed row prow
__host__ __device__ __forceinline__ int64_t pk_off(int64_t pitch, int64_t desc, int64_t prow, int64_t b)
{
    const int64_t tp = pk_tail_pitch(desc);
    return (tp && b >= pitch) ? pk_tail_off(desc) + prow * tp + (b - pitch) : prow * pitch + b;
}

typedef __attribute__((address_space(3))) const double lds_cdouble;
typedef double f64x2_t __attribute__((ext_vector_type(2)));

}  // namespace snpm

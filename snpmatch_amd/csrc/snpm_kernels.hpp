// snpm_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, wave64) behind libsnpmatch_hip.so.
//
// The work is compare-and-count over an int8 SNP x accession panel: HBM-bound byte streaming,
// no MFMA.  Layout: panel is SNP-major [n_snp, pitch] (pitch = n_acc rounded up to 256 B), so one
// SNP row is a coalesced run of accession bytes and the three weights of a row are wave-uniform.
//
// Kernels
//   k_fast     dominant kernel.  Each lane owns BPL adjacent accessions (one 4/8/16-byte load per
//              row), each wave 64*BPL adjacent accession bytes, each workgroup a run of rows (a "part").
//              Per-row weights live in LDS as a 4-entry fp64 LUT {ref, alt, het, 0} indexed by
//              (byte & 3); every element costs one ds_read_b64 + one v_add_f64 instead of three
//              compare/select pairs.  Missing counts are SWAR (packed u8 lanes).  Partials per part
//              are written once; k_reduce adds them in part order (deterministic, no atomics).
//   k_fast_packed_q4 the fast pass on a 2-bit packed panel: 16 accessions per lane, four rows per table lookup
//              (one ds_read_b128 + two v_add_f64 per two comparisons), bit-sliced missing counts, LDS reads
//              issued and waited for by hand.
//   k_fast_bits  hard-call samples (all weights 0 or 1) on a packed panel: bit-plane boolean scoring and
//              bit-sliced counting, no LDS, no fp64.
//   k_strict4 / k_strict / k_strict_sparse(_T)  reference summation order (three per-category sequential fp64
//              sums per segment, core/snpmatch.py:85-87).  Used for cross windows, for SNPM_MODE_STRICT and to
//              re-evaluate the few accessions the fast pass cannot certify (sparse variants; _T reads the
//              accession-major packed copy built by k_pack_transpose).
//   k_scan / k_scan_few  sequential accumulation of segment sums (ScoreList += chunk, core/snpmatch.py:224), with an
//              optional carry-in (totals of earlier SNP slabs).
//   k_fast<..., SEG>  the same fast pass over many independent row ranges (samples of a batch, windows of a cross) in one
//              launch; k_reduce_seg / k_eseg_part + k_eseg_finish / k_strict_pairs / k_scan_pairs: per-segment reduce, error bound, and the
//              reference-order re-evaluation of the (segment, accession) pairs the certificate flags.
//   certificate  k_wprops (weight properties), k_eref / k_efinish (reference-order error bound on the device),
//              flag_if_uncertain inside k_reduce / k_carry_flag; the re-evaluation kernels read the flag count on the
//              device and leave at once when their tier has nothing to do (dense_tier_off, *d_ncols).
//   k_carry_add / k_carry_flag / k_tot_seg  running totals of slab-streamed jobs, totals over windows.
//   k_likelihood  likeliTest + nanmin + ratio on device (core/snpmatch.py:40-55,106-117).
//   k_binom_identity  np_test_identity (core/snpmatch.py:57-72).   k_segregating  --refine support.
//   k_f1_*     in-silico F1 scores in numpy's summation order (core/csmatch.py:115-125).
//   k_build_lut, k_repitch_canon / k_pack_rows / k_unpack_rows (upload / download), k_pack_transpose[_packed]
//   (accession-major copies), k_synth* / k_synth_sample (benchmark data), k_check_rows, k_expand_codes, k_seg_pack,
//   k_patch, k_calib_read: small helpers.
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

typedef __attribute__((address_space(3))) const double lds_cdouble;
typedef double f64x2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// LUT build: W [n,3] (ref, het, alt) -> LUT [n,4] = {ref, alt, het (0 if skip_hets), 0}, entry index = db byte & 3
// (0 -> ref, 1 -> alt, 2 -> het, 3 and 0xFF -> nothing).
// bad (may be NULL): bit 2 is raised when a weight is NaN or infinite (batched samples are vetted here; single queries in k_wprops)
__global__ void k_build_lut(const double *__restrict__ w, double *__restrict__ lut, int64_t n, int skip_hets, int *__restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double w0 = w[3 * i + 0], w1 = w[3 * i + 1], w2 = w[3 * i + 2];
    if (bad && !(fabs(w0) <= 1.7976931348623157e308 && fabs(w1) <= 1.7976931348623157e308 && fabs(w2) <= 1.7976931348623157e308))
        atomicOr(bad, 4);
    double4 e;
    e.x = w0;
    e.y = w2;
    e.z = skip_hets ? 0.0 : w1;
    e.w = 0.0;
    reinterpret_cast<double4 *>(lut)[i] = e;
}

// ------------------------------------------------------------------------------------------------
// Properties of a sample's weights, computed where the weights live (no host pass over the rows):
//   k_wprops  block partial sums of wmax_r = max_c |W[r,c]| and three flags: bit 0 = some weight is not an integer
//             (or huge), bit 1 = some weight is neither 0 nor 1, bit 2 = some weight is NaN or infinite (refused: the
//             reference multiplies 0/1 masks by the weights, core/snpmatch.py:85-87, so one such weight turns EVERY
//             accession's score into NaN and int(NaN) raises in GenotyperOutput, :96).
//   k_wbits   hard-call samples: one byte of three weight bits per row (ref | het << 1 | alt << 2) for k_fast_bits.
//   k_eref / k_efinish   the reference-order part of the certificate's error bound (DESIGN.md "Exactness"):
//             E_ref = u / (1 - m_max u) * sum_k s_k * (len_k + 3 + K - k + chunks_after),  s_k = sum of wmax over
//             chunk k, rounded up by 1e-7 relative (the fp64 sums of non-negative terms below are good to ~1e-12).
__device__ __forceinline__ double block_sum_256(double v, double *sm)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[wave] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__global__ void __launch_bounds__(256)
k_wprops(const double *__restrict__ w, int64_t n, double *__restrict__ partial, int *__restrict__ flags)
{
    __shared__ double sm[4];
    double acc = 0.0;
    int f = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const double a = fabs(w[3 * r]), b = fabs(w[3 * r + 1]), c = fabs(w[3 * r + 2]);
        acc += fmax(a, fmax(b, c));
        if (!(a == floor(a) && b == floor(b) && c == floor(c)) || !(a < 1e300 && b < 1e300 && c < 1e300)) f |= 1;
        const double x = w[3 * r], y = w[3 * r + 1], z = w[3 * r + 2];
        if (!((x == 0.0 || x == 1.0) && (y == 0.0 || y == 1.0) && (z == 0.0 || z == 1.0))) f |= 2;
        if (!(a <= 1.7976931348623157e308 && b <= 1.7976931348623157e308 && c <= 1.7976931348623157e308)) f |= 4;
    }
    const double tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    if (f) atomicOr(flags, f);
}

__global__ void k_wbits(const double *__restrict__ w, int64_t n, int64_t n_padded, uint8_t *__restrict__ wbits)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_padded) return;
    uint8_t b = 0;
    if (r < n) b = (uint8_t)((w[3 * r] == 1.0 ? 1 : 0) | (w[3 * r + 1] == 1.0 ? 2 : 0) | (w[3 * r + 2] == 1.0 ? 4 : 0));
    wbits[r] = b;
}

__global__ void __launch_bounds__(256)
k_eref(const double *__restrict__ w, int64_t n, int64_t chunk, int64_t chunks_after, double *__restrict__ partial)
{
    __shared__ double sm[4];
    const int64_t K = (n + chunk - 1) / chunk;
    double acc = 0.0;                                     // meaningful in thread 0
    for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
        const int64_t r0 = k * chunk, r1 = (r0 + chunk < n) ? r0 + chunk : n;
        double s = 0.0;
        for (int64_t r = r0 + threadIdx.x; r < r1; r += 256)
            s += fmax(fabs(w[3 * r]), fmax(fabs(w[3 * r + 1]), fabs(w[3 * r + 2])));
        s = block_sum_256(s, sm);
        acc += s * (double)((r1 - r0) + 3 + (K - k) + chunks_after);
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256)
k_efinish(const double *__restrict__ partial, int n_partial, int64_t n, int64_t chunk, int64_t chunks_after,
          double *__restrict__ eref)
{
    __shared__ double sm[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) v += partial[i];
    v = block_sum_256(v, sm);
    if (threadIdx.x == 0) {
        const double u = 1.1102230246251565e-16;
        const int64_t K = (n + chunk - 1) / chunk;
        const double mmax = (double)(chunk + 3 + K + chunks_after);
        eref[0] = (v * u / (1.0 - mmax * u)) * 1.0000001;
    }
}

// row lists that arrive without a host pass over them (batches): entries outside the panel are replaced by row 0 --
// no kernel ever reads outside the panel -- and reported through *bad (the call then fails after its synchronisation).
// src32 != NULL: the list crossed PCIe as int32 (half the bytes; -1 stands for any value that does not fit) and is
// widened into `rows` here.
__global__ void k_check_rows(int64_t *__restrict__ rows, const int32_t *__restrict__ src32, int64_t n, int64_t n_snp,
                             int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = src32 ? (int64_t)src32[i] : rows[i];
    if (r < 0 || r >= n_snp) {
        r = 0;
        atomicOr(bad, 1);
    }
    rows[i] = r;
}

// dictionary-coded weights (batches from VCFs whose PLs are small integers): wei[i] = table[codes[i]], i over n * 3
__global__ void k_expand_codes(const uint16_t *__restrict__ codes, const double *__restrict__ table, int64_t n3,
                               double *__restrict__ wei)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) wei[i] = table[codes[i]];
}

// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int BPL>
struct LoadT;
template <>
struct LoadT<4> { typedef uint32_t type; };
template <>
struct LoadT<8> { typedef u32x2 type; };
template <>
struct LoadT<16> { typedef u32x4 type; };

__device__ __forceinline__ uint32_t dword_of(const uint32_t &v, int) { return v; }
__device__ __forceinline__ uint32_t dword_of(const u32x2 &v, int k) { return k == 0 ? v.x : v.y; }
__device__ __forceinline__ uint32_t dword_of(const u32x4 &v, int k)
{
    return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
}

// One SNP row for one lane: NDW dwords of accession bytes.
//   address of element j of dword x = group_base | byte,  byte = (code & 3) * 8 + roff, built by ONE
//   v_perm_b32 (group_base is 256-B aligned and wave-uniform; roff in {0,128} selects the half of the
//   256-B block, the row's offset inside its 4-row group, U*32, goes into the ds_read immediate);
//   then one ds_read_b64 and one v_add_f64 per element.
template <int NDW, bool SKIP, int U>
__device__ __forceinline__ void fast_row(const uint32_t (&x)[NDW], uint32_t group_base, uint32_t roff4,
                                         double (&acc)[NDW * 4], uint32_t (&miss8)[NDW])
{
#ifdef SNPM_FAST_PATTERN_ONLY
    // diagnostic build: k_fast's loads, geometry and barriers without its arithmetic (what the access pattern alone reaches);
    // -DSNPM_FAST_PATTERN_ONLY=2 also drops the LUT staging and the per-tile barriers (measured: 3 % SLOWER than with them)
#pragma unroll
    for (int k = 0; k < NDW; ++k) miss8[k] ^= x[k];
    (void)group_base; (void)roff4; (void)acc;
    return;
#endif
    double w[NDW * 4];
#pragma unroll
    for (int k = 0; k < NDW; ++k) {
        const uint32_t tix = ((x[k] << 3) & 0x18181818u) | roff4;   // byte j = (code & 3) * 8 + roff
        if (SKIP)
            miss8[k] += ((x[k] >> 7) | ((x[k] >> 1) & ~x[k])) & 0x01010101u;   // negative, or het (code 2 = 0b10; 3 is not)
        else
            miss8[k] += (x[k] >> 7) & 0x01010101u;                   // negative
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // v_perm_b32: D.b0 = tix.b[j] (selector 4+j: src0 bytes), D.b1 = group_base.b1, D.b2 = D.b3 = 0
            const uint32_t addr = __builtin_amdgcn_perm(tix, group_base, 0x0c0c0100u | (uint32_t)(4 + j));
            lds_cdouble *ptr = (lds_cdouble *)(uintptr_t)addr;
            w[4 * k + j] = ptr[U * 4];                               // + U*32 bytes: immediate offset
        }
    }
#pragma unroll
    for (int e = 0; e < NDW * 4; ++e) acc[e] += w[e];
}

template <int BPL, bool NT>
__device__ __forceinline__ void load_row(const int8_t *p, uint32_t (&x)[BPL / 4])
{
    typedef typename LoadT<BPL>::type load_t;
    load_t v;
    if (NT)
        v = __builtin_nontemporal_load(reinterpret_cast<const load_t *>(p));
    else
        v = *reinterpret_cast<const load_t *>(p);
#pragma unroll
    for (int k = 0; k < BPL / 4; ++k) x[k] = dword_of(v, k);
}

// one SNP row of one lane
#define SCORE_ROW(U, X, GROUP_BASE, ROFF4, ROFF) fast_row<NDW, SKIP, U>((X), (GROUP_BASE), (ROFF4), acc, miss8)

// Fast pass.
//   grid.x = column blocks (blockDim.x/64 waves x 64 lanes x BPL bytes), grid.y = P parts.
//   Rows are cut into tiles of TILE_ROWS; part p scores tiles p, p+P, p+2P, ... so that at any time the
//   resident workgroups sweep ONE contiguous window of the panel (DRAM-page friendly, like a streaming
//   copy) and every part gets the same number of tiles (+-1).
//   The row loop is software-pipelined: the G rows of the next group are requested from HBM before the
//   G rows of the current group are scored, so every wave keeps 2*G row loads in flight; prefetches are
//   unconditional (straight-line code lets the compiler count vmcnt exactly) and may run up to 2*G rows
//   past the last row: the panel and the row-index list carry PREFETCH_PAD_ROWS extra rows/entries.
//   Partial sums are written out and restarted every EPOCH_TILES tiles of a part (epoch e of part p goes
//   to slot e*P + p): accumulation chains stay short (tight rounding bound, u16 counters never overflow).
//   out_score [n_epochs*P, ld] fp64, out_miss [n_epochs*P, ld] u32 (ld = pitch).
// launch bound: <= 512 threads and (for the 4 B/lane layout) >= 6 waves per SIMD, i.e. <= 80 VGPRs: the
// kernel is latency-bound and 5-wave blocks only fit 4 per CU with 6 wave slots per SIMD (measured:
// 79-82 % of HBM peak at 80 VGPRs vs 67-70 % at 88; asking for 7 waves changes nothing -- the kernel already
// needs only 70 VGPRs -- and 8 waves (64 VGPRs, 7 spilled) costs 1-8 %: -DSNPM_FAST_MIN_WAVES=n to re-measure)
// SEG (segmented pass: batches of samples, windows of a cross): part p scores the CONTIGUOUS rows
// [part_desc[3p], part_desc[3p+1]) of the (concatenated) matched list -- never more than EPOCH_TILES tiles, all inside
// one segment -- and writes its partial sums to slot part_desc[3p+2]; k_reduce_seg adds the slots of a segment in
// order.  Without SEG the arguments part_desc is unused and the code is the tile-interleaved pass described above.
template <int BPL, bool SKIP, bool GATHER, bool NT, bool SEG = false, int TR = TILE_ROWS>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK, (BPL <= 4 ? SNPM_FAST_MIN_WAVES : 1))
k_fast(const int8_t *__restrict__ db, int64_t pitch, const int64_t *__restrict__ row_idx, int64_t row0, int64_t n,
       const double *__restrict__ lut, double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld,
       const int64_t *__restrict__ part_desc = nullptr)
{
    // BPL = bytes (= accessions = accumulators) per lane and row; packed panels have their own kernels below
    static_assert(BPL == 4 || BPL == 8 || BPL == 16, "int8 panels: 4, 8 or 16 bytes per lane");
    constexpr int NDW = BPL / 4;
    constexpr int EPL = BPL;
    // rows per prefetch group: 4 (G*32 B = half a 256-B LUT block); 8-row groups spill at the 80-VGPR budget and measured no gain
    constexpr int G = GATHER ? SNPM_FAST_G_GATHER : SNPM_FAST_G;
    __shared__ __attribute__((aligned(256))) double s_lut[2][TR * 4];

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int64_t byte0 = ((int64_t)blockIdx.x * nthr + tid) * BPL;     // byte offset of the lane inside a row
    const int64_t col0 = byte0;                                          // first accession of the lane
    // a lane works when its bytes lie inside the row (pitch is a multiple of 256 B) and its accessions inside the result
    // arrays (ld); blocks may carry spare waves
    const bool lane_on = byte0 < pitch && col0 < ld;
    const int64_t p = SEG ? (int64_t)blockIdx.y + (int64_t)gridDim.y * blockIdx.z : (int64_t)blockIdx.y;
    if (SEG && p >= n) return;             // SEG: n is the number of parts (grid.y * grid.z may exceed it)
    // tiles of this block: T = T0, T0 + TS, ... < n_tiles_total; tile T = rows [rbase + T * TR, ...) up to rend
    const int64_t rbase = SEG ? part_desc[3 * p] : 0;
    const int64_t rend = SEG ? part_desc[3 * p + 1] : n;
    const int64_t P = SEG ? 1 : (int64_t)gridDim.y;               // tile stride
    const int64_t T0 = SEG ? 0 : p;
    const int64_t slot0 = SEG ? part_desc[3 * p + 2] : p;         // partial slot of epoch 0 (epoch e: slot0 + e * slot_stride)
    const int64_t slot_stride = SEG ? 1 : (int64_t)gridDim.y;
    const int64_t n_tiles_total = (rend - rbase + TR - 1) / TR;

    double acc[EPL];
    uint32_t miss16[NDW * 2];               // packed 2 x u16 per register, flushed from packed u8 every tile
    uint32_t miss8[NDW];
#pragma unroll
    for (int i = 0; i < EPL; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NDW * 2; ++i) miss16[i] = 0;
#pragma unroll
    for (int i = 0; i < NDW; ++i) miss8[i] = 0;

    // address = wave-uniform row base (scalar registers) + 32-bit lane offset: global_load saddr form,
    // no per-lane 64-bit address arithmetic
    const uint32_t lane_off = lane_on ? (uint32_t)byte0 : 0u;
    auto row_ptr = [&](int64_t rr) -> const int8_t * {
        const int64_t prow = GATHER ? row_idx[rr] : (row0 + rr);
        const int8_t *rowbase = db + prow * pitch;                 // wave-uniform: scalar registers
        // The empty asm keeps the zero-extension of the lane offset inside the loop body, where the
        // instruction selector can fold it: global_load_dword v, v_off, s[base:base+1] (saddr form), no
        // per-lane 64-bit address arithmetic and no address VGPR pairs.
        uint32_t off = lane_off;
        asm volatile("" : "+v"(off));
        return rowbase + off;
    };

    // write this lane's partial sums to slot (epoch, part) and restart them
    auto store_partials = [&](int64_t epoch) {
        if (lane_on) {
            double *os = out_score + (slot0 + epoch * slot_stride) * ld + col0;
            uint32_t *om = out_miss + (slot0 + epoch * slot_stride) * ld + col0;
#pragma unroll
            for (int i = 0; i < EPL; i += 2) *reinterpret_cast<double2 *>(os + i) = make_double2(acc[i], acc[i + 1]);
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                uint4 m;
                m.x = miss16[2 * k + 0] & 0xffffu;
                m.y = miss16[2 * k + 1] & 0xffffu;
                m.z = miss16[2 * k + 0] >> 16;
                m.w = miss16[2 * k + 1] >> 16;
                *reinterpret_cast<uint4 *>(om + 4 * k) = m;
            }
        }
#pragma unroll
        for (int i = 0; i < EPL; ++i) acc[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NDW * 2; ++i) miss16[i] = 0;
    };
    int64_t last_epoch = 0;

    if (T0 < n_tiles_total) {
        // first LUT tile -> LDS; first group in flight
        {
            const int64_t tr0 = rbase + T0 * TR;
            const int rows2 = 2 * (int)((rend - tr0 < TR) ? (rend - tr0) : TR);
            const double2 *src = reinterpret_cast<const double2 *>(lut + 4 * tr0);
            double2 *dst = reinterpret_cast<double2 *>(&s_lut[0][0]);
            for (int i = tid; i < rows2; i += nthr) dst[i] = src[i];
        }
        uint32_t xa[G][NDW], xb[G][NDW];
#pragma unroll
        for (int u = 0; u < G; ++u) load_row<BPL, NT>(row_ptr(rbase + T0 * TR + u), xa[u]);
        __syncthreads();

        int buf = 0;
        int tiles_in_epoch = 0;
        int64_t epoch = 0;
        for (int64_t T = T0; T < n_tiles_total; T += P, buf ^= 1) {
            if (tiles_in_epoch == EPOCH_TILES) {
                store_partials(epoch);
                ++epoch;
                tiles_in_epoch = 0;
            }
            ++tiles_in_epoch;
            const int64_t tr0 = rbase + T * TR;
            const int rows = (int)((rend - tr0 < TR) ? (rend - tr0) : TR);
            const bool more = (T + P < n_tiles_total);
            const int64_t ntr0 = more ? rbase + (T + P) * TR : tr0;      // my next tile (or a harmless re-read)
            // stage the next LUT tile (256 double2) in ONE register pair per thread when the block has
            // >= 256 threads; narrower blocks copy it synchronously at the end of the tile instead
            double2 pre0 = make_double2(0.0, 0.0);
            const bool staged = more && nthr >= TR * 2;
            const int nrows2 = more ? 2 * (int)((rend - ntr0 < TR) ? (rend - ntr0) : TR) : 0;
#if !defined(SNPM_FAST_PATTERN_ONLY) || SNPM_FAST_PATTERN_ONLY != 2
            if (staged && tid < nrows2) pre0 = reinterpret_cast<const double2 *>(lut + 4 * ntr0)[tid];
#endif

            const uint32_t lds_base =
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)(&s_lut[buf][0]);
            const int full_groups = rows / G;
            // one group of G rows: LUT rows [gi*G, gi*G + G) of the tile = 256-B block (gi*G*32)/256, offset
            // (gi*G*32)%256 inside it (0 or 128 for G = 4, always 0 for G = 8)
#define SCORE_GROUP(X, GI)                                                                  \
    do {                                                                                    \
        const uint32_t goff_ = (uint32_t)(GI) * (uint32_t)(G * LUT_ROW_BYTES);              \
        const uint32_t gbase_ = lds_base + (goff_ & ~255u);                                 \
        const uint32_t roff_ = goff_ & 255u;                                                \
        const uint32_t roff4_ = roff_ * 0x01010101u;                                        \
        SCORE_ROW(0, (X)[0], gbase_, roff4_, roff_);                                        \
        SCORE_ROW(1, (X)[1], gbase_, roff4_, roff_);                                        \
        if constexpr (G > 2) {                                                              \
            SCORE_ROW(2, (X)[G > 2 ? 2 : 0], gbase_, roff4_, roff_);                        \
            SCORE_ROW(3, (X)[G > 2 ? 3 : 0], gbase_, roff4_, roff_);                        \
        }                                                                                   \
        if constexpr (G > 4) {                                                              \
            SCORE_ROW(4, (X)[G > 4 ? 4 : 0], gbase_, roff4_, roff_);                        \
            SCORE_ROW(5, (X)[G > 4 ? 5 : 0], gbase_, roff4_, roff_);                        \
            SCORE_ROW(6, (X)[G > 4 ? 6 : 0], gbase_, roff4_, roff_);                        \
            SCORE_ROW(7, (X)[G > 4 ? 7 : 0], gbase_, roff4_, roff_);                        \
        }                                                                                   \
    } while (0)
            // two groups per iteration so that the xa/xb roles are static (no register copies)
            int g = 0;
            for (; g + 2 <= full_groups; g += 2) {
                const int64_t rnext = tr0 + (int64_t)(g + 1) * G;
                // the group after the pair: inside this tile, or the first group of my next tile
                const int64_t rafter = (g + 2 < TR / G) ? rnext + G : ntr0;
                // ---- group g (data in xa); request group g+1 into xb
#pragma unroll
                for (int u = 0; u < G; ++u) load_row<BPL, NT>(row_ptr(rnext + u), xb[u]);
                SCORE_GROUP(xa, g);
                // ---- group g+1 (data in xb); request the following group into xa
#pragma unroll
                for (int u = 0; u < G; ++u) load_row<BPL, NT>(row_ptr(rafter + u), xa[u]);
                SCORE_GROUP(xb, g + 1);
            }
            if (g < full_groups) {                               // odd group count: only in the last tile of all
                SCORE_GROUP(xa, g);
            }
#undef SCORE_GROUP
            for (int r = full_groups * G; r < rows; ++r) {      // at most G-1 rows: last tile of all
                uint32_t x[NDW];
                load_row<BPL, NT>(row_ptr(tr0 + r), x);
                const uint32_t group_base = lds_base + (uint32_t)(r >> 3) * 256u;
                const uint32_t roff4 = (uint32_t)(r & 7) * 0x20202020u;        // (r & 7) * 32 in every byte
                SCORE_ROW(0, x, group_base, roff4, (uint32_t)(r & 7) * 32u);
            }
            // flush packed u8 counters (<= TR <= 255 per byte) into packed u16 pairs
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                miss16[2 * k + 0] += miss8[k] & 0x00ff00ffu;          // bytes 0 and 2
                miss16[2 * k + 1] += (miss8[k] >> 8) & 0x00ff00ffu;   // bytes 1 and 3
                miss8[k] = 0;
            }
#if !defined(SNPM_FAST_PATTERN_ONLY) || SNPM_FAST_PATTERN_ONLY != 2
            if (more) {
                double2 *dst = reinterpret_cast<double2 *>(&s_lut[buf ^ 1][0]);
                if (staged) {
                    if (tid < TR * 2) dst[tid] = pre0;
                } else {
                    const double2 *src = reinterpret_cast<const double2 *>(lut + 4 * ntr0);
                    for (int i = tid; i < nrows2; i += nthr) dst[i] = src[i];
                }
            }
            __syncthreads();
#endif
        }
        last_epoch = epoch;
    }
    // the last (possibly only, possibly empty) epoch; epochs a short part never reaches are zeroed by the host
    store_partials(last_epoch);
}

#undef SCORE_ROW

// ------------------------------------------------------------------------------------------------
// Fast pass on a PACKED panel (2 bits per call): 16 accessions per lane, FOUR ROWS PER LOOKUP (k_fast_packed_q4).
//   (Round 1's one-byte-per-lane instantiation of k_fast spent ~5.5 VALU operations per comparison and was VALU-bound at a
//   quarter of the HBM rate.)  A lane loads a dword per SNP row (16 calls; a wave covers 1024 accessions with 256
//   contiguous bytes, the access shape of the int8 kernel), and the block builds, per group of four rows, a 256-entry
//   table  entry[c0 | c1 << 2 | c2 << 4 | c3 << 6] = ((lut[r][c0] + lut[r+1][c1]) + lut[r+2][c2]) + lut[r+3][c3]
//   (2 KiB per four rows), so that a lane scores four rows of one accession with ONE 8-byte LDS read and ONE addition.
//   The index byte of every accession comes from a 4 x 16 transpose of the rows' 2-bit fields (24 integer operations
//   per 64 comparisons: fields -> nibbles -> bytes).
//   (Round 1-2's k_fast_packed16 looked up pairs of accessions of ONE row, a 16-byte read and two additions per two
//   comparisons: 51 ms on the packed 10k x 50M panel, this kernel 36 ms, the loads alone 22 ms.)
//   The pre-added entries only change the summation tree of the fast pass (every term still passes through fewer
//   additions than efast_bound assumes); the reference-order paths never use this kernel.
//   Rows past the end of the matched list read as table rows of 0.0 and as call code 0 (not missing).
//   Missing calls (code 3, or 2/3 with skip_hets) are one bit per call after  x & (x >> 1) & 0x55555555;
//   the 16 per-accession counts are kept bit-sliced (planes 1, 2, 4 ... 64) and updated for 8 rows at a time
//   with carry-save adders (3 operations each), i.e. ~0.4 operations per comparison instead of one.
//   Geometry, tile-interleaved parts, epochs and the prefetch pipeline are those of k_fast.
constexpr int BITS_TILE_ROWS = 256;     // rows per tile of k_fast_bits (no LDS: only the unit in which parts interleave; 128: +2.4 %, 512: -1.4 %)
constexpr int BITS_FLUSH_ROWS = 64;     // its bit-sliced counters (7 planes) are flushed into 16-bit counters every 64 rows
constexpr int Q4_TILE_ROWS = 64;        // 16 four-row tables x 2 KiB = 32 KiB of LDS per block
constexpr int Q4_G = 8;                 // rows per prefetch group (two tables)
#ifndef SNPM_Q4_RUN
#define SNPM_Q4_RUN 4
#endif
constexpr int Q4_RUN = SNPM_Q4_RUN;     // tiles a part scores in a row before it jumps ahead (the host's tile = Q4_RUN * Q4_TILE_ROWS rows)
// Epoch sizes are coupled across three places: the 16-bit missing-call counters of the packed kernels (flushed once per
// epoch), and the host's fast-pass error bound (efast_bound, snpm_api.hip), which counts at most EPOCH_TILES * TILE_ROWS
// additions per term inside a part.  k_fast_packed_q4 adds pre-summed quads of rows (a quarter of its rows + 3 table additions
// per term); k_fast_bits only runs on all-integer weights (bound 0), but its counters still have to hold an epoch.
static_assert(EPOCH_TILES * BITS_TILE_ROWS <= 65535, "k_fast_bits: an epoch overflows the 16-bit counters");
static_assert(EPOCH_TILES * Q4_RUN * Q4_TILE_ROWS <= 65535, "k_fast_packed_q4: an epoch overflows the 16-bit counters");
static_assert(EPOCH_TILES * Q4_RUN * Q4_TILE_ROWS / 4 + 3 + 8 <= EPOCH_TILES * TILE_ROWS,          // + the phase additions of a phased wave
              "k_fast_packed_q4: more additions per term and epoch than efast_bound assumes");
static_assert(Q4_TILE_ROWS % (2 * Q4_G) == 0 && Q4_TILE_ROWS <= 127, "two register sets per iteration; 7-bit missing counters per tile");

// carry-save adder of three bit vectors: two v_bitop3_b32 (majority 0xE8, parity 0x96)
#define Q4_CSA(H, L, A, B, C)                                                   \
    do {                                                                        \
        const uint32_t a_ = (A), b_ = (B), c_ = (C);                            \
        (H) = __builtin_amdgcn_bitop3_b32(a_, b_, c_, 0xE8);                    \
        (L) = __builtin_amdgcn_bitop3_b32(a_, b_, c_, 0x96);                    \
    } while (0)

// SEG (batches of samples, windows of a cross: as in k_fast): part p scores the contiguous rows [part_desc[3p],
// part_desc[3p+1]) of the concatenated matched list -- all inside one segment, never more than EPOCH_TILES tiles of k_fast
// (8192 rows: one slot, no epochs) -- and writes its partial sums to slot part_desc[3p+2]; n is the number of parts.
// TR_: rows per tile = TR_ / 4 tables of 2 KiB in LDS.  64 for blocks of four waves and more; narrow panels run blocks of one
// to three waves, and with 34 KiB each only four of those fit a CU (1135 accessions: 8 resident waves, 512 and fewer: 4 -- one
// per SIMD): their tiles have 16 (one wave) or 32 rows (two, three), so that LDS stops bounding the resident waves.
template <bool SKIP, bool GATHER, bool NT, bool SEG = false, int TR_ = Q4_TILE_ROWS>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK, SNPM_Q4_MIN_WAVES)
k_fast_packed_q4(const int8_t *__restrict__ db, int64_t pitch, const int64_t *__restrict__ row_idx, int64_t row0, int64_t n,
                 const double *__restrict__ lut, double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld,
                 int64_t n_acc, const int64_t *__restrict__ part_desc = nullptr)
{
    constexpr int G = Q4_G;
    constexpr int TR = TR_;
    constexpr int NQ = TR / 4;
    constexpr int NG = TR / G;
    static_assert(TR % (2 * G) == 0 && TR <= Q4_TILE_ROWS && Q4_TILE_ROWS % TR == 0, "two register sets per iteration; whole tiles per run");
    __shared__ __attribute__((aligned(256))) double s_tab[NQ * 256];
    __shared__ __attribute__((aligned(16))) double s_l4[TR * 4];       // the tile's 4-entry LUT rows (table build only)

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    // PHASED waves (narrow panels and the ragged end of any panel): a wave whose first dword lies t <= 32 dwords before the
    // end of the row would run its whole instruction stream for t lanes -- on the 1135 accessions of the 1001 Genomes panel
    // (71 dwords) the second wave scored 7 lanes at the price of 64.  Such a wave instead gives lane l the dword l % t of
    // the row groups (8 rows) ph_j + ph * i of every tile, ph_j = l / t < ph = min(8, 64 / t): it walks a tile in
    // ceil(8 / ph) steps instead of 8, looks its four-row tables up at a per-lane offset, and adds its ph phases together
    // (lanes of phase 0, in phase order) before it writes partial sums.  Every term passes through FEWER additions than
    // in an ordinary wave (its phase's share of the quads + ph - 1 <= 7 phase additions).
    // Values that are needed once per tile or once per epoch (lane, phase, column, LDS / LUT addresses of the tile refill) are
    // recomputed from a thread index the compiler cannot hoist (tid_now): kept alive across the scoring loop they were
    // spilled, and a reload in the wrong place waits for every row load in flight -- or worse: this compiler placed spill
    // stores in front of the s_or that ends a divergent region (the SEG / dense / 32-row-tile build lost ph_j and col0 of the
    // lanes that had been inactive there and wrote garbage counts; tests/test_gpu_batch.py::test_dense_windows_on_narrow_packed_panels).
    auto tid_now = [&]() -> int { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; };
    const int64_t dw_first = (int64_t)blockIdx.x * nthr + __builtin_amdgcn_readfirstlane(tid & ~(WAVE - 1));   // first dword of my wave (scalar)
    const int64_t dw_end = (pitch / 4 < (n_acc + 15) / 16) ? pitch / 4 : (n_acc + 15) / 16;   // dwords of a row that hold accessions
    int ph_t = 0, ph = 1;                               // wave-uniform
    if (SNPM_Q4_PHASES && dw_first < dw_end && dw_end - dw_first <= WAVE / 2) {
        ph_t = (int)(dw_end - dw_first);
        ph = (WAVE / ph_t < NG) ? WAVE / ph_t : NG;
    }
    const bool phased = ph > 1;
    const int n_it = (NG + ph - 1) / ph;                // steps per tile of a phased wave
    // (my phase, my dword, am I inside the panel) -- cheap enough to derive again wherever they are needed
    auto my_phase = [&]() -> int { return phased ? (tid_now() & (WAVE - 1)) / ph_t : 0; };
    auto my_dword = [&](int phj) -> int64_t {
        const int t = tid_now();
        return phased ? dw_first + ((t & (WAVE - 1)) - phj * ph_t) : (int64_t)blockIdx.x * nthr + t;
    };
    auto lane_inside = [&](int phj, int64_t dw) -> bool { return phased ? (phj < ph) : (dw * 4 < pitch && dw * 16 < ld); };
    bool lane_on;
    uint32_t lane_off;                                  // byte offset of my dword in a row (+ the first row of my phase's group: 32 bits hold 56 rows of <= 2^25 bytes)
    {
        const int phj = my_phase();
        const int64_t dw = my_dword(phj);
        lane_on = lane_inside(phj, dw);
        lane_off = lane_on ? (uint32_t)(dw * 4) + (uint32_t)(phj * G) * (uint32_t)pitch : 0u;
    }
    const bool wave_on = __any(lane_on) != 0;           // wave-uniform
    const int64_t p = SEG ? (int64_t)blockIdx.y + (int64_t)gridDim.y * blockIdx.z : (int64_t)blockIdx.y;
    if (SEG && p >= n) return;             // whole blocks leave (grid.y * grid.z may exceed the number of parts)
    // tiles of this block: T = T0, T0 + P, ... ; tile T = rows [rbase + T * TR, ...) up to rend
    const int64_t rbase = SEG ? part_desc[3 * p] : 0;
    const int64_t rend = SEG ? part_desc[3 * p + 1] : n;
    const int P = SEG ? 1 : (int)gridDim.y;            // tile indices are 32-bit (a scalar 64-bit compare costs a vector register pair)
    const int T0 = SEG ? 0 : (int)p;
    const int64_t slot0 = SEG ? part_desc[3 * p + 2] : p;          // partial slot of epoch 0 (epoch e: slot0 + e * slot_stride)
    const int64_t slot_stride = SEG ? 1 : (int64_t)gridDim.y;
    // parts interleave in RUNS of Q4_RUN tiles (the host's tile = one run: k_fast_bits gained 2-4 % from longer contiguous
    // pieces per part); my tiles are k = 0, 1, 2, ...: run T0 + (k / RUN) * P, tile k % RUN inside it
    constexpr int RUN = SEG ? 1 : Q4_RUN * (Q4_TILE_ROWS / TR);
    const int n_tiles_total = (int)((rend - rbase + TR - 1) / TR);
    auto tile_of = [&](int k) -> int { return (T0 + (k / RUN) * P) * RUN + (k % RUN); };

    double acc[16];
    uint32_t miss16[8];                 // miss16[d]: accession d (low half) and d + 8 (high half)
    uint32_t p1 = 0, p2 = 0, p4 = 0, p8 = 0, p16 = 0, p32 = 0, p64 = 0;   // bit-sliced counts of the current tile
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) miss16[i] = 0;

    uint32_t three = 3u;                                   // shift count of the table index (SDWA takes no literal)
    asm volatile("" : "+v"(three));
    auto load = [&](int64_t rr) -> uint32_t {
        const int64_t prow = GATHER ? row_idx[rr] : (row0 + rr);
        const int8_t *rowbase = db + prow * pitch;
        uint32_t off = lane_off;
        asm volatile("" : "+v"(off));                     // keeps the saddr form of the load (see k_fast)
        const uint32_t *ptr = reinterpret_cast<const uint32_t *>(rowbase + off);
        return NT ? __builtin_nontemporal_load(ptr) : *ptr;
    };
    // G consecutive rows starting at matched row r: one buffer resource per group (scalar registers), the row inside the group
    // in the scalar offset, the lane's bytes in the vector offset -- no per-load vector instruction (the saddr form of
    // global_load costs a v_mov per load here); gathered rows keep the global loads
    auto load_group = [&](uint32_t (&x)[G], int64_t r) {
#if SNPM_Q4_PROTO_NOLOAD                         // timing experiment only (wrong results): the pass without its row loads
#pragma unroll
        for (int u = 0; u < G; ++u) x[u] = ((uint32_t)r + (uint32_t)u) * 2654435761u + lane_off * 40503u;
        return;
#endif
        if constexpr (!GATHER) {
            const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<int8_t *>(db + (row0 + r) * pitch), 0, (int)(G * pitch), 0x00020000);
#pragma unroll
            for (int u = 0; u < G; ++u)
                x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, lane_off, (int)(u * pitch), NT ? 2 : 0);
        } else {
#pragma unroll
            for (int u = 0; u < G; ++u) x[u] = load(r + u);
        }
    };
    auto missing_bits = [](uint32_t x) -> uint32_t {
        return SKIP ? ((x >> 1) & 0x55555555u) : __builtin_amdgcn_bitop3_b32(x, x >> 1, 0x55555555u, 0x80);
    };
    // tables of one tile from its LUT rows in s_l4
    auto build_tables = [&]() {
        // one thread per (table, c0, c1, c2): the three-row prefix once, then the four entries that differ in c3
        for (int i = tid_now(); i < NQ * 64; i += nthr) {
            const double *L = s_l4 + 16 * (i >> 6);
            const int e = i & 63;
            const double pre3 = (L[e & 3] + L[4 + ((e >> 2) & 3)]) + L[8 + (e >> 4)];
            const double2 l3a = *reinterpret_cast<const double2 *>(L + 12), l3b = *reinterpret_cast<const double2 *>(L + 14);
#if SNPM_Q4_BITIDX
            const int c0 = e & 3, c1 = (e >> 2) & 3, c2 = e >> 4;
            const int eb = (c0 & 1) | ((c1 & 1) << 1) | ((c2 & 1) << 2) | ((c0 >> 1) << 4) | ((c1 >> 1) << 5) | ((c2 >> 1) << 6);
            double *dst = s_tab + 256 * (i >> 6) + eb;
            dst[0] = pre3 + l3a.x;          // c3 = 0
            dst[8] = pre3 + l3a.y;          // c3 = 1: bit 3
            dst[128] = pre3 + l3b.x;        // c3 = 2: bit 7
            dst[136] = pre3 + l3b.y;        // c3 = 3
#elif SNPM_Q4_SWZ
            // swizzled positions (see score_quad): entry (c0, c1, c2, c3) lives at index  e ^ ((c3 & 1) << 1) ^ (((c2 ^ c3) >> 1) << 3)  + 64 c3
            double *tb = s_tab + 256 * (i >> 6);
            const int h2 = (SNPM_Q4_SWZ >= 2) ? ((e >> 5) & 1) << 3 : 0;      // c2's high bit -> bit 3
            const int h3 = (SNPM_Q4_SWZ >= 2) ? 8 : 0;                        // c3's high bit -> bit 3
            tb[(e ^ h2)] = pre3 + l3a.x;                       // c3 = 0
            tb[(e ^ h2 ^ 2) + 64] = pre3 + l3a.y;              // c3 = 1
            tb[(e ^ h2 ^ h3) + 128] = pre3 + l3b.x;            // c3 = 2
            tb[(e ^ h2 ^ h3 ^ 2) + 192] = pre3 + l3b.y;        // c3 = 3
#else
            double *dst = s_tab + 256 * (i >> 6) + e;
            dst[0] = pre3 + l3a.x;
            dst[64] = pre3 + l3a.y;
            dst[128] = pre3 + l3b.x;
            dst[192] = pre3 + l3b.y;
#endif
        }
    };
    auto flush_planes = [&]() {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            uint32_t c = (p1 >> (2 * d)) & 0x00010001u;
            c += ((p2 >> (2 * d)) & 0x00010001u) << 1;
            c += ((p4 >> (2 * d)) & 0x00010001u) << 2;
            c += ((p8 >> (2 * d)) & 0x00010001u) << 3;
            c += ((p16 >> (2 * d)) & 0x00010001u) << 4;
            c += ((p32 >> (2 * d)) & 0x00010001u) << 5;
            c += ((p64 >> (2 * d)) & 0x00010001u) << 6;
            miss16[d] += c;
        }
        p1 = p2 = p4 = p8 = p16 = p32 = p64 = 0;
    };
    auto store_partials = [&](int64_t epoch) {
        const int ph_j = my_phase();
        if (phased) {                           // wave-uniform: every lane takes part in the shuffles
            const int lane = tid_now() & (WAVE - 1);
            for (int sft = 1; sft < ph; ++sft) {
                const int src = lane + sft * ph_t;          // phase 0 receives phase sft (src < 64 there)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double v = __shfl(acc[i], src);
                    if (ph_j == 0) acc[i] += v;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t v = (uint32_t)__shfl((int)miss16[i], src);
                    if (ph_j == 0) miss16[i] += v;
                }
            }
        }
        if (lane_on && ph_j == 0) {
            const int64_t col0 = my_dword(ph_j) * 16;
            double *os = out_score + (slot0 + epoch * slot_stride) * ld + col0;
            uint32_t *om = out_miss + (slot0 + epoch * slot_stride) * ld + col0;
#pragma unroll
            for (int i = 0; i < 16; i += 2) *reinterpret_cast<double2 *>(os + i) = make_double2(acc[i], acc[i + 1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {                 // accessions 8k .. 8k+7
                uint4 a, b;
                a.x = k ? (miss16[0] >> 16) : (miss16[0] & 0xffffu);
                a.y = k ? (miss16[1] >> 16) : (miss16[1] & 0xffffu);
                a.z = k ? (miss16[2] >> 16) : (miss16[2] & 0xffffu);
                a.w = k ? (miss16[3] >> 16) : (miss16[3] & 0xffffu);
                b.x = k ? (miss16[4] >> 16) : (miss16[4] & 0xffffu);
                b.y = k ? (miss16[5] >> 16) : (miss16[5] & 0xffffu);
                b.z = k ? (miss16[6] >> 16) : (miss16[6] & 0xffffu);
                b.w = k ? (miss16[7] >> 16) : (miss16[7] & 0xffffu);
                *reinterpret_cast<uint4 *>(om + 8 * k) = a;
                *reinterpret_cast<uint4 *>(om + 8 * k + 4) = b;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) miss16[i] = 0;
    };
    // four rows of 16 accessions against one table: index bytes by a transpose of the rows' 2-bit fields
    //   e01 nibble k = accession 2k, rows 0-1;  o01 nibble k = accession 2k+1;  then nibbles -> bytes:
    //   w[0] byte m = accession 4m, w[1] byte m = accession 4m+1, w[2]: 4m+2, w[3]: 4m+3
    // (an index with the calls' low bits in bits 0-3 -- ref and alt entries of all four rows in distinct LDS banks -- costs
    // the same 24 operations and measured 8 % SLOWER: the pass is bound by instruction issue, not by the LDS array)
    auto score_quad = [&](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, const double *tab, uint32_t lane_tab = 0u) {
        // the pass is bound by VALU issue: every step below is pinned to the one instruction it needs (the compiler expands
        // the merges into and / and / or and the byte extractions into shift + and: 6.2 instead of 4.2 VALU per lookup)
        const uint32_t M3 = 0x33333333u, MF = 0x0F0F0F0Fu;
        auto bfi = [](uint32_t m, uint32_t a, uint32_t b) -> uint32_t {        // (a & m) | (b & ~m)
            uint32_t d;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(m), "v"(a), "v"(b));
            return d;
        };
#if SNPM_Q4_BITIDX
        // index byte = low bits of the four calls in bits 0-3, high bits in bits 4-7 (one transpose stage more): the LDS
        // bank of an entry is its index mod 32, so entries that differ only in ref / alt calls never share a bank.
        // With the field index below 63 % of the LDS-array cycles are bank-conflict cycles (SQ_LDS_BANK_CONFLICT /
        // SQ_LDS_IDX_ACTIVE, profiles/r02b_sq_fast_packed_q4.txt) -- and yet this form is not faster: 31.9 vs 32.0 ms on
        // 10 000 accessions, 5.9 vs 5.2 ms on 1135: the eight extra VALU instructions cost what the conflicts cost.
        const uint32_t M1 = 0x55555555u;
        const uint32_t l01 = bfi(M1, x0, x1 << 1), h01 = bfi(M1, x0 >> 1, x1);
        const uint32_t l23 = bfi(M1, x2, x3 << 1), h23 = bfi(M1, x2 >> 1, x3);
        const uint32_t e01 = bfi(M3, l01, l23 << 2), o01 = bfi(M3, l01 >> 2, l23);      // nibbles [lo0 lo1 lo2 lo3]
        const uint32_t e23 = bfi(M3, h01, h23 << 2), o23 = bfi(M3, h01 >> 2, h23);      // nibbles [hi0 hi1 hi2 hi3]
#else
#if SNPM_Q4_SWZ && !SNPM_Q4_PROTO_QUAD
        // The LDS bank pair of an entry is its index mod 32 = c0 | c1 << 2 | (c2 & 1) << 4: the sixteen entries whose calls are
        // all ref / alt (3 of 4 lookups on real panels) share EIGHT bank pairs, entries that differ in the fourth row's call
        // always collide -- 63 % of the LDS-array cycles were bank-conflict cycles (profiles/r02b_sq_fast_packed_q4.txt).
        // Swizzle: index bit 1 (row 1's high bit, rarely set) ^= the low bit of row 4's call: two VALU operations per 16
        // lookups here, the table is built at the swizzled positions.  Level 2 also folds the high bits of rows 3 and 4 into
        // bit 3 (row 2's high bit), so that entries with a het / missing call in those rows leave the frequent banks.
        x0 = __builtin_amdgcn_bitop3_b32(x0, x3 << 1, 0xAAAAAAAAu, 0x78);             // x0 ^ ((x3 << 1) & 0xAAAA...)
#if SNPM_Q4_SWZ >= 2
        x1 = __builtin_amdgcn_bitop3_b32(x1, x2 ^ x3, 0xAAAAAAAAu, 0x78);             // x1 ^ ((x2 ^ x3) & 0xAAAA...)
#endif
#endif
        const uint32_t e01 = bfi(M3, x0, x1 << 2), o01 = bfi(M3, x0 >> 2, x1);
        const uint32_t e23 = bfi(M3, x2, x3 << 2), o23 = bfi(M3, x2 >> 2, x3);
#endif
        uint32_t w[4];
#if SNPM_Q4_PROTO_QUAD
        // TIMING EXPERIMENT ONLY (wrong results): the four dwords taken as ready-made index bytes, i.e. what the lookup would
        // cost on a panel stored four rows per byte (tools/ab_q4_proto_quad.sh)
        (void)e01; (void)o01; (void)e23; (void)o23; (void)MF;
        w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3;
#if SNPM_Q4_SWZ
        // the bank swizzle of the shipped kernel on ready-made index bytes (a real four-rows-per-byte panel would store
        // the swizzled bytes: these 8 operations per 16 lookups would not exist)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_bitop3_b32(w[j], w[j] >> 5, 0x02020202u, 0x78);
#endif
#else
        w[0] = bfi(MF, e01, e23 << 4);
        w[1] = bfi(MF, o01, o23 << 4);
        w[2] = bfi(MF, e01 >> 4, e23);
        w[3] = bfi(MF, o01 >> 4, o23);
#endif
        const char *tabc = reinterpret_cast<const char *>(tab);
#define Q4_IDX(D, W, SEL)                                                                                             \
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL              \
        : "=v"(D) : "v"(three), "v"(W))
#pragma unroll
        for (int h = 0; h < 2; ++h) {                     // eight lookups in flight, then their additions
            uint32_t a8[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // byte 2h of w[j] -> accession 8h + j, byte 2h+1 -> 8h + 4 + j
                if (h == 0) { Q4_IDX(a8[j], w[j], "BYTE_0"); Q4_IDX(a8[4 + j], w[j], "BYTE_1"); }
                else        { Q4_IDX(a8[j], w[j], "BYTE_2"); Q4_IDX(a8[4 + j], w[j], "BYTE_3"); }
            }
            double t[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) t[c] = *reinterpret_cast<const double *>(tabc + (a8[c] + lane_tab));
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[8 * h + c] += t[c];
        }
#undef Q4_IDX
    };
    // the 8 missing-bit words of a group into the bit-sliced counters
    auto count_group = [&](const uint32_t (&x)[G]) {
        uint32_t t2a, t2b, t4a, t4b, t8;
        Q4_CSA(t2a, p1, p1, missing_bits(x[0]), missing_bits(x[1]));
        Q4_CSA(t2b, p1, p1, missing_bits(x[2]), missing_bits(x[3]));
        Q4_CSA(t4a, p2, p2, t2a, t2b);
        Q4_CSA(t2a, p1, p1, missing_bits(x[4]), missing_bits(x[5]));
        Q4_CSA(t2b, p1, p1, missing_bits(x[6]), missing_bits(x[7]));
        Q4_CSA(t4b, p2, p2, t2a, t2b);
        Q4_CSA(t8, p4, p4, t4a, t4b);
        uint32_t c = p8 & t8;  p8 ^= t8;
        uint32_t d = p16 & c;  p16 ^= c;
        c = p32 & d;           p32 ^= d;
        p64 ^= c;
    };
    auto score_group = [&](uint32_t (&x)[G], int gi) {
#pragma unroll
        for (int u = 0; u < G; ++u) asm volatile("" : "+v"(x[u]));       // the group's rows become visible here, not earlier
        score_quad(x[0], x[1], x[2], x[3], s_tab + (2 * gi) * 256);
        score_quad(x[4], x[5], x[6], x[7], s_tab + (2 * gi + 1) * 256);
        count_group(x);
    };
    // LUT rows of the tile that starts at matched row tr: TR * 4 doubles, one per thread (0.0 past the end of the list);
    // blocks with fewer than TR * 4 threads copy the rest synchronously
    auto fetch_l4 = [&](int64_t tr, bool on) -> double {
        const int t = tid_now();
        return (on && t < TR * 4 && tr + (t >> 2) < rend) ? (lut + 4 * tr)[t] : 0.0;
    };
    auto store_l4 = [&](int64_t tr, double pre) {
        const int t = tid_now();
        if (t < TR * 4) s_l4[t] = pre;
        for (int i = t + nthr; i < TR * 4; i += nthr) s_l4[i] = (tr + (i >> 2) < rend) ? (lut + 4 * tr)[i] : 0.0;
    };
    // phased waves: the G rows of my phase's group in step `it` of the tile that starts at matched row tr (`rows` of it exist)
    auto ph_load = [&](uint32_t (&x)[G], int64_t tr, int rows, int it, int ph_j) {
        const int grp = it * ph + ph_j;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const bool on = lane_on && grp * G + u < rows;              // grp < NG follows (rows <= TR)
            if constexpr (GATHER) {
                x[u] = 0u;
                if (on) {
                    const uint32_t *ptr = reinterpret_cast<const uint32_t *>(db + row_idx[tr + grp * G + u] * pitch + my_dword(ph_j) * 4);
                    x[u] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
                }
            } else {
                x[u] = on ? load(tr + (int64_t)it * ph * G + u) : 0u;   // lane_off carries my phase's rows
            }
        }
    };
    int64_t last_epoch = 0;

    // The walk over my tiles, in two exclusive copies: ordinary waves and phased waves (PH).  One loop with both kinds of
    // scoring inside kept the scalars and row registers of both alive at once (72 scalar and 47 vector registers spilled, and
    // every reload waits for ALL loads in flight: the kernel ran at a quarter of its speed); the barriers of the two copies
    // pair up because every wave of a block walks the same tiles.
    auto walk = [&](auto ph_tag) {
        constexpr bool PH = decltype(ph_tag)::value;
        if (!(tile_of(0) < n_tiles_total)) return;
        const int ph_j = PH ? my_phase() : 0;
        // two register sets of G rows (a third one, 16 to 24 row loads in flight per lane, measured no gain)
        uint32_t xa[G], xb[G];
        {
            const int64_t tr_first = rbase + (int64_t)tile_of(0) * TR;
            const double pre = fetch_l4(tr_first, true);
            const int rows0 = (int)((rend - tr_first < TR) ? (rend - tr_first) : TR);
            if constexpr (PH) {
                ph_load(xa, tr_first, rows0, 0, ph_j);
            } else {
#pragma unroll
                for (int u = 0; u < G; ++u) xa[u] = (wave_on && u < rows0) ? load(tr_first + u) : 0u;
            }
            store_l4(tr_first, pre);
        }
        __syncthreads();
        build_tables();
        __syncthreads();

        int tiles_in_epoch = 0;
        int64_t epoch = 0;
        for (int k = 0, T = tile_of(0); T < n_tiles_total; T = tile_of(++k)) {
            if (!SEG && tiles_in_epoch == EPOCH_TILES * RUN) {
                store_partials(epoch);
                ++epoch;
                tiles_in_epoch = 0;
            }
            ++tiles_in_epoch;
            const int64_t tr0 = rbase + (int64_t)T * TR;
            const int rows = (int)((rend - tr0 < TR) ? (rend - tr0) : TR);
            const int Tn = tile_of(k + 1);
            const bool more = (Tn < n_tiles_total);
            const int64_t ntr0 = more ? rbase + (int64_t)Tn * TR : tr0;             // my next tile (or a harmless re-read)
            const double pre = fetch_l4(ntr0, more);               // waited for at the end of this tile

            if constexpr (PH) {
                // xa holds step 0 of this tile; the step after the current one (or step 0 of my next tile) is requested first
                const int nrows = more ? (int)((rend - ntr0 < TR) ? (rend - ntr0) : TR) : 0;
                for (int it = 0; it < n_it; ++it) {
                    if (it + 1 < n_it) ph_load(xb, tr0, rows, it + 1, ph_j);
                    else ph_load(xb, ntr0, nrows, 0, ph_j);
#pragma unroll
                    for (int u = 0; u < G; ++u) asm volatile("" : "+v"(xa[u]));
                    const int grp = it * ph + ph_j;
                    if (lane_on && grp * G < rows) {          // rows of the group past the end read as 0 against table rows of 0.0
                        const uint32_t lane_tab = (uint32_t)grp * (2u * 256u * (uint32_t)sizeof(double));
                        score_quad(xa[0], xa[1], xa[2], xa[3], s_tab, lane_tab);
                        score_quad(xa[4], xa[5], xa[6], xa[7], s_tab + 256, lane_tab);
                        count_group(xa);
                    }
#pragma unroll
                    for (int u = 0; u < G; ++u) xa[u] = xb[u];
                }
            } else if (!wave_on) {
                // a wave whose lanes all lie past the last accession only helps to build the tables
            } else if (rows == TR) {
#pragma unroll
                for (int g = 0; g < TR / G; g += 2) {
                    const int64_t rnext = tr0 + (int64_t)(g + 1) * G;
                    // the group after the pair: inside this tile, or the first group of my next tile (a partial last tile
                    // is followed by PREFETCH_PAD_ROWS >= G readable rows)
                    const int64_t rafter = (g + 2 < TR / G) ? rnext + G : ntr0;
                    load_group(xb, rnext);
                    score_group(xa, g);
                    load_group(xa, rafter);
                    score_group(xb, g + 1);
                }
            } else {
                // a partial tile (the last one of the list or of a part): group by group, rows past the end read as 0
                for (int g = 0; g * G < rows; ++g) {
#pragma unroll
                    for (int u = 0; u < G; ++u) xb[u] = (g * G + u < rows) ? load(tr0 + g * G + u) : 0u;
#pragma unroll
                    for (int u = 0; u < G; ++u) asm volatile("" : "+v"(xb[u]));
                    score_quad(xb[0], xb[1], xb[2], xb[3], s_tab + (2 * g) * 256);
                    score_quad(xb[4], xb[5], xb[6], xb[7], s_tab + (2 * g + 1) * 256);
                    count_group(xb);
                }
            }
            flush_planes();
            if (more) store_l4(ntr0, pre);
#if !SNPM_Q4_PROTO_ONE_BARRIER                   // (1: timing experiment only, results are wrong -- what a second table set would save)
            __syncthreads();                      // every wave is done with this tile's tables; s_l4 holds the next rows
#endif
            if (more) build_tables();
            __syncthreads();
        }
        last_epoch = epoch;
    };
    if (phased) walk(std::true_type{});
    else walk(std::false_type{});
    store_partials(last_epoch);
}
#undef Q4_CSA

// ------------------------------------------------------------------------------------------------
// Fast pass for HARD-CALL samples on a packed panel: every weight is 0 or 1 (BED input, VCF without PL:
// ParseInputs.get_wei_from_GT, core/parsers.py:118-127), so the score of an accession is a COUNT of SNPs and
// needs neither the LUT nor fp64: with the two bit planes of the 16 calls of a dword,
//     lo = x & 0x5555..., hi = (x >> 1) & 0x5555...      (code = lo + 2 hi: 0 ref, 1 alt, 2 het, 3 missing)
// a call scores when  (~lo & ~hi & R_ref) | (lo & ~hi & R_alt) | (~lo & hi & R_het),  R_c = 0x5555... or 0 from the
// row's three weight bits (wave-uniform, scalar registers), and is missing when lo & hi (or hi with skip_hets).
// Both bit vectors are counted per accession with the bit-sliced carry-save scheme of k_fast_packed_q4, flushed
// into 16-bit counters every 64 rows.  ~1.1 integer operations and 0.25 B per comparison, no LDS, no barriers.
// Geometry (16 accessions per lane, tile-interleaved parts, epochs, prefetch pipeline) as k_fast_packed_q4;
// partial scores are written as fp64 counts so that the reduce kernels are shared.  wbits[r] = ref | het << 1 |
// alt << 2 for query row r, padded to a multiple of 8 entries.
// carry-save adder of three bit vectors: two v_bitop3_b32 (majority 0xE8, parity 0x96)
#define BITS_CSA(H, L, A, B, C)                                                 \
    do {                                                                        \
        const uint32_t a_ = (A), b_ = (B), c_ = (C);                            \
        (H) = __builtin_amdgcn_bitop3_b32(a_, b_, c_, 0xE8);                    \
        (L) = __builtin_amdgcn_bitop3_b32(a_, b_, c_, 0x96);                    \
    } while (0)

template <bool SKIP, bool GATHER, bool NT>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK, 6)
k_fast_bits(const int8_t *__restrict__ db, int64_t pitch, const int64_t *__restrict__ row_idx, int64_t row0, int64_t n,
            const uint8_t *__restrict__ wbits, double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld,
            int64_t n_acc)
{
    // Rows are requested in groups of 4 (two register sets: 4 to 8 row loads in flight per lane) and counted in groups of 8;
    // 256-row tiles.  Measured with the arithmetic stripped (tools/micro/read_patterns.hip, this panel's 2560-B rows,
    // one-wave blocks): groups of 8 / 64-row tiles 21.0 ms, groups of 4 / 128-row tiles 19.2 ms per 125 GB; the kernel's own
    // pattern-only build 22.3 -> 20.5 ms.  (Unpipelined groups of 8 -- request, wait, count -- measured the same as this.)
    constexpr int H = 4;                    // rows per load group (load_rows)
    constexpr int TR = BITS_TILE_ROWS;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    // PHASED waves, as in k_fast_packed_q4: a wave that starts t <= 32 dwords before the end of the row (the second wave of the
    // 1135-accession panel holds 7) gives lane l the dword l % t of the 8-row groups ph_j + ph * i of every tile, ph_j = l / t <
    // ph = min(8, 64 / t), classifies its rows with per-lane weight masks and adds its phases together before it writes its
    // counts (1024 and 1040 accessions x 40M rows took 1.7 and 3.3 ms: the nearly empty wave cost as much as the full one).
    constexpr int PH_MAX = 8;
    const int lane = tid & (WAVE - 1);
    const int64_t dw_first = (int64_t)blockIdx.x * nthr + __builtin_amdgcn_readfirstlane(tid - lane);   // first dword of my wave (scalar)
    const int64_t dw_end = (pitch / 4 < (n_acc + 15) / 16) ? pitch / 4 : (n_acc + 15) / 16;   // dwords of a row that hold accessions
    int ph_t = 0, ph = 1;                               // wave-uniform
    if (SNPM_Q4_PHASES && dw_first < dw_end && dw_end - dw_first <= WAVE / 2) {
        ph_t = (int)(dw_end - dw_first);
        ph = (WAVE / ph_t < PH_MAX) ? WAVE / ph_t : PH_MAX;
    }
    const bool phased = ph > 1;
    const int ph_j = phased ? lane / ph_t : 0;           // my phase
    const int64_t my_dw = phased ? dw_first + (lane - ph_j * ph_t) : (int64_t)blockIdx.x * nthr + tid;
    const int64_t byte0 = my_dw * 4;
    const int64_t col0 = byte0 * 4;
    const bool lane_on = phased ? (ph_j < ph) : (byte0 < pitch && col0 < ld);
    const int64_t p = blockIdx.y;
    const int64_t P = gridDim.y;
    const int64_t n_tiles_total = (n + TR - 1) / TR;

    uint32_t hit16[8], miss16[8];       // [d]: accession d (low half) and d + 8 (high half)
    uint32_t h1 = 0, h2 = 0, h4 = 0, h8 = 0, h16 = 0, h32 = 0, h64 = 0;   // bit-sliced hit counts of the current tile
    uint32_t m1 = 0, m2 = 0, m4 = 0, m8 = 0, m16 = 0, m32 = 0, m64 = 0;   // ... and missing counts
#pragma unroll
    for (int i = 0; i < 8; ++i) hit16[i] = miss16[i] = 0;

    // phased lanes: + the first row of my phase's group
    const uint32_t lane_off = lane_on ? (uint32_t)byte0 + (uint32_t)(ph_j * 8) * (uint32_t)pitch : 0u;
    auto load = [&](int64_t rr) -> uint32_t {
        const int64_t prow = GATHER ? row_idx[rr] : (row0 + rr);
        const int8_t *rowbase = db + prow * pitch;
        uint32_t off = lane_off;
        asm volatile("" : "+v"(off));
        const uint32_t *ptr = reinterpret_cast<const uint32_t *>(rowbase + off);
        return NT ? __builtin_nontemporal_load(ptr) : *ptr;
    };
    // H consecutive rows through a buffer resource (see k_fast_packed_q4::load_group)
    auto load_rows = [&](uint32_t (&x)[4], int64_t r) {
        if constexpr (!GATHER) {
            const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<int8_t *>(db + (row0 + r) * pitch), 0, (int)(4 * pitch), 0x00020000);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, lane_off, (int)(u * pitch), NT ? 2 : 0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = load(r + u);
        }
    };
    // (hit bits, missing bits) of one row
    auto classify = [&](uint32_t x, uint32_t wb, uint32_t &hit, uint32_t &mis) {
        // the row's truth table T[code] = (ref, alt, het, 0) in algebraic normal form over (lo, hi):
        //   hit = D ^ (lo & A) ^ (hi & B) ^ (lo & hi & C),  D = ref, A = ref ^ alt, B = ref ^ het, C = ref ^ alt ^ het
        // (four wave-uniform masks; three 3-input boolean operations per row instead of eight 2-input ones)
        // every step one v_bitop3_b32 (truth-table immediates: (a & b) ^ c = 0x6A, a & b & c = 0x80); the masks carry the
        // 0x5555... themselves, so x and x >> 1 need no masking of their own
        const uint32_t M = 0x55555555u, s1 = x >> 1;
        const uint32_t r = wb & 1u, h = SKIP ? 0u : ((wb >> 1) & 1u), a = (wb >> 2) & 1u;
        const uint32_t md = r ? M : 0u, ma = (r ^ a) ? M : 0u;
        const uint32_t mb = (r ^ h) ? M : 0u, mc = (r ^ a ^ h) ? M : 0u;
        const uint32_t both = __builtin_amdgcn_bitop3_b32(x, s1, M, 0x80);          // lo & hi
        uint32_t t = __builtin_amdgcn_bitop3_b32(x, ma, md, 0x6A);                  // (lo & ma) ^ md
        t = __builtin_amdgcn_bitop3_b32(s1, mb, t, 0x6A);                           // ^ (hi & mb)
        hit = __builtin_amdgcn_bitop3_b32(both, mc, t, 0x6A);                       // ^ (lo & hi & mc)
        mis = SKIP ? (s1 & M) : both;
    };
    auto ripple = [&](uint32_t c, uint32_t &a1, uint32_t &a2, uint32_t &a4, uint32_t &a8, uint32_t &a16, uint32_t &a32, uint32_t &a64) {
        uint32_t t;
        t = a1 & c; a1 ^= c; c = t;
        t = a2 & c; a2 ^= c; c = t;
        t = a4 & c; a4 ^= c; c = t;
        t = a8 & c; a8 ^= c; c = t;
        t = a16 & c; a16 ^= c; c = t;
        t = a32 & c; a32 ^= c; c = t;
        a64 ^= c;
    };
    auto flush = [&]() {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const uint32_t k = 0x00010001u;
            hit16[d] += ((h1 >> (2 * d)) & k) + (((h2 >> (2 * d)) & k) << 1) + (((h4 >> (2 * d)) & k) << 2) +
                        (((h8 >> (2 * d)) & k) << 3) + (((h16 >> (2 * d)) & k) << 4) + (((h32 >> (2 * d)) & k) << 5) +
                        (((h64 >> (2 * d)) & k) << 6);
            miss16[d] += ((m1 >> (2 * d)) & k) + (((m2 >> (2 * d)) & k) << 1) + (((m4 >> (2 * d)) & k) << 2) +
                         (((m8 >> (2 * d)) & k) << 3) + (((m16 >> (2 * d)) & k) << 4) + (((m32 >> (2 * d)) & k) << 5) +
                         (((m64 >> (2 * d)) & k) << 6);
        }
        h1 = h2 = h4 = h8 = h16 = h32 = h64 = 0;
        m1 = m2 = m4 = m8 = m16 = m32 = m64 = 0;
    };
    auto store_partials = [&](int64_t epoch) {
        if (phased) {                           // wave-uniform: every lane takes part in the shuffles
            for (int sft = 1; sft < ph; ++sft) {
                const int src = lane + sft * ph_t;          // phase 0 receives phase sft (src < 64 there)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t vh = (uint32_t)__shfl((int)hit16[i], src), vm = (uint32_t)__shfl((int)miss16[i], src);
                    if (ph_j == 0) { hit16[i] += vh; miss16[i] += vm; }
                }
            }
        }
        if (lane_on && ph_j == 0) {
            double *os = out_score + (epoch * P + p) * ld + col0;
            uint32_t *om = out_miss + (epoch * P + p) * ld + col0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {                 // accessions 8k .. 8k+7
#pragma unroll
                for (int d = 0; d < 8; d += 2) {
                    const uint32_t c0 = k ? (hit16[d] >> 16) : (hit16[d] & 0xffffu);
                    const uint32_t c1 = k ? (hit16[d + 1] >> 16) : (hit16[d + 1] & 0xffffu);
                    *reinterpret_cast<double2 *>(os + 8 * k + d) = make_double2((double)c0, (double)c1);
                }
                uint4 a, b;
                a.x = k ? (miss16[0] >> 16) : (miss16[0] & 0xffffu);
                a.y = k ? (miss16[1] >> 16) : (miss16[1] & 0xffffu);
                a.z = k ? (miss16[2] >> 16) : (miss16[2] & 0xffffu);
                a.w = k ? (miss16[3] >> 16) : (miss16[3] & 0xffffu);
                b.x = k ? (miss16[4] >> 16) : (miss16[4] & 0xffffu);
                b.y = k ? (miss16[5] >> 16) : (miss16[5] & 0xffffu);
                b.z = k ? (miss16[6] >> 16) : (miss16[6] & 0xffffu);
                b.w = k ? (miss16[7] >> 16) : (miss16[7] & 0xffffu);
                *reinterpret_cast<uint4 *>(om + 8 * k) = a;
                *reinterpret_cast<uint4 *>(om + 8 * k + 4) = b;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) hit16[i] = miss16[i] = 0;
    };
    int64_t last_epoch = 0;

    if (!phased && p < n_tiles_total) {
        uint32_t xa[H], xb[H];
        // the eight weight bytes of a counting group (one scalar dwordx2) travel with its rows, one group ahead
        auto wbits8 = [&](int64_t r) -> uint64_t { return *reinterpret_cast<const uint64_t *>(wbits + r); };
        uint64_t w8;
#pragma unroll
        for (int u = 0; u < H; ++u) xa[u] = load(p * TR + u);
        w8 = wbits8(p * TR);

        int tiles_in_epoch = 0;
        int64_t epoch = 0;
        for (int64_t T = p; T < n_tiles_total; T += P) {
            if (tiles_in_epoch == EPOCH_TILES) {              // EPOCH_TILES * BITS_TILE_ROWS = 16384 rows per epoch: the 16-bit counters hold them (static_assert below the tile constants)
                store_partials(epoch);
                ++epoch;
                tiles_in_epoch = 0;
            }
            ++tiles_in_epoch;
            const int64_t tr0 = T * TR;
            const int rows = (int)((n - tr0 < TR) ? (n - tr0) : TR);
            const bool more = (T + P < n_tiles_total);
            const int64_t ntr0 = more ? (T + P) * TR : tr0;
            const int full8 = rows / 8;

            for (int it = 0; it < full8; ++it) {
                // rows rb .. rb+3 are in xa (requested one step ago), rb+4 .. rb+7 are requested now
                const int64_t rb = tr0 + (int64_t)it * 8;
                const int64_t rn = (it + 1 < TR / 8) ? rb + 8 : ntr0;        // the next counting group: in this tile or my next tile
                load_rows(xb, rb + H);
                const uint64_t wcur = w8;
#pragma unroll
                for (int u = 0; u < H; ++u) asm volatile("" : "+v"(xa[u]));
#ifndef SNPM_FAST_PATTERN_ONLY
                uint32_t hb[8], mb[8];
#pragma unroll
                for (int u = 0; u < H; ++u) classify(xa[u], (uint32_t)(wcur >> (8 * u)) & 0xffu, hb[u], mb[u]);
#else       // diagnostic build: the loads and the loop structure without the arithmetic
                h1 ^= xa[0] ^ xa[1] ^ xa[2] ^ xa[3];
#endif
                load_rows(xa, rn);
                w8 = wbits8(rn);
#pragma unroll
                for (int u = 0; u < H; ++u) asm volatile("" : "+v"(xb[u]));
#ifndef SNPM_FAST_PATTERN_ONLY
#pragma unroll
                for (int u = 0; u < H; ++u) classify(xb[u], (uint32_t)(wcur >> (8 * (H + u))) & 0xffu, hb[H + u], mb[H + u]);
                {   // 8 rows: both bit vectors into their bit-sliced counters with carry-save adders
                    uint32_t t2a_, t2b_, t4a_, t4b_, t8_, c_, d_;
                    BITS_CSA(t2a_, h1, h1, hb[0], hb[1]); BITS_CSA(t2b_, h1, h1, hb[2], hb[3]);
                    BITS_CSA(t4a_, h2, h2, t2a_, t2b_);
                    BITS_CSA(t2a_, h1, h1, hb[4], hb[5]); BITS_CSA(t2b_, h1, h1, hb[6], hb[7]);
                    BITS_CSA(t4b_, h2, h2, t2a_, t2b_);
                    BITS_CSA(t8_, h4, h4, t4a_, t4b_);
                    c_ = h8 & t8_;  h8 ^= t8_;  d_ = h16 & c_;  h16 ^= c_;  c_ = h32 & d_;  h32 ^= d_;  h64 ^= c_;
                    BITS_CSA(t2a_, m1, m1, mb[0], mb[1]); BITS_CSA(t2b_, m1, m1, mb[2], mb[3]);
                    BITS_CSA(t4a_, m2, m2, t2a_, t2b_);
                    BITS_CSA(t2a_, m1, m1, mb[4], mb[5]); BITS_CSA(t2b_, m1, m1, mb[6], mb[7]);
                    BITS_CSA(t4b_, m2, m2, t2a_, t2b_);
                    BITS_CSA(t8_, m4, m4, t4a_, t4b_);
                    c_ = m8 & t8_;  m8 ^= t8_;  d_ = m16 & c_;  m16 ^= c_;  c_ = m32 & d_;  m32 ^= d_;  m64 ^= c_;
                }
#else
                h1 ^= xb[0] ^ xb[1] ^ xb[2] ^ xb[3];
                m1 ^= (uint32_t)wcur;
#endif
                if (((it + 1) * 8) % BITS_FLUSH_ROWS == 0) flush();
            }
            for (int r = full8 * 8; r < rows; ++r) {            // at most 7 rows: last tile of all
                uint32_t hb, mb;
                classify(load(tr0 + r), wbits[tr0 + r], hb, mb);
                ripple(hb, h1, h2, h4, h8, h16, h32, h64);
                ripple(mb, m1, m2, m4, m8, m16, m32, m64);
            }
            flush();
        }
        last_epoch = epoch;
    }
    // the same walk for a phased wave: per step the 8 rows of my phase's group (no second register set: one-wave blocks, the
    // other resident waves cover the wait), weight bytes and masks per lane
    if (phased && p < n_tiles_total) {
        constexpr int NGT = TR / 8;                     // counting groups per tile
        const int n_it = (NGT + ph - 1) / ph;
        int tiles_in_epoch = 0;
        int64_t epoch = 0;
        for (int64_t T = p; T < n_tiles_total; T += P) {
            if (tiles_in_epoch == EPOCH_TILES) {
                store_partials(epoch);
                ++epoch;
                tiles_in_epoch = 0;
            }
            ++tiles_in_epoch;
            const int64_t tr0 = T * TR;
            const int rows = (int)((n - tr0 < TR) ? (n - tr0) : TR);
            for (int it = 0; it < n_it && it * ph * 8 < rows; ++it) {
                const int grp = it * ph + ph_j;
                uint32_t x[8];
                uint64_t w8 = 0;
                if (lane_on && grp * 8 < rows) w8 = *reinterpret_cast<const uint64_t *>(wbits + tr0 + grp * 8);   // padded to 8 entries
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool on = lane_on && grp * 8 + u < rows;
                    if constexpr (GATHER) {
                        x[u] = 0u;
                        if (on) {
                            const uint32_t *ptr = reinterpret_cast<const uint32_t *>(db + row_idx[tr0 + grp * 8 + u] * pitch + byte0);
                            x[u] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
                        }
                    } else {
                        x[u] = on ? load(tr0 + (int64_t)it * ph * 8 + u) : 0u;      // lane_off carries my phase's rows
                    }
                }
                uint32_t hb[8], mb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // rows past the end: x = 0 and weight byte 0 (a partial last group reads the padding of wbits, which is 0)
                    const bool on = grp * 8 + u < rows;
                    classify(x[u], on ? (uint32_t)(w8 >> (8 * u)) & 0xffu : 0u, hb[u], mb[u]);
                }
                {
                    uint32_t t2a_, t2b_, t4a_, t4b_, t8_, c_, d_;
                    BITS_CSA(t2a_, h1, h1, hb[0], hb[1]); BITS_CSA(t2b_, h1, h1, hb[2], hb[3]);
                    BITS_CSA(t4a_, h2, h2, t2a_, t2b_);
                    BITS_CSA(t2a_, h1, h1, hb[4], hb[5]); BITS_CSA(t2b_, h1, h1, hb[6], hb[7]);
                    BITS_CSA(t4b_, h2, h2, t2a_, t2b_);
                    BITS_CSA(t8_, h4, h4, t4a_, t4b_);
                    c_ = h8 & t8_;  h8 ^= t8_;  d_ = h16 & c_;  h16 ^= c_;  c_ = h32 & d_;  h32 ^= d_;  h64 ^= c_;
                    BITS_CSA(t2a_, m1, m1, mb[0], mb[1]); BITS_CSA(t2b_, m1, m1, mb[2], mb[3]);
                    BITS_CSA(t4a_, m2, m2, t2a_, t2b_);
                    BITS_CSA(t2a_, m1, m1, mb[4], mb[5]); BITS_CSA(t2b_, m1, m1, mb[6], mb[7]);
                    BITS_CSA(t4b_, m2, m2, t2a_, t2b_);
                    BITS_CSA(t8_, m4, m4, t4a_, t4b_);
                    c_ = m8 & t8_;  m8 ^= t8_;  d_ = m16 & c_;  m16 ^= c_;  c_ = m32 & d_;  m32 ^= d_;  m64 ^= c_;
                }
                if (((it + 1) * 8) % BITS_FLUSH_ROWS == 0) flush();
            }
            flush();
        }
        last_epoch = epoch;
    }
    store_partials(last_epoch);
}
#undef BITS_CSA

// Blocked summation of the M = n_epochs*P partial slots (deterministic, no atomics):
//   k_reduce_groups: group g = slots [g*REDUCE_GROUP, ...) added sequentially   -> grp [n_groups, ld]
//   k_reduce       : groups added sequentially                                  -> score, ninfo = n - miss
// Every term passes through at most REDUCE_GROUP + n_groups additions here.
__global__ void k_reduce_groups(const double *__restrict__ part_score, const uint32_t *__restrict__ part_miss,
                                int64_t n_slots, int64_t ld, int64_t n_acc, double *__restrict__ grp_score,
                                uint32_t *__restrict__ grp_miss)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = blockIdx.y;
    if (a >= n_acc) return;
    const int64_t s0 = g * REDUCE_GROUP;
    const int64_t s1 = (s0 + REDUCE_GROUP < n_slots) ? s0 + REDUCE_GROUP : n_slots;
    double s = 0.0;
    uint32_t m = 0;
    int64_t k = s0;
    for (; k + 8 <= s1; k += 8) {
        double v[8];
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = part_score[(k + u) * ld + a];
            c[u] = part_miss[(k + u) * ld + a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s + v[u];
            m += c[u];
        }
    }
    for (; k < s1; ++k) {
        s = s + part_score[k * ld + a];
        m += part_miss[k * ld + a];
    }
    grp_score[g * ld + a] = s;
    grp_miss[g * ld + a] = m;
}

// Certificate of SNPM_MODE_EXACT, fused into the last reduce step: accession a is appended to `cols` when an
// integer lies inside [score - E, score + E] (or the interval reaches below zero, or a < force_first: tests),
// E = *d_eref + efast.  The list order depends on the atomics; what is computed from it does not.  *count may
// exceed `cap` (then only the first cap entries were stored and the caller's dense fallback runs).
__device__ __forceinline__ void flag_if_uncertain(double s, int64_t a, const double *__restrict__ d_eref, double efast,
                                                  int force_first, int32_t *__restrict__ cols, int *__restrict__ count, int cap)
{
    const double E = *d_eref + efast;
    const double lo = s - E, hi = s + E;
    if (!(lo >= 0.0) || floor(lo) != floor(hi) || a < force_first) {
        const int k = atomicAdd(count, 1);
        if (k < cap) cols[k] = (int32_t)a;
    }
}

__global__ void k_reduce(const double *__restrict__ part_score, const uint32_t *__restrict__ part_miss,
                         int64_t n_parts, int64_t ld, int64_t n_acc, int64_t n_rows, double *__restrict__ score,
                         int64_t *__restrict__ ninfo, const double *__restrict__ d_eref, double efast, int force_first,
                         int32_t *__restrict__ cols, int *__restrict__ count, int cap)
{
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_acc) return;
    double s = 0.0;
    int64_t m = 0;
    int64_t p = 0;
    for (; p + 8 <= n_parts; p += 8) {
        double v[8];
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = part_score[(p + u) * ld + a];
            c[u] = part_miss[(p + u) * ld + a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s + v[u];
            m += c[u];
        }
    }
    for (; p < n_parts; ++p) {
        s = s + part_score[p * ld + a];
        m += part_miss[p * ld + a];
    }
    score[a] = s;
    ninfo[a] = n_rows - m;
    if (d_eref) flag_if_uncertain(s, a, d_eref, efast, force_first, cols, count, cap);
}

// ------------------------------------------------------------------------------------------------
// Jobs scored SNP slab after SNP slab (panels larger than HBM): running totals.
//   k_carry_add   totals += this slab's fast-pass results (one fp64 addition per slab and accession, in slab
//                 order); thread 0 adds the slab's error bound (*d_eref + efast) onto the job's.
//   k_carry_flag  the certificate over the whole job, after the last slab: as flag_if_uncertain with the summed bound.
__global__ void k_carry_add(double *__restrict__ tot_score, int64_t *__restrict__ tot_ninfo, const double *__restrict__ score,
                            const int64_t *__restrict__ ninfo, int64_t n_acc, double *__restrict__ tot_E,
                            const double *__restrict__ d_eref, double efast)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0) tot_E[0] += (d_eref ? d_eref[0] : 0.0) + efast;
    if (a >= n_acc) return;
    tot_score[a] = tot_score[a] + score[a];
    tot_ninfo[a] += ninfo[a];
}

__global__ void k_carry_flag(const double *__restrict__ tot_score, int64_t n_acc, const double *__restrict__ tot_E,
                             double e_extra, int force_first, int32_t *__restrict__ cols, int *__restrict__ count, int cap)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_acc) return;
    flag_if_uncertain(tot_score[a], a, tot_E, e_extra, force_first, cols, count, cap);
}

// A + (match ? w : 0.0) in the reference's per-class sums, evaluated as fma(m, w, A) with m = 1.0 or 0.0: the
// same bits (1.0 * w is w and 0.0 * w is 0 exactly for finite w, then ONE rounding of the sum), but the select
// acts on the high dword of m only: v_cmp + v_cndmask + v_fmac_f64 instead of v_cmp + 2 v_cndmask + v_add_f64
// (measured on k_strict4: 2.5 -> 3.8 TB/s).
__device__ __forceinline__ double add_if(double acc, bool match, double w)
{
    return __builtin_fma(__hiloint2double(match ? 0x3FF00000 : 0, 0), w, acc);
}
// the plain form for the latency-bound sparse kernels (one lane per segment and column, per-lane weights),
// where the fma form measured slower (k_strict_sparse_T: 0.53 vs 0.36 ms per re-evaluated accession at 50M SNPs)
__device__ __forceinline__ double add_sel(double acc, bool match, double w) { return acc + (match ? w : 0.0); }

// genotype code of (row, accession) in either panel format: int8 -> the byte (negative = missing);
// packed -> 2-bit field, 3 = missing (returned as -1)
__device__ __forceinline__ int code_at(const int8_t *__restrict__ db, int64_t pitch, int64_t prow, int64_t col, int packed)
{
    if (packed) {
        const int v = (((const uint8_t *)db)[prow * pitch + (col >> 2)] >> (2 * (int)(col & 3))) & 3;
        return v == 3 ? -1 : v;
    }
    return db[prow * pitch + col];      // (a non-temporal load here: 3.80 -> 3.69 ms for the second pass of the N = 1 bench: not worth a variant)
}

// ------------------------------------------------------------------------------------------------
// Segmented scoring (k_fast<..., SEG>): many independent row ranges ("segments": the samples of a batch, the
// windows of a cross) of one concatenated matched list in ONE launch.
//   k_eseg_*        per segment: the certificate's error bound.  Segment s = rows [seg_off[s], seg_off[s+1]) scored by
//                   the reference in `chunk`-row matchGTsAccs calls (a window: one call, chunk >= its length):
//                   E_s = (sum_k s_k (len_k + 3 + K_s - k)) u / (1 - m u) + wsum_s gamma(fast adds), 0 when every
//                   weight of the segment is an integer (any order is exact then).  One block per segment.
//   k_reduce_seg    adds the partial slots [slot0[s], slot0[s+1]) of segment s in order -> score / ninfo [n_seg, ldo];
//                   optional certificate: pairs (s, a) whose int(score) is not proven are appended to `pairs`.
//   k_strict_pairs  reference-order chunk sums of the flagged pairs: block = pair, lane = chunk of its segment.
//   k_scan_pairs    the chain of additions over a pair's chunk sums (ScoreList += chunk) and the patch.
// two steps so that long segments (a 200k-row sample = 200 chunks) do not run on one block: a WAVE per chunk, four chunks
// per block, partial[(seg * npart + blockIdx.x) * 3 + {0, 1, 2}] = {sum_k s_k * factor_k, sum_k s_k, non-integer flag};
// the finish kernel adds a segment's partials in a fixed order (the bound is the same in every run).
__global__ void __launch_bounds__(256)
k_eseg_part(const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t seg_base, int npart,
            double *__restrict__ partial)
{
    __shared__ double sm[4][3];
    const int64_t s = blockIdx.y + seg_base;
    const int64_t r0 = seg_off[s], r1 = seg_off[s + 1];
    const int64_t len = r1 - r0;
    const int64_t K = (len + chunk - 1) / chunk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + wave;
    double v = 0.0;
    int nonint = 0;
    int64_t c0 = 0, c1 = 0;
    if (k < K) {
        c0 = r0 + k * chunk;
        c1 = (c0 + chunk < r1) ? c0 + chunk : r1;
        for (int64_t r = c0 + lane; r < c1; r += 64) {
            const double a = fabs(w[3 * r]), b = fabs(w[3 * r + 1]), c = fabs(w[3 * r + 2]);
            v += fmax(a, fmax(b, c));
            if (!(a == floor(a) && b == floor(b) && c == floor(c)) || !(a < 1e300 && b < 1e300 && c < 1e300)) nonint = 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor(v, o);
        nonint |= __shfl_xor(nonint, o);
    }
    if (lane == 0) {
        sm[wave][0] = (k < K) ? v * (double)((c1 - c0) + 3 + (K - k)) : 0.0;
        sm[wave][1] = v;
        sm[wave][2] = (double)nonint;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int q = threadIdx.x;
        const double t = (q == 2) ? fmax(fmax(sm[0][2], sm[1][2]), fmax(sm[2][2], sm[3][2]))
                                  : ((sm[0][q] + sm[1][q]) + sm[2][q]) + sm[3][q];
        partial[((int64_t)blockIdx.y * npart + blockIdx.x) * 3 + q] = t;
    }
}

__global__ void __launch_bounds__(256)
k_eseg_finish(const double *__restrict__ partial, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t fast_adds,
              int64_t seg_base, int npart, double *__restrict__ eseg)
{
    __shared__ double sm[4];
    const int64_t s = blockIdx.x + seg_base;
    const int64_t len = seg_off[s + 1] - seg_off[s];
    const int64_t K = (len + chunk - 1) / chunk;
    const int np = (int)((K + 3) / 4);                  // blocks of k_eseg_part that held chunks of this segment
    const double *p = partial + (int64_t)blockIdx.x * npart * 3;
    double acc = 0.0, wsum = 0.0, flag = 0.0;
    for (int i = threadIdx.x; i < np && i < npart; i += 256) {
        acc += p[3 * i];
        wsum += p[3 * i + 1];
        flag = fmax(flag, p[3 * i + 2]);
    }
    acc = block_sum_256(acc, sm);
    wsum = block_sum_256(wsum, sm);
    flag = block_sum_256(flag, sm);
    if (threadIdx.x == 0) {
        const double u = 1.1102230246251565e-16;
        const double mmax = (double)((chunk < len ? chunk : len) + 3 + K);
        const double mf = (double)fast_adds;
        double e = (acc * u / (1.0 - mmax * u) + wsum * (mf * u / (1.0 - mf * u))) * 1.0000001;
        if (flag == 0.0 && wsum < 9.0e15) e = 0.0;
        eseg[s] = e;
    }
}

__global__ void k_reduce_seg(const double *__restrict__ part_score, const uint32_t *__restrict__ part_miss,
                             const int64_t *__restrict__ slot0, const int64_t *__restrict__ seg_off, int64_t ld, int64_t n_acc,
                             double *__restrict__ score, int64_t *__restrict__ ninfo, int64_t ldo,
                             const double *__restrict__ eseg, int force_first, int32_t *__restrict__ pairs,
                             int *__restrict__ count, int cap, int64_t seg_base)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t sg = blockIdx.y + seg_base;
    if (a >= n_acc) return;
    double v = 0.0;
    int64_t m = 0;
    for (int64_t k = slot0[sg]; k < slot0[sg + 1]; ++k) {
        v = v + part_score[k * ld + a];
        m += part_miss[k * ld + a];
    }
    score[sg * ldo + a] = v;
    ninfo[sg * ldo + a] = (seg_off[sg + 1] - seg_off[sg]) - m;
    if (eseg) {
        const double E = eseg[sg];
        const double lo = v - E, hi = v + E;
        if (!(lo >= 0.0) || floor(lo) != floor(hi) || a < force_first) {
            const int k = atomicAdd(count, 1);
            if (k < cap) {
                pairs[2 * k] = (int32_t)sg;
                pairs[2 * k + 1] = (int32_t)a;
            }
        }
    }
}

// One WAVE per (flagged pair, chunk of its segment): the lanes fetch 64 rows' calls and weights at once, then every lane adds
// them in row order from broadcast values (v_readlane) -- the reference's three sequential per-category sums with the loads
// of 64 rows in flight instead of one dependent gather per row.  (Round 3 gave every chunk ONE lane: a window of `cross` is a
// single chunk, so a flagged (window, accession) pair walked its ~500 gathered rows on one lane -- 0.24 ms of a 0.06-ms pass.)
//   grid.x walks the chunks of a segment, grid.y the flagged pairs (both bounded: no pair flagged = a launch of microseconds)
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(64)
k_strict_pairs(const int8_t *__restrict__ db, int64_t pitch, int packed, const int64_t *__restrict__ row_idx, int64_t row0,
               const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk,
               const int32_t *__restrict__ pairs, const int *__restrict__ count, int cap, int64_t kmax,
               double *__restrict__ sums)
{
    const int np = *count < cap ? *count : cap;
    const int lane = threadIdx.x;
    for (int pr = blockIdx.y; pr < np; pr += gridDim.y) {
        const int64_t sg = pairs[2 * pr], col = pairs[2 * pr + 1];
        const int64_t s0 = seg_off[sg], s1 = seg_off[sg + 1];
        int64_t K = (s1 - s0 + chunk - 1) / chunk;
        if (K < 1) K = 1;                                  // an empty segment: one matchGTsAccs call on no rows
        for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
            const int64_t r0 = s0 + k * chunk;
            const int64_t r1 = (r0 + chunk < s1) ? r0 + chunk : s1;
            double a_ref = 0.0, a_het = 0.0, a_alt = 0.0;
            for (int64_t rb = r0; rb < r1; rb += WAVE) {
                const int64_t r = rb + lane;
                int b = -1;
                double w0 = 0.0, w1 = 0.0, w2 = 0.0;
                if (r < r1) {
                    const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
                    b = code_at(db, pitch, prow, col, packed);
                    w0 = w[3 * r + 0];
                    w1 = w[3 * r + 1];
                    w2 = w[3 * r + 2];
                }
                const int cnt = (int)((r1 - rb < WAVE) ? (r1 - rb) : WAVE);
                for (int i = 0; i < cnt; ++i) {
                    const int bi = __shfl(b, i);
                    a_ref = add_sel(a_ref, bi == 0, __shfl(w0, i));
                    if (!SKIP) a_het = add_sel(a_het, bi == 2, __shfl(w1, i));
                    a_alt = add_sel(a_alt, bi == 1, __shfl(w2, i));
                }
            }
            if (lane == 0) sums[(int64_t)pr * kmax + k] = ((0.0 + a_ref) + a_het) + a_alt;
        }
    }
}

// one wave per pair: lanes fetch 64 chunk sums at a time, lane 0 adds them in order; then score[seg, acc] = total
__global__ void __launch_bounds__(64)
k_scan_pairs(const double *__restrict__ sums, const int64_t *__restrict__ seg_off, int64_t chunk,
             const int32_t *__restrict__ pairs, const int *__restrict__ count, int cap, int64_t kmax,
             double *__restrict__ score, int64_t ldo)
{
    __shared__ double tile[64];
    const int np = *count < cap ? *count : cap;
    for (int pr = blockIdx.x; pr < np; pr += gridDim.x) {
    const int64_t sg = pairs[2 * pr], col = pairs[2 * pr + 1];
    const int64_t len = seg_off[sg + 1] - seg_off[sg];
    int64_t K = (len + chunk - 1) / chunk;
    if (K < 1) K = 1;                                  // an empty segment: one matchGTsAccs call on no rows
    double s = 0.0;
    for (int64_t k0 = 0; k0 < K; k0 += 64) {
        if (k0 + threadIdx.x < K) tile[threadIdx.x] = sums[(int64_t)pr * kmax + k0 + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = (int)((K - k0 < 64) ? (K - k0) : 64);
            for (int i = 0; i < cnt; ++i) s = s + tile[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) score[sg * ldo + col] = s;
    __syncthreads();
    }
}

// totals over the segments in order (TotScoreList += ScoreList per window, core/csmatch.py:88-90) from [n_seg, ldo]
// results; block 0 also leaves sum_s eseg[s] in etot[0] (the totals' share of the per-window bounds)
__global__ void k_tot_seg(const double *__restrict__ score, const int64_t *__restrict__ ninfo, int64_t n_seg, int64_t ldo,
                          int64_t n_acc, double *__restrict__ tot_score, int64_t *__restrict__ tot_ninfo,
                          const double *__restrict__ eseg, double *__restrict__ etot)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0 && eseg) {
        double e = 0.0;
        for (int64_t s = 0; s < n_seg; ++s) e += eseg[s];
        etot[0] = e * 1.0000001;
    }
    if (a >= n_acc) return;
    double t = 0.0;
    int64_t m = 0;
    // the additions stay in window order; the loads of 16 windows are requested together (one dependent load per window made
    // the 399 windows of a `cross` cost 0.13 ms)
    constexpr int U = 16;
    int64_t s = 0;
    for (; s + U <= n_seg; s += U) {
        double v[U];
        int64_t c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = score[(s + u) * ldo + a];
            c[u] = ninfo[(s + u) * ldo + a];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            t = t + v[u];
            m += c[u];
        }
    }
    for (; s < n_seg; ++s) {
        t = t + score[s * ldo + a];
        m += ninfo[s * ldo + a];
    }
    tot_score[a] = t;
    tot_ninfo[a] = m;
}

// Rows of segment `seg`: explicit offsets (windows of a cross) or implicit `chunk`-row pieces of [0, n)
// (the reference's chunk loop, core/snpmatch.py:218-222) -- no offset table to build or upload.
__device__ __forceinline__ void seg_bounds(const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n, int64_t seg,
                                           int64_t &r0, int64_t &r1)
{
    if (seg_off) {
        r0 = seg_off[seg];
        r1 = seg_off[seg + 1];
    } else {
        r0 = seg * chunk;
        r1 = (r0 + chunk < n) ? r0 + chunk : n;
    }
}

// Re-evaluation kernels are launched before the host knows how many accessions the certificate flagged; they
// read the count on the device and leave at once when their tier is not the one that has to run:
//   sparse tier: 1 <= *count <= cap   (k_strict_sparse / _T, k_scan_few, k_patch)
//   dense tier : *count > cap         (k_strict4 / k_strict, k_scan: every accession in reference order)
__device__ __forceinline__ bool dense_tier_off(const int *__restrict__ gate, int cap) { return gate && *gate <= cap; }

// ------------------------------------------------------------------------------------------------
// Strict (reference-order) segment sums.
//   grid.x = segment, grid.y = column blocks of blockDim.x lanes
//   cols: optional list of accession indices (NULL = dense 0..ncols-1)
//   out_score [n_seg, ld] fp64 = ((0 + A_ref) + A_het) + A_alt, out_miss [n_seg, ld] u32
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(256)
k_strict(const int8_t *__restrict__ db, int64_t pitch, int packed, const int64_t *__restrict__ row_idx, int64_t row0,
         const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n, int64_t seg0,
         int64_t n_seg, const int32_t *__restrict__ cols, int64_t ncols, double *__restrict__ out_score,
         uint32_t *__restrict__ out_miss, int64_t ld, const int *__restrict__ gate, int gate_cap)
{
    if (dense_tier_off(gate, gate_cap)) return;
    const int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    const int64_t col = cols ? (int64_t)cols[i] : i;
    // seg = output row (the segment itself is seg0 + seg when the pieces are implicit); a gated launch uses a
    // bounded grid and walks the segments, so that a launch that has nothing to do costs a few microseconds
    for (int64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    int64_t r0, r1;
    seg_bounds(seg_off, chunk, n, seg_off ? seg : seg0 + seg, r0, r1);
    double a_ref = 0.0, a_het = 0.0, a_alt = 0.0;
    uint32_t miss = 0;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {
        int b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t prow = GATHER ? row_idx[r + u] : (row0 + r + u);
            b[u] = code_at(db, pitch, prow, col, packed);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double w0 = w[3 * (r + u) + 0], w1 = w[3 * (r + u) + 1], w2 = w[3 * (r + u) + 2];
            a_ref = add_if(a_ref, b[u] == 0, w0);
            if (!SKIP) a_het = add_if(a_het, b[u] == 2, w1);
            a_alt = add_if(a_alt, b[u] == 1, w2);
            miss += SKIP ? (b[u] < 0 || b[u] == 2) : (b[u] < 0);
        }
    }
    for (; r < r1; ++r) {
        const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
        const int b = code_at(db, pitch, prow, col, packed);
        const double w0 = w[3 * r + 0], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        a_ref = add_if(a_ref, b == 0, w0);
        if (!SKIP) a_het = add_if(a_het, b == 2, w1);
        a_alt = add_if(a_alt, b == 1, w2);
        miss += SKIP ? (b < 0 || b == 2) : (b < 0);
    }
    out_score[seg * ld + i] = ((0.0 + a_ref) + a_het) + a_alt;
    out_miss[seg * ld + i] = miss;
    }
}

// Dense strict kernel: 4 adjacent accession columns per lane (int8 panel: one dword per row; packed panel: one
// byte = four 2-bit calls), same arithmetic and order as k_strict (three sequential per-category sums per column
// and segment).
//   grid.x = segment, grid.y = blocks of blockDim.x lanes x 4 columns;  out_* [n_seg, ld]
//   MASKS: two compares per call instead of three (int8 panels whose calls are all in {0, 1, 2, missing}, and packed
//   panels, whose row byte goes through a 256-entry table of compare-ready bits), see below.

// Table entry of a packed row byte e (four 2-bit calls) for the MASKS form: per call j two predicates,
// G = call in {alt, het} and U = call in {ref, alt}, each at the SIGN bit of a byte (the SDWA compare picks the byte and
// sign-extends it): .x holds calls 0, 1 (bytes G0 U0 G1 U1), .y calls 2, 3; .z = one byte per call, 1 where the call
// counts as missing (code 3, or 2 / 3 with skip_hets), calls 0, 1; .w the same for calls 2, 3: .z + .w is the increment of
// the packed missing counters.
template <bool SKIP>
__device__ __forceinline__ uint4 strict_lut_entry(uint32_t e)
{
    uint32_t d[2] = {0u, 0u}, m[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t code = (e >> (2 * j)) & 3u;
        const uint32_t g = (code == 1u || code == 2u) ? 0x80u : 0u, u = (code < 2u) ? 0x80u : 0u;
        d[j >> 1] |= (g | (u << 8)) << (16 * (j & 1));
        m[j >> 1] |= (uint32_t)(SKIP ? code >= 2u : code == 3u) << (8 * j);
    }
    return make_uint4(d[0], d[1], m[0], m[1]);      // all four words are used: one ds_read_b128 (a b96 takes twice the cycles)
}

template <bool SKIP, bool GATHER, bool PACKED, bool MASKS>
__device__ __forceinline__ void
strict4_segments(const int8_t *__restrict__ db, int64_t pitch, const int64_t *__restrict__ row_idx, int64_t row0,
                 const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n, int64_t seg0,
                 int64_t n_seg, int64_t ncols, double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld,
                 int64_t c0, const uint4 *lut)
{
    for (int64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {          // one pass unless the launch is gated (see k_strict)
    int64_t r0, r1;
    seg_bounds(seg_off, chunk, n, seg_off ? seg : seg0 + seg, r0, r1);
    double a_ref[4] = {0.0, 0.0, 0.0, 0.0}, a_het[4] = {0.0, 0.0, 0.0, 0.0}, a_alt[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t miss8 = 0, miss[4] = {0, 0, 0, 0};
    int since_flush = 0;
#if SNPM_STRICT_EXEC
    const uint64_t exec_all = __builtin_amdgcn_read_exec();       // the lanes of this wave that own columns
    uint32_t k0 = 0u, k1 = 1u, k2 = 2u;                            // class codes in VGPRs (SDWA takes no literals)
    asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2));
#endif
    auto one_row = [&](uint32_t x, double w0, double w1, double w2) {
        // (skipping the classes whose weight is zero -- two of three on a hard-call row -- was tried in round 2: the
        // scalar branches cost more than the additions they save, 30.8 instead of 17.1 ms on 10k x 6.25M)
#if SNPM_STRICT_EXEC
        uint32_t lut_miss = 0;
        if constexpr (PACKED && MASKS) {
            // packed row byte -> table entry (one ds_read_b128), then per call two sign tests that give the scalar masks
            // G and U; ref = U & ~G, het = G & ~U, alt = U & G are written to EXEC by scalar instructions and one
            // v_add_f64 runs under each: 8 + 12 + 2 VALU per row of 4 calls (the select form below takes about 50)
            const uint4 e = lut[x];
            lut_miss = e.z + e.w;
#define STRICT_COLUMN_LUT(J, P, SELG, SELU)                                                                 \
    do {                                                                                                    \
        uint64_t mg, mu;                                                                                    \
        if (!SKIP)                                                                                          \
            asm volatile("v_cmp_lt_i32_sdwa %[g], sext(%[x]), %[k0] src0_sel:" SELG " src1_sel:DWORD\n\t"    \
                         "v_cmp_lt_i32_sdwa %[u], sext(%[x]), %[k0] src0_sel:" SELU " src1_sel:DWORD\n\t"    \
                         "s_andn2_b64 exec, %[u], %[g]\n\t"                                                 \
                         "v_add_f64 %[ar], %[ar], %[w0]\n\t"                                                \
                         "s_andn2_b64 exec, %[g], %[u]\n\t"                                                 \
                         "v_add_f64 %[ah], %[ah], %[w1]\n\t"                                                \
                         "s_and_b64 exec, %[u], %[g]\n\t"                                                   \
                         "v_add_f64 %[aa], %[aa], %[w2]\n\t"                                                \
                         "s_mov_b64 exec, %[sv]"                                                             \
                         : [ar] "+v"(a_ref[J]), [ah] "+v"(a_het[J]), [aa] "+v"(a_alt[J]), [g] "=&s"(mg), [u] "=&s"(mu) \
                         : [x] "v"(P), [k0] "v"(k0), [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [sv] "s"(exec_all) \
                         : "scc");                                                                           \
        else                                                                                                \
            asm volatile("v_cmp_lt_i32_sdwa %[g], sext(%[x]), %[k0] src0_sel:" SELG " src1_sel:DWORD\n\t"    \
                         "v_cmp_lt_i32_sdwa %[u], sext(%[x]), %[k0] src0_sel:" SELU " src1_sel:DWORD\n\t"    \
                         "s_andn2_b64 exec, %[u], %[g]\n\t"                                                 \
                         "v_add_f64 %[ar], %[ar], %[w0]\n\t"                                                \
                         "s_and_b64 exec, %[u], %[g]\n\t"                                                   \
                         "v_add_f64 %[aa], %[aa], %[w2]\n\t"                                                \
                         "s_mov_b64 exec, %[sv]"                                                             \
                         : [ar] "+v"(a_ref[J]), [aa] "+v"(a_alt[J]), [g] "=&s"(mg), [u] "=&s"(mu)             \
                         : [x] "v"(P), [k0] "v"(k0), [w0] "s"(w0), [w2] "s"(w2), [sv] "s"(exec_all)           \
                         : "scc");                                                                           \
    } while (0)
            STRICT_COLUMN_LUT(0, e.x, "BYTE_0", "BYTE_1");
            STRICT_COLUMN_LUT(1, e.x, "BYTE_2", "BYTE_3");
            STRICT_COLUMN_LUT(2, e.y, "BYTE_0", "BYTE_1");
            STRICT_COLUMN_LUT(3, e.y, "BYTE_2", "BYTE_3");
#undef STRICT_COLUMN_LUT
        } else if constexpr (!PACKED) {
            // EXEC-masked additions: v_cmpx selects the lanes whose call is this class (byte select inside the compare),
            // one v_add_f64 with the row's weight from scalar registers runs on exactly those lanes, EXEC is restored:
            // 1 + 2 VALU issue slots per class instead of 1 + 1 + 2 (no select), one scalar move more.
#define STRICT_CLASS(ACC, CODE_REG, WREG, SEL)                                                              \
    asm volatile("v_cmpx_eq_u32_sdwa vcc, %[x], %[c] src0_sel:" SEL " src1_sel:DWORD\n\t"                   \
                 "v_add_f64 %[a], %[a], %[w]\n\t"                                                          \
                 "s_mov_b64 exec, %[sv]"                                                                     \
                 : [a] "+v"(ACC)                                                                             \
                 : [x] "v"(x), [c] "v"(CODE_REG), [w] "s"(WREG), [sv] "s"(exec_all)                          \
                 : "vcc")
            // MASKS: two compares per call instead of three -- G = {1, 2} (signed byte > 0), U = {0, 1} (unsigned byte < 2);
            // ref = U & ~G, het = G & ~U, alt = U & G are scalar operations that write EXEC directly.  A call code 3
            // ("other": informative, matches no class) would land in the het class, so this form runs only on panels
            // that hold none (snpm_panel::d_other, raised by the upload kernel).
#define STRICT_COLUMN_MASKS(J, SEL)                                                                         \
    do {                                                                                                    \
        uint64_t mg, mu;                                                                                    \
        if (!SKIP)                                                                                          \
            asm volatile("v_cmp_gt_i32_sdwa %[g], sext(%[x]), %[k0] src0_sel:" SEL " src1_sel:DWORD\n\t"     \
                         "v_cmp_lt_u32_sdwa %[u], %[x], %[k2] src0_sel:" SEL " src1_sel:DWORD\n\t"           \
                         "s_andn2_b64 exec, %[u], %[g]\n\t"                                                 \
                         "v_add_f64 %[ar], %[ar], %[w0]\n\t"                                                \
                         "s_andn2_b64 exec, %[g], %[u]\n\t"                                                 \
                         "v_add_f64 %[ah], %[ah], %[w1]\n\t"                                                \
                         "s_and_b64 exec, %[u], %[g]\n\t"                                                   \
                         "v_add_f64 %[aa], %[aa], %[w2]\n\t"                                                \
                         "s_mov_b64 exec, %[sv]"                                                             \
                         : [ar] "+v"(a_ref[J]), [ah] "+v"(a_het[J]), [aa] "+v"(a_alt[J]), [g] "=&s"(mg), [u] "=&s"(mu) \
                         : [x] "v"(x), [k0] "v"(k0), [k2] "v"(k2), [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2),   \
                           [sv] "s"(exec_all)                                                                \
                         : "scc");                                                                           \
        else                                                                                                \
            asm volatile("v_cmp_gt_i32_sdwa %[g], sext(%[x]), %[k0] src0_sel:" SEL " src1_sel:DWORD\n\t"     \
                         "v_cmp_lt_u32_sdwa %[u], %[x], %[k2] src0_sel:" SEL " src1_sel:DWORD\n\t"           \
                         "s_andn2_b64 exec, %[u], %[g]\n\t"                                                 \
                         "v_add_f64 %[ar], %[ar], %[w0]\n\t"                                                \
                         "s_and_b64 exec, %[u], %[g]\n\t"                                                   \
                         "v_add_f64 %[aa], %[aa], %[w2]\n\t"                                                \
                         "s_mov_b64 exec, %[sv]"                                                             \
                         : [ar] "+v"(a_ref[J]), [aa] "+v"(a_alt[J]), [g] "=&s"(mg), [u] "=&s"(mu)             \
                         : [x] "v"(x), [k0] "v"(k0), [k2] "v"(k2), [w0] "s"(w0), [w2] "s"(w2), [sv] "s"(exec_all) \
                         : "scc");                                                                           \
    } while (0)
#define STRICT_COLUMN(J, SEL)                                                                               \
    do {                                                                                                    \
        if constexpr (MASKS) {                                                                              \
            STRICT_COLUMN_MASKS(J, SEL);                                                                    \
        } else {                                                                                            \
            STRICT_CLASS(a_ref[J], k0, w0, SEL);                                                            \
            if (!SKIP) STRICT_CLASS(a_het[J], k2, w1, SEL);                                                 \
            STRICT_CLASS(a_alt[J], k1, w2, SEL);                                                            \
        }                                                                                                   \
    } while (0)
            STRICT_COLUMN(0, "BYTE_0");
            STRICT_COLUMN(1, "BYTE_1");
            STRICT_COLUMN(2, "BYTE_2");
            STRICT_COLUMN(3, "BYTE_3");
#undef STRICT_COLUMN
#undef STRICT_COLUMN_MASKS
#undef STRICT_CLASS
        } else
#endif
        {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t b = PACKED ? ((x >> (2 * j)) & 3u) : ((x >> (8 * j)) & 0xffu);
            a_ref[j] = add_if(a_ref[j], b == 0u, w0);
            if (!SKIP) a_het[j] = add_if(a_het[j], b == 2u, w1);
            a_alt[j] = add_if(a_alt[j], b == 1u, w2);
        }
        }
#if SNPM_STRICT_EXEC
        if (PACKED && MASKS) {
            miss8 += lut_miss;
        } else
#endif
        if (PACKED) {     // code 3 (or 2 / 3 with skip_hets): one bit per call, spread to one byte per call
            const uint32_t m = SKIP ? ((x >> 1) & 0x55u) : (x & (x >> 1) & 0x55u);
            miss8 += (m * 0x41041u) & 0x01010101u;
        } else {
            miss8 += SKIP ? (((x >> 7) | ((x >> 1) & ~x)) & 0x01010101u) : ((x >> 7) & 0x01010101u);
        }
    };
    // the byte counters of miss8 take 255 rows: callers announce the rows they are about to score
    auto flush_before = [&](int rows) {
        if (since_flush + rows > 255) {
#pragma unroll
            for (int j = 0; j < 4; ++j) miss[j] += (miss8 >> (8 * j)) & 0xffu;
            miss8 = 0;
            since_flush = 0;
        }
        since_flush += rows;
    };
    auto load = [&](const int8_t *ptr) -> uint32_t {
        return PACKED ? (uint32_t) * reinterpret_cast<const uint8_t *>(ptr) : *reinterpret_cast<const uint32_t *>(ptr);
    };
    // row address = wave-uniform row base + the lane's 32-bit offset (global_load with a scalar base: no address VALU)
    const uint32_t coff = (uint32_t)(PACKED ? c0 / 4 : c0);
    int64_t r = r0;
    if constexpr ((!PACKED || MASKS) && SNPM_STRICT_EXEC) {
        // Batches of SB rows, two register sets: the next batch is requested before the current one is scored (up to
        // 2 * SB row loads in flight per wave), and the batch's weights arrive in a few wide scalar loads.
        constexpr int SB = SNPM_STRICT_BATCH;
        auto load_batch = [&](uint32_t (&x)[SB], int64_t rb) {
            if constexpr (!GATHER) {
                // consecutive rows: one buffer resource per batch (scalar registers), the row inside the batch in the scalar
                // offset, the lane's column in the vector offset -- no address arithmetic on the vector unit, which this
                // kernel saturates (and reads past the batch would return 0 instead of faulting).  32-bit byte counts: panels
                // hold at most 2^27 accessions (snpm_panel_create)
                const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<int8_t *>(db + (row0 + rb) * pitch), 0, (int)(SB * pitch), 0x00020000);
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    x[u] = PACKED ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rows, coff, (int)(u * pitch), 0)
                                  : (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, coff, (int)(u * pitch), 0);
            } else {
#pragma unroll
                for (int u = 0; u < SB; ++u) x[u] = load(db + row_idx[rb + u] * pitch + coff);
            }
        };
        auto score_batch = [&](const uint32_t (&x)[SB], int64_t rb) {
            // the batch's weights first (wave-uniform: a few wide scalar loads, one wait), then the rows: with a scalar load
            // and a wait per row the waves spent 62 % of their cycles parked in s_waitcnt (SQ_WAIT_ANY)
            double wb[SB * 3];
#pragma unroll
            for (int i = 0; i < SB * 3; ++i) wb[i] = w[3 * rb + i];
            flush_before(SB);
#pragma unroll
            for (int u = 0; u < SB; ++u) one_row(x[u], wb[3 * u], wb[3 * u + 1], wb[3 * u + 2]);
        };
        const int64_t nb = (r1 - r0) / SB;
        uint32_t xa[SB], xb[SB];
        if (nb > 0) load_batch(xa, r0);
        for (int64_t b = 0; b < nb; b += 2) {
            if (b + 1 < nb) load_batch(xb, r0 + (b + 1) * SB);
            score_batch(xa, r0 + b * SB);
            if (b + 1 < nb) {
                if (b + 2 < nb) load_batch(xa, r0 + (b + 2) * SB);
                score_batch(xb, r0 + (b + 1) * SB);
            }
        }
        r = r0 + nb * SB;
    } else {
        // select form: round 1's loop -- four rows requested, then scored (the batched form above is slower with it:
        // 24.5 vs 18.5 ms on a packed 10 000 x 6.25M panel)
        for (; r + 4 <= r1; r += 4) {
            uint32_t x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t prow = GATHER ? row_idx[r + u] : (row0 + r + u);
                x[u] = load(db + prow * pitch + coff);
            }
            flush_before(4);
#pragma unroll
            for (int u = 0; u < 4; ++u) one_row(x[u], w[3 * (r + u)], w[3 * (r + u) + 1], w[3 * (r + u) + 2]);
        }
    }
    for (; r < r1; ++r) {
        const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
        flush_before(1);
        one_row(load(db + prow * pitch + coff), w[3 * r], w[3 * r + 1], w[3 * r + 2]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (c0 + j < ncols) {
            out_score[seg * ld + c0 + j] = ((0.0 + a_ref[j]) + a_het[j]) + a_alt[j];
            out_miss[seg * ld + c0 + j] = miss[j] + ((miss8 >> (8 * j)) & 0xffu);
        }
    }
    }
}

//   other_codes: the int8 panel's flag "a call code > 2 was stored" (nullptr for packed panels)
template <bool SKIP, bool GATHER, bool PACKED>
__global__ void __launch_bounds__(256)
k_strict4(const int8_t *__restrict__ db, int64_t pitch, const int64_t *__restrict__ row_idx, int64_t row0,
          const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n, int64_t seg0,
          int64_t n_seg, int64_t ncols, double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld,
          const int *__restrict__ other_codes, const int *__restrict__ gate, int gate_cap)
{
    if (dense_tier_off(gate, gate_cap)) return;
    const int64_t c0 = ((int64_t)blockIdx.y * blockDim.x + threadIdx.x) * 4;
    if constexpr (PACKED && SNPM_STRICT_EXEC) {
        __shared__ uint4 s_lut[256];
        for (uint32_t e = threadIdx.x; e < 256u; e += blockDim.x) s_lut[e] = strict_lut_entry<SKIP>(e);
        __syncthreads();
        if (c0 >= ncols) return;
        strict4_segments<SKIP, GATHER, true, true>(db, pitch, row_idx, row0, w, seg_off, chunk, n, seg0, n_seg, ncols,
                                                   out_score, out_miss, ld, c0, s_lut);
        return;
    }
    if (c0 >= ncols) return;                 // pitch is a multiple of 256: the dword / byte of an active lane is inside the row
    if constexpr (!PACKED && SNPM_STRICT_EXEC) {
        if (*other_codes == 0) {             // wave-uniform
            strict4_segments<SKIP, GATHER, false, true>(db, pitch, row_idx, row0, w, seg_off, chunk, n, seg0, n_seg, ncols,
                                                        out_score, out_miss, ld, c0, nullptr);
            return;
        }
    }
    strict4_segments<SKIP, GATHER, PACKED, false>(db, pitch, row_idx, row0, w, seg_off, chunk, n, seg0, n_seg, ncols,
                                                  out_score, out_miss, ld, c0, nullptr);
}

// Strict segment sums for a SHORT list of columns (the accessions SNPM_MODE_EXACT has to re-evaluate):
// one lane per (segment, column) pair so that every lane of a wave is busy and 8 independent byte
// loads per lane are in flight (each is its own cache line: this path is latency-bound).
// Same arithmetic and order as k_strict.  out_* [n_seg, ld].
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(256)
k_strict_sparse(const int8_t *__restrict__ db, int64_t pitch, int packed, const int64_t *__restrict__ row_idx,
                int64_t row0, const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n,
                int64_t n_seg, const int32_t *__restrict__ cols, const int *__restrict__ d_ncols, int cap,
                double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld)
{
    const int64_t ncols = *d_ncols;           // flagged accessions (device-side count): sparse tier only
    if (ncols > cap) return;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_seg * ncols; id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t seg = id / ncols;
    const int64_t i = id - seg * ncols;
    const int64_t col = cols[i];
    int64_t r0, r1;
    seg_bounds(seg_off, chunk, n, seg, r0, r1);
    double a_ref = 0.0, a_het = 0.0, a_alt = 0.0;
    uint32_t miss = 0;
    constexpr int U = 32;                 // byte loads in flight per lane (each its own cache line)
    int64_t r = r0;
    for (; r + U <= r1; r += U) {
        int b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t prow = GATHER ? row_idx[r + u] : (row0 + r + u);
            b[u] = code_at(db, pitch, prow, col, packed);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double w0 = w[3 * (r + u) + 0], w1 = w[3 * (r + u) + 1], w2 = w[3 * (r + u) + 2];
            a_ref = add_sel(a_ref, b[u] == 0, w0);
            if (!SKIP) a_het = add_sel(a_het, b[u] == 2, w1);
            a_alt = add_sel(a_alt, b[u] == 1, w2);
            miss += SKIP ? (b[u] < 0 || b[u] == 2) : (b[u] < 0);
        }
    }
    for (; r < r1; ++r) {
        const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
        const int b = code_at(db, pitch, prow, col, packed);
        const double w0 = w[3 * r + 0], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        a_ref = add_sel(a_ref, b == 0, w0);
        if (!SKIP) a_het = add_sel(a_het, b == 2, w1);
        a_alt = add_sel(a_alt, b == 1, w2);
        miss += SKIP ? (b < 0 || b == 2) : (b < 0);
    }
    out_score[seg * ld + i] = ((0.0 + a_ref) + a_het) + a_alt;
    out_miss[seg * ld + i] = miss;
    }
}

// ------------------------------------------------------------------------------------------------
// Accession-major packed copy of the panel (the reference keeps a column-chunked second HDF5 file for
// the same purpose, core/makedb.py:64-81): dT [n_acc, pitchT] bytes, 4 SNPs per byte, 2 bits per call
// (0 ref, 1 alt, 2 het, 3 missing).  A column of the panel becomes one contiguous run (n_snp/4 bytes), so
// re-evaluating an accession in reference order no longer fetches a cache line per SNP.  Code 3 of the
// SNP-major panel ("informative, matches nothing") has no 2-bit encoding: *bad is set and the copy is not
// used.  Tile: 256 SNPs x 64 accessions through LDS.
constexpr int PT_ROWS = 256;
constexpr int PT_COLS = 64;
__global__ void __launch_bounds__(256)
k_pack_transpose(const int8_t *__restrict__ db, int64_t pitch, int64_t n_snp, int64_t n_acc,
                 uint8_t *__restrict__ dT, int64_t pitchT, int *__restrict__ bad)
{
    __shared__ uint32_t tile[PT_ROWS][PT_COLS / 4 + 1];       // +1 dword per row: conflict-free column reads
    const int64_t snp0 = (int64_t)blockIdx.x * PT_ROWS;
    const int64_t acc0 = (int64_t)blockIdx.y * PT_COLS;
    const int t = threadIdx.x;
    // load: 16 lanes x 4 B cover the 64 accession bytes of one SNP row; 16 rows per pass
    for (int pass = 0; pass < PT_ROWS / 16; ++pass) {
        const int r = pass * 16 + (t >> 4);
        const int64_t row = snp0 + r;
        uint32_t v = 0xffffffffu;                               // rows past the end: missing
        if (row < n_snp) v = *reinterpret_cast<const uint32_t *>(db + row * pitch + acc0 + (t & 15) * 4);
        tile[r][t & 15] = v;
    }
    __syncthreads();
    // pack: thread = (accession c, quarter q of the 256 SNPs): 64 calls -> 16 bytes
    const int c = t & 63, q = t >> 6;
    if (acc0 + c < n_acc) {
        uint32_t out[4] = {0, 0, 0, 0};
        int saw3 = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const uint32_t w = tile[q * 64 + k][c >> 2];
            const uint32_t b = (w >> (8 * (c & 3))) & 0xffu;
            saw3 |= (b == 3u);
            const uint32_t code = (b & 0x80u) ? 3u : (b & 3u);
            out[k >> 4] |= code << (2 * (k & 15));
        }
        if (saw3) atomicOr(bad, 1);
        uint4 o;
        o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
        *reinterpret_cast<uint4 *>(dT + (acc0 + c) * pitchT + snp0 / 4 + q * 16) = o;
    }
}

// The same copy from a PACKED panel (2 bits per call on both sides; the copy is as large as the panel, so it is
// only built when it fits): tile of 256 SNPs x 256 accessions = 64 B per SNP row through LDS; thread = one
// accession, 256 SNPs -> 64 contiguous bytes of its row in the copy.
constexpr int PTP_ROWS = 256;
constexpr int PTP_COLS = 256;
__global__ void __launch_bounds__(256)
k_pack_transpose_packed(const uint8_t *__restrict__ db, int64_t pitch, int64_t n_snp, int64_t n_acc,
                        uint8_t *__restrict__ dT, int64_t pitchT)
{
    __shared__ uint32_t tile[PTP_ROWS][PTP_COLS / 16 + 1];     // 16 dwords of 16 calls per row (+1: conflict-free columns)
    const int64_t snp0 = (int64_t)blockIdx.x * PTP_ROWS;
    const int64_t acc0 = (int64_t)blockIdx.y * PTP_COLS;
    const int t = threadIdx.x;
    for (int pass = 0; pass < PTP_ROWS / 16; ++pass) {
        const int r = pass * 16 + (t >> 4);
        const int64_t row = snp0 + r;
        const int64_t byte = acc0 / 4 + (t & 15) * 4;
        uint32_t v = 0xffffffffu;                                // rows / bytes past the end: missing
        if (row < n_snp && byte < pitch) v = *reinterpret_cast<const uint32_t *>(db + row * pitch + byte);
        tile[r][t & 15] = v;
    }
    __syncthreads();
    const int c = t;
    if (acc0 + c < n_acc) {
        uint32_t out[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) out[k] = 0;
#pragma unroll 16
        for (int k = 0; k < PTP_ROWS; ++k) {
            const uint32_t code = (tile[k][c >> 4] >> (2 * (c & 15))) & 3u;
            out[k >> 4] |= code << (2 * (k & 15));
        }
        uint4 *dst = reinterpret_cast<uint4 *>(dT + (acc0 + c) * pitchT + snp0 / 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint4 o;
            o.x = out[4 * k]; o.y = out[4 * k + 1]; o.z = out[4 * k + 2]; o.w = out[4 * k + 3];
            dst[k] = o;
        }
    }
}

// k_strict_sparse on the accession-major packed copy: same arithmetic and order.
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(256)
k_strict_sparse_T(const uint8_t *__restrict__ dT, int64_t pitchT, const int64_t *__restrict__ row_idx, int64_t row0,
                  const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n,
                  int64_t n_seg, const int32_t *__restrict__ cols, const int *__restrict__ d_ncols, int cap,
                  double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld)
{
    const int64_t ncols = *d_ncols;
    if (ncols > cap) return;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_seg * ncols; id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t seg = id / ncols;
    const int64_t i = id - seg * ncols;
    const uint8_t *colp = dT + (int64_t)cols[i] * pitchT;
    int64_t r0, r1;
    seg_bounds(seg_off, chunk, n, seg, r0, r1);
    double a_ref = 0.0, a_het = 0.0, a_alt = 0.0;
    uint32_t miss = 0;
#ifndef SNPM_SPARSE_T_U
#define SNPM_SPARSE_T_U 8
#endif
    constexpr int U = SNPM_SPARSE_T_U;      // rows whose code bytes are in flight per lane
    int64_t r = r0;
    for (; r + U <= r1; r += U) {
        int b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t prow = GATHER ? row_idx[r + u] : (row0 + r + u);
            b[u] = (colp[prow >> 2] >> (2 * (int)(prow & 3))) & 3;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double w0 = w[3 * (r + u) + 0], w1 = w[3 * (r + u) + 1], w2 = w[3 * (r + u) + 2];
            a_ref = add_sel(a_ref, b[u] == 0, w0);
            if (!SKIP) a_het = add_sel(a_het, b[u] == 2, w1);
            a_alt = add_sel(a_alt, b[u] == 1, w2);
            miss += SKIP ? (b[u] >= 2) : (b[u] == 3);
        }
    }
    for (; r < r1; ++r) {
        const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
        const int b = (colp[prow >> 2] >> (2 * (int)(prow & 3))) & 3;
        const double w0 = w[3 * r + 0], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        a_ref = add_sel(a_ref, b == 0, w0);
        if (!SKIP) a_het = add_sel(a_het, b == 2, w1);
        a_alt = add_sel(a_alt, b == 1, w2);
        miss += SKIP ? (b >= 2) : (b == 3);
    }
    out_score[seg * ld + i] = ((0.0 + a_ref) + a_het) + a_alt;
    out_miss[seg * ld + i] = miss;
    }
}

// total[i] = (((0 + s0) + s1) + ...) over the segments in order; ninfo[i] = n_rows - sum(miss).
// The adds are sequential by contract (ScoreList += chunk, core/snpmatch.py:224); the loads are not:
// 8 are issued ahead of the adds that consume them.
// carry_score / carry_ninfo (may be NULL): totals of the SNP slabs scored before this one -- the chain of
// additions continues from them, as the reference's loop does over the whole SNP axis.
__global__ void k_scan(const double *__restrict__ seg_score, const uint32_t *__restrict__ seg_miss,
                       int64_t n_rows, int64_t n_seg, int64_t ld, int64_t ncols,
                       double *tot_score, int64_t *tot_ninfo, const double *carry_score, const int64_t *carry_ninfo,
                       const int *__restrict__ gate, int gate_cap)     // carry_* may alias tot_* (in-place continuation)
{
    if (dense_tier_off(gate, gate_cap)) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    double s = carry_score ? carry_score[i] : 0.0;
    int64_t m = 0;
    int64_t k = 0;
    for (; k + 8 <= n_seg; k += 8) {
        double v[8];
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = seg_score[(k + u) * ld + i];
            c[u] = seg_miss[(k + u) * ld + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s + v[u];
            m += c[u];
        }
    }
    for (; k < n_seg; ++k) {
        s = s + seg_score[k * ld + i];
        m += seg_miss[k * ld + i];
    }
    tot_score[i] = s;
    if (tot_ninfo) tot_ninfo[i] = (carry_ninfo ? carry_ninfo[i] : 0) + n_rows - m;
}

// Same result for a SHORT column list (ncols <= 64, one block of 4 waves).  The chain of additions is
// sequential by contract, and a single wave issues in order, so what bounds it is the number of instructions
// per addition.  Wave 0 only adds: lane c owns column c, whose segment sums lie contiguously in an LDS tile
// (column-major, stride = tile length + 2 doubles: conflict-free 16-B reads), fetched with ds_read_b128 at
// immediate offsets (0.5 LDS instruction and no address arithmetic per addition) two 16-value sets ahead.
// Waves 1-3 meanwhile load the next tile from global memory into the other LDS buffer.
// The LDS reads are issued from inline asm, so their completion is waited for by hand: LDS operations of a
// wave complete in order, lgkmcnt(8) therefore means "everything but the 8 reads just issued has arrived";
// the registers are operands of the wait so that the additions cannot be scheduled before it.
constexpr int SCAN_TILE_ELEMS = 4096;
typedef double f64x2 __attribute__((ext_vector_type(2)));

#define SCAN_READ8(S, ADDR, OFF)                                                                             \
    asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%9+16\n\t"                   \
                 "ds_read_b128 %2, %8 offset:%9+32\n\tds_read_b128 %3, %8 offset:%9+48\n\t"                 \
                 "ds_read_b128 %4, %8 offset:%9+64\n\tds_read_b128 %5, %8 offset:%9+80\n\t"                 \
                 "ds_read_b128 %6, %8 offset:%9+96\n\tds_read_b128 %7, %8 offset:%9+112"                    \
                 : "=v"(S##0), "=v"(S##1), "=v"(S##2), "=v"(S##3), "=v"(S##4), "=v"(S##5), "=v"(S##6), "=v"(S##7) \
                 : "v"(ADDR), "n"(OFF))
#define SCAN_WAIT8(S, N)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                               \
                 : "+v"(S##0), "+v"(S##1), "+v"(S##2), "+v"(S##3), "+v"(S##4), "+v"(S##5), "+v"(S##6), "+v"(S##7))
#define SCAN_ADD16(S)                                                                                        \
    do {                                                                                                     \
        s = s + (S##0).x; s = s + (S##0).y; s = s + (S##1).x; s = s + (S##1).y; s = s + (S##2).x; s = s + (S##2).y;      \
        s = s + (S##3).x; s = s + (S##3).y; s = s + (S##4).x; s = s + (S##4).y; s = s + (S##5).x; s = s + (S##5).y;      \
        s = s + (S##6).x; s = s + (S##6).y; s = s + (S##7).x; s = s + (S##7).y;                                      \
    } while (0)

__global__ void __launch_bounds__(256)
k_scan_few(const double *__restrict__ seg_score, int64_t n_seg, int64_t ld, const int *__restrict__ d_ncols, int cap,
           double *__restrict__ tot_score, const double *__restrict__ carry)
{
    __shared__ __attribute__((aligned(16))) double tile[2][SCAN_TILE_ELEMS + 2 * WAVE];
    const int ncols = *d_ncols;
    if (ncols < 1 || ncols > cap) return;          // block-uniform
    const int ts = (SCAN_TILE_ELEMS / ncols) & ~31;  // segments per tile: a multiple of 32, >= 64
    const int cs = ts + 2;                           // column stride: 16 B more than a multiple of 256 B
    const int wave = threadIdx.x / WAVE;
    auto load_tile = [&](int buf, int64_t base, int first, int nthr) {
        const int nseg = (int)((n_seg - base < ts) ? (n_seg - base) : ts);
        const int n_elem = nseg * ncols;
        for (int e0 = first; e0 < n_elem; e0 += 8 * nthr) {           // 8 independent loads in flight per thread
            double v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nthr;
                if (e < n_elem) {
                    const int sg = e / ncols, c = e - sg * ncols;
                    v8[u] = seg_score[(base + sg) * ld + c];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nthr;
                if (e < n_elem) {
                    const int sg = e / ncols, c = e - sg * ncols;
                    tile[buf][c * cs + sg] = v8[u];
                }
            }
        }
    };
    if (n_seg > 0) load_tile(0, 0, threadIdx.x, 256);
    __syncthreads();
    double s = (carry && (int)threadIdx.x < ncols) ? carry[threadIdx.x] : 0.0;     // totals of earlier slabs
    int buf = 0;
    for (int64_t base = 0; base < n_seg; base += ts, buf ^= 1) {
        if (wave > 0) {
            if (base + ts < n_seg) load_tile(buf ^ 1, base + ts, threadIdx.x - WAVE, 256 - WAVE);
        } else if ((int)threadIdx.x < ncols) {
            const int nseg = (int)((n_seg - base < ts) ? (n_seg - base) : ts);
            const double *col = &tile[buf][threadIdx.x * cs];
            uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double *)col;
            int sg = 0;
            f64x2 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
            if (nseg >= 16) SCAN_READ8(a, addr, 0);                  // a = [0, 16)
            while (sg + 48 <= nseg) {
                SCAN_READ8(b, addr, 128);                             // b = [sg + 16, sg + 32)
                SCAN_WAIT8(a, 8);
                SCAN_ADD16(a);
                SCAN_READ8(a, addr, 256);                             // a = [sg + 32, sg + 48)
                SCAN_WAIT8(b, 8);
                SCAN_ADD16(b);
                addr += 256;
                sg += 32;
            }
            if (sg + 16 <= nseg) {                                    // a holds [sg, sg + 16)
                SCAN_WAIT8(a, 0);
                SCAN_ADD16(a);
                sg += 16;
            }
            for (; sg < nseg; ++sg) s = s + col[sg];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < ncols) tot_score[threadIdx.x] = s;
}
#undef SCAN_READ8
#undef SCAN_WAIT8
#undef SCAN_ADD16

// per-segment ninfo [n_seg, n_acc] i64 and score copy-out into a dense [n_seg, n_acc] host-shaped layout
__global__ void k_seg_pack(const double *__restrict__ seg_score, const uint32_t *__restrict__ seg_miss,
                           const int64_t *__restrict__ seg_off, int64_t n_seg, int64_t ld, int64_t n_acc,
                           double *__restrict__ score, int64_t *__restrict__ ninfo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t k = blockIdx.y;
    if (i >= n_acc || k >= n_seg) return;
    int64_t len = seg_off[k + 1] - seg_off[k];
    score[k * n_acc + i] = seg_score[k * ld + i];
    ninfo[k * n_acc + i] = len - (int64_t)seg_miss[k * ld + i];
}

// score[cols[i]] = strict_total[i]
__global__ void k_patch(const double *__restrict__ strict_total, const int32_t *__restrict__ cols,
                        const int *__restrict__ d_ncols, int cap, double *__restrict__ score)
{
    const int ncols = *d_ncols;
    if (ncols > cap) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncols) score[cols[i]] = strict_total[i];
}

// ------------------------------------------------------------------------------------------------
// likeliTest (core/snpmatch.py:40-55) row-wise, nanmin per row, ratio (core/snpmatch.py:106-117).
// grid.x = row; one block per row.  flags[0] |= 1 when some y > n (the reference asserts).
__device__ __forceinline__ double likeli_one(double y, double n, int *bad)
{
    const double p = 0.99999999;
    if (y > n) { *bad = 1; return __builtin_nan(""); }
    if (n == 0.0) return __builtin_nan("");
    if (y == n) return 1.0;
    if (y > 0.0) {
        const double ps = y / n;
        const double a = y * log(ps / p);
        const double b = (n - y) * log((1.0 - ps) / (1.0 - p));
        return a + b;
    }
    return __builtin_nan("");
}

__global__ void __launch_bounds__(1024)
k_likelihood(const double *__restrict__ y, const int64_t *__restrict__ n, int64_t len, int truncate,
             double amin_or_nan, double *__restrict__ lik, double *__restrict__ lrt, int *__restrict__ flags)
{
    __shared__ double s_min[16];
    __shared__ double s_top;
    const int64_t row = blockIdx.x;
    const double *yr = y + row * len;
    const int64_t *nr = n + row * len;
    double *lr = lik + row * len;
    double *rr = lrt + row * len;
    double mn = __builtin_inf();
    int bad = 0;
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) {
        double yy = yr[i];
        if (truncate) yy = trunc(yy);
        const double l = likeli_one(yy, (double)nr[i], &bad);
        lr[i] = l;
        if (l == l && l < mn) mn = l;
    }
    if (bad) atomicOr(flags, 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_xor(mn, o);
        mn = other < mn ? other : mn;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_min[wave] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = __builtin_inf();
        const int nw = (blockDim.x + 63) >> 6;
        for (int k = 0; k < nw; ++k) m = s_min[k] < m ? s_min[k] : m;
        if (m == __builtin_inf()) m = __builtin_nan("");      // all-NaN row: np.nanmin -> nan
        if (amin_or_nan == amin_or_nan) m = amin_or_nan;
        s_top = m;
    }
    __syncthreads();
    const double top = s_top;
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) {
        // get_fraction(x, y): nan when y <= 0 (core/snpmatch.py:25-28); y = nan falls through to x / nan
        rr[i] = (top <= 0.0) ? __builtin_nan("") : lr[i] / top;
    }
}

// ------------------------------------------------------------------------------------------------
// packed panel upload: int8 rows (staging slab on the device, row stride src_pitch) -> 2 bits per call.
// One thread per output byte.  Codes outside {-1 (any negative), 0, 1, 2} cannot be encoded: *bad |= 1.
__global__ void k_pack_rows(const int8_t *__restrict__ src, int64_t src_pitch, int64_t nrows, int64_t n_acc,
                            uint8_t *__restrict__ dst, int64_t dst_pitch, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * dst_pitch) return;
    const int64_t r = i / dst_pitch, b = i - r * dst_pitch;
    uint32_t out = 0;
    int saw = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int64_t a = b * 4 + f;
        int v = -1;
        if (a < n_acc) v = src[r * src_pitch + a];
        saw |= (v > 2);
        out |= (uint32_t)(v < 0 ? 3 : (v & 3)) << (2 * f);
    }
    if (saw) atomicOr(bad, 1);
    dst[i] = (uint8_t)out;
}

// int8 panel upload: tightly staged rows (row stride src_pitch) -> panel rows (256-B pitch), codes
// canonicalised on the way (negative -> 0xFF, > 2 -> 3, which raises *other_codes), pad bytes = 0xFF.  One thread per
// destination dword.
__global__ void k_repitch_canon(const int8_t *__restrict__ src, int64_t src_pitch, int64_t nrows, int64_t n_acc,
                                uint32_t *__restrict__ dst, int64_t dst_pitch, int *__restrict__ other_codes)
{
    const int64_t dwords_per_row = dst_pitch / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * dwords_per_row) return;
    const int64_t r = i / dwords_per_row, d = i - r * dwords_per_row;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t a = d * 4 + j;
        int v = -1;
        if (a < n_acc) v = src[r * src_pitch + a];
        const uint32_t c = v < 0 ? 0xffu : (v > 2 ? 3u : (uint32_t)v);
        out |= c << (8 * j);
    }
    if ((out & (out >> 1) & ~(out >> 7)) & 0x01010101u) atomicOr(other_codes, 1);     // a byte == 3 (k_strict4 needs to know)
    dst[r * dwords_per_row + d] = out;
}

// packed rows -> int8 (download / checks): one thread per accession byte of the destination
__global__ void k_unpack_rows(const uint8_t *__restrict__ src, int64_t src_pitch, int64_t nrows, int64_t n_acc,
                              int8_t *__restrict__ dst, int64_t dst_pitch)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * n_acc) return;
    const int64_t r = i / n_acc, a = i - r * n_acc;
    const int v = (src[r * src_pitch + (a >> 2)] >> (2 * (int)(a & 3))) & 3;
    dst[r * dst_pitch + a] = (int8_t)(v == 3 ? -1 : v);
}

// ------------------------------------------------------------------------------------------------
// synthetic panel fill: counter-based, element (snp, acc) depends only on (seed, snp, acc).
// One splitmix64 hash per 4 adjacent accessions (16 random bits each).
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ uint32_t synth_quad(uint64_t seed, uint64_t snp, uint64_t acc_quad)
{
    // thresholds on 16-bit uniforms: P(-1)=3277/65536, P(0)=39321/65536, P(1)=21627/65536, P(2)=1311/65536
    const uint64_t h = splitmix64(splitmix64(seed ^ (snp * 0xD6E8FEB86659FD93ull)) + acc_quad);
    uint32_t out = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t u = (uint32_t)(h >> (16 * j)) & 0xffffu;
        const uint32_t c = u < 3277u ? 0xffu : (u < 42598u ? 0u : (u < 64225u ? 1u : 2u));
        out |= c << (8 * j);
    }
    return out;
}

// grid.x = blocks of 256 accession quads, grid.y = row lanes (a block walks rows blockIdx.y, + gridDim.y, ...): the
// row's hash is wave-uniform (scalar unit), a thread pays one splitmix64 per quad and no index division.
__device__ __forceinline__ uint32_t synth_quad_row(uint64_t row_hash, uint64_t acc_quad)
{
    const uint64_t h = splitmix64(row_hash + acc_quad);
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t u = (uint32_t)(h >> (16 * j)) & 0xffffu;
        const uint32_t c = u < 3277u ? 0xffu : (u < 42598u ? 0u : (u < 64225u ? 1u : 2u));
        out |= c << (8 * j);
    }
    return out;
}

__global__ void __launch_bounds__(256)
k_synth(uint32_t *__restrict__ db, int64_t pitch, int64_t n_snp, int64_t n_acc, uint64_t seed,
        int64_t snp0, int64_t acc0)
{
    // acc0 must be a multiple of 4 so that a shard sees the same quads as the full panel
    const int64_t quads_per_row = pitch / 4;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads_per_row) return;
    const int64_t c = q * 4;
    uint32_t pad = 0;                    // pad bytes are "missing"
    for (int j = 0; j < 4; ++j)
        if (c + j >= n_acc) pad |= 0xffu << (8 * j);
    for (int64_t row = blockIdx.y; row < n_snp; row += gridDim.y) {
        const uint64_t rh = splitmix64(seed ^ ((uint64_t)(snp0 + row) * 0xD6E8FEB86659FD93ull));
        db[row * quads_per_row + q] = synth_quad_row(rh, (uint64_t)((acc0 >> 2) + q)) | pad;
    }
}

// packed counterpart of k_synth: the same values, one byte (= one accession quad) per thread and row
__global__ void __launch_bounds__(256)
k_synth_packed(uint8_t *__restrict__ db, int64_t pitch, int64_t n_snp, int64_t n_acc, uint64_t seed,
               int64_t snp0, int64_t acc0)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= pitch) return;
    for (int64_t row = blockIdx.y; row < n_snp; row += gridDim.y) {
        const uint64_t rh = splitmix64(seed ^ ((uint64_t)(snp0 + row) * 0xD6E8FEB86659FD93ull));
        const uint32_t v = synth_quad_row(rh, (uint64_t)((acc0 >> 2) + q));
        uint32_t out = 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint32_t c = (v >> (8 * f)) & 0xffu;                      // 0, 1, 2 or 0xff
            const bool pad = (q * 4 + f) >= n_acc;
            out |= ((c == 0xffu || pad) ? 3u : c) << (2 * f);
        }
        db[row * pitch + q] = (uint8_t)out;
    }
}

// synthetic SAMPLE (benchmarks; SURVEY 8d recipe) generated where it is used: weights [n,3] (ref, het, alt) of a
// sample planted on accession `planted` of the synthetic panel `seed`, rows snp0 .. snp0 + n - 1.  Counter-based
// like the panel: row s depends on (seed, s) only.  exp_tab[k] = exp(-k/10) comes from the host so that the
// numpy twin (snpmatch_amd.synth.sample_weights_twin) reproduces the bits.
__global__ void k_synth_sample(uint64_t seed, int64_t snp0, int64_t n, int64_t planted, uint32_t err_permille,
                               uint32_t pl_permille, const double *__restrict__ exp_tab, double *__restrict__ wei)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t s = (uint64_t)(snp0 + r);
    const uint32_t quad = synth_quad(seed, s, (uint64_t)(planted >> 2));
    uint32_t code = (quad >> (8 * (uint32_t)(planted & 3))) & 0xffu;          // 0, 1, 2 or 0xff (missing)
    const uint64_t h = splitmix64(splitmix64((seed ^ 0x5851F42D4C957F2Dull) + s * 0x9FB21C651E98DF25ull));
    const uint64_t h2 = splitmix64(h + 0x2545F4914F6CDD1Dull);
    if (code == 0xffu) code = (uint32_t)(h & 1u);                               // missing in the DB: ref or alt
    if ((uint32_t)((h >> 8) & 0xFFFFFFu) % 1000u < err_permille) code = (uint32_t)((h >> 40) & 0xFFFFu) % 3u;
    const int called = code == 0u ? 0 : (code == 2u ? 1 : 2);                   // column of the called genotype
    double w[3];
    if ((uint32_t)(h2 & 0xFFFFFFu) % 1000u < pl_permille) {
        const uint32_t pa = 1u + (uint32_t)((h2 >> 24) & 0xFFFFu) % 255u, pb = 1u + (uint32_t)((h2 >> 40) & 0xFFFFu) % 255u;
        w[called] = exp_tab[0];
        w[(called + 1) % 3] = exp_tab[pa];
        w[(called + 2) % 3] = exp_tab[pb];
    } else {
        w[0] = w[1] = w[2] = 0.0;
        w[called] = 1.0;
    }
    wei[3 * r] = w[0];
    wei[3 * r + 1] = w[1];
    wei[3 * r + 2] = w[2];
}

// ------------------------------------------------------------------------------------------------
// binom.sf(k, n, p) = P(X > floor(k)), X ~ Binomial(n, p) (np_test_identity, core/snpmatch.py:57-72):
// the smaller tail is summed term by term from its largest term outward (same algorithm as the host
// reference implementation in snpm_api.hip, used by the tests to cross-check).
__host__ __device__ inline double binom_sf_eval(double k, double n, double p)
{
    if (!(n >= 0) || !(p >= 0.0 && p <= 1.0) || k != k) return __builtin_nan("");
    const double kf = floor(k);
    if (kf < 0) return 1.0;
    if (kf >= n) return 0.0;
    if (p == 0.0) return 0.0;
    if (p == 1.0) return 1.0;
    const double lp = log(p), lq = log1p(-p);
    const double lg_n1 = lgamma(n + 1.0);
    const double mean = n * p;
    if (kf + 1 > mean) {                 // upper tail j = kf+1 .. n, terms decrease
        double j = kf + 1;
        double t = exp(lg_n1 - lgamma(j + 1.0) - lgamma(n - j + 1.0) + j * lp + (n - j) * lq);
        double s = 0;
        while (j <= n && t > 0) {
            s += t;
            if (t < s * 1e-18) break;
            t *= (n - j) / (j + 1.0) * (p / (1.0 - p));
            j += 1;
        }
        return s > 1.0 ? 1.0 : s;
    }
    double j = kf;                       // lower tail j = kf .. 0, terms decrease going down
    double t = exp(lg_n1 - lgamma(j + 1.0) - lgamma(n - j + 1.0) + j * lp + (n - j) * lq);
    double s = 0;
    while (j >= 0 && t > 0) {
        s += t;
        if (t < s * 1e-18) break;
        t *= j / (n - j + 1.0) * ((1.0 - p) / p);
        j -= 1;
    }
    const double sf = 1.0 - s;
    return sf < 0 ? 0.0 : sf;
}

// out[i] = (sf((n[i] - x[i]) - 1, n[i], error_rate) >= pthres); sf[i] optional
__global__ void k_binom_identity(const double *__restrict__ x, const int64_t *__restrict__ n, int64_t len,
                                 double error_rate, double pthres, int64_t *__restrict__ out, double *__restrict__ sf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const double nn = (double)n[i];
    const double v = binom_sf_eval(nn - x[i] - 1.0, nn, error_rate);
    if (sf) sf[i] = v;
    out[i] = (v >= pthres) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// --refine support (identify_segregating_snps, core/snp_genotype.py:188-211): mask[row] = 1 when the
// informative (non-negative) calls of the listed accessions in that SNP row are not all identical.
__global__ void k_segregating(const int8_t *__restrict__ db, int64_t pitch, int packed, int64_t n_snp,
                              const int32_t *__restrict__ cols, int ncols, uint8_t *__restrict__ mask,
                              uint8_t *__restrict__ first_out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_snp) return;
    int first = -1;
    int seg = 0;
    for (int i = 0; i < ncols; ++i) {
        const int b = code_at(db, pitch, r, cols[i], packed);
        if (b < 0) continue;
        if (first < 0) first = b;
        else if (b != first) seg = 1;
    }
    mask[r] = (uint8_t)seg;
    // accession-sharded DBs: the first informative call (0xFF = none) lets the ranks decide together whether the
    // row segregates among columns that live on different GPUs
    if (first_out) first_out[r] = (uint8_t)(first < 0 ? 0xFF : first);
}

// ------------------------------------------------------------------------------------------------
// In-silico F1s (match_insilico_f1s, core/csmatch.py:115-125).  For a pair (i, j) of accession columns a
// matched SNP is "alt" when both calls are 1, "ref" when both are 0, "het" when both are informative and
// differ; the pair's score is np.sum(W[alt, 2]) + np.sum(W[ref, 0]) + np.sum(W[het, 1]) and the reference
// prints it as a float, so the bits of numpy's summation matter.  np.sum of a contiguous fp64 vector is
//   res = 0.0;  for every 8192-element chunk (the ufunc buffer):  res += pairwise(chunk)
// with pairwise() = numpy's DOUBLE_pairwise_sum: < 8 elements sequential; <= 128 elements eight strided
// accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail sequentially; otherwise split at
// n/2 rounded down to a multiple of 8.  The kernels below compact each category's weights in SNP order
// (count -> scan -> scatter) and evaluate exactly that tree.
constexpr int F1_BLOCK = 256;
constexpr int F1_ROWS_PER_THREAD = 8;
constexpr int F1_ROWS_PER_BLOCK = F1_BLOCK * F1_ROWS_PER_THREAD;   // 2048
constexpr int NP_SUM_CHUNK = 8192;
constexpr int NP_PW_LEAF = 128;
constexpr int F1_MAX_SEL = 32;

// codes[c][s] = call of selected accession c at matched SNP s (0 ref, 1 alt, 2 het, 3 other, 0xFF missing);
// rows n..stride-1 are padding (missing)
__global__ void __launch_bounds__(256)
k_f1_gather(const int8_t *__restrict__ db, int64_t pitch, int packed, const int64_t *__restrict__ row_idx, int64_t row0,
            int64_t n, const int32_t *__restrict__ acc, int n_sel, uint8_t *__restrict__ codes, int64_t stride)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= stride) return;
    const int64_t prow = (s < n) ? (row_idx ? row_idx[s] : row0 + s) : 0;
    for (int c = 0; c < n_sel; ++c)
        codes[(int64_t)c * stride + s] = (s < n) ? (uint8_t)code_at(db, pitch, prow, acc[c], packed) : (uint8_t)0xFF;
}

// category of one SNP for a pair: 0 alt, 1 ref, 2 het, 3 not informative
__device__ __forceinline__ int f1_cat(uint32_t a, uint32_t b)
{
    if (a == 1u && b == 1u) return 0;
    if (a == 0u && b == 0u) return 1;
    if (a != 0xFFu && b != 0xFFu && a != b) return 2;
    return 3;
}

// per-thread category counts of its 8 consecutive SNPs, packed in 16-bit fields (alt | ref << 16 | het << 32)
__device__ __forceinline__ uint64_t f1_thread_counts(uint64_t xa, uint64_t xb)
{
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < F1_ROWS_PER_THREAD; ++k) {
        const int cat = f1_cat((uint32_t)(xa >> (8 * k)) & 0xFFu, (uint32_t)(xb >> (8 * k)) & 0xFFu);
        if (cat < 3) c += 1ull << (16 * cat);
    }
    return c;
}

// grid (stride / 2048, pairs of this batch): cnt[(pair_local * 3 + cat) * nblk + blk]
__global__ void __launch_bounds__(F1_BLOCK)
k_f1_count(const uint8_t *__restrict__ codes, int64_t stride, const int32_t *__restrict__ pair_ij, int pair0,
           uint32_t *__restrict__ cnt, int64_t nblk)
{
    __shared__ uint64_t wave_tot[F1_BLOCK / WAVE];
    const int pair = pair0 + blockIdx.y;
    const uint8_t *ci = codes + (int64_t)pair_ij[2 * pair] * stride;
    const uint8_t *cj = codes + (int64_t)pair_ij[2 * pair + 1] * stride;
    const int64_t base = (int64_t)blockIdx.x * F1_ROWS_PER_BLOCK + (int64_t)threadIdx.x * F1_ROWS_PER_THREAD;
    uint64_t c = f1_thread_counts(*(const uint64_t *)(ci + base), *(const uint64_t *)(cj + base));
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & (WAVE - 1)) == 0) wave_tot[threadIdx.x / WAVE] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < F1_BLOCK / WAVE; ++w) t += wave_tot[w];
        for (int cat = 0; cat < 3; ++cat)
            cnt[((int64_t)blockIdx.y * 3 + cat) * nblk + blockIdx.x] = (uint32_t)((t >> (16 * cat)) & 0xFFFFu);
    }
}

// one block per list: cnt -> exclusive prefix (in place), total -> m[list]
__global__ void __launch_bounds__(256)
k_f1_scan(uint32_t *__restrict__ cnt, int64_t nblk, uint32_t *__restrict__ m)
{
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;
    uint32_t *c = cnt + (int64_t)blockIdx.x * nblk;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblk; b0 += 256) {
        const int64_t b = b0 + threadIdx.x;
        const uint32_t v = (b < nblk) ? c[b] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d);
            if ((int)(threadIdx.x & (WAVE - 1)) >= d) inc += t;
        }
        if ((threadIdx.x & (WAVE - 1)) == WAVE - 1) wave_tot[threadIdx.x / WAVE] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (int w = 0; w < (int)(threadIdx.x / WAVE); ++w) before += wave_tot[w];
        if (b < nblk) c[b] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) m[blockIdx.x] = carry_s;
}

// first element of list (pair_local, cat) inside the pair's compacted-weight slab
__device__ __forceinline__ int64_t f1_list_base(const uint32_t *__restrict__ m, int pair_local, int cat, int64_t n)
{
    int64_t off = (int64_t)pair_local * n;
    if (cat >= 1) off += m[pair_local * 3];
    if (cat >= 2) off += m[pair_local * 3 + 1];
    return off;
}

// same grid as k_f1_count: cw[list base + rank of the SNP inside its list] = weight of the SNP's category
__global__ void __launch_bounds__(F1_BLOCK)
k_f1_compact(const uint8_t *__restrict__ codes, int64_t stride, const int32_t *__restrict__ pair_ij, int pair0,
             const double *__restrict__ w, int64_t n, const uint32_t *__restrict__ blk_off, int64_t nblk,
             const uint32_t *__restrict__ m, double *__restrict__ cw)
{
    __shared__ uint64_t wave_tot[F1_BLOCK / WAVE];
    const int pair = pair0 + blockIdx.y;
    const uint8_t *ci = codes + (int64_t)pair_ij[2 * pair] * stride;
    const uint8_t *cj = codes + (int64_t)pair_ij[2 * pair + 1] * stride;
    const int64_t base = (int64_t)blockIdx.x * F1_ROWS_PER_BLOCK + (int64_t)threadIdx.x * F1_ROWS_PER_THREAD;
    const uint64_t xa = *(const uint64_t *)(ci + base), xb = *(const uint64_t *)(cj + base);
    const uint64_t mine = f1_thread_counts(xa, xb);
    uint64_t inc = mine;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t t = __shfl_up(inc, d);
        if ((int)(threadIdx.x & (WAVE - 1)) >= d) inc += t;
    }
    if ((threadIdx.x & (WAVE - 1)) == WAVE - 1) wave_tot[threadIdx.x / WAVE] = inc;
    __syncthreads();
    uint64_t before = inc - mine;
    for (int wv = 0; wv < (int)(threadIdx.x / WAVE); ++wv) before += wave_tot[wv];
    int64_t pos[3];
#pragma unroll
    for (int cat = 0; cat < 3; ++cat)
        pos[cat] = f1_list_base(m, blockIdx.y, cat, n) + blk_off[((int64_t)blockIdx.y * 3 + cat) * nblk + blockIdx.x] +
                   (int64_t)((before >> (16 * cat)) & 0xFFFFu);
#pragma unroll
    for (int k = 0; k < F1_ROWS_PER_THREAD; ++k) {
        const int cat = f1_cat((uint32_t)(xa >> (8 * k)) & 0xFFu, (uint32_t)(xb >> (8 * k)) & 0xFFu);
        if (cat == 0) cw[pos[0]++] = w[(base + k) * 3 + 2];
        else if (cat == 1) cw[pos[1]++] = w[(base + k) * 3 + 0];
        else if (cat == 2) cw[pos[2]++] = w[(base + k) * 3 + 1];
    }
}

// child of the pairwise-sum node [lo, lo + len) that contains element p (len > NP_PW_LEAF)
__device__ __forceinline__ void np_pw_descend(int p, int &lo, int &len)
{
    int n2 = len / 2;
    n2 -= n2 % 8;
    if (p < lo + n2) len = n2;
    else { lo += n2; len -= n2; }
}

// grid (max chunks, lists of this batch): chunk_sum[list * max_chunks + c] = numpy pairwise sum of the c-th
// 8192-element chunk of the list.  Blocks past the list's last chunk exit.
__global__ void __launch_bounds__(256)
k_f1_chunks(const double *__restrict__ cw, const uint32_t *__restrict__ m, int64_t n, int64_t max_chunks,
            double *__restrict__ chunk_sum)
{
    __shared__ double val[NP_SUM_CHUNK / 8];
    const int list = blockIdx.y;
    const int64_t mm = m[list];
    const int64_t first = (int64_t)blockIdx.x * NP_SUM_CHUNK;
    if (first >= mm) return;
    const int len = (int)((mm - first < NP_SUM_CHUNK) ? (mm - first) : NP_SUM_CHUNK);
    const double *a = cw + f1_list_base(m, list / 3, list % 3, n) + first;
    double *out = chunk_sum + (int64_t)list * max_chunks + blockIdx.x;
    if (len < 8) {
        if (threadIdx.x == 0) {
            double r = 0.0;
            for (int i = 0; i < len; ++i) r = r + a[i];
            *out = r;
        }
        return;
    }
    const int ncand = (len + 7) / 8;          // every leaf starts at a multiple of 8
    const int j = threadIdx.x & 7;
    for (int cand = threadIdx.x >> 3; cand < ncand; cand += 256 / 8) {
        const int p = cand * 8;
        int lo = 0, ln = len;
        while (ln > NP_PW_LEAF) np_pw_descend(p, lo, ln);
        if (lo == p) {                        // the 8 lanes of the group own the 8 accumulators of this leaf
            const int k8 = ln - ln % 8;
            double r = a[lo + j];
            for (int i = 8; i < k8; i += 8) r = r + a[lo + i + j];
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            if (j == 0) {
                for (int i = k8; i < ln; ++i) r = r + a[lo + i];
                val[cand] = r;
            }
        }
    }
    // inner nodes, deepest level first: a node's value replaces its left child's slot
    for (int d = 7; d >= 0; --d) {
        __syncthreads();
        for (int cand = threadIdx.x; cand < ncand; cand += 256) {
            const int p = cand * 8;
            int lo = 0, ln = len;
            bool inner = true;
            for (int lvl = 0; lvl < d; ++lvl) {
                if (ln <= NP_PW_LEAF) { inner = false; break; }
                np_pw_descend(p, lo, ln);
            }
            if (inner && ln > NP_PW_LEAF && lo == p) {
                int n2 = ln / 2;
                n2 -= n2 % 8;
                val[lo / 8] = val[lo / 8] + val[(lo + n2) / 8];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *out = val[0];
}

// one block per pair of the batch: add the chunk sums of its three lists in order, then
// score = (S_alt + S_ref) + S_het and ninfo = the three list lengths
__global__ void __launch_bounds__(192)
k_f1_finish(const double *__restrict__ chunk_sum, const uint32_t *__restrict__ m, int64_t max_chunks, int pair0,
            double *__restrict__ score, int64_t *__restrict__ ninfo)
{
    __shared__ double tile[3][1024];
    __shared__ double total[3];
    const int cat = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const int list = blockIdx.x * 3 + cat;
    const int64_t nch = ((int64_t)m[list] + NP_SUM_CHUNK - 1) / NP_SUM_CHUNK;
    int64_t nch_max = 0;
    for (int c = 0; c < 3; ++c) {
        const int64_t t = ((int64_t)m[blockIdx.x * 3 + c] + NP_SUM_CHUNK - 1) / NP_SUM_CHUNK;
        nch_max = t > nch_max ? t : nch_max;
    }
    const double *cs = chunk_sum + (int64_t)list * max_chunks;
    double r = 0.0;
    for (int64_t c0 = 0; c0 < nch_max; c0 += 1024) {
        const int64_t left = nch - c0;
        const int cnt = (int)(left < 0 ? 0 : (left < 1024 ? left : 1024));
        for (int i = lane; i < cnt; i += WAVE) tile[cat][i] = cs[c0 + i];
        __syncthreads();
        if (lane == 0)
            for (int i = 0; i < cnt; ++i) r = r + tile[cat][i];
        __syncthreads();
    }
    if (lane == 0) total[cat] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        score[pair0 + blockIdx.x] = (total[0] + total[1]) + total[2];
        ninfo[pair0 + blockIdx.x] = (int64_t)m[blockIdx.x * 3] + m[blockIdx.x * 3 + 1] + m[blockIdx.x * 3 + 2];
    }
}

// snpm_genotype_once: (score, ninfo, likelihood, lrt) of one sample and the two status words (re-evaluated accessions, y > n
// flag of k_likelihood) in ONE buffer of 8-byte words [4 * n_acc + 2], copied back in one piece
__global__ void k_once_pack(const double *__restrict__ score, const int64_t *__restrict__ ninfo, const double *__restrict__ lik,
                            const double *__restrict__ lrt, const int *__restrict__ count, const int *__restrict__ domain_flag,
                            int64_t n_acc, int64_t *__restrict__ out)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0) {
        out[4 * n_acc] = count ? (int64_t)*count : 0;
        out[4 * n_acc + 1] = domain_flag ? (int64_t)*domain_flag : 0;
    }
    if (a >= n_acc) return;
    out[a] = __double_as_longlong(score[a]);
    out[n_acc + a] = ninfo[a];
    out[2 * n_acc + a] = lik ? __double_as_longlong(lik[a]) : 0;
    out[3 * n_acc + a] = lrt ? __double_as_longlong(lrt[a]) : 0;
}


// ------------------------------------------------------------------------------------------------
// PMC calibration: reads `n_dwords` dwords exactly once with the access shape of k_fast (one dword
// per lane, 256 contiguous bytes per wave instruction, non-temporal), so that FETCH_SIZE can be
// calibrated on a known byte count (MI355X_MICROARCH.md, HBM section).  The xor keeps the loads live.
__global__ void __launch_bounds__(256)
k_calib_read(const uint32_t *__restrict__ p, int64_t n_dwords, uint32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n_dwords; i += 4 * stride) {
        const uint32_t a = __builtin_nontemporal_load(p + i);
        const uint32_t b = __builtin_nontemporal_load(p + i + stride);
        const uint32_t c = __builtin_nontemporal_load(p + i + 2 * stride);
        const uint32_t d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n_dwords; i += stride) acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x9e3779b9u) out[0] = acc;       // practically never true; prevents dead-code elimination
}

}  // namespace snpm

#include "snpm_kernels_single.hpp"

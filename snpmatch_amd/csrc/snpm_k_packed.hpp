// snpm_k_packed.hpp -- the fast passes over 2-bit packed panels: k_fast_packed_q4 (PL-weighted samples, four-row LDS tables) and k_fast_bits (hard-call samples, bit-sliced counting).
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
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
                 int64_t n_acc, int64_t desc = 0, const int64_t *__restrict__ part_desc = nullptr)
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
    // SPLIT layout (desc, snpm_k_common.hpp): a wave past the main part's columns reads the tail matrix -- its own base, its own
    // (narrow) pitch, dwords counted from the tail's first one.  All wave-uniform: scalar registers.
    const int64_t tail_p = pk_tail_pitch(desc);
    const bool tail_wave = tail_p != 0 && dw_first * 4 >= pitch;
    const int8_t *dbw = tail_wave ? db + pk_tail_off(desc) : db;
    const int64_t pw = tail_wave ? tail_p : pitch;
    const int64_t dw0 = tail_wave ? pitch / 4 : 0;
    const int64_t row_dwords = (pitch + tail_p) / 4;
    const int64_t dw_end = (row_dwords < (n_acc + 15) / 16) ? row_dwords : (n_acc + 15) / 16;   // dwords of a row that hold accessions
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
    auto lane_inside = [&](int phj, int64_t dw) -> bool { return phased ? (phj < ph) : (dw < row_dwords && dw * 16 < ld); };
    bool lane_on;
    uint32_t lane_off;                                  // byte offset of my dword in a row (+ the first row of my phase's group: 32 bits hold 56 rows of <= 2^25 bytes)
    {
        const int phj = my_phase();
        const int64_t dw = my_dword(phj);
        lane_on = lane_inside(phj, dw);
        lane_off = lane_on ? (uint32_t)((dw - dw0) * 4) + (uint32_t)(phj * G) * (uint32_t)pw : 0u;
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
        const int8_t *rowbase = dbw + prow * pw;
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
                const_cast<int8_t *>(dbw + (row0 + r) * pw), 0, (int)(G * pw), 0x00020000);
#pragma unroll
            for (int u = 0; u < G; ++u)
                x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, lane_off, (int)(u * pw), NT ? 2 : 0);
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
                    const uint32_t *ptr = reinterpret_cast<const uint32_t *>(dbw + row_idx[tr + grp * G + u] * pw + (my_dword(ph_j) - dw0) * 4);
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
            int64_t n_acc, int64_t desc = 0)
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
    // Grid = (parts, column blocks): the PART is the fast block index.  Workgroups are dealt to the eight XCDs round-robin in
    // linear block order; with the column block as the fast index (rounds 1-3) a panel of two column blocks put every block of the
    // full column on the four even XCDs and the light tail blocks on the four odd ones -- 1040 ... 2048 accessions scanned at half
    // the chip (1024 accessions x 40M rows 1.72 ms, 1040 accessions 2.9 ms).
    // (same-box A/B of the two orders on nine widths: profiles/r04_bits_grid_order_ab.txt)
    const int64_t cb = blockIdx.y;
    const int64_t dw_first = cb * nthr + __builtin_amdgcn_readfirstlane(tid - lane);   // first dword of my wave (scalar)
    // SPLIT layout (desc): a wave past the main part's columns reads the tail matrix (see k_fast_packed_q4)
    const int64_t tail_p = pk_tail_pitch(desc);
    const bool tail_wave = tail_p != 0 && dw_first * 4 >= pitch;
    const int8_t *dbw = tail_wave ? db + pk_tail_off(desc) : db;
    const int64_t pw = tail_wave ? tail_p : pitch;
    const int64_t dw0 = tail_wave ? pitch / 4 : 0;
    const int64_t row_dwords = (pitch + tail_p) / 4;
    const int64_t dw_end = (row_dwords < (n_acc + 15) / 16) ? row_dwords : (n_acc + 15) / 16;   // dwords of a row that hold accessions
    int ph_t = 0, ph = 1;                               // wave-uniform
    if (SNPM_Q4_PHASES && dw_first < dw_end && dw_end - dw_first <= WAVE / 2) {
        ph_t = (int)(dw_end - dw_first);
        ph = (WAVE / ph_t < PH_MAX) ? WAVE / ph_t : PH_MAX;
    }
    const bool phased = ph > 1;
    const int ph_j = phased ? lane / ph_t : 0;           // my phase
    const int64_t my_dw = phased ? dw_first + (lane - ph_j * ph_t) : cb * nthr + tid;
    const int64_t byte0 = (my_dw - dw0) * 4;            // my byte inside a row of the matrix my wave reads
    const int64_t col0 = my_dw * 16;
    const bool lane_on = phased ? (ph_j < ph) : (my_dw < row_dwords && col0 < ld);
    const int64_t p = blockIdx.x;
    const int64_t P = gridDim.x;
    const int64_t n_tiles_total = (n + TR - 1) / TR;

    uint32_t hit16[8], miss16[8];       // [d]: accession d (low half) and d + 8 (high half)
    uint32_t h1 = 0, h2 = 0, h4 = 0, h8 = 0, h16 = 0, h32 = 0, h64 = 0;   // bit-sliced hit counts of the current tile
    uint32_t m1 = 0, m2 = 0, m4 = 0, m8 = 0, m16 = 0, m32 = 0, m64 = 0;   // ... and missing counts
#pragma unroll
    for (int i = 0; i < 8; ++i) hit16[i] = miss16[i] = 0;

    // phased lanes: + the first row of my phase's group
    const uint32_t lane_off = lane_on ? (uint32_t)byte0 + (uint32_t)(ph_j * 8) * (uint32_t)pw : 0u;
    auto load = [&](int64_t rr) -> uint32_t {
        const int64_t prow = GATHER ? row_idx[rr] : (row0 + rr);
        const int8_t *rowbase = dbw + prow * pw;
        uint32_t off = lane_off;
        asm volatile("" : "+v"(off));
        const uint32_t *ptr = reinterpret_cast<const uint32_t *>(rowbase + off);
        return NT ? __builtin_nontemporal_load(ptr) : *ptr;
    };
    // H consecutive rows through a buffer resource (see k_fast_packed_q4::load_group)
    auto load_rows = [&](uint32_t (&x)[4], int64_t r) {
        if constexpr (!GATHER) {
            const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<int8_t *>(dbw + (row0 + r) * pw), 0, (int)(4 * pw), 0x00020000);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, lane_off, (int)(u * pw), NT ? 2 : 0);
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
                            const uint32_t *ptr = reinterpret_cast<const uint32_t *>(dbw + row_idx[tr0 + grp * 8 + u] * pw + byte0);
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

}  // namespace snpm

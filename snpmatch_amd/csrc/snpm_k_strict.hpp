// snpm_k_strict.hpp -- reference summation order: k_strict / k_strict4 / k_strict_sparse(_T), the accession-major packed copy they read, the chains of chunk totals (k_scan, k_scan_few) and the patch of re-evaluated accessions.
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
// ------------------------------------------------------------------------------------------------
// Strict (reference-order) segment sums.
//   grid.x = segment, grid.y = column blocks of blockDim.x lanes
//   cols: optional list of accession indices (NULL = dense 0..ncols-1)
//   out_score [n_seg, ld] fp64 = ((0 + A_ref) + A_het) + A_alt, out_miss [n_seg, ld] u32
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(256)
k_strict(const int8_t *__restrict__ db, int64_t pitch, int64_t packed, const int64_t *__restrict__ row_idx, int64_t row0,
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
                 int64_t c0, const uint4 *lut, uint32_t coff_sub = 0u)
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
    // (coff_sub: a block of a SPLIT packed panel that lies in the tail matrix -- db / pitch are that matrix's, the lane's byte
    // counts from the tail's first byte)
    const uint32_t coff = (uint32_t)(PACKED ? c0 / 4 : c0) - coff_sub;
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
          const int *__restrict__ other_codes, const int *__restrict__ gate, int gate_cap, int64_t desc = 0)
{
    if (dense_tier_off(gate, gate_cap)) return;
    const int64_t c0 = ((int64_t)blockIdx.y * blockDim.x + threadIdx.x) * 4;
    if constexpr (PACKED && SNPM_STRICT_EXEC) {
        __shared__ uint4 s_lut[256];
        for (uint32_t e = threadIdx.x; e < 256u; e += blockDim.x) s_lut[e] = strict_lut_entry<SKIP>(e);
        __syncthreads();
        if (c0 >= ncols) return;
        // split layout: a block covers blockDim.x (64 / 128 / 256, dividing 256) consecutive row bytes, so it lies in the main
        // matrix or in the tail matrix as a whole (block-uniform)
        const int64_t tp = pk_tail_pitch(desc);
        const int64_t byte0 = (int64_t)blockIdx.y * blockDim.x;
        if (tp && byte0 >= pitch) {
            strict4_segments<SKIP, GATHER, true, true>(db + pk_tail_off(desc), tp, row_idx, row0, w, seg_off, chunk, n, seg0, n_seg, ncols,
                                                       out_score, out_miss, ld, c0, s_lut, (uint32_t)pitch);
            return;
        }
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
k_strict_sparse(const int8_t *__restrict__ db, int64_t pitch, int64_t packed, const int64_t *__restrict__ row_idx,
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
                        uint8_t *__restrict__ dT, int64_t pitchT, int64_t desc)
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
        // (a dword of four row bytes never straddles the main / tail boundary of a split panel: both parts are multiples of 4 bytes)
        if (row < n_snp && byte < pitch + pk_tail_pitch(desc)) v = *reinterpret_cast<const uint32_t *>(db + pk_off(pitch, desc, row, byte));
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

// the body of k_scan_few (one block of 256 threads; `tile` is the kernel's LDS): also the first half of k_once_tail, which adds the
// one-call path's likelihood / ratio / status step behind it in the same launch
__device__ __forceinline__ void scan_few_body(double (&tile)[2][SCAN_TILE_ELEMS + 2 * WAVE], const double *__restrict__ seg_score,
                                              int64_t n_seg, int64_t ld, const int *__restrict__ d_ncols, int cap,
                                              double *__restrict__ tot_score, const double *__restrict__ carry,
                                              const int32_t *__restrict__ patch_cols, double *__restrict__ patch_score)
{
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
    if ((int)threadIdx.x < ncols) {
        tot_score[threadIdx.x] = s;
        if (patch_score) patch_score[patch_cols[threadIdx.x]] = s;        // k_patch fused in: the totals replace the fast-pass scores
    }
}

__global__ void __launch_bounds__(256)
k_scan_few(const double *__restrict__ seg_score, int64_t n_seg, int64_t ld, const int *__restrict__ d_ncols, int cap,
           double *__restrict__ tot_score, const double *__restrict__ carry, const int32_t *__restrict__ patch_cols = nullptr,
           double *__restrict__ patch_score = nullptr)
{
    __shared__ __attribute__((aligned(16))) double tile[2][SCAN_TILE_ELEMS + 2 * WAVE];
    scan_few_body(tile, seg_score, n_seg, ld, d_ncols, cap, tot_score, carry, patch_cols, patch_score);
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

}  // namespace snpm

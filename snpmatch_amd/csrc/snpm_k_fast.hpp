// snpm_k_fast.hpp -- k_fast -- the int8 fast pass (dense / gathered / segmented instantiations).
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
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
       const int64_t *__restrict__ part_desc = nullptr, int tile_rows_rt = 0)
{
    // BPL = bytes (= accessions = accumulators) per lane and row; packed panels have their own kernels below
    static_assert(BPL == 4 || BPL == 8 || BPL == 16, "int8 panels: 4, 8 or 16 bytes per lane");
    constexpr int NDW = BPL / 4;
    constexpr int EPL = BPL;
    // rows per prefetch group: 4 (G*32 B = half a 256-B LUT block); 8-row groups spill at the 80-VGPR budget and measured no gain
    constexpr int G = GATHER ? SNPM_FAST_G_GATHER : SNPM_FAST_G;
    __shared__ __attribute__((aligned(256))) double s_lut[2][TR * 4];
    // rows per tile at run time (a multiple of 8, <= TR; 0 = TR): short gathered queries choose it so that every part gets the same
    // number of tiles -- a 200k-row sample on 1024 resident blocks is 1563 tiles of 128 rows, i.e. half of the blocks walk two tiles
    // and half one (the launch takes the time of two: 76 % efficient); tiles of 104 rows give 1924 = 1.88 per block (94 %)
    const int TRR = (tile_rows_rt > 0 && tile_rows_rt <= TR) ? tile_rows_rt : TR;

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
    const int64_t n_tiles_total = (rend - rbase + TRR - 1) / TRR;

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
            const int64_t tr0 = rbase + T0 * TRR;
            const int rows2 = 2 * (int)((rend - tr0 < TRR) ? (rend - tr0) : TRR);
            const double2 *src = reinterpret_cast<const double2 *>(lut + 4 * tr0);
            double2 *dst = reinterpret_cast<double2 *>(&s_lut[0][0]);
            for (int i = tid; i < rows2; i += nthr) dst[i] = src[i];
        }
        uint32_t xa[G][NDW], xb[G][NDW];
#pragma unroll
        for (int u = 0; u < G; ++u) load_row<BPL, NT>(row_ptr(rbase + T0 * TRR + u), xa[u]);
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
            const int64_t tr0 = rbase + T * TRR;
            const int rows = (int)((rend - tr0 < TRR) ? (rend - tr0) : TRR);
            const bool more = (T + P < n_tiles_total);
            const int64_t ntr0 = more ? rbase + (T + P) * TRR : tr0;      // my next tile (or a harmless re-read)
            // stage the next LUT tile (256 double2) in ONE register pair per thread when the block has
            // >= 256 threads; narrower blocks copy it synchronously at the end of the tile instead
            double2 pre0 = make_double2(0.0, 0.0);
            const bool staged = more && nthr >= TRR * 2;
            const int nrows2 = more ? 2 * (int)((rend - ntr0 < TRR) ? (rend - ntr0) : TRR) : 0;
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
                const int64_t rafter = (g + 2 < TRR / G) ? rnext + G : ntr0;
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
                    if (tid < TRR * 2) dst[tid] = pre0;
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

}  // namespace snpm

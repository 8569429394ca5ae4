// snpm_k_reduce.hpp -- ordered reduction of the fast passes' partial sums with the certificate fused in, slab carries, the per-call helpers of the reference-order kernels (add_if / add_sel / code_at / seg_bounds), segmented passes (batches, windows): per-segment bound, reduce, pair re-evaluation, totals.
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
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

// k_reduce_groups and k_reduce in ONE launch (round 5: a launch costs ~5 us of stream time, the two reduce steps of a 200k-SNP
// sample took 10 of its 130): grid (column blocks, groups) as k_reduce_groups; every block leaves its group's sums and takes a
// ticket of its column block, the block that draws the last ticket adds the groups in order (the additions, and therefore the
// bits, of the two-kernel form), runs the certificate and puts the ticket back to zero.  Hand-off without cache maintenance: the
// group sums leave as write-through (sc1) stores, the wave waits for them (vmcnt(0)), one lane takes the ticket with an agent-scope
// atomic; the block whose ticket came last reads the sums with sc1 loads (L2-served, never a stale L1 line).
__global__ void __launch_bounds__(64)
k_reduce_all(const double *__restrict__ part_score, const uint32_t *__restrict__ part_miss, int64_t n_slots, int64_t ld,
             int64_t n_acc, int64_t n_rows, double *__restrict__ grp_score, uint32_t *__restrict__ grp_miss,
             double *__restrict__ score, int64_t *__restrict__ ninfo, const double *__restrict__ d_eref, double efast,
             int force_first, int32_t *__restrict__ cols, int *__restrict__ count, int cap, unsigned *__restrict__ tickets)
{
    __shared__ int s_last;
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = blockIdx.y;
    const int64_t n_groups = gridDim.y;
    if (a < n_acc) {
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
        // write-through stores (sc1): the sums are in memory, not in this XCD's L2, once the wave's vmcnt is zero -- no
        // cache write-back per block (a __threadfence() here made the fused kernel slower than the two launches it replaces)
        __hip_atomic_store(&grp_score[g * ld + a], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&grp_miss[g * ld + a], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // one wave per block: its own stores are all there is to wait for
    if (threadIdx.x == 0)
        s_last = (__hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_groups - 1)) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a >= n_acc) return;
    double s = 0.0;
    int64_t m = 0;
    for (int64_t p = 0; p < n_groups; ++p) {
        s = s + __hip_atomic_load(&grp_score[p * ld + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m += __hip_atomic_load(&grp_miss[p * ld + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
constexpr int64_t STRICT_PAIRS_LANE_FORM = 2048;       // (pair, chunk) chains from which k_strict_pairs gives every chain a lane instead of a wave

// genotype code of (row, accession) in either panel format: int8 -> the byte (negative = missing);
// packed -> 2-bit field, 3 = missing (returned as -1)
__device__ __forceinline__ int code_at(const int8_t *__restrict__ db, int64_t pitch, int64_t prow, int64_t col, int64_t packed)
{
    if (packed) {               // `packed` is the panel's layout descriptor (snpm_k_common.hpp): row-major or split rows
        const int v = (((const uint8_t *)db)[pk_off(pitch, packed, prow, col >> 2)] >> (2 * (int)(col & 3))) & 3;
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
//   Two forms in one launch, chosen by the number of (pair, chunk) chains the device finds flagged:
//   few chains    a WAVE per chain (latency: a lone flagged pair should cost microseconds): 64 rows loaded side by side, their
//                 three per-class contributions formed in parallel (the reference adds 0.0 where the class does not match,
//                 add_sel), the chain of additions walked from broadcast values: 6 v_readlane + 3 v_add_f64 per row;
//   many chains   a LANE per chain (throughput: every lane of the wave form executes the same additions, 64 pairs of a
//   (>= 2048)     200-chunk sample kept the chip busy for 0.93 ms): a lane walks its own chunk, eight rows' loads in flight --
//                 ~0.13 ms however many chains there are, up to tens of thousands.
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(64)
k_strict_pairs(const int8_t *__restrict__ db, int64_t pitch, int64_t packed, const int64_t *__restrict__ row_idx, int64_t row0,
               const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk,
               const int32_t *__restrict__ pairs, const int *__restrict__ count, int cap, int64_t kmax,
               double *__restrict__ sums)
{
    const int np = *count < cap ? *count : cap;
    const int lane = threadIdx.x;
    if ((int64_t)np * kmax >= STRICT_PAIRS_LANE_FORM) {
        const int64_t n_items = (int64_t)np * kmax;
        const int64_t n_waves = (int64_t)gridDim.x * gridDim.y;
        for (int64_t it0 = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVE; it0 < n_items; it0 += n_waves * WAVE) {
            const int64_t it = it0 + lane;
            if (it >= n_items) continue;
            const int pr = (int)(it / kmax);
            const int64_t k = it - (int64_t)pr * kmax;
            const int64_t sg = pairs[2 * pr], col = pairs[2 * pr + 1];
            const int64_t s0 = seg_off[sg], s1 = seg_off[sg + 1];
            int64_t K = (s1 - s0 + chunk - 1) / chunk;
            if (K < 1) K = 1;
            if (k >= K) continue;
            const int64_t r0 = s0 + k * chunk;
            const int64_t r1 = (r0 + chunk < s1) ? r0 + chunk : s1;
            // where this lane's column lies in a row, and how its code comes out of the byte there, settled ONCE: code_at decides
            // the panel format per call, and eight loads behind eight branches were eight round trips, one after the other
            // (codes are compared with 0, 1, 2 only: an int8 byte taken as 0..255 and a 2-bit field 0..3 answer alike)
            int64_t base, stride;
            int sh = 0, mask = 0xff;
            if (packed) {
                const int64_t bcol = col >> 2, tp = pk_tail_pitch(packed);
                if (tp && bcol >= pitch) { base = pk_tail_off(packed) + (bcol - pitch); stride = tp; }
                else { base = bcol; stride = pitch; }
                sh = 2 * (int)(col & 3);
                mask = 3;
            } else {
                base = col;
                stride = pitch;
            }
            const uint8_t *cell = (const uint8_t *)db + base;
            double a_ref = 0.0, a_het = 0.0, a_alt = 0.0;
            int64_t r = r0;
            for (; r + 8 <= r1; r += 8) {
                int b[8];
                double w0[8], w1[8], w2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t prow = GATHER ? row_idx[r + u] : (row0 + r + u);
                    b[u] = ((int)cell[prow * stride] >> sh) & mask;
                    w0[u] = w[3 * (r + u) + 0];
                    w1[u] = SKIP ? 0.0 : w[3 * (r + u) + 1];
                    w2[u] = w[3 * (r + u) + 2];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a_ref = add_sel(a_ref, b[u] == 0, w0[u]);
                    if (!SKIP) a_het = add_sel(a_het, b[u] == 2, w1[u]);
                    a_alt = add_sel(a_alt, b[u] == 1, w2[u]);
                }
            }
            for (; r < r1; ++r) {
                const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
                const int bb = ((int)cell[prow * stride] >> sh) & mask;
                a_ref = add_sel(a_ref, bb == 0, w[3 * r + 0]);
                if (!SKIP) a_het = add_sel(a_het, bb == 2, w[3 * r + 1]);
                a_alt = add_sel(a_alt, bb == 1, w[3 * r + 2]);
            }
            sums[(int64_t)pr * kmax + k] = ((0.0 + a_ref) + a_het) + a_alt;
        }
        return;
    }
    for (int pr = blockIdx.y; pr < np; pr += gridDim.y) {
        const int64_t sg = pairs[2 * pr], col = pairs[2 * pr + 1];
        const int64_t s0 = seg_off[sg], s1 = seg_off[sg + 1];
        int64_t K = (s1 - s0 + chunk - 1) / chunk;
        if (K < 1) K = 1;                                  // an empty segment: one matchGTsAccs call on no rows
        for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
            const int64_t r0 = s0 + k * chunk;
            const int64_t r1 = (r0 + chunk < s1) ? r0 + chunk : s1;
            double a_ref = 0.0, a_het = 0.0, a_alt = 0.0;
            // the next 64 rows are requested before the current 64 are added
            int bn = -1;
            double wn0 = 0.0, wn1 = 0.0, wn2 = 0.0;
            if (r0 + lane < r1) {
                const int64_t r = r0 + lane;
                const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
                bn = code_at(db, pitch, prow, col, packed);
                wn0 = w[3 * r + 0];
                wn1 = w[3 * r + 1];
                wn2 = w[3 * r + 2];
            }
            for (int64_t rb = r0; rb < r1; rb += WAVE) {
                // this lane's row: what the reference adds to each class sum (0.0 where the class does not match: add_sel)
                const double c0 = bn == 0 ? wn0 : 0.0, c1 = bn == 2 ? wn1 : 0.0, c2 = bn == 1 ? wn2 : 0.0;
                bn = -1;
                wn0 = wn1 = wn2 = 0.0;
                if (rb + WAVE + lane < r1) {
                    const int64_t r = rb + WAVE + lane;
                    const int64_t prow = GATHER ? row_idx[r] : (row0 + r);
                    bn = code_at(db, pitch, prow, col, packed);
                    wn0 = w[3 * r + 0];
                    wn1 = w[3 * r + 1];
                    wn2 = w[3 * r + 2];
                }
                const int cnt = (int)((r1 - rb < WAVE) ? (r1 - rb) : WAVE);
                for (int i = 0; i < cnt; ++i) {
                    a_ref = a_ref + __shfl(c0, i);
                    if (!SKIP) a_het = a_het + __shfl(c1, i);
                    a_alt = a_alt + __shfl(c2, i);
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

}  // namespace snpm

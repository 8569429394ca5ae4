// snpm_kernels_single.hpp -- reference summation order for panels of ONE accession.
//
// matchGTsAccs multiplies an [N_acc, n] mask (a transposed view, so its n axis is strided) by the weights and calls
// .sum(axis = 1) (core/snpmatch.py:85-87).  numpy walks a strided axis row after row -- the order k_strict* restate --
// but with N_acc == 1 the [1, n] product is contiguous along the reduced axis as well and numpy takes its vector
// route: res = 0.0; for every 8192-element buffer piece: res += DOUBLE_pairwise_sum(piece).  The three per-category
// sums of one call are therefore pairwise trees over ALL n rows of the call (zeros of non-matching rows included,
// positions matter), then score = ((0 + A_ref) + A_het) + A_alt as everywhere else.  Pinned by the reference-generated
// goldens tests/golden/g1b_single_acc.npz and g2b_g5b_single_acc.npz.
//
// k_strict_single stands in for k_strict4 / k_strict / k_strict_sparse(_T) / k_strict_pairs on such panels (the host
// switches on snpm_panel::n_acc_total == 1: the width of the panel the REFERENCE would see, not of a rank's shard).
// One 256-thread block per matchGTsAccs call; an edge case, written for clarity and exactness, not for bandwidth.
#pragma once

namespace snpm {

constexpr int SINGLE_THREADS = 256;
constexpr int SINGLE_LEAVES = NP_SUM_CHUNK / 8;     // leaf candidates of one buffer piece: every leaf starts at a multiple of 8

// One call over rows [r0, r1) of the matched list, evaluated by the whole block; thread 0 returns the call's score and
// missing count.  sm_code [8192], sm_val [3][1024], sm_miss [1] are the block's LDS.
template <bool SKIP, bool GATHER>
__device__ __forceinline__ void single_call(const int8_t *__restrict__ db, int64_t pitch, int64_t packed,
                                            const int64_t *__restrict__ row_idx, int64_t row0, const double *__restrict__ w,
                                            int64_t r0, int64_t r1, int8_t *sm_code, double (*sm_val)[SINGLE_LEAVES],
                                            uint32_t *sm_miss, double &score, uint32_t &miss)
{
    const int tid = threadIdx.x;
    double res[3] = {0.0, 0.0, 0.0};            // thread 0: A_ref, A_het, A_alt
    if (tid == 0) *sm_miss = 0u;
    for (int64_t rb = r0; rb < r1; rb += NP_SUM_CHUNK) {
        const int len = (int)((r1 - rb < NP_SUM_CHUNK) ? (r1 - rb) : NP_SUM_CHUNK);
        __syncthreads();                        // the previous piece's codes / values are no longer read
        uint32_t my_miss = 0;
        for (int i = tid; i < len; i += SINGLE_THREADS) {
            const int64_t prow = GATHER ? row_idx[rb + i] : (row0 + rb + i);
            int b = code_at(db, pitch, prow, 0, packed);
            if (SKIP && b == 2) b = -1;
            my_miss += (b < 0);
            sm_code[i] = (int8_t)b;
        }
        if (my_miss) atomicAdd(sm_miss, my_miss);
        __syncthreads();
        // element i of category c: the row's weight where the call is that category's code, else 0.0
        auto elem = [&](int c, int i) -> double {
            const int code = (c == 0) ? 0 : (c == 1 ? 2 : 1);          // ref -> W[:,0], het -> W[:,1], alt -> W[:,2]
            return (sm_code[i] == code) ? w[3 * (rb + i) + c] : 0.0;
        };
        if (len < 8) {
            if (tid == 0)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double r = 0.0;
                    for (int i = 0; i < len; ++i) r = r + elem(c, i);
                    res[c] = res[c] + r;
                }
            continue;
        }
        const int ncand = (len + 7) / 8;
        const int j = tid & 7;
        for (int cand = tid >> 3; cand < ncand; cand += SINGLE_THREADS / 8) {
            const int p = cand * 8;
            int lo = 0, ln = len;
            while (ln > NP_PW_LEAF) np_pw_descend(p, lo, ln);
            if (lo == p) {                      // these 8 lanes own the 8 accumulators of the leaf [lo, lo + ln)
                const int k8 = ln - ln % 8;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double r = elem(c, lo + j);
                    for (int i = 8; i < k8; i += 8) r = r + elem(c, lo + i + j);
                    r = r + __shfl_xor(r, 1);
                    r = r + __shfl_xor(r, 2);
                    r = r + __shfl_xor(r, 4);
                    if (j == 0) {
                        for (int i = k8; i < ln; ++i) r = r + elem(c, lo + i);
                        sm_val[c][cand] = r;
                    }
                }
            }
        }
        // inner nodes, deepest level first: a node's value replaces its left child's slot
        for (int d = 7; d >= 0; --d) {
            __syncthreads();
            for (int cand = tid; cand < ncand; cand += SINGLE_THREADS) {
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
#pragma unroll
                    for (int c = 0; c < 3; ++c) sm_val[c][lo / 8] = sm_val[c][lo / 8] + sm_val[c][(lo + n2) / 8];
                }
            }
        }
        __syncthreads();
        if (tid == 0)
#pragma unroll
            for (int c = 0; c < 3; ++c) res[c] = res[c] + sm_val[c][0];
    }
    __syncthreads();
    if (tid == 0) {
        score = ((0.0 + res[0]) + res[1]) + res[2];
        miss = *sm_miss;
    }
}

// tier: 0 = always; 1 = sparse re-evaluation tier (runs iff 1 <= *count <= cap; the one flagged column is column 0);
//       2 = dense tier (runs iff *count > cap -- never true on a one-column panel, kept for symmetry)
// pairs == NULL: calls are the segments [seg0, seg0 + n_seg) (explicit seg_off or implicit chunk-row pieces of [0, n)):
//       out_score / out_miss [seg * ld].  grid.x walks the segments.
// pairs != NULL: blockIdx.y = flagged (segment, accession) pair, calls are the chunk-row pieces of that segment:
//       out_score [pair * kmax + k] (k_scan_pairs adds them in order).  grid.x walks the pieces.
template <bool SKIP, bool GATHER>
__global__ void __launch_bounds__(SINGLE_THREADS)
k_strict_single(const int8_t *__restrict__ db, int64_t pitch, int64_t packed, const int64_t *__restrict__ row_idx, int64_t row0,
                const double *__restrict__ w, const int64_t *__restrict__ seg_off, int64_t chunk, int64_t n, int64_t seg0,
                int64_t n_seg, const int32_t *__restrict__ pairs, const int *__restrict__ count, int cap, int tier,
                int64_t kmax, double *__restrict__ out_score, uint32_t *__restrict__ out_miss, int64_t ld)
{
    __shared__ int8_t sm_code[NP_SUM_CHUNK];
    __shared__ double sm_val[3][SINGLE_LEAVES];
    __shared__ uint32_t sm_miss;
    if (tier == 1 && !(*count >= 1 && *count <= cap)) return;
    if (tier == 2 && !(*count > cap)) return;
    double score = 0.0;
    uint32_t miss = 0;
    if (pairs) {
        const int np = *count < cap ? *count : cap;
        for (int pr = blockIdx.y; pr < np; pr += gridDim.y) {
        const int64_t sg = pairs[2 * pr];
        const int64_t s0 = seg_off[sg], s1 = seg_off[sg + 1];
        int64_t K = (s1 - s0 + chunk - 1) / chunk;
        if (K < 1) K = 1;                       // an empty segment: one call on no rows
        for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
            const int64_t r0 = s0 + k * chunk;
            const int64_t r1 = (r0 + chunk < s1) ? r0 + chunk : s1;
            single_call<SKIP, GATHER>(db, pitch, packed, row_idx, row0, w, r0, r1, sm_code, sm_val, &sm_miss, score, miss);
            if (threadIdx.x == 0) out_score[(int64_t)pr * kmax + k] = score;
        }
        }
        return;
    }
    for (int64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        int64_t r0, r1;
        seg_bounds(seg_off, chunk, n, seg_off ? seg : seg0 + seg, r0, r1);
        single_call<SKIP, GATHER>(db, pitch, packed, row_idx, row0, w, r0, r1, sm_code, sm_val, &sm_miss, score, miss);
        if (threadIdx.x == 0) {
            out_score[seg * ld] = score;
            out_miss[seg * ld] = miss;
        }
    }
}

}  // namespace snpm

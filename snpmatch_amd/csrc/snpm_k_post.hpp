// snpm_k_post.hpp -- after the scores: likelihood / nanmin / ratio, the binomial identity test, --refine's segregating-site scan, the in-silico F1s in numpy's summation order, the one-copy result pack of snpm_genotype_once.
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
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
__global__ void k_segregating(const int8_t *__restrict__ db, int64_t pitch, int64_t packed, int64_t n_snp,
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
k_f1_gather(const int8_t *__restrict__ db, int64_t pitch, int64_t packed, const int64_t *__restrict__ row_idx, int64_t row0,
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

// the flags a batch call reads at its end -- the count of unproven pairs, the likelihoods' domain flag, the row / weight flags of
// the inputs, the contraction's flags -- gathered into the status line in front of the results, so that they ride in the one copy
// that brings the results back (a null source reads as 0)
__global__ void k_batch_status(const int *__restrict__ pair_count, const int *__restrict__ lik_flag, const int *__restrict__ in_flags,
                               const unsigned long long *__restrict__ contraction_flags, int *__restrict__ status)
{
    const int t = threadIdx.x;
    if (t >= 16) return;
    int v = 0;
    if (t == 0 && pair_count) v = *pair_count;
    if (t == 1 && lik_flag) v = *lik_flag;
    if (t == 2 && in_flags) v = *in_flags;
    if ((t == 4 || t == 5) && contraction_flags) v = (int)(*contraction_flags >> (32 * (t - 4)));
    status[t] = v;
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

// The same in one launch with the likelihoods (k_likelihood of ONE row on the truncated counts, nanmin, ratio: core/snpmatch.py:
// 96, 106-117) and written straight into `out` -- which may be the pinned host slab: no copy engine, no fill of the flag word
// (profiles/r04_once_timeline_*).  ONE block.  state[1] = bad-input bits of k_once_prep (read, reported in word 4 n_acc + 1 bits
// 1-2, cleared for the next call); lik_tmp [n_acc] device scratch (the ratio pass re-reads the likelihoods: not from the host).
__device__ __forceinline__ void once_finish_body(const double *__restrict__ score, const int64_t *__restrict__ ninfo, int64_t n_acc,
                                                 int want_lik, const int *__restrict__ count, unsigned *__restrict__ state,
                                                 double *__restrict__ lik_tmp, int64_t *__restrict__ out, double (&s_min)[16],
                                                 double &s_top, int &s_bad)
{
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    double mn = __builtin_inf();
    int bad = 0;
    for (int64_t i = threadIdx.x; i < n_acc; i += blockDim.x) {
        const double y = score[i];
        const int64_t ni = ninfo[i];
        out[i] = __double_as_longlong(y);
        out[n_acc + i] = ni;
        double l = 0.0;
        if (want_lik) {
            l = likeli_one(trunc(y), (double)ni, &bad);
            lik_tmp[i] = l;
            if (l == l && l < mn) mn = l;
        }
        out[2 * n_acc + i] = want_lik ? __double_as_longlong(l) : 0;
    }
    if (bad) atomicOr(&s_bad, 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(mn, o);
        mn = other < mn ? other : mn;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = __builtin_inf();
        const int nw = (blockDim.x + 63) >> 6;
        for (int k = 0; k < nw; ++k) m = s_min[k] < m ? s_min[k] : m;
        if (m == __builtin_inf()) m = __builtin_nan("");      // all-NaN: np.nanmin -> nan
        s_top = m;
        out[4 * n_acc] = count ? (int64_t)*count : 0;
        out[4 * n_acc + 1] = (int64_t)(s_bad & 1) | ((int64_t)(state[1] & 3u) << 1);
        state[1] = 0u;
    }
    __syncthreads();
    const double top = s_top;
    for (int64_t i = threadIdx.x; i < n_acc; i += blockDim.x)
        out[3 * n_acc + i] = want_lik ? __double_as_longlong((top <= 0.0) ? __builtin_nan("") : lik_tmp[i] / top) : 0;
}

__global__ void __launch_bounds__(1024)
k_once_finish(const double *__restrict__ score, const int64_t *__restrict__ ninfo, int64_t n_acc, int want_lik,
              const int *__restrict__ count, unsigned *__restrict__ state, double *__restrict__ lik_tmp, int64_t *__restrict__ out)
{
    __shared__ double s_min[16];
    __shared__ double s_top;
    __shared__ int s_bad;
    once_finish_body(score, ninfo, n_acc, want_lik, count, state, lik_tmp, out, s_min, s_top, s_bad);
}

// The tail of the one-call path in ONE launch (round 5; it was two: k_scan_few and k_once_finish, ~5 us of stream time each): the
// chain of the flagged accessions' chunk sums with the patch of their scores (scan_few_body: nothing to do when no accession was
// flagged), then the likelihood / ratio / status step on the patched scores.  One block of 256 threads.
__global__ void __launch_bounds__(256)
k_once_tail(const double *__restrict__ seg_score, int64_t n_seg, int64_t ld, const int *__restrict__ d_ncols, int cap,
            double *__restrict__ tot_score, const int32_t *__restrict__ patch_cols, double *__restrict__ score,
            const int64_t *__restrict__ ninfo, int64_t n_acc, int want_lik, unsigned *__restrict__ state, double *__restrict__ lik_tmp,
            int64_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double tile[2][SCAN_TILE_ELEMS + 2 * WAVE];
    __shared__ double s_min[16];
    __shared__ double s_top;
    __shared__ int s_bad;
    scan_few_body(tile, seg_score, n_seg, ld, d_ncols, cap, tot_score, nullptr, patch_cols, score);
    __syncthreads();                // the patched scores are visible to every wave of the block
    once_finish_body(score, ninfo, n_acc, want_lik, d_ncols, state, lik_tmp, out, s_min, s_top, s_bad);
}

}  // namespace snpm

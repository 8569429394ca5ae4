// snpm_k_shared.hpp -- the SHARED-ROW scan of a batch of samples ("batching many samples per launch turns the scan
// into a small-N contraction; only then would MFMA be worth revisiting").
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
//
// snpm_score_batch scores sample b as its own segment: every sample gathers and reads its own DB rows (B x 233 MB for 64
// samples of 200k SNPs on 1135 accessions).  When the samples of a batch are genotyped on largely the same markers -- the
// production use, many samples against one panel -- each DB row can be read ONCE and scored against every sample:
//
//     score[s, a] = sum_r sum_c W[s, r, c] * [db[r, a] == c]                       (core/snpmatch.py:84-88, per sample)
//
// is a contraction over k = (row, class) of a weight matrix [samples, k] with the one-hot expansion of the panel [k, accessions].
// fp64 arithmetic cannot use it (gfx950 runs v_mfma_f64 at the fp64 VALU rate, 3-4 x the flops of the LUT form), but the
// EXACT mode only needs a fast pass with a rigorous error bound (DESIGN.md "Exactness contract"), so the weights are taken as
// FIXED-POINT numbers: w in [0, 1] -> Q = floor(w * 2^F), written as L balanced base-256 digits d_0 .. d_{L-1} in [-128, 127],
// w ~ sum_j d_j 256^(L-1-j) 2^-F, F = 8 (L - 1) + 6.  Digit j of every sample is one ROW of an int8 matrix A; the panel's calls become
// indicator bytes {0, 1} per class (B); v_mfma_i32_32x32x32_i8 adds the products EXACTLY in int32.  One further row per sample holds
// a 1 in every informative class: its product counts the sample's informative sites (ninfo).  The error of the pass is the
// quantisation alone, one-sided: fixed <= exact <= fixed + n 2^-F over a sample's n matched SNPs, and the certificate flags the few (sample, accession) pairs whose int() is not proven, as the fp64
// fast pass does; those are re-scored in reference order by k_strict_pairs.
//
// Geometry of the contraction (k_sh_mfma):
//   K          = (row, class) over the THREE informative classes (ref, alt, het): a missing call scores nothing and counts
//                nothing, so it needs no K slot (the first version spent a quarter of its MFMAs on it).
//   round      = 32 union rows = three K = 32 blocks, one per class; lane (c = lane & 31, h = lane >> 5) holds rows 16h .. 16h+15 of
//                the round, byte i of a fragment = row 16h + i  [A and B use the same slots, so the hardware's k order inside a
//                lane does not matter].  The host counts in STEPS of 8 rows; tiles and pads are multiples of 8 steps = two rounds.
//   A fragment = 32 matrix rows (sample digits) x 32 rows of one class, 16 B per lane, stored in fragment order by k_sh_expand:
//                A[group][round][class < 3][tile t < 4][lane][16 B], group = 128 matrix rows (3 KiB per group and step)
//   B fragment = 32 accessions x 32 rows of one class, built in registers: the lane's sixteen row dwords (4 accessions each) are
//                transposed with v_perm_b32 into the codes of each accession at four consecutive rows, and the indicator of a
//                class is one 3-input boolean per dword (v_bitop3 on the code bits 0 and 1)
//   wave tile  = 128 matrix rows x 128 accessions = 4 x 4 MFMA tiles, 256 accumulator registers, one wave per SIMD;
//                a wave reads its A fragments and panel dwords straight into registers (no LDS, no barrier): panel dwords two
//                rounds ahead, row indices three, the A fragments of a class reloaded for the next round right behind its MFMAs
//   grid       = row tiles x (groups x accession groups / 4); blocks of one row tile are dealt to ONE XCD so that the L2 of that
//                XCD serves the re-reads (A by every accession group, panel rows by every matrix-row group)
#pragma once

namespace snpm {

constexpr int SH_GROUP_ROWS = 128;      // matrix rows (sample digits) per wave tile
constexpr int SH_WAVE_ACCS = 128;       // accessions per wave tile
constexpr int SH_STEP_ROWS = 8;         // union rows per STEP, the unit of the host's geometry (tiles, parts, pads)
constexpr int SH_ROUND_ROWS = 32;       // a ROUND = 4 steps = 32 union rows: what a wave contracts with 3 x 16 MFMAs (K = 32 rows of ONE class each)
constexpr int SH_DEPTH = 8;             // steps a tile is a multiple of: two rounds (the contraction's loop is unrolled over two)
constexpr int SH_PAD_STEPS = 3 * SH_DEPTH;   // steps a wave may read (never score) past the last one: row list three rounds, panel rows two, A fragments one
constexpr int SH_A_STEP_BYTES = 3072;   // bytes of A per group of 128 matrix rows and step: a round holds 3 classes x 4 tiles x 64 lanes x 16 B
constexpr int SH_MAX_RPS = 8;           // matrix rows per sample: digits + the count row

typedef int sh_v4i __attribute__((ext_vector_type(4)));
typedef int sh_v16i __attribute__((ext_vector_type(16)));

__host__ __device__ __forceinline__ int sh_frac_bits(int digits) { return 8 * (digits - 1) + 6; }

// meta block of a shared-row pass (device, int64 [8]): [0] union rows U, [1] bad-input bits (1 a sample's rows are not strictly
// increasing, 2 a weight lies outside [0, 1] or is not finite -- found by k_sh_expand, i.e. AFTER the pass was taken: the caller
// then scores the batch again through the per-sample pass --, 4 a row index outside the panel), [3] entries the probe marked
// ---------------------------------------------------------------------------------------------------------------
// k_sh_mark: one pass over the batch's ROW LISTS (blockIdx.y = sample; the weights are not read here: k_sh_expand vets them while
// it converts them): marks the rows in the bitmap of panel rows (the bit is read first: most rows of a batch on one marker set are
// marked already, and 64 samples would otherwise send 64 atomics to every word), checks that every index lies inside the panel and
// that every sample's list is strictly increasing (the union holds every (sample, row) at most once).  Four entries per lane in
// flight, and the grid is sized so that those four are ALL a lane has (ceil(longest sample / 1024) blocks per sample: 41 us for
// 64 x 194k entries; three rounds of four on a quarter of the blocks took 70).
__device__ __forceinline__ void sh_mark_row(int64_t r, int64_t r_prev, bool has_prev, int64_t n_snp, uint32_t *__restrict__ bitmap, int &bad)
{
    if (r < 0 || r >= n_snp) {
        bad |= 4;
        return;
    }
    if (has_prev && r_prev >= r) bad |= 1;
    uint32_t *wp = &bitmap[r >> 5];
    const uint32_t bit = 1u << (r & 31);
    if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(wp, bit);      // (a plain test load: same time)
}

__global__ void __launch_bounds__(256)
k_sh_mark(const int64_t *__restrict__ rows, const int64_t *__restrict__ seg_off /* pinned host memory */, int64_t *__restrict__ seg_off_dev,
          int64_t n_snp, uint32_t *__restrict__ bitmap, long long *__restrict__ meta)
{
    // the sample offsets are read where the host left them (two 8-byte reads over the bus per block) and stored for the kernels that
    // follow: an upload of their own was one more operation in front of the first kernel of the call
    const int64_t s = blockIdx.y;
    const int64_t r0 = seg_off[s], r1 = seg_off[s + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        seg_off_dev[s] = r0;
        if (s + 1 == (int64_t)gridDim.y) seg_off_dev[s + 1] = r1;
    }
    int bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = r0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < r1; i += 4 * stride) {
        int64_t rw[4], rp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rw[u] = rows[i + u * stride];
            rp[u] = (i + u * stride > r0) ? rows[i + u * stride - 1] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sh_mark_row(rw[u], rp[u], i + u * stride > r0, n_snp, bitmap, bad);
    }
    for (; i < r1; i += stride) sh_mark_row(rows[i], i > r0 ? rows[i - 1] : -1, i > r0, n_snp, bitmap, bad);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicOr((unsigned long long *)&meta[1], (unsigned long long)bad);
}

// sh_eseg_of: the reference-order error bound of a sample in closed form.  The contraction takes weights in [0, 1] only, so
// wmax_r <= 1 (and = 1 on every row whose genotype likelihoods are normalised: min PL = 0) and the sum k_eseg_part forms from the
// weights is at most  sum_k len_k (len_k + 3 + K - k)  (chunk k of K, len_k rows) -- no pass over the weights.  A sample whose
// weights are all integers (`nonint` unset by k_sh_expand) has bound 0: any summation order is exact.  + the conversions of
// k_sh_finish, which evaluates this per (sample, accession) -- some thirty flops -- instead of reading it from a kernel of its own.
__device__ __forceinline__ double sh_eseg_of(int64_t len, int64_t chunk, int nonint)
{
    if (!nonint || len == 0) return 0.0;
    const int64_t K = (len + chunk - 1) / chunk;
    const int64_t last = len - (K - 1) * chunk;                  // rows of the last chunk
    // sum over the K - 1 full chunks k = 0 .. K-2 of chunk (chunk + 3 + K - k), + the last one
    const double kf = (double)(K - 1);
    const double full = (double)chunk * (kf * (double)(chunk + 3) + (kf * (double)K - kf * (kf - 1.0) * 0.5));
    const double acc = full + (double)last * (double)(last + 3 + 1);
    const double u = 1.1102230246251565e-16;
    const double mmax = (double)((chunk < len ? chunk : len) + 3 + K);
    const double mf = 4.0;
    return (acc * u / (1.0 - mmax * u) + (double)len * (mf * u / (1.0 - mf * u))) * 1.0000001;
}

// ---------------------------------------------------------------------------------------------------------------
// k_sh_probe: a cheap look at the batch before the full pass (automatic policy): block = sample; the entries whose rows lie below
// `row_limit` (a prefix of a sorted row list: binary search) are marked and counted.  calls / (distinct rows x samples) of that
// slice of the panel estimates the density of the whole batch; a batch on scattered marker sets is declined after ~40 us
// instead of after the full pass over its rows and weights.  meta[3] += entries marked.
__global__ void __launch_bounds__(256)
k_sh_probe(const int64_t *__restrict__ rows, const int64_t *__restrict__ seg_off, int64_t row_limit, int64_t n_snp,
           uint32_t *__restrict__ bitmap, long long *__restrict__ meta)
{
    __shared__ long long s_len;
    const int64_t s = blockIdx.x;
    const int64_t r0 = seg_off[s], r1 = seg_off[s + 1];
    if (threadIdx.x == 0) {
        int64_t lo = r0, hi = r1;                   // first entry with row >= row_limit
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (rows[mid] < row_limit) lo = mid + 1; else hi = mid;
        }
        s_len = lo - r0;
        if (lo > r0) atomicAdd((unsigned long long *)&meta[3], (unsigned long long)(lo - r0));
    }
    __syncthreads();
    const int64_t len = s_len;
    for (int64_t i = r0 + threadIdx.x; i < r0 + len; i += 256) {
        const int64_t r = rows[i];
        if (r < 0 || r >= n_snp || r >= row_limit) continue;      // an unsorted or invalid list: the full pass reports it
        uint32_t *wp = &bitmap[r >> 5];
        const uint32_t bit = 1u << (r & 31);
        if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(wp, bit);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// rank of the marked rows: a block owns 4096 bitmap words (16 consecutive words per thread).
//   k_sh_count   popcount of a block's words -> block_sum[b]
//   k_sh_scan    ONE block: exclusive prefix of block_sum -> block_base, total -> meta[0]; zero rows behind the union
//   k_sh_fill    word_base[w] = marked rows before word w; union_rows[rank] = row for every marked row
constexpr int SH_WORDS_PER_THREAD = 16;
constexpr int SH_WORDS_PER_BLOCK = 256 * SH_WORDS_PER_THREAD;

__global__ void __launch_bounds__(256)
k_sh_count(const uint32_t *__restrict__ bitmap, int64_t n_words, uint32_t *__restrict__ block_sum)
{
    __shared__ uint32_t sm[4];
    const int64_t w0 = (int64_t)blockIdx.x * SH_WORDS_PER_BLOCK + (int64_t)threadIdx.x * SH_WORDS_PER_THREAD;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < SH_WORDS_PER_THREAD; ++i)
        if (w0 + i < n_words) c += __popc(bitmap[w0 + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void __launch_bounds__(1024)
k_sh_scan(const uint32_t *__restrict__ block_sum, int64_t n_blocks, uint32_t *__restrict__ block_base, long long *__restrict__ meta,
          int32_t *__restrict__ union_rows /* or null */, int pad_entries, const int *__restrict__ other_codes /* or null */,
          int *__restrict__ pair_count /* cleared here, or null */)
{
    if (threadIdx.x == 0 && pair_count) *pair_count = 0;
    __shared__ uint32_t sm[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const int64_t b = b0 + threadIdx.x;
        const uint32_t v = b < n_blocks ? block_sum[b] : 0;
        uint32_t x = v;                                  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) sm[wave] = x;
        __syncthreads();
        uint32_t before = carry;
        for (int i = 0; i < wave; ++i) before += sm[i];
        if (b < n_blocks) block_base[b] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        meta[0] = (long long)carry;
        meta[2] = other_codes ? (long long)*other_codes : 0;        // "the int8 panel holds codes besides -1, 0, 1, 2": read with the total
    }
    // row 0 in the entries behind the union (k_sh_fill writes [0, total)): the steps that pad it and the contraction's read-ahead
    if (union_rows)
        for (int i = threadIdx.x; i < pad_entries; i += 1024) union_rows[(int64_t)carry + i] = 0;
}

__global__ void __launch_bounds__(256)
k_sh_fill(const uint32_t *__restrict__ bitmap, int64_t n_words, const uint32_t *__restrict__ block_base,
          uint32_t *__restrict__ word_base, int32_t *__restrict__ union_rows)
{
    __shared__ uint32_t sm[4];
    const int64_t w0 = (int64_t)blockIdx.x * SH_WORDS_PER_BLOCK + (int64_t)threadIdx.x * SH_WORDS_PER_THREAD;
    uint32_t word[SH_WORDS_PER_THREAD];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < SH_WORDS_PER_THREAD; ++i) {
        word[i] = (w0 + i < n_words) ? bitmap[w0 + i] : 0;
        c += __popc(word[i]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) sm[wave] = x;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x] + x - c;
    for (int i = 0; i < wave; ++i) base += sm[i];
#pragma unroll
    for (int i = 0; i < SH_WORDS_PER_THREAD; ++i) {
        if (w0 + i >= n_words) break;
        word_base[w0 + i] = base;
        uint32_t m = word[i];
        while (m) {
            const int b = __ffs(m) - 1;
            m &= m - 1;
            union_rows[base++] = (int32_t)((w0 + i) * 32 + b);
        }
    }
}

// position of every (sample, row) entry in the union: pos[s_local, u] = entry index + 1 (0 = the sample has no call at union
// row u).  blockIdx.y = sample of this pass (s_base + blockIdx.y).  Every element of the row [0, ld_pos) is written here, none by
// a memset (51 MB for 64 samples): a sample's rows ascend, so the 256 entries of a chunk own the stretch of the row from the
// chunk's first rank to the next chunk's first rank (the first chunk from 0, the last to ld_pos) -- zeroed by the block, then
// the entries stored on top.
__global__ void __launch_bounds__(256)
k_sh_pos(const int64_t *__restrict__ rows, const int64_t *__restrict__ seg_off, int64_t s_base, const uint32_t *__restrict__ bitmap,
         const uint32_t *__restrict__ word_base, uint32_t *__restrict__ pos, int64_t ld_pos)
{
    __shared__ int64_t range[2];
    const int64_t s = s_base + blockIdx.y;
    const int64_t r0 = seg_off[s], r1 = seg_off[s + 1];
    uint32_t *row = pos + (int64_t)blockIdx.y * ld_pos;
    if (r1 <= r0) {                                 // a sample without calls: all zeros
        for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < ld_pos; u += (int64_t)gridDim.x * 256) row[u] = 0;
        return;
    }
    for (int64_t c0 = r0 + (int64_t)blockIdx.x * 256; c0 < r1; c0 += (int64_t)gridDim.x * 256) {
        const int64_t i = c0 + threadIdx.x;
        int64_t u = -1;
        if (i < r1) {
            const int64_t r = rows[i];
            u = word_base[r >> 5] + __popc(bitmap[r >> 5] & ((1u << (r & 31)) - 1u));
        }
        if (threadIdx.x == 0) range[0] = c0 == r0 ? 0 : u;
        if (threadIdx.x == 1) {
            const int64_t nx = c0 + 256;
            if (nx < r1) {
                const int64_t r = rows[nx];
                range[1] = word_base[r >> 5] + __popc(bitmap[r >> 5] & ((1u << (r & 31)) - 1u));
            } else {
                range[1] = ld_pos;
            }
        }
        __syncthreads();
        const int64_t u0 = range[0], u1 = range[1];
        for (int64_t v = u0 + threadIdx.x; v < u1; v += 256) row[v] = 0;
        __syncthreads();                            // (also: range[] is rewritten in the next round)
        if (i < r1) row[u] = (uint32_t)(i + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_sh_expand: the A matrix in fragment order.  Block = (sample s of the pass, run of 128 K steps); a thread converts the sample's
// weights at ONE union row at a time to fixed point and splits them into balanced digits; sixteen rows of one class make the 16 B
// of a digit row (and of the count row) that lane (m, h) of tile t will load, assembled through LDS and stored in A's order.
//   matrix row of (sample s, digit j): M = s * RPS + j, RPS = DIGITS + 1 (samples follow each other without gaps: a sample may
//   lie across two groups of 128 rows); group M >> 7, tile (M >> 5) & 3, lane row M & 31
//   balanced digits: Q' = floor(w 2^F) + sum_{p < DIGITS-1} 128 * 256^p; digit at position p < DIGITS-1 = byte p of Q' - 128
//   (stored byte = byte ^ 0x80), top digit = Q' >> 8 (DIGITS - 1) (0 .. 65)
//   class order: 0 (ref) = W[:, 0], 1 (alt) = W[:, 2], 2 (het) = W[:, 1] (0 when skip_hets); the count row holds 1 in every class
//   that is an informative call (not in het when skip_hets): its product is ninfo.
template <int DIGITS>
__global__ void __launch_bounds__(256)
k_sh_expand(const uint32_t *__restrict__ pos, int64_t ld_pos, const double *__restrict__ w, int64_t n_samples_pass, int skip_hets,
            int64_t step0, int64_t n_steps, int64_t steps_ld, sh_v4i *__restrict__ A, int *__restrict__ nonint /* of the pass's samples */,
            long long *__restrict__ meta)
{
    // K steps [step0, step0 + n_steps) of the pass (a launch per part; both multiples of 8 = two rounds).
    // blockIdx.x = sample, blockIdx.y walks runs of 128 K steps = 32 rounds = 1024 union rows.
    //   phase 1  a thread converts ONE union row at a time (consecutive lanes = consecutive rows: positions 4 B and weights 24 B per
    //            lane, a wave's loads cover 12 cache lines each) and leaves one dword per digit row -- the bytes (ref, alt, het, 0) --
    //            in LDS, digit row major;
    //   phase 2  a thread takes the four dwords of four consecutive rows of one digit row (one ds_read_b128), transposes them into one
    //            dword per CLASS (rows 4q .. 4q+3 of that class: a quarter of a 16-B fragment piece) and stores the three where lane
    //            (m, h) of tile t will load them: the six digit rows of a sample are 96 contiguous bytes of every class block.
    // (The first version gave a thread the four rows of a piece: every load and store instruction of a wave touched 48 - 64 lines,
    // and the address path, not HBM, set its 0.22 ms.)
    constexpr int RPS = DIGITS + 1;
    __shared__ uint32_t stage[RPS * 1024];
    const int64_t s = blockIdx.x;
    const int64_t rounds_ld = steps_ld >> 2;
    bool out_of_range = false, fractional = false;
    constexpr int frac_bits = 8 * (DIGITS - 1) + 6;
    const double scale = __builtin_ldexp(1.0, frac_bits);
    unsigned long long bias = 0;
#pragma unroll
    for (int p = 0; p < DIGITS - 1; ++p) bias |= 128ull << (8 * p);
    for (int64_t kb = blockIdx.y; kb * 128 < n_steps; kb += gridDim.y) {
        const int64_t steps_here = (n_steps - kb * 128 < 128) ? n_steps - kb * 128 : 128;
        // every position and weight of the thread's four rows is requested before the first one is used (rows the sample has no
        // call at re-read entry 0: harmless, their digits are cleared below)
        uint32_t e[4];
        double wv[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ul = i * 256 + (int)threadIdx.x;                         // union row of the run
            e[i] = (ul >> 3) < steps_here ? pos[s * ld_pos + (step0 + kb * 128) * SH_STEP_ROWS + ul] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double *wr = w + 3 * (int64_t)(e[i] ? e[i] - 1 : 0);
            wv[i][0] = wr[0]; wv[i][1] = wr[2]; wv[i][2] = wr[1];              // class order: ref, alt, het
        }
        // the weights are vetted where they are read: outside [0, 1] (or not finite) -> the whole batch goes back to the per-sample
        // pass (meta[1] bit 1, read by the host with the results); a weight that is not an integer -> the sample's reference-order
        // bound is not zero
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ul = i * 256 + (int)threadIdx.x;
            uint32_t lo[3], hi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double x = (c == 2 && skip_hets) ? 0.0 : wv[i][c];
                if (e[i]) {
                    if (!(x >= 0.0 && x <= 1.0)) out_of_range = true;
                    else if (x != 0.0 && x != 1.0) fractional = true;
                }
                x = (x >= 0.0 && x <= 1.0) ? x : 0.0;                          // keep the conversion defined
                const unsigned long long Q = (unsigned long long)floor(x * scale) + bias;
                lo[c] = (uint32_t)Q;
                hi[c] = (uint32_t)(Q >> 32);
            }
#pragma unroll
            for (int j = 0; j < DIGITS; ++j) {
                const int p = DIGITS - 1 - j;                                  // byte position of digit j (j = 0: the top digit)
                const uint32_t *src = (p < 4) ? lo : hi;
                const uint32_t b = (uint32_t)(p & 3);
                // (ref.b, alt.b, 0, 0) then (.., .., het.b, 0)
                const uint32_t t = __builtin_amdgcn_perm(src[1], src[0], 0x0c0c0000u | ((4u + b) << 8) | b);
                uint32_t d = __builtin_amdgcn_perm(src[2], t, 0x0c000100u | ((4u + b) << 16));
                if (j > 0) d ^= 0x00808080u;
                stage[j * 1024 + ul] = e[i] ? d : 0u;
            }
            // the count row: 1 in every class that is an informative call (het is not when the DB's hets are skipped)
            stage[DIGITS * 1024 + ul] = e[i] ? (skip_hets ? 0x00000101u : 0x00010101u) : 0u;
        }
        __syncthreads();
        const int rounds_here = (int)(steps_here >> 2);
        const int64_t round0 = (step0 + kb * 128) >> 2;
#pragma unroll
        for (int it = 0; it < RPS; ++it) {
            const int item = it * 256 + (int)threadIdx.x;                      // (((round, half), digit row), quarter)
            const int q = item & 3;
            const int j = (item >> 2) % RPS;
            const int rh = (item >> 2) / RPS;
            const int h = rh & 1, rd = rh >> 1;
            if (rd < rounds_here) {
                const sh_v4i d = *reinterpret_cast<const sh_v4i *>(stage + j * 1024 + rd * 32 + h * 16 + q * 4);
                const uint32_t lo01 = __builtin_amdgcn_perm((uint32_t)d.y, (uint32_t)d.x, 0x05010400u);       // (x.b0, y.b0, x.b1, y.b1)
                const uint32_t hi01 = __builtin_amdgcn_perm((uint32_t)d.w, (uint32_t)d.z, 0x05010400u);
                const uint32_t lo2 = __builtin_amdgcn_perm((uint32_t)d.y, (uint32_t)d.x, 0x0c0c0602u);        // (x.b2, y.b2, 0, 0)
                const uint32_t hi2 = __builtin_amdgcn_perm((uint32_t)d.w, (uint32_t)d.z, 0x0c0c0602u);
                const uint32_t o0 = __builtin_amdgcn_perm(hi01, lo01, 0x05040100u);                           // class 0 of rows 4q .. 4q+3
                const uint32_t o1 = __builtin_amdgcn_perm(hi01, lo01, 0x07060302u);
                const uint32_t o2 = __builtin_amdgcn_perm(hi2, lo2, 0x05040100u);
                const int64_t M = s * RPS + j;
                const int64_t g = M >> 7;
                const int t = (int)((M >> 5) & 3), m = (int)(M & 31);
                uint32_t *dst = reinterpret_cast<uint32_t *>(A + (((g * rounds_ld + round0 + rd) * 3) * 4 + t) * 64 + h * 32 + m) + q;
                dst[0] = o0;
                dst[4 * 64 * 4] = o1;                                          // the next class block: 4 tiles x 64 lanes x 4 dwords
                dst[2 * 4 * 64 * 4] = o2;
            }
        }
        __syncthreads();                                                       // the next run refills the stage
    }
    // one flag update per block (the sample is block-uniform)
    const int any_bad = __syncthreads_or(out_of_range ? 1 : 0);
    const int any_frac = __syncthreads_or(fractional ? 1 : 0);
    if (threadIdx.x == 0) {
        if (any_bad) atomicOr((unsigned long long *)&meta[1], 2ull);
        if (any_frac && !__hip_atomic_load(&nonint[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&nonint[s], 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// B fragments.  A lane holds 16 rows of a round (rows 16h .. 16h+15), one dword per row with the calls of its four accessions
// (int8 panel: the bytes themselves, -1 = 0xff; packed panel: its byte of 2-bit fields spread to four bytes).  Only bits 0 and 1 of a
// byte matter: 0 ref, 1 alt, 2 het, 3 missing.  sh_transpose turns them into V[j][q] = the codes of accession j at rows 4q .. 4q+3;
// the fragment of class C for accession j is then the four dwords sh_ind<C>(V[j][q]): bytes 1 where the call is C -- one 3-input
// boolean per dword (t = v >> 1 brings bit 1 of every byte to bit 0).
template <int C>
__device__ __forceinline__ uint32_t sh_ind(uint32_t v, uint32_t t)
{
    return C == 0 ? (~(v | t) & 0x01010101u) : (C == 1 ? (v & ~t & 0x01010101u) : (~v & t & 0x01010101u));
}

template <bool PACKED>
__device__ __forceinline__ void sh_transpose(const uint32_t (&x)[16], uint32_t byte_shift, uint32_t (&V)[4][4], uint32_t (&T)[4][4])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (PACKED) {
                const uint32_t xb = __builtin_amdgcn_ubfe(x[4 * q + i], byte_shift, 8u);      // this lane's byte of the dword
                const uint32_t t1 = xb | (xb << 6);                                             // field k at bits 8k .. 8k+1 (junk elsewhere)
                y[i] = t1 | (t1 << 12);
            } else {
                y[i] = x[4 * q + i];
            }
        }
        const uint32_t p0 = __builtin_amdgcn_perm(y[1], y[0], 0x05010400u);   // (y0.b0, y1.b0, y0.b1, y1.b1)
        const uint32_t p1 = __builtin_amdgcn_perm(y[1], y[0], 0x07030602u);   // (y0.b2, y1.b2, y0.b3, y1.b3)
        const uint32_t p2 = __builtin_amdgcn_perm(y[3], y[2], 0x05010400u);
        const uint32_t p3 = __builtin_amdgcn_perm(y[3], y[2], 0x07030602u);
        V[0][q] = __builtin_amdgcn_perm(p2, p0, 0x05040100u);                 // accession 0: rows 4q .. 4q+3
        V[1][q] = __builtin_amdgcn_perm(p2, p0, 0x07060302u);
        V[2][q] = __builtin_amdgcn_perm(p3, p1, 0x05040100u);
        V[3][q] = __builtin_amdgcn_perm(p3, p1, 0x07060302u);
#pragma unroll
        for (int j = 0; j < 4; ++j) T[j][q] = V[j][q] >> 1;
    }
}

// k_sh_mfma: see the head of this file.  partial [n_tiles, ldn, n_groups * 128] int32 (accession-major), ldn = n_accgroups * 128: every
// element is written by exactly one wave.  steps_per_tile K steps per row tile (a multiple of 8); union_rows carries SH_PAD_STEPS * 8
// valid entries past the last step and A SH_PAD_STEPS steps past the last group.
template <bool PACKED>
__global__ void __launch_bounds__(256, 1)
k_sh_mfma(const int8_t *__restrict__ db, int64_t pitch, int64_t desc, const int32_t *__restrict__ union_rows,
          const sh_v4i *__restrict__ A, int64_t n_steps, int64_t steps_ld, int64_t steps_per_tile, int tile0, int n_tiles, int n_groups,
          int n_accgroups, int blocks_per_tile, int aligned_tiles, int fill_per_xcd, int *__restrict__ partial, int64_t ldn)
{
    // blocks of one row tile on ONE XCD: workgroups are dealt round-robin over the 8 XCDs in launch order.  This launch owns the
    // row tiles [tile0, n_tiles): the first `aligned_tiles` (a multiple of 8) are dealt one per XCD in turn; the blocks of the
    // remaining FILLER tiles take the compute units the aligned tiles leave idle -- `fill_per_xcd` per XCD, so a filler tile lies on
    // two or three XCDs (the host plans the fillers so that every CU holds exactly one block: 27 wave tiles x 32 aligned row tiles
    // fill 864 of 1024 wave slots, four filler tiles the rest).
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int aligned_blocks = (aligned_tiles >> 3) * blocks_per_tile;          // per XCD
    int tile, wb;
    if (q8 < aligned_blocks) {
        tile = tile0 + (q8 / blocks_per_tile) * 8 + xcd;
        wb = q8 % blocks_per_tile;
    } else {
        const int ex = xcd * fill_per_xcd + (q8 - aligned_blocks);
        tile = tile0 + aligned_tiles + ex / blocks_per_tile;
        wb = ex % blocks_per_tile;
    }
    if (tile >= n_tiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wt = wb * 4 + wave;
    if (wt >= n_groups * n_accgroups) return;
    const int g = wt / n_accgroups, ng = wt % n_accgroups;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    const int64_t k0 = (int64_t)tile * steps_per_tile;
    int64_t k1 = k0 + steps_per_tile;
    if (k1 > n_steps) k1 = n_steps;
    const int64_t R0 = k0 >> 2, R1 = k1 >> 2;                   // rounds of this tile: an even number, never none

    // byte offset of this lane's accessions inside a row (and the matrix it lies in, for split packed panels)
    int64_t col_off, row_stride, mat_off = 0;
    if (PACKED) {
        const int64_t b = (int64_t)ng * 32 + c;
        const int64_t tp = pk_tail_pitch(desc);
        if (tp && b >= pitch) {                 // the ragged tail of a split panel: its own matrix and pitch
            int64_t bb = b - pitch;
            if (bb > tp - 1) bb = tp - 1;       // lanes past the tail re-read its last byte (accessions >= n_acc: never reported)
            col_off = bb; row_stride = tp; mat_off = pk_tail_off(desc);
        } else {
            col_off = (!tp && b > pitch - 1) ? pitch - 1 : b;
            row_stride = pitch;
        }
    } else {
        col_off = (int64_t)ng * 128 + 4 * c;
        if (col_off > pitch - 4) col_off = pitch - 4;
        row_stride = pitch;
    }
    // packed panels: the lane's byte of a row is read as the aligned dword around it and cut out when it is used (every packed pitch
    // and matrix offset is a multiple of 4).  A byte load's value travels round the loop as an 8-bit quantity, and the compiler
    // widened ALL of a round's at the loop's head: one wait for the youngest load per round, the prefetch depth gone.
    const uint32_t byte_shift = PACKED ? (uint32_t)(col_off & 3) * 8u : 0u;
    const int8_t *base = db + mat_off + (PACKED ? (col_off & ~(int64_t)3) : col_off);
    const sh_v4i *Ag = A + (int64_t)g * (steps_ld >> 2) * 768 + lane;          // a round of a group: 3 x 4 x 64 fragments of 16 B

    sh_v16i acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0;

    sh_v4i a[3][4];                          // A fragments of the round being contracted, one set per class: reloaded for the next round
                                             // as soon as their sixteen MFMAs are issued
    uint32_t xb[2][16];                      // panel dwords: round r lives in xb[r & 1], requested two rounds ahead
    int32_t rn[16];                          // union rows of the round whose panel dwords are requested next
    uint32_t V[4][4], T[4][4];               // codes of the current round by accession (and shifted by one bit)
    sh_v4i bf[2][4];                         // B fragments of a class: built for the next class while this one's MFMAs run

    // row address = base + row * stride: rows are non-negative 32-bit numbers and a row stride is below 4 GiB, so ONE v_mad_u64_u32
    const uint32_t stride32 = (uint32_t)row_stride;
#define SH_LOAD_RN(R)                                                                         \
    do {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
            const sh_v4i rr = *reinterpret_cast<const sh_v4i *>(union_rows + (R) * SH_ROUND_ROWS + 16 * h + 4 * i); \
            rn[4 * i] = rr.x; rn[4 * i + 1] = rr.y; rn[4 * i + 2] = rr.z; rn[4 * i + 3] = rr.w; \
        }                                                                                     \
    } while (0)
#define SH_LOAD_X(BUF)                                                                        \
    do {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 16; ++i)                                        \
            xb[BUF][i] = *reinterpret_cast<const uint32_t *>(base + (uint64_t)(uint32_t)rn[i] * (uint64_t)stride32); \
    } while (0)
#define SH_LOAD_A(C, R)                                                                       \
    do {                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) a[C][t] = Ag[(((R) * 3 + (C)) * 4 + t) * 64]; \
    } while (0)
#define SH_IND(C, BUF)                                                                        \
    do {                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
            bf[BUF][j].x = (int)sh_ind<C>(V[j][0], T[j][0]); bf[BUF][j].y = (int)sh_ind<C>(V[j][1], T[j][1]); \
            bf[BUF][j].z = (int)sh_ind<C>(V[j][2], T[j][2]); bf[BUF][j].w = (int)sh_ind<C>(V[j][3], T[j][3]); \
        }                                                                                     \
    } while (0)
#define SH_MFMA(C, BUF)                                                                       \
    do {                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                acc[t][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[C][t], bf[BUF][j], acc[t][j], 0, 0, 0); \
    } while (0)
    // sixteen MFMAs with NV vector instructions and NL loads between two of them; the fence keeps what belongs to a phase inside it
    // (without it the scheduler sinks the loads to their uses a round later and waits with vmcnt(0))
#define SH_PHASE_END(NV, NL)                                                                  \
    do {                                                                                      \
        _Pragma("unroll") for (int g16 = 0; g16 < 16; ++g16) {                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                               \
            __builtin_amdgcn_sched_group_barrier(0x020, NL, 0);                               \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
    // One round = three phases of sixteen MFMAs (K = 32 rows of one class).  P / Q: the fragment buffers (class 0 of the round is in
    // P when it starts and class 0 of the next round in Q when it ends: the roles swap every round, hence two rounds per loop body);
    // XL: the buffer the panel dwords of round R + 2 go to (= the one round R came from), XT: the one holding round R + 1.
#define SH_ROUND(R, P, Q, XL, XT)                                                             \
    do {                                                                                      \
        SH_IND(1, Q);                                                                         \
        SH_MFMA(0, P);                                                                        \
        SH_LOAD_X(XL);                      /* rn holds the rows of round R + 2 */            \
        SH_LOAD_RN((R) + 3);                                                                  \
        SH_LOAD_A(0, (R) + 1);                                                                \
        SH_PHASE_END(2, 2);                                                                   \
        SH_IND(2, P);                                                                         \
        SH_MFMA(1, Q);                                                                        \
        SH_LOAD_A(1, (R) + 1);                                                                \
        SH_PHASE_END(2, 1);                                                                   \
        sh_transpose<PACKED>(xb[XT], byte_shift, V, T);                                       \
        SH_IND(0, Q);                                                                         \
        SH_MFMA(2, P);                                                                        \
        SH_LOAD_A(2, (R) + 1);                                                                \
        SH_PHASE_END(7, 1);                                                                   \
    } while (0)

    // prologue: the panel dwords of the first two rounds, the rows of the third, the A fragments of the first; everything is waited
    // for here (the wait the compiler places at the loop head is the merge of "entered from the prologue" and "came round the
    // loop": a full drain in EVERY iteration unless the first path arrives with nothing pending)
    SH_LOAD_RN(R0);
    SH_LOAD_X(0);
    SH_LOAD_RN(R0 + 1);
    SH_LOAD_X(1);
    SH_LOAD_RN(R0 + 2);
    SH_LOAD_A(0, R0);
    SH_LOAD_A(1, R0);
    SH_LOAD_A(2, R0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0) expcnt(7) lgkmcnt(15)
    sh_transpose<PACKED>(xb[0], byte_shift, V, T);
    SH_IND(0, 0);
    int64_t R = R0;
    do {                                    // a tile is never empty and holds an even number of rounds
        SH_ROUND(R, 0, 1, 0, 1);
        SH_ROUND(R + 1, 1, 0, 1, 0);
        R += 2;
    } while (R < R1);
#undef SH_ROUND
#undef SH_PHASE_END
#undef SH_MFMA
#undef SH_IND
#undef SH_LOAD_A
#undef SH_LOAD_X
#undef SH_LOAD_RN

    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): registers 4 q .. 4 q + 3 of a
    // tile are four adjacent matrix rows of one accession.  The partial sums are kept accession-major ([.., accession, matrix
    // row]) so that those four registers leave as ONE 16-B store straight from the accumulators (no repacking, no copies).
    // Column c of accession tile j is accession ng * 128 + 4 c + j.
    const int64_t ldm = (int64_t)n_groups * SH_GROUP_ROWS;
    int *out = partial + ((int64_t)tile * ldn + (int64_t)ng * SH_WAVE_ACCS + 4 * c) * ldm + (int64_t)g * SH_GROUP_ROWS + 4 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                sh_v4i v;
                v.x = acc[t][j][4 * q4]; v.y = acc[t][j][4 * q4 + 1]; v.z = acc[t][j][4 * q4 + 2]; v.w = acc[t][j][4 * q4 + 3];
                *reinterpret_cast<sh_v4i *>(out + (int64_t)j * ldm + 32 * t + 8 * q4) = v;
            }
}

// ---------------------------------------------------------------------------------------------------------------
// k_sh_finish: block = accession a, four waves; lane = sample of the pass (the digit sums of one sample are adjacent in the
// accession-major partial array, samples follow each other), wave w adds the row tiles w, w + 4, ... (int64), the waves' sums meet
// in LDS.  Then the digits are put together and converted to fp64 (hi / lo parts: each conversion is exact or rounds once far
// below the bound), ninfo = the count row (informative calls), and the certificate runs: the reference's score lies in
// [v - E, v + E + Eq], E = eseg[s] (reference-order bound + the conversion), Eq = len[s] 2^-F (the quantisation, one-sided).
__global__ void __launch_bounds__(256)
k_sh_finish(const int *__restrict__ partial, int n_tiles, int n_groups, int64_t ldn, int digits, const int64_t *__restrict__ seg_off,
            int64_t s_base, int64_t n_samples_pass, int64_t n_acc, int64_t chunk, const int *__restrict__ nonint /* of the batch's samples; null: no certificate */,
            int force_first, double *__restrict__ score, int64_t *__restrict__ ninfo, int64_t ldo, int32_t *__restrict__ pairs,
            int *__restrict__ count, int cap)
{
    __shared__ long long sm[3][64][SH_MAX_RPS];
    const int64_t a = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t sp = (int64_t)blockIdx.y * 64 + lane;
    const bool on = sp < n_samples_pass;
    const int64_t sg = s_base + sp;
    const int rps = digits + 1;
    const int64_t ldm = (int64_t)n_groups * SH_GROUP_ROWS;
    long long dsum[SH_MAX_RPS];
#pragma unroll
    for (int j = 0; j < SH_MAX_RPS; ++j) dsum[j] = 0;
    if (on) {
        for (int t = wave; t < n_tiles; t += 4) {
            const int *p = partial + ((int64_t)t * ldn + a) * ldm + sp * rps;
#pragma unroll
            for (int j = 0; j < SH_MAX_RPS; ++j)
                if (j < rps) dsum[j] += p[j];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < SH_MAX_RPS; ++j) sm[wave - 1][lane][j] = dsum[j];
    }
    __syncthreads();
    if (wave > 0 || !on) return;
#pragma unroll
    for (int j = 0; j < SH_MAX_RPS; ++j) dsum[j] += sm[0][lane][j] + sm[1][lane][j] + sm[2][lane][j];
    // digits 0 .. 2 form the high part (units of 256^(digits-3) 2^-F), the others the low part (units of 2^-F)
    long long hi = 0, lo = 0;
#pragma unroll
    for (int j = 0; j < SH_MAX_RPS - 1; ++j) {
        if (j >= digits) break;
        if (j < 3) hi = hi * 256 + dsum[j];
        else lo = lo * 256 + dsum[j];
    }
    const int frac_bits = sh_frac_bits(digits);
    const int n_hi = digits < 3 ? digits : 3;
    const double v = __builtin_ldexp((double)hi, 8 * (digits - n_hi) - frac_bits) + __builtin_ldexp((double)lo, -frac_bits);
    const int64_t len = seg_off[sg + 1] - seg_off[sg];
    long long informative = 0;
#pragma unroll
    for (int j = 0; j < SH_MAX_RPS; ++j)
        if (j == digits) informative = dsum[j];
    score[sg * ldo + a] = v;
    ninfo[sg * ldo + a] = informative;
    if (nonint) {
        const double E = sh_eseg_of(len, chunk, nonint[sg]) + 8.0 * 1.1102230246251565e-16 * (fabs(v) + 1.0);
        // quantisation, one-sided: every matched SNP may lose up to 2^-F (counted for all of them: an upward-only slack cannot
        // flag an exact-integer score, floor(v) = floor(v + Eq) as long as Eq < 1)
        const double Eq = __builtin_ldexp((double)len, -frac_bits) * 1.0000001;
        const double lo_v = v - E, hi_v = (v + Eq) + E;
        if (!(lo_v >= 0.0) || floor(lo_v) != floor(hi_v) || a < force_first) {
            const int k = atomicAdd(count, 1);
            if (k < cap) {
                pairs[2 * k] = (int32_t)sg;
                pairs[2 * k + 1] = (int32_t)a;
            }
        }
    }
}

}  // namespace snpm

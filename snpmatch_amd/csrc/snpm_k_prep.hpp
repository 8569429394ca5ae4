// snpm_k_prep.hpp -- what runs once per sample before the scoring: LUT rows, weight properties, weight bits, the reference-order error bound, row-list checks, dictionary-coded weights.
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
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
// snpm_genotype_once: everything between "the sample lies in the staging slab" and the fast pass in ONE launch (it was eight:
// k_check_rows, k_expand_codes, two fills, k_build_lut, k_eref, k_efinish, the fill of the flag count -- 4.8 us each on a
// path whose scoring kernel takes 53 us, profiles/r04_once_timeline_before.txt).  The slab is read where it lies: src_rows /
// src_wei may be the PINNED HOST slab itself (no copy engine in the path: a copy costs ~25 us before it moves a byte) or a
// device staging copy of it.
//   block b walks the chunks k = b, b + grid, ... of `chunk` rows, as k_eref does -- same rows per thread, same order of
//   additions, so the bound is the one k_eref / k_efinish leave: per row it checks the row index (outside the panel: row 0,
//   bad |= 1), expands the weight codes (CODED: wei = table[code], a code outside the table: bad |= 2) and writes rows [n]
//   int64, w [n, 3], lut [n, 4] (k_build_lut's entries); weights go through LDS in sub-tiles of ONCE_SUB rows so that every
//   source byte crosses the bus once, in full dwords.  Block 0 also clears the prefetch pad behind the row list and the
//   flag count.  The LAST block to finish (ticket counter) adds the partials up in k_efinish's order and resets the ticket.
constexpr int ONCE_SUB = 1024;
constexpr int64_t ONCE_MAX_CHUNK = 4096;        // longer chunks (few blocks, long walks) take the unfused kernels

template <bool CODED>
__global__ void __launch_bounds__(256)
k_once_prep(const void *src_rows, const void *src_wei, const double *__restrict__ table, int table_len,
            int64_t n, int64_t n_snp, int64_t chunk, int skip_hets, int64_t *rows, double *w,      // fp64 samples: src_rows == rows, src_wei == w (in place)
            double *__restrict__ lut, double *__restrict__ partial, double *__restrict__ eref, int *__restrict__ cert_count,
            unsigned *__restrict__ state /* [0] ticket, [1] bad */, int pad_rows)
{
    __shared__ double sm[4];
    __shared__ __attribute__((aligned(16))) double s_w[CODED ? (ONCE_SUB * 3 / 4 + 2) : ONCE_SUB * 3];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int64_t K = (n + chunk - 1) / chunk;
    if (blockIdx.x == 0) {
        for (int i = tid; i < pad_rows; i += 256) rows[n + i] = 0;
        if (tid == 0) *cert_count = 0;
    }
    int bad = 0;
    double acc = 0.0;                                     // meaningful in thread 0
    for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
        const int64_t r0 = k * chunk, r1 = (r0 + chunk < n) ? r0 + chunk : n;
        double s = 0.0;
        for (int64_t t0 = r0; t0 < r1; t0 += ONCE_SUB) {
            const int64_t t1 = (t0 + ONCE_SUB < r1) ? t0 + ONCE_SUB : r1;
            __syncthreads();                              // the previous sub-tile has been consumed
            // every load of the sub-tile is issued before the first one is waited for: over the bus a load takes ~2 us, and
            // one at a time (row index, then staged word after staged word) they made the kernel latency-bound
            constexpr int RPT = ONCE_SUB / 256;           // rows per thread and sub-tile
            int64_t prow_in[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int64_t r = t0 + tid + 256 * j;
                prow_in[j] = r < t1 ? (CODED ? (int64_t)reinterpret_cast<const int32_t *>(src_rows)[r] : reinterpret_cast<const int64_t *>(src_rows)[r]) : 0;
            }
            int shift = 0;                                // CODED: first code of the sub-tile inside its first staged dword (0 or 1)
            if constexpr (CODED) {
                const int64_t b0 = 6 * t0, b1 = 6 * t1, a0 = b0 & ~(int64_t)3;
                const int n_dw = (int)((b1 - a0 + 3) / 4);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(src_wei) + a0);
                uint32_t *dst = reinterpret_cast<uint32_t *>(s_w);
                constexpr int WPT = (ONCE_SUB * 6 / 4 + 1 + 255) / 256;        // staged dwords per thread
                uint32_t v[WPT];
#pragma unroll
                for (int j = 0; j < WPT; ++j) v[j] = (tid + 256 * j < n_dw) ? src[tid + 256 * j] : 0u;
#pragma unroll
                for (int j = 0; j < WPT; ++j)
                    if (tid + 256 * j < n_dw) dst[tid + 256 * j] = v[j];
                shift = (int)((b0 - a0) / 2);
            } else {
                const double *src = reinterpret_cast<const double *>(src_wei) + 3 * t0;
                const int n_el = (int)(3 * (t1 - t0));
                constexpr int WPT = ONCE_SUB * 3 / 256;
                double v[WPT];
#pragma unroll
                for (int j = 0; j < WPT; ++j) v[j] = (tid + 256 * j < n_el) ? src[tid + 256 * j] : 0.0;
#pragma unroll
                for (int j = 0; j < WPT; ++j)
                    if (tid + 256 * j < n_el) s_w[tid + 256 * j] = v[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int64_t r = t0 + tid + 256 * j;
                if (r >= t1) break;
                int64_t prow = prow_in[j];
                if (prow < 0 || prow >= n_snp) { prow = 0; bad |= 1; }
                double w0, w1, w2;
                const int l = (int)(r - t0);
                if constexpr (CODED) {
                    const uint16_t *c = reinterpret_cast<const uint16_t *>(s_w) + shift + 3 * l;
                    int c0 = c[0], c1 = c[1], c2 = c[2];
                    if (c0 >= table_len || c1 >= table_len || c2 >= table_len) {
                        bad |= 2;
                        c0 = c0 < table_len ? c0 : 0; c1 = c1 < table_len ? c1 : 0; c2 = c2 < table_len ? c2 : 0;
                    }
                    w0 = table[c0]; w1 = table[c1]; w2 = table[c2];
                } else {
                    w0 = s_w[3 * l]; w1 = s_w[3 * l + 1]; w2 = s_w[3 * l + 2];
                }
                rows[r] = prow;
                w[3 * r] = w0; w[3 * r + 1] = w1; w[3 * r + 2] = w2;
                double4 e;
                e.x = w0; e.y = w2; e.z = skip_hets ? 0.0 : w1; e.w = 0.0;
                reinterpret_cast<double4 *>(lut)[r] = e;
                s += fmax(fabs(w0), fmax(fabs(w1), fabs(w2)));
            }
        }
        s = block_sum_256(s, sm);
        acc += s * (double)((r1 - r0) + 3 + (K - k));
    }
    if (bad) atomicOr(&state[1], (unsigned)bad);
    if (tid == 0) {
        partial[blockIdx.x] = acc;
        __threadfence();
        s_last = (atomicAdd(&state[0], 1u) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double v = 0.0;
    for (int i = tid; i < (int)gridDim.x; i += 256) v += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = block_sum_256(v, sm);
    if (tid == 0) {
        const double u = 1.1102230246251565e-16;
        const double mmax = (double)(chunk + 3 + K);
        eref[0] = (v * u / (1.0 - mmax * u)) * 1.0000001;
        state[0] = 0u;
    }
}

}  // namespace snpm

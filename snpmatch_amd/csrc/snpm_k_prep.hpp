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

}  // namespace snpm

// snpm_k_io.hpp -- panel rows in and out (repitch / pack / unpack), the counter-based synthetic panel and sample, the PMC calibration read.
// One of the kernel-family headers behind snpm_kernels.hpp (include that one: the families share constants and helpers in this order).
#pragma once

namespace snpm {
// ------------------------------------------------------------------------------------------------
// packed panel upload: int8 rows (staging slab on the device, row stride src_pitch) -> 2 bits per call.
// One thread per output byte.  Codes outside {-1 (any negative), 0, 1, 2} cannot be encoded: *bad |= 1.
__global__ void k_pack_rows(const int8_t *__restrict__ src, int64_t src_pitch, int64_t nrows, int64_t n_acc,
                            uint8_t *__restrict__ db, int64_t pitch, int64_t desc, int64_t row0, int *__restrict__ bad)
{
    // db / pitch / desc: the packed panel (row-major or split, snpm_k_common.hpp); rows row0 .. row0 + nrows - 1 are written
    const int64_t row_bytes = pitch + pk_tail_pitch(desc);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * row_bytes) return;
    const int64_t r = i / row_bytes, b = i - r * row_bytes;
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
    db[pk_off(pitch, desc, row0 + r, b)] = (uint8_t)out;
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
__global__ void k_unpack_rows(const uint8_t *__restrict__ db, int64_t pitch, int64_t desc, int64_t row0, int64_t nrows, int64_t n_acc,
                              int8_t *__restrict__ dst, int64_t dst_pitch)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * n_acc) return;
    const int64_t r = i / n_acc, a = i - r * n_acc;
    const int v = (db[pk_off(pitch, desc, row0 + r, a >> 2)] >> (2 * (int)(a & 3))) & 3;
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
k_synth_packed(uint8_t *__restrict__ db, int64_t pitch, int64_t desc, int64_t row0, int64_t n_snp, int64_t n_acc, uint64_t seed,
               int64_t snp0, int64_t acc0)
{
    // rows row0 .. row0 + n_snp - 1 of the packed panel (db, pitch, desc) <- SNPs snp0 ..; one thread per row byte
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= pitch + pk_tail_pitch(desc)) return;
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
        db[pk_off(pitch, desc, row0 + row, q)] = (uint8_t)out;
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

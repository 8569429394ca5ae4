// Read-only streaming rates on one GPU for a few load shapes (what bounds k_fast: DESIGN.md section 4).
//   build: hipcc -O3 --offload-arch=gfx950 -o read_patterns tools/micro/read_patterns.hip ; run: ./read_patterns [GiB]
// Every kernel reads the same buffer once, grid-stride, `U` loads of `V` bytes per lane in flight; the XOR keeps them live.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ inline unsigned fold(const T &v);
template <> __device__ inline unsigned fold<unsigned>(const unsigned &v) { return v; }
template <> __device__ inline unsigned fold<u32x2>(const u32x2 &v) { return v.x ^ v.y; }
template <> __device__ inline unsigned fold<u32x4>(const u32x4 &v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename T, int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const T *__restrict__ p, long n, unsigned *out)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= fold(v[u]);
    }
    for (; i < n; i += stride) acc ^= fold(p[i]);
    if (acc == 0x9e3779b9u) out[0] = acc;
}

// k_fast's shape: a block of W waves owns W*256 contiguous bytes of every row of a tile; rows are `pitch` apart
template <int U, bool NT>
__global__ void __launch_bounds__(1024) k_rows(const unsigned *__restrict__ p, long pitch_dw, long nrows, long tile_rows, unsigned *out)
{
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;     // dword column
    if (col >= pitch_dw) return;
    const long P = gridDim.y;
    unsigned acc = 0;
    static_assert(128 % U == 0, "a group must not run past its tile");
    const long ntiles = nrows / tile_rows;
    for (long t = blockIdx.y; t < ntiles; t += P) {
        const unsigned *base = p + t * tile_rows * pitch_dw + col;
        for (long r = 0; r < tile_rows; r += U) {
            unsigned v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(base + (r + u) * pitch_dw) : base[(r + u) * pitch_dw];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if (acc == 0x9e3779b9u) out[0] = acc;
}

// the same with k_fast's software pipeline: the next group of G rows is requested before the current one is consumed
template <int G, bool NT>
__global__ void __launch_bounds__(512) k_rows_db(const unsigned *__restrict__ p, long pitch_dw, long nrows, long tile_rows, unsigned *out)
{
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= pitch_dw) return;
    const long P = gridDim.y;
    unsigned acc = 0;
    const long ntiles = nrows / tile_rows;
    auto ld = [&](const unsigned *q) -> unsigned { return NT ? __builtin_nontemporal_load(q) : *q; };
    for (long t = blockIdx.y; t < ntiles; t += P) {
        const unsigned *base = p + t * tile_rows * pitch_dw + col;
        unsigned a[G], b[G];
#pragma unroll
        for (int u = 0; u < G; ++u) a[u] = ld(base + u * pitch_dw);
        for (long r = 0; r < tile_rows; r += 2 * G) {
#pragma unroll
            for (int u = 0; u < G; ++u) b[u] = ld(base + (r + G + u) * pitch_dw);
#pragma unroll
            for (int u = 0; u < G; ++u) acc ^= a[u];
            const long rn = (r + 2 * G < tile_rows) ? r + 2 * G : 0;
#pragma unroll
            for (int u = 0; u < G; ++u) a[u] = ld(base + (rn + u) * pitch_dw);
#pragma unroll
            for (int u = 0; u < G; ++u) acc ^= b[u];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) acc ^= a[u];
    }
    if (acc == 0x9e3779b9u) out[0] = acc;
}

template <typename F> static double time_ms(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    std::vector<double> ts;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[0];
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 48.0;
    const long bytes = (long)(gib * (1 << 30)) / 10240 * 10240;
    void *buf; unsigned *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cu = prop.multiProcessorCount;
    printf("%s, %d CUs, buffer %.1f GB\n", prop.name, cu, bytes / 1e9);
#define RUN(T, U, NT, BPC)                                                                                     \
    do {                                                                                                       \
        const long n = bytes / (long)sizeof(T);                                                                \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_read<T, U, NT>), dim3(cu * BPC), dim3(256), 0, 0, (const T *)buf, n, out); }); \
        printf("grid-stride  %2zu B/lane  %d in flight  %s  %2d blocks/CU : %7.3f ms  %6.0f GB/s\n", sizeof(T), U,    \
               NT ? "nt" : "  ", BPC, ms, bytes / ms / 1e6);                                                   \
    } while (0)
    RUN(unsigned, 4, true, 8);  RUN(unsigned, 8, true, 8);  RUN(unsigned, 4, false, 8); RUN(unsigned, 8, true, 4);
    RUN(u32x2, 4, true, 8);     RUN(u32x4, 4, true, 8);     RUN(u32x4, 2, true, 8);     RUN(u32x4, 4, false, 8);
    RUN(u32x4, 4, true, 4);     RUN(u32x4, 8, true, 2);     RUN(unsigned, 16, true, 4);
    // k_fast's shape on a 10 240-byte pitch: W waves per block, P parts
    const long pitch_dw = 2560, nrows = bytes / 10240 / 128 * 128;
#define ROWS(U, NT, W, OCC)                                                                                    \
    do {                                                                                                       \
        const int colblocks = (int)((pitch_dw + 64 * W - 1) / (64 * W));                                       \
        const int parts = cu * OCC / colblocks;                                                                \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows<U, NT>), dim3(colblocks, parts), dim3(64 * W), 0, 0, (const unsigned *)buf, pitch_dw, nrows, 128L, out); }); \
        printf("row tiles    %d-wave blocks  %d rows in flight  %s  %d blocks/CU : %7.3f ms  %6.0f GB/s\n", W, U,   \
               NT ? "nt" : "  ", OCC, ms, nrows * 10240.0 / ms / 1e6);                                         \
    } while (0)
    ROWS(4, true, 8, 3); ROWS(8, true, 8, 3); ROWS(8, true, 8, 4); ROWS(8, true, 4, 6); ROWS(8, true, 5, 4); ROWS(16, true, 8, 3);
    ROWS(8, false, 8, 3);
    ROWS(2, true, 8, 3); ROWS(4, true, 8, 2); ROWS(8, true, 8, 2); ROWS(4, true, 4, 6);   // U must divide the 128-row tile
    ROWS(4, true, 4, 4); ROWS(4, true, 5, 3); ROWS(4, true, 5, 4); ROWS(2, true, 8, 4);
#define ROWS_DB(G, NT, W, OCC)                                                                                 \
    do {                                                                                                       \
        const int colblocks = (int)((pitch_dw + 64 * W - 1) / (64 * W));                                       \
        const int parts = cu * OCC / colblocks;                                                                \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_db<G, NT>), dim3(colblocks, parts), dim3(64 * W), 0, 0, (const unsigned *)buf, pitch_dw, nrows, 128L, out); }); \
        printf("row tiles, pipelined  %d-wave blocks  groups of %d (<= %d in flight)  %s  %d blocks/CU : %7.3f ms  %6.0f GB/s\n", W, G, 2 * G, \
               NT ? "nt" : "  ", OCC, ms, nrows * 10240.0 / ms / 1e6);                                         \
    } while (0)
    ROWS(4, true, 16, 1); ROWS(4, true, 16, 2); ROWS(8, true, 16, 1); ROWS(4, true, 10, 2); ROWS(4, true, 10, 3); ROWS(2, true, 16, 2);
#define ROWS_DBT(G, W, OCC, TR)                                                                                \
    do {                                                                                                       \
        const int colblocks = (int)((pitch_dw + 64 * W - 1) / (64 * W));                                       \
        const int parts = cu * OCC / colblocks;                                                                \
        const long nr = nrows / TR * TR;                                                                       \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_db<G, true>), dim3(colblocks, parts), dim3(64 * W), 0, 0, (const unsigned *)buf, pitch_dw, nr, (long)TR, out); }); \
        printf("row tiles, pipelined  %d-wave blocks  groups of %d  %4d-row tiles  %d blocks/CU : %7.3f ms  %6.0f GB/s\n", W, G, TR, OCC, ms, \
               nr * 10240.0 / ms / 1e6);                                                                       \
    } while (0)
    ROWS_DBT(4, 8, 3, 64); ROWS_DBT(4, 8, 3, 128); ROWS_DBT(4, 8, 3, 256); ROWS_DBT(4, 8, 3, 512); ROWS_DBT(4, 8, 3, 1024); ROWS_DBT(4, 5, 4, 512);
    ROWS_DB(4, true, 8, 3); ROWS_DB(4, true, 8, 2); ROWS_DB(2, true, 8, 3); ROWS_DB(2, true, 8, 4); ROWS_DB(1, true, 8, 3); ROWS_DB(4, true, 4, 4);
    // the packed panel's rows: 2560 B (10 000 accessions x 2 bits), 64-row tiles, W-wave blocks
    const long ppitch_dw = 640, pnrows = bytes / 2560 / 1024 * 1024;
#define PROWS(U, W, OCC)                                                                                       \
    do {                                                                                                       \
        const int colblocks = (int)((ppitch_dw + 64 * W - 1) / (64 * W));                                      \
        const int parts = cu * OCC / colblocks;                                                                \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows<U, true>), dim3(colblocks, parts), dim3(64 * W), 0, 0, (const unsigned *)buf, ppitch_dw, pnrows, 128L, out); }); \
        printf("packed rows (2560 B)  %2d-wave blocks  %2d rows in flight  nt  %2d blocks/CU : %7.3f ms  %6.0f GB/s\n", W, U, OCC, ms, \
               pnrows * 2560.0 / ms / 1e6);                                                                    \
    } while (0)
    PROWS(8, 1, 24); PROWS(16, 1, 24); PROWS(8, 1, 16); PROWS(4, 1, 24); PROWS(8, 2, 12); PROWS(16, 2, 12); PROWS(8, 5, 4); PROWS(8, 5, 5);
    PROWS(8, 10, 2); PROWS(16, 10, 2); PROWS(4, 10, 3); PROWS(8, 10, 3); PROWS(16, 1, 16); PROWS(16, 1, 32);
    // k_fast_bits' pipeline: groups of 8 rows, two groups in flight, TR-row tiles, one-wave blocks
#define PROWS_DB(G, TR, OCC)                                                                                   \
    do {                                                                                                       \
        const int colblocks = (int)((ppitch_dw + 63) / 64);                                                    \
        const int parts = cu * OCC / colblocks;                                                                \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_db<G, true>), dim3(colblocks, parts), dim3(64), 0, 0, (const unsigned *)buf, ppitch_dw, pnrows, (long)TR, out); }); \
        printf("packed rows, pipelined  1-wave blocks  groups of %d  %3d-row tiles  %2d blocks/CU : %7.3f ms  %6.0f GB/s\n", G, TR, OCC, ms, \
               pnrows * 2560.0 / ms / 1e6);                                                                    \
    } while (0)
    PROWS_DB(8, 64, 24); PROWS_DB(8, 128, 24); PROWS_DB(4, 64, 24); PROWS_DB(8, 64, 16); PROWS_DB(4, 128, 24); PROWS_DB(8, 64, 32);
    PROWS_DB(4, 256, 24); PROWS_DB(4, 512, 24); PROWS_DB(4, 128, 28); PROWS_DB(4, 128, 16); PROWS_DB(2, 128, 24); PROWS_DB(4, 1024, 24);
    return 0;
}

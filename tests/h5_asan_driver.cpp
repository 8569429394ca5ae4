// Driver for the sanitizer build of the HDF5 reader (tests/test_host_sanitizers_cpu.py): every file named on the command line is
// opened, its root group listed, every dataset read whole and row-wise, every attribute read.  Errors are expected on damaged
// files; faults, leaks and undefined behaviour are what the sanitizers look for.  Prints "<file> rc=<open rc> objects=<n> bytes=<n>".
#include "snpmatch_hip.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static long long visit(snpm_h5 *f, const std::string &path, int depth, long long *objects)
{
    long long bytes = 0;
    int kind = 0, rank = 0, tc = 0, es = 0, sg = 0, na = 0;
    int64_t dims[8], chunk[8];
    if (snpm_h5_info(f, path.c_str(), nullptr, &kind, &rank, dims, &tc, &es, &sg, chunk, &na) != SNPM_OK) return 0;
    ++*objects;
    for (int i = 0; i < na; ++i) {
        char name[256];
        if (snpm_h5_attr_name(f, path.c_str(), i, name, sizeof(name)) != SNPM_OK) continue;
        int ar = 0, aes = 0;
        int64_t ad[8];
        if (snpm_h5_info(f, path.c_str(), name, nullptr, &ar, ad, nullptr, &aes, nullptr, nullptr, nullptr) != SNPM_OK) continue;
        long long n = 1;
        for (int d = 0; d < ar; ++d) n *= ad[d];
        if (n < 0 || n * aes > (1ll << 26)) continue;
        std::vector<char> buf((size_t)(n * aes) + 1);
        if (snpm_h5_read(f, path.c_str(), name, buf.data(), n * aes) == SNPM_OK) bytes += n * aes;
    }
    if (kind == 0) {
        if (depth > 4) return bytes;
        int64_t need = 0;
        if (snpm_h5_list(f, path.c_str(), nullptr, 0, &need) != SNPM_OK || need > (1 << 20)) return bytes;
        std::vector<char> names((size_t)need + 1);
        if (snpm_h5_list(f, path.c_str(), names.data(), need, nullptr) != SNPM_OK) return bytes;
        char *save = nullptr;
        for (char *tok = strtok_r(names.data(), "\n", &save); tok; tok = strtok_r(nullptr, "\n", &save))
            bytes += visit(f, path.empty() ? std::string(tok) : path + "/" + tok, depth + 1, objects);
        return bytes;
    }
    long long n = 1;
    for (int d = 0; d < rank; ++d) n *= dims[d];
    if (n < 0 || es <= 0 || n * es > (1ll << 28)) return bytes;
    std::vector<char> buf((size_t)(n * es) + 1);
    if (snpm_h5_read(f, path.c_str(), nullptr, buf.data(), n * es) == SNPM_OK) bytes += n * es;
    if (rank == 2 && tc != 3 && dims[0] > 3 && dims[1] > 1) {
        const int64_t rows[3] = {dims[0] - 1, 0, dims[0] / 2};
        std::vector<char> part((size_t)(3 * (dims[1] - 1) * es) + 1);
        if (snpm_h5_read_rows(f, path.c_str(), rows, 0, 3, 1, dims[1] - 1, part.data(), (dims[1] - 1) * es) == SNPM_OK)
            bytes += 3 * (dims[1] - 1) * es;
    }
    return bytes;
}

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        snpm_h5 *f = nullptr;
        const int rc = snpm_h5_open(argv[i], &f);
        long long objects = 0, bytes = 0;
        if (rc == SNPM_OK) {
            bytes = visit(f, "", 0, &objects);
            snpm_h5_close(f);
        }
        const char *base = strrchr(argv[i], '/');
        printf("%s rc=%d objects=%lld bytes=%lld\n", base ? base + 1 : argv[i], rc, objects, bytes);
    }
    printf("done\n");
    return 0;
}

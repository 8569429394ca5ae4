// Driver for the pure-host sources of libsnpmatch_hip.so (snpm_vcf.cpp, snpm_host.cpp), built by
// tests/test_host_sanitizers_cpu.py with -fsanitize=address,undefined: any out-of-bounds access, leak or undefined
// behaviour on the inputs below ends the run with a non-zero status.  Prints a few summary numbers that the test compares
// with the regular (unsanitised) library.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "snpmatch_hip.h"

static std::string tmpdir;

static std::string write_file(const char *name, const std::string &text, bool gz = false)
{
    const std::string path = tmpdir + "/" + name;
    if (gz) {
        gzFile f = gzopen(path.c_str(), "wb");
        gzwrite(f, text.data(), (unsigned)text.size());
        gzclose(f);
    } else {
        FILE *f = fopen(path.c_str(), "wb");
        fwrite(text.data(), 1, text.size(), f);
        fclose(f);
    }
    return path;
}

static int parse_and_fill(const std::string &path, int sample, int64_t *n_out = nullptr)
{
    snpm_vcf *v = nullptr;
    const int rc = snpm_vcf_parse(path.c_str(), sample, &v);
    if (rc != SNPM_OK) return rc;
    int64_t n = 0;
    int cw = 0, gw = 0, flags = 0, ns = 0;
    snpm_vcf_dims(v, &n, &cw, &gw, &flags, &ns);
    std::vector<char> chr((size_t)n * cw + 1), gt((size_t)n * gw + 1);
    std::vector<int64_t> pos((size_t)n + 1), dp((size_t)n + 1);
    std::vector<double> pl((size_t)n * 3 + 1);
    snpm_vcf_fill(v, chr.data(), pos.data(), gt.data(), pl.data(), dp.data());
    for (int i = -1; i <= ns; ++i) (void)snpm_vcf_sample_name(v, i);
    if (n_out) *n_out = n;
    snpm_vcf_free(v);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string vcf = argv[1];
    tmpdir = argv[2];
    // (1) the reference's sample VCF
    int64_t n = 0;
    int rc = parse_and_fill(vcf, 0, &n);
    printf("sample_vcf rc=%d records=%lld\n", rc, (long long)n);
    printf("sample_vcf_second_column rc=%d\n", parse_and_fill(vcf, 1));
    // (2) hostile inputs: every one must come back with a status, never touch memory it does not own
    const std::string head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\n";
    std::vector<std::pair<const char *, std::string>> cases = {
        {"empty", ""},
        {"header_only", head},
        {"no_newline_at_end", head + "Chr1\t10\t.\tA\tT\t50\tPASS\tDP=7\tGT:PL\t0/1:10,0,20"},
        {"crlf", head + "Chr1\t10\t.\tA\tT\t50\tPASS\tDP=7\tGT:PL\t0/1:10,0,20\r\n"},
        {"short_records", head + "Chr1\t10\n\nChr1\n\t\t\t\nChr1\t11\t.\tA\tT\t.\t.\t.\n"},
        {"bad_pos", head + "Chr1\tten\t.\tA\tT\t50\tPASS\tDP=7\tGT\t0/1\n"},
        {"bad_dp", head + "Chr1\t10\t.\tA\tT\t50\tPASS\tDP=x7;AF=1\tGT\t0/1\n"},
        {"bad_pl", head + "Chr1\t10\t.\tA\tT\t50\tPASS\t.\tGT:PL\t0/1:10,zz,20\n"},
        {"pl_short", head + "Chr1\t10\t.\tA\tT\t50\tPASS\t.\tGT:PL\t0/1:10\nChr1\t11\t.\tA\tT\t50\tPASS\t.\tGT:PL\t0/1:.\nChr1\t12\t.\tA\tT\t50\tPASS\t.\tGT:PL\t.:.,.,.\n"},
        {"more_keys_than_values", head + "Chr1\t10\t.\tA\tT\t50\tPASS\t.\tGT:AD:DP:GQ:PL\t0/1\n"},
        {"huge_gt", head + "Chr1\t10\t.\tA\tT\t50\tPASS\t.\tGT\t" + std::string(100000, '1') + "\n"},
        {"huge_chrom", head + std::string(100000, 'C') + "\t10\t.\tA\tT\t50\tPASS\t.\tGT\t0/1\n"},
        {"many_columns", head + "Chr1\t10\t.\tA\tT\t50\tPASS\t.\tGT" + std::string(20000, '\t') + "0/1\n"},
        {"many_format_keys", head + "Chr1\t10\t.\tA\tT\t50\tPASS\t.\t" + std::string(500, ':') + "\t0/1\n"},
        {"binary_garbage", std::string("\x00\x01\xff\xfe\n\t\t\t\t\t\t\t\t\t\n#CHROM\n\x80\x80", 24)},
        {"long_number", head + "Chr1\t123456789012345678901234567890\t.\tA\tT\t.\t.\tDP=99999999999999999999\tGT\t0/1\n"},
    };
    for (auto &c : cases) {
        const std::string path = write_file(c.first, c.second);
        int64_t k = -1;
        const int r = parse_and_fill(path, 0, &k);
        printf("case %s rc=%d records=%lld\n", c.first, r, (long long)k);
        (void)parse_and_fill(path, 3);                              // a sample column that does not exist
        (void)parse_and_fill(write_file((std::string(c.first) + ".gz").c_str(), c.second, true), 0);
    }
    {   // a file several read blocks long (lines straddle the 4 MiB blocks of the streaming reader), plain and gzip
        std::string big = head;
        for (int i = 0; i < 150000; ++i) {
            char line[160];
            snprintf(line, sizeof(line), "Chr%d\t%d\t.\tA\tT\t50\tPASS\tDP=%d;AF=0.5\tGT:AD:PL\t%s:3,4:%d,0,%d\n", 1 + i % 5, 100 + 7 * i,
                     i % 90, (i % 3) ? "0/1" : "1/1", 10 + i % 200, 20 + i % 100);
            big += line;
        }
        for (int gz = 0; gz < 2; ++gz) {
            const std::string path = write_file(gz ? "big.vcf.gz" : "big.vcf", big, gz != 0);
            snpm_vcf *bv = nullptr;
            const int r = snpm_vcf_parse(path.c_str(), 0, &bv);
            int64_t k = 0;
            int cw = 0, gw = 0, fl = 0, ns = 0;
            long long possum = 0, dpsum = 0;
            double plsum = 0;
            if (r == SNPM_OK) {
                snpm_vcf_dims(bv, &k, &cw, &gw, &fl, &ns);
                std::vector<char> chr((size_t)k * cw + 1), gt((size_t)k * gw + 1);
                std::vector<int64_t> pos((size_t)k + 1), dp((size_t)k + 1);
                std::vector<double> pl((size_t)k * 3 + 1);
                snpm_vcf_fill(bv, chr.data(), pos.data(), gt.data(), pl.data(), dp.data());
                for (int64_t i = 0; i < k; ++i) { possum += pos[(size_t)i]; dpsum += dp[(size_t)i]; plsum += pl[(size_t)i * 3] + pl[(size_t)i * 3 + 2]; }
                snpm_vcf_free(bv);
            }
            printf("big_%s rc=%d records=%lld bytes=%zu possum=%lld dpsum=%lld plsum=%.0f\n", gz ? "gz" : "plain", r, (long long)k,
                   big.size(), possum, dpsum, plsum);
        }
    }
    snpm_vcf *v = nullptr;
    printf("missing_file rc=%d\n", snpm_vcf_parse((tmpdir + "/does_not_exist.vcf").c_str(), 0, &v));
    printf("bad_args rc=%d %d\n", snpm_vcf_parse(nullptr, 0, &v), snpm_vcf_parse(vcf.c_str(), -1, &v));
    snpm_vcf_free(nullptr);

    // (3) sorted-merge / galloping intersection against the obvious quadratic answer
    std::mt19937_64 rng(7);
    long long checked = 0;
    for (int round = 0; round < 400; ++round) {
        const int na = (int)(rng() % 60), nb = (int)(rng() % 40);
        std::vector<int64_t> a, b;
        int64_t x = (int64_t)(rng() % 5) - 2;
        for (int i = 0; i < na; ++i) { x += 1 + (int64_t)(rng() % 4); a.push_back(x); }
        x = (int64_t)(rng() % 5) - 2;
        for (int i = 0; i < nb; ++i) { x += 1 + (int64_t)(rng() % 5); b.push_back(x); }
        std::vector<int64_t> ia((size_t)std::min(na, nb) + 1), ib(ia.size()), ja(ia.size()), jb(ia.size());
        int64_t k1 = -1, k2 = -1;
        const int r1 = snpm_intersect_sorted(a.data(), na, b.data(), nb, ia.data(), ib.data(), &k1);
        const int r2 = snpm_intersect_sorted_search(a.data(), na, b.data(), nb, ja.data(), jb.data(), &k2);
        std::vector<std::pair<int64_t, int64_t>> want;
        for (int i = 0; i < na; ++i)
            for (int j = 0; j < nb; ++j)
                if (a[(size_t)i] == b[(size_t)j]) want.push_back({i, j});
        bool ok = r1 == SNPM_OK && r2 == SNPM_OK && k1 == (int64_t)want.size() && k2 == k1;
        for (int64_t t = 0; ok && t < k1; ++t)
            ok = ia[(size_t)t] == want[(size_t)t].first && ib[(size_t)t] == want[(size_t)t].second &&
                 ja[(size_t)t] == ia[(size_t)t] && jb[(size_t)t] == ib[(size_t)t];
        if (!ok) { printf("intersect mismatch in round %d\n", round); return 1; }
        ++checked;
    }
    // the pooled form of the galloping search (nb >= 4096: ranges of b on the pool's threads), buffers of EXACTLY min(na, nb)
    // entries -- the documented capacity -- including a DB list shorter than the list searched for
    long long pooled = 0;
    const int shapes[][2] = {{10, 20000}, {300000, 20000}, {20000, 20000}, {19999, 20000}, {20001, 20000}, {5000, 9000}, {0, 8192}, {70000, 4096}};
    for (const auto &sh : shapes) {
        const int na = sh[0], nb = sh[1];
        std::vector<int64_t> a, b;
        int64_t x = 0;
        for (int i = 0; i < na; ++i) { x += 1 + (int64_t)(rng() % 3); a.push_back(x); }
        x = 0;
        for (int i = 0; i < nb; ++i) { x += 1 + (int64_t)(rng() % (na > 10 * nb ? 30 : 3)); b.push_back(x); }
        const size_t cap = (size_t)std::min(na, nb);
        std::vector<int64_t> ia(cap), ib(cap), ja(cap), jb(cap);
        int64_t k1 = -1, k2 = -1;
        const int r1 = snpm_intersect_sorted(a.data(), na, b.data(), nb, ia.data(), ib.data(), &k1);
        const int r2 = snpm_intersect_sorted_search(a.data(), na, b.data(), nb, ja.data(), jb.data(), &k2);
        bool ok = r1 == SNPM_OK && r2 == SNPM_OK && k1 == k2 && k1 <= (int64_t)cap;
        for (int64_t t = 0; ok && t < k1; ++t) ok = ia[(size_t)t] == ja[(size_t)t] && ib[(size_t)t] == jb[(size_t)t];
        if (!ok) { printf("pooled intersect mismatch na=%d nb=%d (%d %d, %lld %lld)\n", na, nb, r1, r2, (long long)k1, (long long)k2); return 1; }
        ++pooled;
    }
    printf("intersect_pooled shapes=%lld\n", pooled);
    const int64_t dup[] = {1, 2, 2, 3}, inc[] = {1, 2, 3};
    int64_t o1[4], o2[4], k = 0;
    printf("intersect rounds=%lld not_increasing rc=%d %d %d empty rc=%d\n", checked,
           snpm_intersect_sorted(dup, 4, inc, 3, o1, o2, &k), snpm_intersect_sorted(inc, 3, dup, 4, o1, o2, &k),
           snpm_intersect_sorted_search(inc, 3, dup, 4, o1, o2, &k), snpm_intersect_sorted(nullptr, 0, nullptr, 0, nullptr, nullptr, &k));
    printf("done\n");
    return 0;
}

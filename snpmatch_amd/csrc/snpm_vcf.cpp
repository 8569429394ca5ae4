// Native reader for single-sample SNP-call VCF text (plain or gzip): host-side input of the scoring path.
// The reference delegates this step to scikit-allel (core/parsers.py:178-213, a C extension); here it is a
// single pass over the file in C++.  Semantics follow snpmatch_amd/core/_vcf.py (the Python reader, kept as
// the generic path): per record CHROM, POS, the sample's GT text as written (a bare '.' becomes './.'), the
// first three PL values (-1 where absent) and INFO/DP (-1 where absent).  Anything this reader is not sure
// about (malformed numbers, over-long fields) makes it decline with SNPM_ERR_STATE and the caller parses the
// file with the Python reader instead.
#include <zlib.h>

#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "snpmatch_hip.h"

struct snpm_vcf {
    std::vector<std::string> sample_names;
    std::string chr_text, gt_text;            // concatenated fields
    std::vector<uint32_t> chr_off, gt_off;    // n + 1 offsets each
    std::vector<int64_t> pos, dp;
    std::vector<double> pl;                   // n * 3
    size_t chr_width = 1, gt_width = 1;
    bool any_gt = false, any_pl = false, any_dp = false;
};

namespace {

struct Field {
    const char *p;
    size_t n;
};

// split [s, e) at `sep` into at most `cap` fields; returns the number found (the last one takes the rest)
inline int split(const char *s, const char *e, char sep, Field *out, int cap)
{
    int k = 0;
    const char *a = s;
    while (k < cap - 1) {
        const char *b = (const char *)memchr(a, sep, (size_t)(e - a));
        if (!b) break;
        out[k++] = {a, (size_t)(b - a)};
        a = b + 1;
    }
    out[k++] = {a, (size_t)(e - a)};
    return k;
}

inline bool parse_int(const char *p, size_t n, int64_t *v)
{
    // Python's int(): optional sign, decimal digits, surrounding whitespace allowed; we accept plain digits
    // with an optional sign and decline everything else
    if (n == 0 || n > 18) return false;
    size_t i = 0;
    bool neg = false;
    if (p[0] == '-' || p[0] == '+') { neg = p[0] == '-'; i = 1; }
    if (i == n) return false;
    int64_t x = 0;
    for (; i < n; ++i) {
        if (p[i] < '0' || p[i] > '9') return false;
        x = x * 10 + (p[i] - '0');
    }
    *v = neg ? -x : x;
    return true;
}

inline bool parse_double(const char *p, size_t n, double *v)
{
    if (n == 0 || n > 40) return false;
    int64_t iv;
    if (n <= 15 && parse_int(p, n, &iv)) { *v = (double)iv; return true; }     // the common case: integer phred values
    char buf[48];
    memcpy(buf, p, n);
    buf[n] = 0;
    for (size_t i = 0; i < n; ++i)      // digits, sign, '.', exponent only: float() and strtod agree on these
        if (!((buf[i] >= '0' && buf[i] <= '9') || buf[i] == '.' || buf[i] == '-' || buf[i] == '+' || buf[i] == 'e' || buf[i] == 'E'))
            return false;
    char *end = nullptr;
    errno = 0;
    const double x = strtod(buf, &end);
    if (end != buf + n || errno != 0) return false;
    *v = x;
    return true;
}

}  // namespace

extern "C" {

// No C++ exception leaves this function (ctypes would turn it into std::terminate): running out of host memory
// on a very large file is reported as SNPM_ERR_STATE, i.e. "use the generic reader", like any other declined file.
int snpm_vcf_parse(const char *path, int sample_index, snpm_vcf **out)
try {
    if (!path || !out || sample_index < 0 || sample_index > 4000) return SNPM_ERR_BADARG;
    std::unique_ptr<snpm_vcf> v(new snpm_vcf());
    v->chr_off.push_back(0);
    v->gt_off.push_back(0);
    constexpr int MAXF = 4096;
    std::vector<Field> f(MAXF);
    Field keys[64], vals[64], nums[4];
    bool ok = true;
    // every line of [s, end): the file is consumed block by block (4 MiB of decompressed text at a time, the records
    // kept are a few dozen bytes each), never held in memory as a whole
    auto consume = [&](const char *s, const char *end) {
    while (s < end && ok) {
        const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
        const char *e = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (e > s && e[-1] == '\r') { ok = false; break; }       // CRLF files: leave to the generic reader
        if (e == s) { s = next; continue; }                        // an empty line is a short record: skipped
        if (*s == '#') {
            if ((size_t)(e - s) >= 6 && memcmp(s, "#CHROM", 6) == 0) {
                const int nf = split(s, e, '\t', f.data(), MAXF);
                if (nf == MAXF) { ok = false; break; }
                v->sample_names.clear();
                for (int i = 9; i < nf; ++i) v->sample_names.emplace_back(f[i].p, f[i].n);
            }
            s = next;
            continue;
        }
        const int want = 10 + sample_index;
        const int nf = split(s, e, '\t', f.data(), want + 1);
        s = next;
        if (nf < 8) continue;
        int64_t pos;
        if (!parse_int(f[1].p, f[1].n, &pos)) { ok = false; break; }
        // INFO/DP: first item starting with "DP="
        int64_t dp = -1;
        if (!(f[7].n == 1 && f[7].p[0] == '.')) {
            const char *a = f[7].p, *ie = f[7].p + f[7].n;
            while (a <= ie) {
                const char *b = (const char *)memchr(a, ';', (size_t)(ie - a));
                const char *ee = b ? b : ie;
                if (ee - a >= 3 && a[0] == 'D' && a[1] == 'P' && a[2] == '=') {
                    int64_t d;
                    if (parse_int(a + 3, (size_t)(ee - a - 3), &d)) { dp = d; v->any_dp = true; }
                    else if (ee - a > 3) ok = false;      // int() may still accept it (spaces, underscores): not ours to decide
                    break;
                }
                if (!b) break;
                a = b + 1;
            }
            if (!ok) break;
        }
        std::string gt = "./.";
        double pl[3] = {-1.0, -1.0, -1.0};
        bool has_pl = false;
        if (nf > 8) {
            const int nk = split(f[8].p, f[8].p + f[8].n, ':', keys, 64);
            if (nk == 64) { ok = false; break; }
            for (int k = 0; k < nk; ++k)
                if (keys[k].n == 2 && keys[k].p[0] == 'G' && keys[k].p[1] == 'T') v->any_gt = true;
            if (nf > 9 + sample_index) {
                const Field &col = f[9 + sample_index];       // exact: only field `want` can hold the rest of the line
                const int nv = split(col.p, col.p + col.n, ':', vals, 64);
                if (nv == 64) { ok = false; break; }
                const int m = nk < nv ? nk : nv;
                for (int k = 0; k < m && ok; ++k) {
                    const Field &key = keys[k], &val = vals[k];
                    if (key.n == 2 && key.p[0] == 'G' && key.p[1] == 'T') {
                        gt = (val.n == 1 && val.p[0] == '.') ? std::string("./.") : std::string(val.p, val.n);
                    } else if (key.n == 2 && key.p[0] == 'P' && key.p[1] == 'L' && !(val.n == 1 && val.p[0] == '.')) {
                        const int nn = split(val.p, val.p + val.n, ',', nums, 4);      // fields 0..2 are exact
                        pl[0] = pl[1] = pl[2] = -1.0;
                        for (int j = 0; j < nn && j < 3; ++j) {
                            if (nums[j].n == 1 && nums[j].p[0] == '.') pl[j] = -1.0;
                            else if (!parse_double(nums[j].p, nums[j].n, &pl[j])) ok = false;
                        }
                        has_pl = true;
                    }
                }
                if (!ok) break;
            }
        }
        if (gt.size() > 64 || f[0].n > 256 || v->chr_text.size() + f[0].n >= 0xFFFFFF00u || v->gt_text.size() + gt.size() >= 0xFFFFFF00u) {
            ok = false;
            break;
        }
        v->any_pl |= has_pl;
        v->chr_text.append(f[0].p, f[0].n);
        v->chr_off.push_back((uint32_t)v->chr_text.size());
        v->gt_text.append(gt);
        v->gt_off.push_back((uint32_t)v->gt_text.size());
        if (f[0].n > v->chr_width) v->chr_width = f[0].n;
        if (gt.size() > v->gt_width) v->gt_width = gt.size();
        v->pos.push_back(pos);
        v->dp.push_back(dp);
        v->pl.push_back(pl[0]);
        v->pl.push_back(pl[1]);
        v->pl.push_back(pl[2]);
    }
    };
    gzFile gz = gzopen(path, "rb");          // transparently reads plain text as well
    if (!gz) return SNPM_ERR_BADARG;
    struct Closer { gzFile f; ~Closer() { gzclose(f); } } closer{gz};
    (void)gzbuffer(gz, 1u << 20);
    std::vector<char> buf(4u << 20);
    std::string carry;                       // the unfinished last line of the previous block
    for (;;) {
        const int got = gzread(gz, buf.data(), (unsigned)buf.size());
        if (got < 0) return SNPM_ERR_STATE;
        if (got == 0) break;
        const char *b = buf.data(), *be = b + got;
        const char *last_nl = nullptr;
        for (const char *q = be; q > b; --q)
            if (q[-1] == '\n') { last_nl = q - 1; break; }
        if (!last_nl) {                      // no line ends in this block
            carry.append(b, (size_t)got);
            if (carry.size() > (64u << 20)) return SNPM_ERR_STATE;     // a 64 MiB line: not ours to interpret
            continue;
        }
        if (!carry.empty()) {
            const char *first_nl = (const char *)memchr(b, '\n', (size_t)(be - b));
            carry.append(b, (size_t)(first_nl + 1 - b));
            consume(carry.data(), carry.data() + carry.size());
            carry.clear();
            b = first_nl + 1;
        }
        if (ok && b <= last_nl) consume(b, last_nl + 1);
        carry.assign(last_nl + 1, (size_t)(be - (last_nl + 1)));
        if (!ok) break;
    }
    if (ok && !carry.empty()) consume(carry.data(), carry.data() + carry.size());
    if (!ok) return SNPM_ERR_STATE;
    *out = v.release();
    return SNPM_OK;
} catch (...) {
    return SNPM_ERR_STATE;
}

int snpm_vcf_dims(const snpm_vcf *v, int64_t *n_records, int *chr_width, int *gt_width, int *flags, int *n_samples)
{
    if (!v) return SNPM_ERR_BADARG;
    if (n_records) *n_records = (int64_t)v->pos.size();
    if (chr_width) *chr_width = (int)v->chr_width;
    if (gt_width) *gt_width = (int)v->gt_width;
    if (flags) *flags = (v->any_gt ? 1 : 0) | (v->any_pl ? 2 : 0) | (v->any_dp ? 4 : 0);
    if (n_samples) *n_samples = (int)v->sample_names.size();
    return SNPM_OK;
}

int snpm_vcf_fill(const snpm_vcf *v, char *chr, int64_t *pos, char *gt, double *pl, int64_t *dp)
{
    if (!v) return SNPM_ERR_BADARG;
    const size_t n = v->pos.size();
    if (chr) {
        memset(chr, 0, n * v->chr_width);
        for (size_t i = 0; i < n; ++i)
            memcpy(chr + i * v->chr_width, v->chr_text.data() + v->chr_off[i], v->chr_off[i + 1] - v->chr_off[i]);
    }
    if (gt) {
        memset(gt, 0, n * v->gt_width);
        for (size_t i = 0; i < n; ++i)
            memcpy(gt + i * v->gt_width, v->gt_text.data() + v->gt_off[i], v->gt_off[i + 1] - v->gt_off[i]);
    }
    if (pos && n) memcpy(pos, v->pos.data(), n * sizeof(int64_t));
    if (dp && n) memcpy(dp, v->dp.data(), n * sizeof(int64_t));
    if (pl && n) memcpy(pl, v->pl.data(), n * 3 * sizeof(double));
    return SNPM_OK;
}

const char *snpm_vcf_sample_name(const snpm_vcf *v, int i)
{
    if (!v || i < 0 || (size_t)i >= v->sample_names.size()) return nullptr;
    return v->sample_names[(size_t)i].c_str();
}

int snpm_vcf_free(snpm_vcf *v)
{
    delete v;
    return SNPM_OK;
}

}  // extern "C"

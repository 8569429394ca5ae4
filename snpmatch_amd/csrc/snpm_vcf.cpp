// Native reader for single-sample SNP-call VCF text (plain, gzip or BGZF): host-side input of the scoring path.
// The reference delegates this step to scikit-allel (core/parsers.py:178-213, a C extension); here it is a
// single pass over the file in C++.  Semantics follow snpmatch_amd/core/_vcf.py (the Python reader, kept as
// the generic path): per record CHROM, POS, the sample's GT text as written (a bare '.' becomes './.'), the
// first three PL values (-1 where absent) and INFO/DP (-1 where absent).  Anything this reader is not sure
// about (malformed numbers, over-long fields) makes it decline with SNPM_ERR_STATE and the caller parses the
// file with the Python reader instead.
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "snpmatch_hip.h"

// what one block of lines parses into (blocks are parsed by several threads and kept in file order)
struct VcfChunk {
    std::vector<std::string> sample_names;    // of the last #CHROM line of the block (has_names)
    bool has_names = false;
    std::string chr_text, gt_text;            // concatenated fields
    std::vector<uint32_t> chr_off{0}, gt_off{0};    // n + 1 offsets each
    std::vector<int64_t> pos, dp;
    std::vector<double> pl;                   // n * 3
    size_t chr_width = 1, gt_width = 1;
    bool any_gt = false, any_pl = false, any_dp = false, ok = true, ascii = true;
};

struct snpm_vcf {
    std::vector<std::string> sample_names;
    std::vector<std::unique_ptr<VcfChunk>> chunks;      // in file order
    std::vector<size_t> first;                // record index of every chunk's first record, + the total
    size_t n = 0;
    size_t chr_width = 1, gt_width = 1;
    bool any_gt = false, any_pl = false, any_dp = false, ascii = true;
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

// every line of [s, end) into the chunk
static void parse_block(VcfChunk *v, const char *s, const char *end, int sample_index)
{
    constexpr int MAXF = 4096;
    std::vector<Field> f(MAXF);
    Field keys[64], vals[64], nums[4];
    bool ok = true;
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
                v->has_names = true;
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
        const char *gt_p = "./.";
        size_t gt_n = 3;
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
                        if (val.n == 1 && val.p[0] == '.') { gt_p = "./."; gt_n = 3; }
                        else { gt_p = val.p; gt_n = val.n; }
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
        if (gt_n > 64 || f[0].n > 256 || v->chr_text.size() + f[0].n >= 0xFFFFFF00u || v->gt_text.size() + gt_n >= 0xFFFFFF00u) {
            ok = false;
            break;
        }
        for (size_t i = 0; i < f[0].n; ++i) v->ascii &= ((unsigned char)f[0].p[i] < 0x80);
        for (size_t i = 0; i < gt_n; ++i) v->ascii &= ((unsigned char)gt_p[i] < 0x80);
        v->any_pl |= has_pl;
        v->chr_text.append(f[0].p, f[0].n);
        v->chr_off.push_back((uint32_t)v->chr_text.size());
        v->gt_text.append(gt_p, gt_n);
        v->gt_off.push_back((uint32_t)v->gt_text.size());
        if (f[0].n > v->chr_width) v->chr_width = f[0].n;
        if (gt_n > v->gt_width) v->gt_width = gt_n;
        v->pos.push_back(pos);
        v->dp.push_back(dp);
        v->pl.push_back(pl[0]);
        v->pl.push_back(pl[1]);
        v->pl.push_back(pl[2]);
    }
    v->ok = ok;
}

static int vcf_threads()
{
    cpu_set_t set;
    int n = 4;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (const char *e = getenv("SNPM_VCF_THREADS")) n = atoi(e);
    return std::max(1, std::min(n, 16));
}

// ---- BGZF (bgzip / bcftools output: gzip members of <= 64 KiB, each with its compressed size in a 'BC' extra subfield) ----
// One gzip stream inflates on one thread (0.15 s of the 0.20 s a 1M-record .vcf.gz takes); BGZF members are independent, so the
// reading thread only walks the member headers of a batch of compressed bytes and a team inflates the members side by side,
// each into its own place of the batch's text (sizes from the members' ISIZE trailers, CRCs checked as gzread would).
struct BgzfMember {
    size_t src, src_len;        // raw deflate bytes inside the batch
    size_t dst, isize;          // where its text goes
    uint32_t crc;
};

// size of the member that starts at h (n readable bytes) and where its deflate data starts; 0: not a BGZF member header
inline size_t bgzf_member_size(const unsigned char *h, size_t n, size_t *data_off)
{
    if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
    if (n < 12 + xlen) return 0;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
        const size_t slen = (size_t)h[o + 2] | ((size_t)h[o + 3] << 8);
        if (h[o] == 'B' && h[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
            *data_off = 12 + xlen;
            const size_t total = ((size_t)h[o + 4] | ((size_t)h[o + 5] << 8)) + 1;
            return (h[3] == 4 && total >= 12 + xlen + 8) ? total : 0;      // other header flags (name, comment, hcrc): not ours
        }
        o += 4 + slen;
    }
    return 0;
}

inline bool bgzf_inflate(const unsigned char *src, size_t src_len, unsigned char *dst, size_t isize, uint32_t crc)
{
    if (isize == 0) return true;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(src);
    zs.avail_in = (uInt)src_len;
    zs.next_out = dst;
    zs.avail_out = (uInt)isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == isize;
    inflateEnd(&zs);
    return ok && (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, (uInt)isize) == crc;
}

}  // namespace

extern "C" {

// The file is read (and, for .gz, decompressed) by the calling thread in blocks of whole lines; the blocks are parsed by a
// small team of threads and kept in file order (round 3 parsed on the reading thread: 0.7 s of a 1M-record file).
// No C++ exception leaves this function (ctypes would turn it into std::terminate): running out of host memory
// on a very large file is reported as SNPM_ERR_STATE, i.e. "use the generic reader", like any other declined file.
int snpm_vcf_parse(const char *path, int sample_index, snpm_vcf **out)
try {
    if (!path || !out || sample_index < 0 || sample_index > 4000) return SNPM_ERR_BADARG;
    std::unique_ptr<snpm_vcf> v(new snpm_vcf());
    gzFile gz = gzopen(path, "rb");          // transparently reads plain text as well
    if (!gz) return SNPM_ERR_BADARG;
    struct Closer { gzFile f; ~Closer() { gzclose(f); } } closer{gz};
    (void)gzbuffer(gz, 1u << 20);

    // blocks of whole lines travel to the parsers through a small queue (bounded: the reader never runs far ahead)
    struct Job { std::unique_ptr<std::string> text; size_t seq; };
    std::mutex m;
    std::condition_variable cv_job, cv_room;
    std::deque<Job> queue;
    bool done = false;
    std::atomic<bool> failed{false};
    const int n_thr = vcf_threads();
    const size_t max_queued = (size_t)n_thr * 2;
    auto worker = [&]() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                job = std::move(queue.front());
                queue.pop_front();
            }
            cv_room.notify_one();
            // nothing a worker does may leave it as an exception (a thread that throws ends the process): the chunk's own
            // allocations and the store into v->chunks included
            try {
                std::unique_ptr<VcfChunk> c(new VcfChunk());
                if (!failed.load()) {
                    parse_block(c.get(), job.text->data(), job.text->data() + job.text->size(), sample_index);
                    if (!c->ok) failed.store(true);
                }
                std::lock_guard<std::mutex> lk(m);
                if (v->chunks.size() <= job.seq) v->chunks.resize(job.seq + 1);
                v->chunks[job.seq] = std::move(c);
            } catch (...) {
                failed.store(true);
            }
        }
    };
    // the guard exists BEFORE the first thread does: if a later thread cannot be created (or anything below throws) the ones
    // already running are told to finish and joined -- a joinable std::thread that is destroyed ends the process
    std::vector<std::thread> team;
    auto finish_team = [&]() {
        {
            std::lock_guard<std::mutex> lk(m);
            done = true;
        }
        cv_job.notify_all();
        for (auto &t : team)
            if (t.joinable()) t.join();
    };
    struct TeamGuard { std::function<void()> f; bool armed = true; ~TeamGuard() { if (armed) f(); } } guard{finish_team};
    team.reserve((size_t)n_thr);
    for (int t = 0; t < n_thr; ++t) team.emplace_back(worker);
    size_t seq = 0;
    auto submit = [&](std::unique_ptr<std::string> text) {
        std::unique_lock<std::mutex> lk(m);
        cv_room.wait(lk, [&] { return queue.size() < max_queued; });
        queue.push_back(Job{std::move(text), seq++});
        lk.unlock();
        cv_job.notify_one();
    };
    size_t kBlock = 4u << 20;
    if (const char *e = getenv("SNPM_VCF_BLOCK_KB")) kBlock = (size_t)std::max(4, atoi(e)) << 10;     // tests: many blocks from a small file
    std::string carry;                       // the unfinished last line of the previous block
    bool read_error = false;

    // BGZF input: batches of compressed bytes are inflated member by member on a team of threads into `text`; the loop below
    // then takes its blocks from there instead of from gzread
    bool bgzf = false;
    FILE *raw = nullptr;
    struct RawCloser { FILE **f; ~RawCloser() { if (*f) fclose(*f); } } raw_closer{&raw};
    std::string comp, text;                  // compressed bytes not yet consumed; inflated text not yet handed out
    size_t text_pos = 0;
    bool raw_eof = false;
    size_t kBatch = 16u << 20;
    if (const char *e = getenv("SNPM_VCF_BGZF_BATCH_KB")) kBatch = (size_t)std::max(1, atoi(e)) << 10;   // tests: many batches
    if (!(getenv("SNPM_VCF_BGZF") && atoi(getenv("SNPM_VCF_BGZF")) == 0)) {
        raw = fopen(path, "rb");
        unsigned char head[64];
        const size_t got = raw ? fread(head, 1, sizeof(head), raw) : 0;
        size_t off = 0;
        bgzf = raw && bgzf_member_size(head, got, &off) != 0;
        if (bgzf) comp.assign((const char *)head, got);
        else if (raw) { fclose(raw); raw = nullptr; }
    }
    // next batch: false on a malformed member (the whole file is then declined)
    size_t kMaxBatchText = size_t(192) << 20;            // inflated bytes per batch
    if (const char *e = getenv("SNPM_VCF_BGZF_TEXT_KB")) kMaxBatchText = (size_t)std::max(1, atoi(e)) << 10;   // tests: the cap at work on a small file
    auto bgzf_refill = [&]() -> bool {
        text.clear();
        text_pos = 0;
        bool starved = false;       // the bytes at hand hold no whole member (a member larger than the batch): read on regardless
        while (text.empty() && !(raw_eof && comp.empty())) {
            if (!raw_eof && (comp.size() < kBatch || starved)) {
                const size_t have = comp.size();
                comp.resize(have + kBatch);
                const size_t got = fread(&comp[have], 1, kBatch, raw);
                comp.resize(have + got);
                if (got < kBatch) raw_eof = true;
            }
            std::vector<BgzfMember> mem;
            size_t pos = 0, total = 0;
            const unsigned char *c = (const unsigned char *)comp.data();
            while (pos < comp.size()) {
                size_t data_off = 0;
                const size_t msz = bgzf_member_size(c + pos, comp.size() - pos, &data_off);
                if (msz == 0) {
                    if (comp.size() - pos < 18 + 65536 && !raw_eof) break;        // a header cut by the batch boundary
                    return false;
                }
                if (pos + msz > comp.size()) {
                    if (raw_eof) return false;                                    // a member cut by the end of the file
                    break;
                }
                const unsigned char *t = c + pos + msz - 8;
                const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                const size_t isize = (size_t)t[4] | ((size_t)t[5] << 8) | ((size_t)t[6] << 16) | ((size_t)t[7] << 24);
                if (isize > 65536) return false;
                // a batch is bounded by its INFLATED size too: ~100 B of compressed bytes can announce a 64-KiB member, so 16 MiB of a
                // crafted (or merely very repetitive) file would otherwise expand to ~10 GB in one string.  The members left over
                // stay in `comp` for the next refill.
                if (total + isize > kMaxBatchText && !mem.empty()) break;
                mem.push_back(BgzfMember{pos + data_off, msz - data_off - 8, total, isize, crc});
                total += isize;
                pos += msz;
            }
            if (mem.empty() && pos == 0 && raw_eof && !comp.empty()) return false;     // trailing bytes that are no member
            starved = mem.empty();
            text.resize(total);
            std::atomic<size_t> next{0};
            std::atomic<bool> bad{false};
            auto inflate_some = [&]() {
                for (size_t i = next.fetch_add(1); i < mem.size(); i = next.fetch_add(1)) {
                    const BgzfMember &b = mem[i];
                    if (!bgzf_inflate(c + b.src, b.src_len, (unsigned char *)&text[0] + b.dst, b.isize, b.crc)) bad.store(true);
                }
            };
            const int helpers = (int)std::min<size_t>((size_t)n_thr, mem.size() / 8);
            {
                // joined by the guard however this scope is left (a thread that cannot be created, an exception in inflate_some)
                std::vector<std::thread> inflaters;
                struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } join_all{inflaters};
                inflaters.reserve((size_t)std::max(helpers, 1));
                for (int t = 1; t < helpers; ++t) inflaters.emplace_back(inflate_some);
                inflate_some();
            }
            if (bad.load()) return false;
            comp.erase(0, pos);
        }
        return true;
    };
    for (;;) {
        if (failed.load()) break;
        std::unique_ptr<std::string> blk(new std::string());
        int got;
        if (bgzf) {
            if (text_pos == text.size() && !bgzf_refill()) { read_error = true; break; }
            const size_t take = std::min(kBlock, text.size() - text_pos);
            blk->resize(carry.size() + take);
            memcpy(&(*blk)[0], carry.data(), carry.size());
            memcpy(&(*blk)[carry.size()], text.data() + text_pos, take);
            text_pos += take;
            got = (int)take;
        } else {
            blk->resize(carry.size() + kBlock);
            memcpy(&(*blk)[0], carry.data(), carry.size());
            got = gzread(gz, &(*blk)[carry.size()], (unsigned)kBlock);
        }
        if (got < 0) { read_error = true; break; }
        if (got == 0) break;
        blk->resize(carry.size() + (size_t)got);
        const size_t last_nl = blk->rfind('\n');
        if (last_nl == std::string::npos) {  // no line ends in this block
            carry.swap(*blk);
            if (carry.size() > (64u << 20)) { read_error = true; break; }     // a 64 MiB line: not ours to interpret
            continue;
        }
        carry.assign(*blk, last_nl + 1, std::string::npos);
        blk->resize(last_nl + 1);
        submit(std::move(blk));
    }
    if (!read_error && !failed.load() && !carry.empty()) submit(std::unique_ptr<std::string>(new std::string(std::move(carry))));
    guard.armed = false;
    finish_team();
    if (read_error || failed.load()) return SNPM_ERR_STATE;
    // totals
    v->first.assign(v->chunks.size() + 1, 0);
    for (size_t c = 0; c < v->chunks.size(); ++c) {
        const VcfChunk *ch = v->chunks[c].get();
        if (!ch) return SNPM_ERR_STATE;
        v->first[c + 1] = v->first[c] + ch->pos.size();
        v->chr_width = std::max(v->chr_width, ch->chr_width);
        v->gt_width = std::max(v->gt_width, ch->gt_width);
        v->any_gt |= ch->any_gt;
        v->any_pl |= ch->any_pl;
        v->any_dp |= ch->any_dp;
        v->ascii &= ch->ascii;
        if (ch->has_names) v->sample_names = ch->sample_names;      // the last #CHROM line of the file, as before
    }
    v->n = v->first.back();
    *out = v.release();
    return SNPM_OK;
} catch (...) {
    return SNPM_ERR_STATE;
}

int snpm_vcf_dims(const snpm_vcf *v, int64_t *n_records, int *chr_width, int *gt_width, int *flags, int *n_samples)
{
    if (!v) return SNPM_ERR_BADARG;
    if (n_records) *n_records = (int64_t)v->n;
    if (chr_width) *chr_width = (int)v->chr_width;
    if (gt_width) *gt_width = (int)v->gt_width;
    if (flags) *flags = (v->any_gt ? 1 : 0) | (v->any_pl ? 2 : 0) | (v->any_dp ? 4 : 0) | (v->ascii ? 8 : 0);
    if (n_samples) *n_samples = (int)v->sample_names.size();
    return SNPM_OK;
}

// chunks are copied out by the same small team of threads, each chunk to its own place
static void for_each_chunk(const snpm_vcf *v, const std::function<void(const VcfChunk &, size_t)> &fn)
{
    const size_t nc = v->chunks.size();
    const int n_thr = (int)std::min<size_t>((size_t)vcf_threads(), nc);
    if (n_thr <= 1) {
        for (size_t c = 0; c < nc; ++c) fn(*v->chunks[c], v->first[c]);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> team;
    for (int t = 0; t < n_thr; ++t)
        team.emplace_back([&]() {
            for (size_t c = next.fetch_add(1); c < nc; c = next.fetch_add(1)) fn(*v->chunks[c], v->first[c]);
        });
    for (auto &t : team) t.join();
}

int snpm_vcf_fill(const snpm_vcf *v, char *chr, int64_t *pos, char *gt, double *pl, int64_t *dp)
try {
    if (!v) return SNPM_ERR_BADARG;
    const size_t cw = v->chr_width, gw = v->gt_width;
    for_each_chunk(v, [&](const VcfChunk &c, size_t r0) {
        const size_t n = c.pos.size();
        if (chr) {
            memset(chr + r0 * cw, 0, n * cw);
            for (size_t i = 0; i < n; ++i) memcpy(chr + (r0 + i) * cw, c.chr_text.data() + c.chr_off[i], c.chr_off[i + 1] - c.chr_off[i]);
        }
        if (gt) {
            memset(gt + r0 * gw, 0, n * gw);
            for (size_t i = 0; i < n; ++i) memcpy(gt + (r0 + i) * gw, c.gt_text.data() + c.gt_off[i], c.gt_off[i + 1] - c.gt_off[i]);
        }
        if (pos && n) memcpy(pos + r0, c.pos.data(), n * sizeof(int64_t));
        if (dp && n) memcpy(dp + r0, c.dp.data(), n * sizeof(int64_t));
        if (pl && n) memcpy(pl + 3 * r0, c.pl.data(), n * 3 * sizeof(double));
    });
    return SNPM_OK;
} catch (...) {
    return SNPM_ERR_STATE;
}

// The same records as numpy wants them: CHROM and GT as fixed-width UTF-32 ('<U{width}': one code point per uint32, zero padded),
// so that no per-string conversion runs in Python; called[i] = 0 where the sample's genotype is './.' or '.|.' (the records
// ParseInputs.read_vcf drops, core/parsers.py:141-157), 1 otherwise.  Only for files whose CHROM / GT text is ASCII (flag bit 3 of
// snpm_vcf_dims); SNPM_ERR_STATE otherwise (the caller converts the byte strings itself).
int snpm_vcf_fill_u32(const snpm_vcf *v, uint32_t *chr, int64_t *pos, uint32_t *gt, double *pl, int64_t *dp, uint8_t *called)
try {
    if (!v) return SNPM_ERR_BADARG;
    if (!v->ascii) return SNPM_ERR_STATE;
    const size_t cw = v->chr_width, gw = v->gt_width;
    for_each_chunk(v, [&](const VcfChunk &c, size_t r0) {
        const size_t n = c.pos.size();
        if (chr) {
            memset(chr + r0 * cw, 0, n * cw * sizeof(uint32_t));
            for (size_t i = 0; i < n; ++i) {
                const unsigned char *src = (const unsigned char *)c.chr_text.data() + c.chr_off[i];
                const size_t len = c.chr_off[i + 1] - c.chr_off[i];
                for (size_t k = 0; k < len; ++k) chr[(r0 + i) * cw + k] = src[k];
            }
        }
        if (gt || called) {
            if (gt) memset(gt + r0 * gw, 0, n * gw * sizeof(uint32_t));
            for (size_t i = 0; i < n; ++i) {
                const unsigned char *src = (const unsigned char *)c.gt_text.data() + c.gt_off[i];
                const size_t len = c.gt_off[i + 1] - c.gt_off[i];
                if (gt)
                    for (size_t k = 0; k < len; ++k) gt[(r0 + i) * gw + k] = src[k];
                if (called) called[r0 + i] = !(len == 3 && src[0] == '.' && src[2] == '.' && (src[1] == '/' || src[1] == '|'));
            }
        }
        if (pos && n) memcpy(pos + r0, c.pos.data(), n * sizeof(int64_t));
        if (dp && n) memcpy(dp + r0, c.dp.data(), n * sizeof(int64_t));
        if (pl && n) memcpy(pl + 3 * r0, c.pl.data(), n * 3 * sizeof(double));
    });
    return SNPM_OK;
} catch (...) {
    return SNPM_ERR_STATE;
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

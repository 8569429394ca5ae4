// snpm_group.hpp -- multi-GPU behind the C ABI: accession shards per GPU and ONE RCCL
// all-gather of the per-accession results, inside libsnpmatch_hip.so.  Included at the end of snpm_api.hip (same
// translation unit: it uses snpm_ctx and the helper macros of that file).
//
// Reference: every reduction of matchGTsAccs runs over SNPs (core/snpmatch.py:84-88), accession columns never
// interact; the likelihood step needs the minimum over ALL accessions (core/snpmatch.py:112).  So member r of a
// group of R scores columns [a0_r, a1_r) and the only exchange is the gather of (score fp64, ninfo int64).
//
// RCCL is bound at the first group call with dlopen, not at link time: a process that already holds an RCCL (PyTorch
// wheels bundle theirs) must use that one -- two RCCLs on one HIP runtime do not share their bootstrap / IPC state --
// and a process that never forms a group needs none.  Search order: an RCCL already mapped into the process, then
// the one beside the HIP runtime this library runs on, then librccl.so.1 from the loader's path.
#pragma once

#include <dlfcn.h>
#include <link.h>

#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *handle = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data)
{
    const char *name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char *base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "librccl.so", 10) == 0) {
        *(std::string *)data = name;
        return 1;
    }
    return 0;
}

// dlopen RCCL once per process; returns SNPM_OK or SNPM_ERR_RCCL with the reason in `why`
int rccl_bind(std::string &why)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return SNPM_OK;
    std::vector<std::string> tries;
    if (const char *forced = getenv("SNPMATCH_RCCL_LIB")) tries.push_back(forced);
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) tries.push_back(loaded);
    Dl_info di;
    if (dladdr((void *)(hipError_t (*)(int))&hipSetDevice, &di) && di.dli_fname) {          // the HIP runtime this library is bound to
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            tries.push_back(dir + "/librccl.so.1");
            tries.push_back(dir + "/librccl.so");
        }
    }
    tries.push_back("librccl.so.1");
    tries.push_back("librccl.so");
    void *h = nullptr;
    for (const std::string &t : tries) {
        h = dlopen(t.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) {
            g_rccl.path = t;
            break;
        }
        why += t + ": " + (dlerror() ?: "?") + "; ";
    }
    if (!h) return SNPM_ERR_RCCL;
#define RCCL_SYM(field, name)                                                        \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                           \
    if (!g_rccl.field) { why = std::string("symbol ") + name + " missing in " + g_rccl.path; dlclose(h); return SNPM_ERR_RCCL; }
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    RCCL_SYM(CommInitRank, "ncclCommInitRank")
    RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(AllGather, "ncclAllGather")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(GetVersion, "ncclGetVersion")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    g_rccl.handle = h;
    return SNPM_OK;
}

// (score [m, n_loc] stride in_ld, ninfo likewise) -> send [2][m][per] (8-byte words), tail columns of the shard zero:
// a padded entry (score 0, ninfo 0) has a NaN likelihood and never reaches an output
__global__ void k_group_pack(const double *__restrict__ score, const int64_t *__restrict__ ninfo, int64_t m, int64_t n_loc,
                             int64_t in_ld, int64_t per, uint64_t *__restrict__ send)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * per) return;
    const int64_t w = i / per, c = i - w * per;
    const bool live = c < n_loc;
    send[i] = live ? (uint64_t)__double_as_longlong(score[w * in_ld + c]) : 0ull;
    send[m * per + i] = live ? (uint64_t)ninfo[w * in_ld + c] : 0ull;
}

// recv [world][2][m][per] -> score_all / ninfo_all [m, n_acc] (accession a lives on rank a / per at offset a % per)
__global__ void k_group_unpack(const uint64_t *__restrict__ recv, int64_t m, int64_t n_acc, int64_t per,
                               double *__restrict__ score_all, int64_t *__restrict__ ninfo_all)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * n_acc) return;
    const int64_t w = i / n_acc, a = i - w * n_acc;
    const int64_t r = a / per, c = a - r * per;
    const uint64_t *blk = recv + r * (2 * m * per);
    score_all[i] = __longlong_as_double((long long)blk[w * per + c]);
    ninfo_all[i] = (int64_t)blk[m * per + w * per + c];
}

}  // namespace

struct snpm_group {
    int world = 1, rank0 = 0;                 // ranks of the whole job; global rank of local member 0
    int transport = 0;                        // 0 = RCCL, 1 = loopback (single-process test transport, peer copies)
    bool owns_ctx = false;                    // created by snpm_group_create_local: the contexts go with the group
    std::vector<snpm_ctx *> ctx;              // local members (1 in process-per-GPU jobs)
    std::vector<ncclComm_t> comm;
    struct Bufs { Buf send, recv, score_all, ninfo_all, lik, lrt; };
    std::vector<Bufs> bufs;
    std::vector<hipEvent_t> ev;               // loopback: "member i has packed its send buffer"
    std::vector<hipEvent_t> ev_done;          // loopback: "member i has copied every member's send buffer" (the buffers may be rewritten)
    bool gathered_once = false;
    std::string err;
};

namespace {

thread_local std::string g_group_error;

int group_err(snpm_group *g, int code, const char *fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    g_group_error = buf;
    return code;
}

#define NCCLCHK(g, expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return group_err((g), SNPM_ERR_RCCL, "%s failed: %s (RCCL %s)", #expr, g_rccl.GetErrorString(r_), \
                             g_rccl.path.c_str());                                                         \
    } while (0)

#define GHIPCHK(g, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return group_err((g), (e_ == hipErrorOutOfMemory) ? SNPM_ERR_OOM : SNPM_ERR_HIP, "%s failed: %s", #expr, \
                             hipGetErrorString(e_));                                                       \
    } while (0)

#define GROUP_GUARD(G)                                                                                     \
    catch (const std::bad_alloc &) { return group_err((G), SNPM_ERR_OOM, "out of host memory"); }          \
    catch (const std::exception &e_) { return group_err((G), SNPM_ERR_STATE, "internal error: %s", e_.what()); } \
    catch (...) { return group_err((G), SNPM_ERR_STATE, "internal error"); }

void group_release_member(snpm_group *g, size_t i, bool use_hip)
{
    snpm_ctx *c = g->ctx[i];
    if (!c) return;
    if (use_hip) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (i < g->comm.size() && g->comm[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comm[i]);
        Buf *bufs[] = {&g->bufs[i].send, &g->bufs[i].recv, &g->bufs[i].score_all, &g->bufs[i].ninfo_all, &g->bufs[i].lik, &g->bufs[i].lrt};
        for (Buf *b : bufs) {
            if (b->p) (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
        if (i < g->ev.size() && g->ev[i]) (void)hipEventDestroy(g->ev[i]);
        if (i < g->ev_done.size() && g->ev_done[i]) (void)hipEventDestroy(g->ev_done[i]);
    }
    if (i < g->comm.size()) g->comm[i] = nullptr;
    if (i < g->ev.size()) g->ev[i] = nullptr;
    if (i < g->ev_done.size()) g->ev_done[i] = nullptr;
}

// the context of a rank-style group is being destroyed before the group: release what lives on it, keep the handle valid
void group_forget_ctx(snpm_group *g, snpm_ctx *ctx, bool use_hip)
{
    for (size_t i = 0; i < g->ctx.size(); ++i)
        if (g->ctx[i] == ctx) {
            group_release_member(g, i, use_hip);
            g->ctx[i] = nullptr;
        }
}

#define CHECK_GROUP_ALIVE(G)                                                                               \
    do {                                                                                                   \
        for (snpm_ctx *c_ : (G)->ctx)                                                                      \
            if (!c_) return group_err((G), SNPM_ERR_STATE, "the group outlived a context it was made of (snpm_destroy was called)"); \
    } while (0)

int64_t group_per(const snpm_group *g, int64_t n_acc)
{
    const int64_t per = (n_acc + g->world - 1) / g->world;
    return (per + 3) / 4 * 4;                 // shard boundaries are multiples of 4 accessions (quads of the generator, dwords of k_fast)
}

}  // namespace

extern "C" {

const char *snpm_group_last_error(const snpm_group *g) { return g ? g->err.c_str() : g_group_error.c_str(); }

int snpm_group_unique_id(void *id_bytes)
try {
    if (!id_bytes) return group_err(nullptr, SNPM_ERR_BADARG, "id_bytes is NULL");
    std::string why;
    if (rccl_bind(why)) return group_err(nullptr, SNPM_ERR_RCCL, "cannot load RCCL: %s", why.c_str());
    ncclUniqueId id;
    NCCLCHK(nullptr, g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == SNPM_GROUP_ID_BYTES, "ncclUniqueId size");
    memcpy(id_bytes, &id, sizeof(id));
    return SNPM_OK;
} GROUP_GUARD(nullptr)

// one process per GPU: this process is rank `rank` of `world`, its GPU is the context's
int snpm_group_create_rank(snpm_ctx *ctx, const void *id_bytes, int world, int rank, snpm_group **out)
try {
    if (!ctx || !out) return group_err(nullptr, SNPM_ERR_BADARG, "ctx / out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return group_err(nullptr, SNPM_ERR_BADARG, "need 0 <= rank < world");
    if (!id_bytes) return group_err(nullptr, SNPM_ERR_BADARG, "id_bytes is NULL (snpm_group_unique_id on one rank, then pass it to all)");
    std::string why;
    if (rccl_bind(why)) return group_err(nullptr, SNPM_ERR_RCCL, "cannot load RCCL: %s", why.c_str());
    GHIPCHK(nullptr, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCLCHK(nullptr, g_rccl.CommInitRank(&comm, world, id, rank));
    snpm_group *g = new snpm_group();
    g->world = world;
    g->rank0 = rank;
    g->ctx.push_back(ctx);
    g->comm.push_back(comm);
    g->bufs.resize(1);
    ctx->groups.push_back(g);
    *out = g;
    return SNPM_OK;
} GROUP_GUARD(nullptr)

// one process drives n GPUs (ncclCommInitAll; no launcher): creates the n contexts too (snpm_group_ctx hands them out,
// snpm_group_free destroys them).  flags: SNPM_GROUP_LOOPBACK = exchange by peer copies instead of RCCL -- a test
// transport that also accepts the same device several times (rehearsal of the sharding on a one-GPU box).
int snpm_group_create_local(const int *device_ids, int n, int flags, snpm_group **out)
try {
    if (!out) return group_err(nullptr, SNPM_ERR_BADARG, "out is NULL");
    *out = nullptr;
    if (n < 1 || !device_ids) return group_err(nullptr, SNPM_ERR_BADARG, "need n >= 1 device ids");
    const bool loopback = (flags & SNPM_GROUP_LOOPBACK) != 0;
    if (!loopback)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (device_ids[i] == device_ids[j])
                    return group_err(nullptr, SNPM_ERR_BADARG, "device %d listed twice (RCCL takes one rank per GPU)", device_ids[i]);
    snpm_group *g = new snpm_group();
    g->world = n;
    g->rank0 = 0;
    g->owns_ctx = true;
    g->transport = loopback ? 1 : 0;
    g->bufs.resize((size_t)n);
    int rc = SNPM_OK;
    for (int i = 0; i < n && !rc; ++i) {
        snpm_ctx *c = nullptr;
        rc = snpm_init(device_ids[i], &c);
        if (rc) group_err(nullptr, rc, "snpm_init(device %d): %s", device_ids[i], snpm_last_error(nullptr));
        else g->ctx.push_back(c);
    }
    if (!rc && !loopback) {
        std::string why;
        if (rccl_bind(why)) rc = group_err(nullptr, SNPM_ERR_RCCL, "cannot load RCCL: %s", why.c_str());
        if (!rc) {
            g->comm.assign((size_t)n, nullptr);
            ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), n, device_ids);
            if (r != ncclSuccess) {
                g->comm.clear();
                rc = group_err(nullptr, SNPM_ERR_RCCL, "ncclCommInitAll failed: %s (RCCL %s)", g_rccl.GetErrorString(r), g_rccl.path.c_str());
            }
        }
    }
    if (!rc && loopback) {
        g->ev.assign((size_t)n, nullptr);
        g->ev_done.assign((size_t)n, nullptr);
        for (int i = 0; i < n && !rc; ++i) {
            if (hipSetDevice(device_ids[i]) != hipSuccess || hipEventCreateWithFlags(&g->ev[(size_t)i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g->ev_done[(size_t)i], hipEventDisableTiming) != hipSuccess)
                rc = group_err(nullptr, SNPM_ERR_HIP, "hipEventCreate failed");
        }
    }
    if (rc) {
        const std::string keep = g_group_error;
        snpm_group_free(g);
        g_group_error = keep;
        return rc;
    }
    *out = g;
    return SNPM_OK;
} GROUP_GUARD(nullptr)

int snpm_group_free(snpm_group *g)
{
    if (!g) return SNPM_OK;
    const bool use_hip = hip_alive();
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        snpm_ctx *c = g->ctx[i];
        if (!c) continue;                   // the context went first and took the member with it
        group_release_member(g, i, use_hip);
        if (g->owns_ctx) (void)snpm_destroy(c);
        else c->groups.erase(std::remove(c->groups.begin(), c->groups.end(), g), c->groups.end());
    }
    delete g;
    return SNPM_OK;
}

int snpm_group_info(const snpm_group *g, int *world, int *rank0, int *n_local)
{
    if (!g) return group_err(nullptr, SNPM_ERR_BADARG, "group is NULL");
    if (world) *world = g->world;
    if (rank0) *rank0 = g->rank0;
    if (n_local) *n_local = (int)g->ctx.size();
    return SNPM_OK;
}

int snpm_group_ctx(snpm_group *g, int member, snpm_ctx **ctx)
{
    if (!g || !ctx) return group_err(g, SNPM_ERR_BADARG, "group / ctx is NULL");
    if (member < 0 || member >= (int)g->ctx.size()) return group_err(g, SNPM_ERR_BADARG, "member %d outside the group's local members", member);
    CHECK_GROUP_ALIVE(g);
    *ctx = g->ctx[(size_t)member];
    return SNPM_OK;
}

// accession range [a0, a1) of global rank `rank` for a DB of n_acc accessions (empty when there are more ranks than quads)
int snpm_group_shard(const snpm_group *g, int64_t n_acc, int rank, int64_t *a0, int64_t *a1)
{
    if (!g) return group_err(nullptr, SNPM_ERR_BADARG, "group is NULL");
    if (n_acc < 0 || rank < 0 || rank >= g->world) return group_err(const_cast<snpm_group *>(g), SNPM_ERR_BADARG, "rank outside the group");
    const int64_t per = group_per(g, n_acc);
    if (a0) *a0 = std::min<int64_t>((int64_t)rank * per, n_acc);
    if (a1) *a1 = std::min<int64_t>((int64_t)(rank + 1) * per, n_acc);
    return SNPM_OK;
}

// The collective of the path.  Local member i hands in the results of its shard -- DEVICE pointers d_score[i] float64
// [m, n_loc_i] and d_ninfo[i] int64 [m, n_loc_i], row stride in_ld elements (m = 1: the genome-wide totals; m = n_win:
// per-window rows) -- written by work queued on its context's stream.  Per member: one pack kernel, ONE all-gather
// (RCCL, 16 * m * per bytes per rank) and one unpack kernel on that stream leave score_all / ninfo_all [m, n_acc] on
// every member; with want_lik the likelihood rows (global minimum per row, core/snpmatch.py:106-117) follow on
// the device.  Host outputs (any may be NULL; lik and lrt both or neither) are copied from local member 0 and the call
// then waits for them; with all of them NULL nothing waits on the host (snpm_group_gathered_ptrs gives the device
// arrays).  Every rank of the job must make the call with the same m and n_acc.
int snpm_group_gather_scores(snpm_group *g, const void *const *d_score, const void *const *d_ninfo, int64_t m, int64_t n_acc,
                             int64_t in_ld, int truncate, double *score, int64_t *ninfo, double *lik, double *lrt)
try {
    if (!g) return group_err(nullptr, SNPM_ERR_BADARG, "group is NULL");
    if (!d_score || !d_ninfo || m < 1 || n_acc < 1) return group_err(g, SNPM_ERR_BADARG, "gather needs device pointers, m >= 1, n_acc >= 1");
    if ((lik == nullptr) != (lrt == nullptr)) return group_err(g, SNPM_ERR_BADARG, "lik and lrt: both or neither");
    CHECK_GROUP_ALIVE(g);
    const int nl = (int)g->ctx.size();
    const int64_t per = group_per(g, n_acc);
    const size_t words = (size_t)(2 * m * per);
    const size_t tot = (size_t)m * (size_t)n_acc;
    for (int i = 0; i < nl; ++i) {
        snpm_ctx *c = g->ctx[(size_t)i];
        int64_t a0 = 0, a1 = 0;
        snpm_group_shard(g, n_acc, g->rank0 + i, &a0, &a1);
        const int64_t n_loc = a1 - a0;
        if (n_loc > 0 && (!d_score[i] || !d_ninfo[i])) return group_err(g, SNPM_ERR_BADARG, "member %d: NULL result pointer", i);
        if (in_ld < n_loc) return group_err(g, SNPM_ERR_BADARG, "in_ld smaller than the shard");
        GHIPCHK(g, hipSetDevice(c->device));
        snpm_group::Bufs &b = g->bufs[(size_t)i];
        int rc;
        if ((rc = ensure(c, b.send, words * 8)) || (rc = ensure(c, b.recv, words * 8 * (size_t)g->world)) ||
            (rc = ensure(c, b.score_all, tot * 8)) || (rc = ensure(c, b.ninfo_all, tot * 8)))
            return group_err(g, rc, "member %d: %s", i, c->err.c_str());
        if (g->transport == 1 && g->gathered_once)          // the previous gather's copies out of this send buffer are done (every member's)
            for (int j = 0; j < nl; ++j) GHIPCHK(g, hipStreamWaitEvent(c->stream, g->ev_done[(size_t)j], 0));
        hipLaunchKernelGGL(k_group_pack, dim3((unsigned)((m * per + 255) / 256)), dim3(256), 0, c->stream, (const double *)d_score[i],
                           (const int64_t *)d_ninfo[i], m, n_loc, in_ld, per, (uint64_t *)b.send.p);
        GHIPCHK(g, hipGetLastError());
        if (g->transport == 1) GHIPCHK(g, hipEventRecord(g->ev[(size_t)i], c->stream));
    }
    if (g->transport == 0) {
        if (nl > 1) NCCLCHK(g, g_rccl.GroupStart());
        for (int i = 0; i < nl; ++i) {
            snpm_ctx *c = g->ctx[(size_t)i];
            GHIPCHK(g, hipSetDevice(c->device));
            NCCLCHK(g, g_rccl.AllGather(g->bufs[(size_t)i].send.p, g->bufs[(size_t)i].recv.p, words, ncclUint64, g->comm[(size_t)i], c->stream));
        }
        if (nl > 1) NCCLCHK(g, g_rccl.GroupEnd());
    } else {
        // loopback: every member copies every member's send block into its own receive buffer, after that member packed it
        for (int i = 0; i < nl; ++i) {
            snpm_ctx *c = g->ctx[(size_t)i];
            GHIPCHK(g, hipSetDevice(c->device));
            for (int j = 0; j < nl; ++j) {
                GHIPCHK(g, hipStreamWaitEvent(c->stream, g->ev[(size_t)j], 0));
                GHIPCHK(g, hipMemcpyAsync((uint64_t *)g->bufs[(size_t)i].recv.p + (size_t)j * words, g->bufs[(size_t)j].send.p, words * 8,
                                          hipMemcpyDeviceToDevice, c->stream));
            }
            GHIPCHK(g, hipEventRecord(g->ev_done[(size_t)i], c->stream));
        }
        g->gathered_once = true;
    }
    for (int i = 0; i < nl; ++i) {
        snpm_ctx *c = g->ctx[(size_t)i];
        GHIPCHK(g, hipSetDevice(c->device));
        snpm_group::Bufs &b = g->bufs[(size_t)i];
        hipLaunchKernelGGL(k_group_unpack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, (const uint64_t *)b.recv.p, m,
                           n_acc, per, (double *)b.score_all.p, (int64_t *)b.ninfo_all.p);
        GHIPCHK(g, hipGetLastError());
    }
    snpm_ctx *c0 = g->ctx[0];
    GHIPCHK(g, hipSetDevice(c0->device));
    snpm_group::Bufs &b0 = g->bufs[0];
    if (lik) {
        int rc;
        if ((rc = ensure(c0, b0.lik, tot * 8)) || (rc = ensure(c0, b0.lrt, tot * 8)) || (rc = ensure_pinned(c0, 64)))
            return group_err(g, rc, "%s", c0->err.c_str());
        rc = snpm_likelihood_device(c0, b0.score_all.p, b0.ninfo_all.p, m, n_acc, truncate, NAN, b0.lik.p, b0.lrt.p, nullptr);
        if (rc) return group_err(g, rc, "%s", c0->err.c_str());
        GHIPCHK(g, hipMemcpyAsync(lik, b0.lik.p, tot * 8, hipMemcpyDeviceToHost, c0->stream));
        GHIPCHK(g, hipMemcpyAsync(lrt, b0.lrt.p, tot * 8, hipMemcpyDeviceToHost, c0->stream));
        GHIPCHK(g, hipMemcpyAsync(c0->h_pinned, c0->ws_flags.p, sizeof(int), hipMemcpyDeviceToHost, c0->stream));
    }
    if (score) GHIPCHK(g, hipMemcpyAsync(score, b0.score_all.p, tot * 8, hipMemcpyDeviceToHost, c0->stream));
    if (ninfo) GHIPCHK(g, hipMemcpyAsync(ninfo, b0.ninfo_all.p, tot * 8, hipMemcpyDeviceToHost, c0->stream));
    if (score || ninfo || lik) {
        // the other local members' copies of the gathered vectors are complete too when the call returns
        for (int i = nl - 1; i >= 0; --i) {
            GHIPCHK(g, hipSetDevice(g->ctx[(size_t)i]->device));
            GHIPCHK(g, hipStreamSynchronize(g->ctx[(size_t)i]->stream));
        }
        if (lik && (*(const int *)c0->h_pinned & 1)) return group_err(g, SNPM_ERR_DOMAIN, "provided y is greater than n");
    }
    return SNPM_OK;
} GROUP_GUARD(g)

// device arrays of the last gather on a local member: score_all float64 [m, n_acc], ninfo_all int64 [m, n_acc]
// (valid until the next gather; ordered on that member's stream)
int snpm_group_gathered_ptrs(snpm_group *g, int member, void **d_score_all, void **d_ninfo_all)
{
    if (!g) return group_err(nullptr, SNPM_ERR_BADARG, "group is NULL");
    if (member < 0 || member >= (int)g->ctx.size()) return group_err(g, SNPM_ERR_BADARG, "member outside the group's local members");
    if (d_score_all) *d_score_all = g->bufs[(size_t)member].score_all.p;
    if (d_ninfo_all) *d_ninfo_all = g->bufs[(size_t)member].ninfo_all.p;
    return SNPM_OK;
}

const char *snpm_group_transport(const snpm_group *g)
{
    if (!g) return "";
    return g->transport == 1 ? "loopback" : g_rccl.path.c_str();
}

}  // extern "C"

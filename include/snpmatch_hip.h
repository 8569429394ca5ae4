/*
 * snpmatch_hip.h -- C ABI of libsnpmatch_hip.so, the MI355X (gfx950) scoring engine that replaces
 * the numpy hot path of SNPmatch's Genotyper / CrossIdentifier.
 *
 * Plain C, plain pointers and sizes; no C++ or torch types cross this line.  It is what a ctypes
 * binding inside the reference would call (see INTEGRATION.md for that stub).  Reference
 * interfaces replaced (paths relative to the SNPmatch v5.0.1 tree):
 *
 *   snpm_score_dense_host   <- matchGTsAccs(sampleWei, t1001snps, skip_hets_db)      core/snpmatch.py:74-89
 *   snpm_query_run          <- the chunk loop of Genotyper.genotyper                 core/snpmatch.py:207-225
 *                              (gather g.g.snps[idx,:] :222, matchGTsAccs :223, accumulate :224-225)
 *   snpm_query_run_windows  <- the window loop of CrossIdentifier.window_genotyper   core/csmatch.py:80-90
 *   snpm_likelihood         <- likeliTest + GenotyperOutput.calculate_likelihoods    core/snpmatch.py:40-55,106-117
 *   snpm_binom_identity     <- np_test_identity (binom.sf)                           core/snpmatch.py:57-72
 *   snpm_panel_*            <- HDF5Genotype.snps (int8 [num_snps,num_accessions])    pygwas/genotype.py:534-550
 *                              as an HBM-resident panel fed by a pinned-host staging path
 *
 * Conventions
 *   - every function returns int: SNPM_OK (0) or a negative SNPM_ERR_*; the message is available
 *     from snpm_last_error(ctx) (ctx == NULL: last error of a failed snpm_init in this thread).
 *   - the caller owns every host buffer it passes in (inputs are never modified) and every output
 *     buffer (pre-allocated, C-contiguous); the library owns device memory and pinned staging
 *     buffers behind the opaque handles.  No pointer handed out outlives its handle.
 *   - lifetime: free queries before their panel and panels before their context.  The other orders are
 *     tolerated: snpm_panel_free releases the device memory of the panel's live queries, snpm_destroy that of
 *     the context's live panels and queries; the orphaned handles stay valid for exactly one call, their own
 *     snpm_query_free / snpm_panel_free (a host-side delete), every other entry point refuses them with
 *     SNPM_ERR_STATE.  After the process has started to exit (exit handlers running) free / destroy only drop
 *     host bookkeeping and make no HIP call.
 *   - calls are blocking from the caller's view unless the name says otherwise; one ctx must not
 *     be used from two threads at once; distinct ctx are independent.  One ctx == one GPU; a
 *     multi-GPU job is a snpm_group (below): one process that drives every GPU, or one process per
 *     GPU, each member holding an accession range of the DB, with ONE RCCL all-gather of the
 *     per-accession results inside snpm_group_gather_scores.
 *   - genotype codes: 0 hom-ref, 1 hom-alt, 2 het, negative = missing (stored as -1);
 *     values > 2 are informative but match nothing (stored as 3).
 *   - weights `wei` are float64 [n,3]: column 0 scores db==0, column 1 scores db==2 (het),
 *     column 2 scores db==1, exactly as sampleWei in the reference.
 */
#ifndef SNPMATCH_HIP_H
#define SNPMATCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNPM_OK            0
#define SNPM_ERR_BADARG   -1   /* argument violates the contract (the Python layer raises AssertionError) */
#define SNPM_ERR_HIP      -2   /* a HIP runtime call or kernel launch failed */
#define SNPM_ERR_OOM      -3   /* device or pinned-host allocation failed */
#define SNPM_ERR_STATE    -4   /* handle used in the wrong state */
#define SNPM_ERR_DOMAIN   -5   /* likelihood: a score exceeds its informative count (reference asserts y <= n) */
#define SNPM_ERR_RCCL     -6   /* RCCL could not be loaded, or a communicator / collective call failed */

/* scoring modes of snpm_query_run */
#define SNPM_MODE_EXACT   0    /* fast streaming pass + strict re-evaluation of every accession whose score could
                                  truncate differently from the reference: int(score) and ninfo are bit-exact,
                                  fp64 scores are within the returned error bound of the reference's */
#define SNPM_MODE_STRICT  1    /* reference summation order for every accession: fp64 scores bit-exact */
#define SNPM_MODE_FAST    2    /* fast streaming pass only (benchmarks of the dominant kernel) */

typedef struct snpm_ctx   snpm_ctx;
typedef struct snpm_panel snpm_panel;
typedef struct snpm_query snpm_query;
typedef struct snpm_carry snpm_carry;
typedef struct snpm_group snpm_group;

/* ---------------------------------------------------------------- lifecycle */
int         snpm_version(void);
/* HIP version of the build (HIP_VERSION: major * 10000000 + minor * 100000 + patch); no GPU needed */
int         snpm_hip_build_version(void);
/* 12 hex digits: hash of the sources this library was built from (build_lib.sh).  Recorded measurements (profiles/pmc_traffic.json)
   carry it; bench.py reports such a figure only for the build it was collected on. */
const char *snpm_build_id(void);
int         snpm_device_count(int *count);
int         snpm_init(int device_id, snpm_ctx **out);
int         snpm_destroy(snpm_ctx *ctx);
const char *snpm_last_error(const snpm_ctx *ctx);
/* run the library's kernels on a caller-provided hipStream_t (e.g. torch's current stream);
   NULL restores the library's own stream */
int         snpm_set_stream(snpm_ctx *ctx, void *hip_stream);
int         snpm_synchronize(snpm_ctx *ctx);
/* free / total device memory of the context's GPU in bytes (the host layer plans the residency of a DB with it: int8
   whole -> 2-bit packed whole -> SNP slabs streamed through two half-buffers) */
int         snpm_device_mem_info(snpm_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);

/* ---------------------------------------------------------------- panel (DB genotype matrix in HBM) */
/* SNP-major int8 [n_snp, pitch], pitch = n_acc rounded up to 256 B (to 128 B where that saves 5 % of the row, + 256 B where it
   would be a multiple of 8 KiB: read it with snpm_panel_info / snpm_panel_row_pitch), pad bytes = -1.  1 <= n_acc <= 2^27 (SNPM_ERR_BADARG). */
int snpm_panel_create(snpm_ctx *ctx, int64_t n_snp, int64_t n_acc, snpm_panel **out);
/* Same panel with 2 bits per call (4 accessions per byte: 0 ref, 1 alt, 2 het, 3 missing; rows either whole at a 256-B pitch or
   SPLIT -- the whole 256-B column blocks of a row in a main matrix, its ragged tail at a narrow pitch of its own -- where that saves
   5 % of the row: snpm_panel_row_pitch / snpm_panel_info report main + tail bytes per row; SNPM_PACKED_SPLIT=0 keeps whole rows): 4x less HBM
   capacity and traffic (the 10k x 50M panel is 125 GB and fits one MI355X).  Rows are uploaded as int8
   exactly like an int8 panel and packed on the device; codes > 2 cannot be stored (SNPM_ERR_BADARG from the
   upload).  Every scoring entry point accepts either panel kind and returns identical results. */
int snpm_panel_create_packed(snpm_ctx *ctx, int64_t n_snp, int64_t n_acc, snpm_panel **out);
int snpm_panel_is_packed(const snpm_panel *panel, int *packed);
/* This panel holds only some accession columns (a rank's shard) of a DB of n_acc_total accessions.  Only n_acc_total == 1
   against > 1 matters: for a panel of ONE accession numpy reduces the reference's [1, n] product along a contiguous axis, i.e.
   pairwise inside 8192-element pieces, where every wider panel is summed row after row (core/snpmatch.py:85-87) -- the
   reference-order kernels follow the rule of the width the reference would see.  Default: the panel's own n_acc. */
int snpm_panel_set_total_accessions(snpm_panel *panel, int64_t n_acc_total);
/* Bytes per row a panel of n_acc accessions will have on this context (what snpm_panel_info reports afterwards): for callers
   that size a panel to a memory budget before creating it. */
int snpm_panel_row_pitch(snpm_ctx *ctx, int64_t n_acc, int packed, int64_t *pitch);
int snpm_panel_free(snpm_panel *panel);
int snpm_panel_info(const snpm_panel *panel, int64_t *n_snp, int64_t *n_acc, int64_t *pitch, void **device_ptr);
/* Asynchronous upload of rows [row0, row0+nrows) from host memory (row stride host_pitch bytes,
   >= n_acc): rows are copied into double-buffered pinned staging slabs, sent with hipMemcpyAsync on a side
   stream and written into the panel by a device kernel (the panel's row pitch, codes canonicalised:
   negative -> -1, >2 -> 3; 2-bit packing for packed panels).
   Returns once the last slab is enqueued; scoring calls wait for it on the device. */
int snpm_panel_upload_rows(snpm_panel *panel, int64_t row0, int64_t nrows, const int8_t *host, int64_t host_pitch);
/* Same pipeline fed from a file of tightly packed int8 rows (n_acc bytes per row) that starts at byte
   file_offset (the data section of snps.npy in a native flat panel): pread() straight into the pinned slabs. */
int snpm_panel_load_file(snpm_panel *panel, const char *path, int64_t file_offset, int64_t row0, int64_t nrows);
/* The general file form: the file holds an int8 matrix with file_pitch bytes per row (>= col0 + n_acc) from byte
   file_offset on; panel row row0 + i receives columns [col0, col0 + n_acc) of file row row_idx[i] (row_idx != NULL: the
   matched rows of a sample, the reference's g.g.snps[idx, :], core/snpmatch.py:222) or file_row0 + i.  col0 / file_pitch
   serve an accession shard of a wider DB.  Contiguous reads of >= 1 GiB use O_DIRECT where the file system takes it
   (SNPM_ODIRECT=0/1 overrides); packed panels are packed to 2 bits per call by the host threads that fill the slabs
   (SNPM_HOST_PACK=0: on the device), so a quarter of the bytes cross PCIe. */
int snpm_panel_load_file_rows(snpm_panel *panel, const char *path, int64_t file_offset, int64_t file_pitch, int64_t col0,
                              const int64_t *row_idx, int64_t file_row0, int64_t row0, int64_t nrows);
/* The same from a PACKED flat file (2 bits per call, 4 accessions per byte as in a packed panel; file_pitch = BYTES per file
   row): accessions [acc0, acc0 + n_acc) (acc0 a multiple of 4) of file row row_idx[i] or file_row0 + i.  Either panel kind
   takes it (an int8 panel is unpacked on the device); a quarter of the bytes leave the disk and cross PCIe. */
int snpm_panel_load_file_rows_packed(snpm_panel *panel, const char *path, int64_t file_offset, int64_t file_pitch, int64_t acc0,
                                     const int64_t *row_idx, int64_t file_row0, int64_t row0, int64_t nrows);
/* the loader's host-side packer on its own (no ctx, no GPU): int8 calls [nrows, n_acc] (row stride src_pitch) -> 2 bits
   per call, (n_acc + 3) / 4 bytes per row (row stride dst_pitch; field f of byte b = call 4 b + f: 0 ref, 1 alt, 2 het,
   3 missing = any negative; fields past n_acc are 3).  *bad (may be NULL) = 1 when a call > 2 was met. */
int snpm_pack_rows_host(const int8_t *src, int64_t src_pitch, int64_t nrows, int64_t n_acc, uint8_t *dst, int64_t dst_pitch,
                        int force_scalar, int *bad);
int snpm_panel_upload_wait(snpm_panel *panel);
int snpm_panel_download_rows(snpm_panel *panel, int64_t row0, int64_t nrows, int8_t *host, int64_t host_pitch);
/* Device-side synthetic fill (benchmarks; no PCIe): element (snp, acc) is a pure function of
   (seed, snp0 + row, acc0 + col) with P(-1,0,1,2) = (3277, 39321, 21627, 1311)/65536.
   snpmatch_amd.synth.panel_values() is the numpy twin used by the tests. */
int snpm_panel_fill_synthetic(snpm_panel *panel, uint64_t seed, int64_t snp0, int64_t acc0);
/* the same for panel rows [row0, row0 + nrows) only: row r receives SNP snp0 + (r - row0) of the synthetic panel
   (slab-streamed benchmarks refill a resident buffer with the next SNP slab) */
int snpm_panel_fill_synthetic_rows(snpm_panel *panel, uint64_t seed, int64_t snp0, int64_t acc0, int64_t row0, int64_t nrows);
/* Device-side synthetic sample for benchmarks (the recipe of SURVEY 8d, no host pass over the SNP axis): rows
   [snp0, snp0 + n) of accession `planted` of the synthetic panel `seed`, with a fraction err_permille / 1000 of the
   calls replaced at random; pl_permille / 1000 of the rows get PL-derived weights exp(-PL/10) (integer PL 1..255, 0
   for the called genotype), the others hard 0/1 weights.  d_wei: DEVICE float64 [n, 3].  exp_table: HOST float64
   [256], exp_table[k] = exp(-k / 10) as the caller's libm rounds it (so that a host twin reproduces the bits);
   snpmatch_amd.synth.sample_weights_twin() is that twin. */
int snpm_sample_synthetic(snpm_ctx *ctx, uint64_t seed, int64_t snp0, int64_t n, int64_t planted, int err_permille,
                          int pl_permille, const double *exp_table, void *d_wei);

/* ---------------------------------------------------------------- query (one sample's matched SNPs, resident) */
/* row_idx: int64 [n] panel rows matched by the sample (commonSNPs[0]); NULL = the dense range
   [row0, row0+n).  wei: float64 [n,3] = inputs.wei[commonSNPs[1]].  Both are copied to the device. */
int snpm_query_create(snpm_panel *panel, const int64_t *row_idx, int64_t row0, int64_t n,
                      const double *wei, snpm_query **out);
/* The same query from DEVICE arrays (int64 row list or NULL, float64 [n,3] weights); both are copied.  The row
   indices are not read back: the caller guarantees that they lie inside the panel. */
int snpm_query_create_device(snpm_panel *panel, const void *d_row_idx, int64_t row0, int64_t n, const void *d_wei,
                             snpm_query **out);
int snpm_query_free(snpm_query *query);

/* Genotyper.genotyper over the whole matched list with `chunk`-row matchGTsAccs calls (1000 in the
   reference).  Outputs (host pointers, may be NULL): score float64 [n_acc] (ScoreList before the int
   truncation), ninfo int64 [n_acc].  info (may be NULL), int64 [4]:
   [0] accessions re-evaluated in strict order, [1] 1 if all weights are integers (any order exact),
   [2] how they were re-read (1 accession-major copy, 2 SNP-major strided, 3 every accession: more than 64 flagged);
   the bound itself via snpm_query_error_bound.
   SNPM_MODE_EXACT never waits for the certificate on the host: the accessions the fast pass cannot vouch for
   are flagged by the last reduce kernel, and the re-evaluation kernels queued behind it read that list on the
   device (asking for `info` costs one synchronisation).  */
int snpm_query_run(snpm_query *query, int64_t chunk, int skip_hets, int mode,
                   double *score, int64_t *ninfo, int64_t *info);
/* Same, leaving results in device memory (pointers owned by the query, valid until the next run /
   free): d_score float64 [n_acc], d_ninfo int64 [n_acc].  Work is enqueued on the ctx stream. */
int snpm_query_run_device(snpm_query *query, int64_t chunk, int skip_hets, int mode,
                          void **d_score, void **d_ninfo, int64_t *info);
/* number of accessions the last certified run re-evaluated (synchronises), name of its scoring kernel */
int snpm_query_last_reeval(snpm_query *query, int64_t *n_flagged);
const char *snpm_query_last_kernel(const snpm_query *query);
/* Redirect the results of subsequent runs into caller-owned DEVICE buffers (e.g. torch tensors that
   feed an RCCL all-gather): d_score float64 [n_acc], d_ninfo int64 [n_acc].  NULL, NULL restores the
   query's own buffers. */
int snpm_query_bind_outputs(snpm_query *query, void *d_score, void *d_ninfo);
/* Rigorous bound on |fast-pass score - reference score| used by SNPM_MODE_EXACT for this query. */
int snpm_query_error_bound(snpm_query *query, int64_t chunk, double *bound);

/* CrossIdentifier.window_genotyper: one matchGTsAccs call per window w over matched rows
   [win_off[w], win_off[w+1]) (reference order, fp64 bit-exact).  Outputs (host, may be NULL):
   score float64 [n_win, n_acc], ninfo int64 [n_win, n_acc], totals accumulated window after
   window: tot_score float64 [n_acc], tot_ninfo int64 [n_acc]. */
int snpm_query_run_windows(snpm_query *query, const int64_t *win_off, int64_t n_win, int skip_hets,
                           double *score, int64_t *ninfo, double *tot_score, int64_t *tot_ninfo);

/* The same for one SNP slab of a DB scored slab after slab (slabs hold whole windows): the totals continue in `carry`
   (reference order: the fp64 bits of one pass over all windows; read them with snpm_carry_finish).  The call does NOT
   wait for the device: score / ninfo [n_win, n_acc] should be pinned host memory (snpm_host_alloc) and are complete when
   snpm_carry_finish or snpm_synchronize returns -- the caller loads the next slab meanwhile. */
int snpm_query_run_windows_carry(snpm_query *query, const int64_t *win_off, int64_t n_win, int skip_hets, double *score,
                                 int64_t *ninfo, snpm_carry *carry);

/* The same window loop at streaming speed (one segmented fast pass over all windows + the certificate per
   (window, accession) and for the totals; uncertain entries are re-scored in reference order): int(score), ninfo and
   the totals' counts are bit-exact, fp64 window scores agree with the reference's to ~1e-12 relative (likelihoods
   well inside the 1e-6 of north_star).  snpm_query_run_windows above is the byte-identical mode.
   info (may be NULL) int64 [4]: [0] (window, accession) pairs re-scored, [1] totals re-scored, [2] 1 = fell back to
   the strict pass (more uncertain entries than the sparse tiers take). */
int snpm_query_run_windows_fast(snpm_query *query, const int64_t *win_off, int64_t n_win, int skip_hets,
                                double *score, int64_t *ninfo, double *tot_score, int64_t *tot_ninfo, int64_t *info);

/* ---------------------------------------------------------------- many samples per call */
/* ONE sample against a resident panel in ONE call: Genotyper.genotyper (core/snpmatch.py:207-241: gather of the matched weight rows
   :221, chunk loop :218-225) followed by GenotyperOutput's truncation and likelihoods (:96, :106-117).  row_idx [n]: matched DB rows;
   wei [n_wei, 3]: the sample's weights; sample_idx [n] (or NULL = rows 0..n-1): row of wei for each matched SNP -- the library's host
   threads gather wei[sample_idx] straight into a pinned slab, check the indices and derive the weight properties on the way (no
   device pass, no read-back), the slab goes up, LUT / fast pass / reduce + certificate / gated reference-order tiers / likelihood
   are enqueued back to back, and score, ninfo, lik, lrt [n_acc each] (lik / lrt of the TRUNCATED counts; both NULL = skip) return
   in one copy after one synchronisation.  mode / chunk / skip_hets as snpm_query_run; info (may be NULL) as there.
   SNPM_ERR_BADARG (-> AssertionError) for indices outside the panel / the weights and non-finite weights, SNPM_ERR_DOMAIN when a
   count exceeds its informative sites (the reference's assert, :43). */
int snpm_genotype_once(snpm_panel *panel, const int64_t *row_idx, const double *wei, const int64_t *sample_idx, int64_t n_wei,
                       int64_t n, int64_t chunk, int skip_hets, int mode, double *score, int64_t *ninfo, double *lik,
                       double *lrt, int64_t *info);
/* The same call with DICTIONARY-CODED weights: wei[r, c] = table[codes[3 r + c]], codes uint16 [n_wei, 3], table fp64 [table_len <= 65536]
   (as snpm_score_batch_coded).  A VCF sample's weights are exp(-PL / 10) of small integer PLs (core/parsers.py:141-151): the caller
   computes the table with its own libm, so the device weights carry the fp64 path's bits; the matched rows travel as 32-bit
   indices (n_snp < 2^31), 10 instead of 32 bytes per matched SNP cross PCIe.  A code >= table_len is SNPM_ERR_BADARG. */
int snpm_genotype_once_coded(snpm_panel *panel, const int64_t *row_idx, const uint16_t *codes, const double *table, int64_t table_len,
                             const int64_t *sample_idx, int64_t n_wei, int64_t n, int64_t chunk, int skip_hets, int mode,
                             double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info);

/* B samples against one resident panel.  The reference scores one sample per process (core/snpmatch.py:256-268); here
   sample b owns entries [sample_off[b], sample_off[b+1]) of the concatenated row list (int64, panel rows matched by the
   sample = commonSNPs[0]) and weights (float64 [N,3] = inputs.wei[commonSNPs[1]]); device_inputs != 0: both are DEVICE
   pointers.  One launch scores all samples (sample on the grid), the certificate runs per (sample, accession) on the
   device, likelihood rows [B, n_acc] come from one launch (scores truncated first, as GenotyperOutput does), results
   return in one copy.  Outputs (host, may be NULL; lik and lrt both or neither): score / ninfo / lik / lrt [B, n_acc].
   mode as snpm_query_run: EXACT = int(score) and ninfo bit-exact per sample.
   info (may be NULL) int64 [4]: [0] (sample, accession) pairs re-scored in reference order, [1] 1 = every sample went
   through the strict pass (more uncertain pairs than the sparse tier takes), [2] 1 = scored by the shared-row pass
   (snpm_batch_configure below), [3] its union rows. */
int snpm_score_batch(snpm_panel *panel, int64_t n_samples, const int64_t *sample_off, const void *row_idx, const void *wei,
                     int device_inputs, int64_t chunk, int skip_hets, int mode, double *score, int64_t *ninfo,
                     double *lik, double *lrt, int64_t *info);

/* The same with dictionary-coded weights, wei[r, c] = table[codes[r, c]] (codes uint16 [N, 3], table float64
   [table_len <= 65536], host; codes >= table_len read 0.0): for samples whose weights are exp(-PL/10) of integer PLs
   (core/parsers.py:141-151) the caller fills the table with its own libm, the device weights then carry the bits the
   fp64 path would have received, and 6 + 4 instead of 24 + 8 bytes per matched SNP cross PCIe (the link bounds a batch
   from host memory). */
int snpm_score_batch_coded(snpm_panel *panel, int64_t n_samples, const int64_t *sample_off, const int64_t *row_idx,
                           const uint16_t *codes, const double *table, int64_t table_len, int64_t chunk, int skip_hets,
                           int mode, double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info);

/* Batches whose samples SHARE DB rows (many samples genotyped on largely the same markers, the production use of the reference:
   one `snpmatch inbred` process per sample over the same panel, core/snpmatch.py:218-225 per sample): instead of one gathered pass
   per sample, the union of the batch's matched rows is formed on the device, every DB row is read ONCE and scored against all
   samples as an int8 MFMA contraction -- the weights as fixed-point numbers of `digits` base-256 digits (w in [0, 1] ->
   floor(w 2^F), F = 8 (digits - 1) + 6), the panel rows as one-hot class bytes, products added exactly in int32.  The pass has
   no rounding error, only the one-sided quantisation n_inexact 2^-F per sample; the certificate of SNPM_MODE_EXACT covers it and
   the unproven (sample, accession) pairs are re-scored in reference order as before: int(score) and ninfo stay bit-exact.
   Taken when every sample's row list is strictly increasing, all weights lie in [0, 1] and the panel holds no call code > 2;
   otherwise (and in SNPM_MODE_STRICT) the per-sample pass runs.  The weights are vetted while they are converted, i.e. after
   the pass was taken: a batch with a weight outside [0, 1] is scored AGAIN through the per-sample pass inside the same call
   (results as if the shared-row pass had never run; stats [1] = 4, info[2] = 0).
     shared_rows  -1 (default; SNPM_BATCH_SHARED) automatic: batches whose inputs are already on the device (device_inputs != 0),
                  of at least 4 samples, with at least min_density calls per (sample, union row) slot;  0 never;
                  1 whenever the batch allows it (host batches are uploaded whole first)
     digits       3..7, or -1 (default; SNPM_SHARED_DIGITS=0): the fewest digits that keep a sample's quantisation below 2^-20 --
                  5 up to 262 144 matched SNPs per sample, 6 up to 2^26, else 7; 0 keeps the current value
     min_density  threshold of the automatic choice (default 0.11 on int8, 0.21 on packed panels, 0.5 for fewer than 8 samples; SNPM_SHARED_MIN_DENSITY); negative keeps the current value
   snpm_score_batch[_coded] report in info[2] whether the shared-row pass scored the batch and in info[3] its union rows. */
int snpm_batch_configure(snpm_ctx *ctx, int shared_rows, int digits, double min_density);
/* stats int64 [8] of the context's last snpm_score_batch[_coded] call: [0] 1 = shared-row pass taken, [1] else why not (1 policy,
   2 too few samples or rows, 3 a row list not strictly increasing, 4 a weight outside [0, 1], 5 call codes > 2 in the panel,
   6 overlap below min_density, 7 sizes beyond 32-bit indices or more than 65 535 samples, 8 a row index outside the panel), [2] union rows, [3] density * 1e6, [4] row tiles, [5] groups of
   128 matrix rows, [6] passes over groups, [7] digits */
int snpm_batch_last_stats(snpm_ctx *ctx, int64_t *stats);

/* pinned host memory (hipHostMalloc): batch inputs built in it go to the device at full PCIe speed without the
   staging copy (snpm_score_batch recognises pinned pointers) */
int snpm_host_alloc(snpm_ctx *ctx, int64_t bytes, void **out);
int snpm_host_free(snpm_ctx *ctx, void *ptr);

/* ---------------------------------------------------------------- jobs larger than HBM: SNP slab after SNP slab */
/* The reference walks the whole SNP axis in `chunk`-row pieces and adds every piece onto ScoreList / NumInfoSites
   (core/snpmatch.py:218-225).  When the panel does not fit in HBM the caller loads (or regenerates) one SNP slab
   after the other into a resident panel, makes a query of the slab's rows and weights, and scores it with a
   snpm_carry that holds the running totals:
     STRICT  the chain of additions continues across slabs: fp64 bits of the reference over the whole job;
     EXACT / FAST  fast pass per slab, totals added in slab order; the error bounds of the slabs add up
             (chunks_after = number of `chunk`-row pieces in the slabs still to come, they lengthen the reference's
             chain) and snpm_carry_finish certifies the TOTALS: it returns the accessions whose int(score) is not
             yet proven.  For those the caller streams the slabs once more through a second, column-list carry
             (snpm_carry_set_columns + snpm_query_run_carry in STRICT mode) and patches the result in
             (snpm_carry_patch); more than 64 flagged: a second pass in STRICT mode over all accessions.
   Every slab but the last must hold a multiple of `chunk` rows.  snpmatch_amd.engine.SlabScorer drives this. */
int snpm_carry_create(snpm_ctx *ctx, int64_t n_acc, snpm_carry **out);
int snpm_carry_reset(snpm_carry *carry);
int snpm_carry_free(snpm_carry *carry);
/* keep the totals in caller-owned DEVICE buffers (float64 [n_acc], int64 [n_acc]); resets the carry */
int snpm_carry_bind_outputs(snpm_carry *carry, void *d_score, void *d_ninfo);
int snpm_carry_set_columns(snpm_carry *carry, const int32_t *cols, int64_t ncols);
int snpm_query_run_carry(snpm_query *query, int64_t chunk, int skip_hets, int mode, int64_t chunks_after,
                         snpm_carry *carry);
/* score / ninfo (host, may be NULL): the totals; flagged [cap] / n_flagged: see above (0 unless EXACT) */
int snpm_carry_finish(snpm_carry *carry, double *score, int64_t *ninfo, int32_t *flagged, int64_t cap,
                      int64_t *n_flagged);
int snpm_carry_patch(snpm_carry *totals, const snpm_carry *cols_pass);
/* Rigorous bound on |total - reference total| of the slabs scored so far in SNPM_MODE_EXACT (the sum of the slabs'
   bounds -- reference-order term of every slab, integer-weight slabs included, plus the fast-pass term of the
   non-integer ones -- and of the slab-order additions); 0 when every slab had integer weights or in other modes. */
int snpm_carry_error_bound(snpm_carry *carry, double *bound);
/* device pointers of the totals (float64 [n_acc], int64 [n_acc]): input of snpm_likelihood_device / an all-gather */
int snpm_carry_device_ptrs(snpm_carry *carry, void **d_score, void **d_ninfo);

/* ---------------------------------------------------------------- multi-GPU: accession shards + one RCCL all-gather */
/* Every reduction of matchGTsAccs runs over SNPs (core/snpmatch.py:84-88): accession columns never interact, so
   member r of a group of R GPUs holds columns [a0_r, a1_r) of every SNP row (snpm_group_shard; boundaries are
   multiples of 4 accessions), scores them with the entry points above on its own context, and the only exchange is
   the gather of (score float64, ninfo int64) per accession -- the likelihood step needs the minimum over ALL
   accessions (core/snpmatch.py:112).  Two ways to form a group:
     snpm_group_create_local   ONE process drives n GPUs (ncclCommInitAll, no launcher): the group creates the n
                               contexts (snpm_group_ctx) and destroys them with snpm_group_free;
     snpm_group_create_rank    one process per GPU: rank 0 calls snpm_group_unique_id, hands the 128 bytes to the
                               other ranks by any channel (file, environment, socket), every rank then joins with
                               its own context (ncclCommInitRank).
   RCCL is bound at the first group call (dlopen: the copy already in the process, else the one beside the HIP
   runtime in use, else librccl.so.1; SNPMATCH_RCCL_LIB overrides); SNPM_ERR_RCCL when none loads.
   Errors of group calls: snpm_group_last_error(group) (group == NULL: last failed creation in this thread). */
#define SNPM_GROUP_ID_BYTES 128
#define SNPM_GROUP_LOOPBACK 1   /* snpm_group_create_local flag: exchange by device-to-device copies instead of RCCL; a TEST
                                   transport (it also accepts one device several times: rehearsal on a one-GPU box) */
int snpm_group_unique_id(void *id_bytes /* [SNPM_GROUP_ID_BYTES] */);
int snpm_group_create_rank(snpm_ctx *ctx, const void *id_bytes, int world, int rank, snpm_group **out);
int snpm_group_create_local(const int *device_ids, int n, int flags, snpm_group **out);
int snpm_group_free(snpm_group *group);
const char *snpm_group_last_error(const snpm_group *group);
/* world = ranks of the job, rank0 = global rank of local member 0, n_local = members this process drives */
int snpm_group_info(const snpm_group *group, int *world, int *rank0, int *n_local);
int snpm_group_ctx(snpm_group *group, int member, snpm_ctx **ctx);
/* accession range [a0, a1) of global rank `rank` in a DB of n_acc accessions (equal shards of ceil(n_acc / world)
   rounded up to 4; the last ranks may be short or empty) */
int snpm_group_shard(const snpm_group *group, int64_t n_acc, int rank, int64_t *a0, int64_t *a1);
/* The collective.  d_score[i] / d_ninfo[i] (i < n_local): DEVICE results of local member i's shard, float64 / int64
   [m, a1_i - a0_i] with row stride in_ld elements (m = 1: genome-wide totals, e.g. the pointers of
   snpm_query_run_device / snpm_carry_device_ptrs; m = n_win: window rows), produced by work queued on that member's
   stream.  Per member one pack kernel, ONE ncclAllGather (16 * m * per bytes per rank) and one unpack kernel run on
   that stream; with lik / lrt the likelihood rows over all accessions follow on member 0 (`truncate` as in
   snpm_likelihood).  Host outputs [m, n_acc] (any may be NULL; lik and lrt both or neither) come from member 0 and
   the call waits for them; with every host output NULL nothing waits on the host.  Every rank of the job makes the
   same call (same m, n_acc). */
int snpm_group_gather_scores(snpm_group *group, const void *const *d_score, const void *const *d_ninfo, int64_t m,
                             int64_t n_acc, int64_t in_ld, int truncate, double *score, int64_t *ninfo, double *lik,
                             double *lrt);
/* device arrays the last gather left on a local member: float64 / int64 [m, n_acc] (valid until the next gather) */
int snpm_group_gathered_ptrs(snpm_group *group, int member, void **d_score_all, void **d_ninfo_all);
/* path of the RCCL library in use, or "loopback" */
const char *snpm_group_transport(const snpm_group *group);

/* ---------------------------------------------------------------- one-shot forms */
/* matchGTsAccs on host arrays: db int8 [n, n_acc] (row stride db_pitch), wei float64 [n,3].
   fp64 bit-exact with the reference (strict order). */
int snpm_score_dense_host(snpm_ctx *ctx, const int8_t *db, int64_t db_pitch, int64_t n, int64_t n_acc,
                          const double *wei, int skip_hets, double *score, int64_t *ninfo);

/* likeliTest over rows: y float64 [m, len] (matches; truncated toward zero first when
   truncate != 0, as GenotyperOutput does for inbred), n int64 [m, len].  Per row r:
   lik[r,:] = likeliTest, lrt[r,:] = lik / nanmin(lik[r,:]) (amin_or_nan: use this TopHit
   instead when it is not NaN).  Host pointers.  SNPM_ERR_DOMAIN if some y > n. */
int snpm_likelihood(snpm_ctx *ctx, const double *y, const int64_t *n, int64_t m, int64_t len,
                    int truncate, double amin_or_nan, double *lik, double *lrt);
/* device-pointer form used after snpm_query_run_device (m = 1) */
int snpm_likelihood_device(snpm_ctx *ctx, const void *d_y, const void *d_n, int64_t m, int64_t len,
                           int truncate, double amin_or_nan, void *d_lik, void *d_lrt, int *domain_error);

/* np_test_identity on the device: out[i] = (binom.sf((n[i]-x[i]) - 1, n[i], error_rate) >= pthres).  Host
   pointers; sf (may be NULL) receives the survival function values. */
int snpm_binom_identity(snpm_ctx *ctx, const double *x, const int64_t *n, int64_t len, double error_rate,
                        double pthres, int64_t *out, double *sf);
/* host twin of the device arithmetic of snpm_binom_identity: sf[i] = binom.sf(k[i], n[i], p) (no ctx, no GPU) */
int snpm_binom_sf_host(const double *k, const double *n, int64_t len, double p, double *sf);

/* ---------------------------------------------------------------- caller-side index preparation */
/* Sorted-merge intersection of two strictly increasing int64 arrays (the positions of one chromosome in the
   DB and in the sample): ia / ib receive the indices of the common values (capacity min(na, nb)), *n_out
   their number.  Replaces the two np.in1d calls per chromosome of Genotype.get_common_positions
   (core/snp_genotype.py:66-67).  Pure host code, needs no ctx / GPU.  SNPM_ERR_STATE when an input is not
   strictly increasing (the caller then takes its generic path). */
int snpm_intersect_sorted(const int64_t *a, int64_t na, const int64_t *b, int64_t nb, int64_t *ia, int64_t *ib,
                          int64_t *n_out);
/* Same result for a short list b against a long list a: galloping search, O(nb log(na / nb)); from 4096 values of b on the search
   runs in ranges of b on a pool of host threads.  Capacity of ia / ib: min(na, nb), as above.  a must be
   strictly increasing and is NOT checked here (DB positions are verified once, when the DB is opened);
   b is checked (SNPM_ERR_STATE). */
int snpm_intersect_sorted_search(const int64_t *a, int64_t na, const int64_t *b, int64_t nb, int64_t *ia, int64_t *ib,
                                 int64_t *n_out);
/* Genotype.identify_segregating_snps on the resident panel (core/snp_genotype.py:188-211, used by --refine):
   mask [n_snp] (host, uint8) = 1 where the informative calls of accessions cols[0..ncols) are not all equal. */
int snpm_panel_segregating(snpm_panel *panel, const int32_t *cols, int64_t ncols, uint8_t *mask);
/* the same scan for an accession-SHARDED DB (ncols may be 0): besides the local mask, first [n_snp] receives the
   first informative call of the listed local accessions in every row (0xFF = none); a row segregates when some
   rank's mask is set or two ranks report different calls (snpmatch_amd.dist) */
int snpm_panel_segregating_first(snpm_panel *panel, const int32_t *cols, int64_t ncols, uint8_t *mask, uint8_t *first);
/* calls of the listed accessions at the query's matched rows, codes [ncols, n] (uint8: 0 ref, 1 alt, 2 het, 3 other
   code, 0xFF missing; host): the g_acc.snps[:, i] column reads of the reference (core/csmatch.py:116-117), used to
   bring the columns of an accession-sharded DB together for the in-silico crosses */
int snpm_query_gather_columns(snpm_query *query, const int32_t *acc_idx, int ncols, uint8_t *codes);
/* CrossIdentifier.match_insilico_f1s (core/csmatch.py:115-125): every pair (i < j, in the order of
   itertools.combinations) of the n_sel (<= 32) accessions acc_idx is crossed in silico over the query's
   matched SNPs.  Per pair: "alt" SNPs (both calls 1) take W[:, 2], "ref" SNPs (both 0) take W[:, 0], SNPs
   where both calls are informative and differ take W[:, 1];
       score = (np.sum(alt weights) + np.sum(ref weights)) + np.sum(het weights),  ninfo = the three counts.
   The reference prints these scores as floats, so each np.sum is evaluated in numpy's order (pairwise
   summation inside 8192-element chunks, chunks added in sequence): fp64 bit patterns of the reference.
   score / ninfo: host arrays [n_sel * (n_sel - 1) / 2].  DB codes other than -1, 0, 1, 2 count as one
   further class (differs from 0/1/2, equal to itself). */
int snpm_query_f1_pairs(snpm_query *query, const int32_t *acc_idx, int n_sel, double *score, int64_t *ninfo);

/* ---------------------------------------------------------------- profiling (HIP events on the ctx stream) */
/* PMC calibration: reads the whole panel once with the access shape of the scoring kernel (4 B per
   lane, non-temporal); *bytes_read = n_snp * pitch.  Used by tools/pmc_traffic.py to calibrate
   FETCH_SIZE on a known byte count. */
int snpm_debug_stream_read(snpm_panel *panel, int64_t *bytes_read);
int snpm_profile_enable(snpm_ctx *ctx, int on);
int snpm_profile_reset(snpm_ctx *ctx);
/* kernel: "fast", "strict", "reduce", "scan", "likelihood", "synth", "lut".  Synchronises the stream. */
int snpm_profile_read(snpm_ctx *ctx, const char *kernel, int64_t *launches, double *total_ms);

/* ---------------------------------------------------------------- sample input: VCF text (host only, no GPU) */
/* Single pass over a (plain or gzip) VCF: what ParseInputs.read_vcf (core/parsers.py:178-213, scikit-allel in
   the reference) extracts for sample column `sample_index`: CHROM, POS, the GT text as written, the first three
   PL values (-1 where absent) and INFO/DP (-1 where absent).  The calling thread reads (and inflates) the file in blocks of
   whole lines, a team of threads (the process's cores, at most 16; SNPM_VCF_THREADS) parses the blocks, file order is kept.
   SNPM_ERR_STATE = the file holds something this reader does not want to interpret (odd numbers, CRLF, very wide records):
   use the generic reader. */
typedef struct snpm_vcf snpm_vcf;
int snpm_vcf_parse(const char *path, int sample_index, snpm_vcf **out);
/* flags: bit 0 some FORMAT has GT, bit 1 some record has PL, bit 2 some record has INFO/DP, bit 3 CHROM / GT text is pure ASCII */
int snpm_vcf_dims(const snpm_vcf *vcf, int64_t *n_records, int *chr_width, int *gt_width, int *flags, int *n_samples);
/* chr [n * chr_width] and gt [n * gt_width]: NUL-padded fixed-width bytes; pos [n]; pl [n * 3]; dp [n] */
int snpm_vcf_fill(const snpm_vcf *vcf, char *chr, int64_t *pos, char *gt, double *pl, int64_t *dp);
/* The same with CHROM / GT as fixed-width UTF-32 (numpy '<U{width}': one code point per uint32_t, zero padded: no per-string
   conversion on the Python side) and called [n] = 0 where the genotype is './.' or '.|.' (the records ParseInputs.read_vcf
   drops, core/parsers.py:141-157).  SNPM_ERR_STATE when the text is not ASCII (flag bit 3). */
int snpm_vcf_fill_u32(const snpm_vcf *vcf, uint32_t *chr, int64_t *pos, uint32_t *gt, double *pl, int64_t *dp, uint8_t *called);
const char *snpm_vcf_sample_name(const snpm_vcf *vcf, int i);
int snpm_vcf_free(snpm_vcf *vcf);

/* ---------------------------------------------------------------- DB input: the reference's HDF5 files (host only, no GPU) */
/* A reader for the files the reference keeps its DBs in -- `snps` int8 [num_snps, num_accessions] in lzf chunks of (1000,
   num_accessions), `positions` with the attributes `chrs` / `chr_regions`, `accessions` (pygwas/genotype.py:310-326), and the
   accession-major twin in gzip chunks (core/makedb.py:64-81) -- for hosts without h5py / libhdf5.  Scope: files of the default
   ("earliest") format of HDF5 1.8 / 1.10: superblock 0-3 with version-1 object headers, old-style groups, compact / contiguous /
   chunked (version-1 B-tree) layouts, filters gzip, shuffle, lzf, integer / float / fixed- and variable-length string data,
   little-endian.  Anything else returns SNPM_ERR_BADARG with a message in snpm_h5_last_error(file) (file == NULL: of the last
   failed snpm_h5_open in this thread).  snpm_h5 handles are independent of contexts; reads are thread-safe. */
typedef struct snpm_h5 snpm_h5;
int snpm_h5_open(const char *path, snpm_h5 **out);
int snpm_h5_close(snpm_h5 *file);
const char *snpm_h5_last_error(const snpm_h5 *file);
/* member names of a group ("" = root), '\n'-separated; *needed = bytes incl. the terminating NUL */
int snpm_h5_list(snpm_h5 *file, const char *group, char *buf, int64_t cap, int64_t *needed);
/* object `path` (attr == NULL) or its attribute `attr`: kind 0 group / 1 data; rank, dims [8], type_class 0 integer / 1 float /
   3 string (variable-length strings are presented as fixed-length ones of elem_size = the longest), elem_size bytes, chunk [8]
   (0s: not chunked), number of attributes.  Any output may be NULL. */
int snpm_h5_info(snpm_h5 *file, const char *path, const char *attr, int *kind, int *rank, int64_t *dims, int *type_class,
                 int *elem_size, int *is_signed, int64_t *chunk, int *n_attrs);
int snpm_h5_attr_name(snpm_h5 *file, const char *path, int index, char *buf, int64_t cap);
/* the whole dataset / attribute in C order; out_bytes = elements * elem_size as snpm_h5_info reports them */
int snpm_h5_read(snpm_h5 *file, const char *path, const char *attr, void *out, int64_t out_bytes);
/* rows row_idx[i] (or file_row0 + i when row_idx is NULL), columns [col0, col0 + ncols) of a 1-D / 2-D dataset of fixed-size
   elements (the reference's g.g.snps[idx, :] / g.g_acc.snps[:, i]) -> out, row stride out_pitch BYTES */
int snpm_h5_read_rows(snpm_h5 *file, const char *path, const int64_t *row_idx, int64_t file_row0, int64_t nrows, int64_t col0,
                      int64_t ncols, void *out, int64_t out_pitch);
/* snpm_panel_load_file_rows for a 2-D int8 dataset of an open HDF5 file: chunks are read and decompressed by the loader's host
   threads straight into the pinned staging slabs (packed panels: packed there too) */
int snpm_panel_load_h5(snpm_panel *panel, snpm_h5 *file, const char *dataset, int64_t col0, const int64_t *row_idx,
                       int64_t file_row0, int64_t row0, int64_t nrows);

#ifdef __cplusplus
}
#endif
#endif /* SNPMATCH_HIP_H */

// snpm_h5.hpp -- internal interface between the HDF5 reader (snpm_h5.cpp, host only) and the panel loader (snpm_loader.hpp)
#pragma once
#include <stdint.h>

struct snpm_h5;

// the 2-D int8 dataset `path` (the reference's `snps`): opaque handle valid while the file is open, NULL on error (message in
// snpm_h5_last_error(f)); chunk_rows = rows per chunk (0: not chunked)
const void *snpm_h5_int8_matrix(snpm_h5 *f, const char *path, int64_t *n_rows, int64_t *n_cols, int64_t *chunk_rows);
// rows row_idx[i] (or file_row0 + i), columns [col0, col0 + ncols) -> out (row stride out_pitch bytes).  Thread-safe: every
// thread keeps its own decompressed chunk.  Errors: SNPM_ERR_* with the message in snpm_h5_thread_error() of the CALLING thread.
int snpm_h5_rows_raw(snpm_h5 *f, const void *dataset, const int64_t *row_idx, int64_t file_row0, int64_t nrows, int64_t col0,
                     int64_t ncols, int8_t *out, int64_t out_pitch);
const char *snpm_h5_thread_error();

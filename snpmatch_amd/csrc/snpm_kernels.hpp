// snpm_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, wave64) behind libsnpmatch_hip.so.
//
// The work is compare-and-count over an int8 SNP x accession panel: HBM-bound byte streaming,
// no MFMA.  Layout: panel is SNP-major [n_snp, pitch] (pitch = n_acc rounded up to 256 B), so one
// SNP row is a coalesced run of accession bytes and the three weights of a row are wave-uniform.
//
// Kernels
//   k_fast     dominant kernel.  Each lane owns BPL adjacent accessions (one 4/8/16-byte load per
//              row), each wave 64*BPL adjacent accession bytes, each workgroup a run of rows (a "part").
//              Per-row weights live in LDS as a 4-entry fp64 LUT {ref, alt, het, 0} indexed by
//              (byte & 3); every element costs one ds_read_b64 + one v_add_f64 instead of three
//              compare/select pairs.  Missing counts are SWAR (packed u8 lanes).  Partials per part
//              are written once; k_reduce adds them in part order (deterministic, no atomics).
//   k_fast_packed_q4 the fast pass on a 2-bit packed panel: 16 accessions per lane, four rows per table lookup
//              (one ds_read_b128 + two v_add_f64 per two comparisons), bit-sliced missing counts, LDS reads
//              issued and waited for by hand.
//   k_fast_bits  hard-call samples (all weights 0 or 1) on a packed panel: bit-plane boolean scoring and
//              bit-sliced counting, no LDS, no fp64.
//   k_strict4 / k_strict / k_strict_sparse(_T)  reference summation order (three per-category sequential fp64
//              sums per segment, core/snpmatch.py:85-87).  Used for cross windows, for SNPM_MODE_STRICT and to
//              re-evaluate the few accessions the fast pass cannot certify (sparse variants; _T reads the
//              accession-major packed copy built by k_pack_transpose).
//   k_scan / k_scan_few  sequential accumulation of segment sums (ScoreList += chunk, core/snpmatch.py:224), with an
//              optional carry-in (totals of earlier SNP slabs).
//   k_fast<..., SEG>  the same fast pass over many independent row ranges (samples of a batch, windows of a cross) in one
//              launch; k_reduce_seg / k_eseg_part + k_eseg_finish / k_strict_pairs / k_scan_pairs: per-segment reduce, error bound, and the
//              reference-order re-evaluation of the (segment, accession) pairs the certificate flags.
//   certificate  k_wprops (weight properties), k_eref / k_efinish (reference-order error bound on the device),
//              flag_if_uncertain inside k_reduce / k_carry_flag; the re-evaluation kernels read the flag count on the
//              device and leave at once when their tier has nothing to do (dense_tier_off, *d_ncols).
//   k_carry_add / k_carry_flag / k_tot_seg  running totals of slab-streamed jobs, totals over windows.
//   k_likelihood  likeliTest + nanmin + ratio on device (core/snpmatch.py:40-55,106-117).
//   k_binom_identity  np_test_identity (core/snpmatch.py:57-72).   k_segregating  --refine support.
//   k_f1_*     in-silico F1 scores in numpy's summation order (core/csmatch.py:115-125).
//   k_build_lut, k_repitch_canon / k_pack_rows / k_unpack_rows (upload / download), k_pack_transpose[_packed]
//   (accession-major copies), k_synth* / k_synth_sample (benchmark data), k_check_rows, k_expand_codes, k_seg_pack,
//   k_patch, k_calib_read: small helpers.
//
// The kernels live in one header per family; this file only puts them together (order matters: later families use the
// constants and helpers of earlier ones):
#pragma once
#include "snpm_k_common.hpp"      // build switches, constants
#include "snpm_k_prep.hpp"        // k_build_lut, k_wprops, k_wbits, k_eref / k_efinish, k_check_rows, k_expand_codes
#include "snpm_k_fast.hpp"        // k_fast
#include "snpm_k_packed.hpp"      // k_fast_packed_q4, k_fast_bits
#include "snpm_k_reduce.hpp"      // k_reduce*, k_carry_*, k_eseg_*, k_reduce_seg, k_strict_pairs, k_scan_pairs, k_tot_seg, helpers
#include "snpm_k_strict.hpp"      // k_strict, k_strict4, k_strict_sparse(_T), k_pack_transpose*, k_scan, k_scan_few, k_seg_pack, k_patch
#include "snpm_k_shared.hpp"      // k_sh_*: the shared-row scan of a batch (int8 MFMA contraction of fixed-point weight digits with the one-hot panel)
#include "snpm_k_post.hpp"        // k_likelihood, k_binom_identity, k_segregating, k_f1_*, k_once_pack
#include "snpm_k_io.hpp"          // k_pack_rows, k_repitch_canon, k_unpack_rows, k_synth*, k_calib_read
#include "snpm_kernels_single.hpp"   // k_strict_single (panels of one accession: numpy's pairwise order)

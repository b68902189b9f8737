/*
 * hicmi.h - C ABI of libhicmi.so: the MI355X (gfx950) hot path of Hi-C contact-map clustering
 * (Part 1) and scaffold ordering (Part 2).
 *
 * The reference (AO33/HiC_Genome_Assembler) is pure Python and has no FFI of its own
 * (SURVEY.md section 8b); the boundary a maintainer would bind is therefore defined here, one entry
 * point per native routine the reference reaches through NumPy / SciPy / Numba.  Each
 * declaration cites the reference call site it replaces (paths relative to
 * HIC_ASSEMBLER/; S2C = scaffoldToChromosomes.py, OG = orderGenome.py).  INTEGRATION.md shows
 * the ctypes stub for each.
 *
 * Conventions: every function returns 0 on success and a negative HICMI_E* code on failure;
 * hicmi_last_error() returns a message for the calling thread.  Host buffers are caller-owned,
 * plain pointers and sizes only.  A context belongs to one host thread and one GPU; all work
 * is issued on the context's own HIP stream and every call that returns host data has
 * synchronised that stream before returning.  Matrices are row-major.
 *
 * There is NO CPU fallback anywhere in this library: without a HIP device hicmi_create fails.
 */
#ifndef HICMI_H
#define HICMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HICMI_ABI_VERSION 1

#define HICMI_OK            0
#define HICMI_EINVAL       -1   /* bad argument / call order */
#define HICMI_EHIP         -2   /* HIP runtime error (message has the hipError string) */
#define HICMI_ENOMEM       -3
#define HICMI_ESTATE       -4   /* kernel reported an internal inconsistency (e.g. nn-chain guard) */
#define HICMI_EUNSUPPORTED -5   /* e.g. more than 65536 bins (rank matrix is uint16 in this version) */

typedef struct hicmi_ctx hicmi_ctx;

int         hicmi_abi_version(void);
const char *hicmi_last_error(void);
int         hicmi_device_count(int *count);

/* One context per GPU / per process rank. */
int hicmi_create(int device, hicmi_ctx **out);
int hicmi_destroy(hicmi_ctx *ctx);
/* The HIP stream all kernels of this context are launched on (hipStream_t as void*), so a caller
 * can bracket launches with its own events. */
int hicmi_stream(hicmi_ctx *ctx, void **stream_out);
int hicmi_synchronize(hicmi_ctx *ctx);

/* ---- contact matrix -------------------------------------------------------------------------
 * Replaces the dense fp64 matrix built by buildAdjacencyMatrix (S2C:70-98, OG:65-93).
 * _host copies n*n doubles to the GPU; _device adopts a caller-owned device allocation (leading
 * dimension ld >= n, in elements) without copying - the caller keeps it alive and unchanged. */
int hicmi_set_contacts_host(hicmi_ctx *ctx, const double *contacts, int64_t n);
/* The same from an fp32 host matrix (BASELINE.json configs[4]: 64,000 bins stored as fp32): half the host memory and
 * PCIe traffic; the values are widened to fp64 on the device and every stage computes on exactly those values. */
int hicmi_set_contacts_host_f32(hicmi_ctx *ctx, const float *contacts, int64_t n);
int hicmi_set_contacts_device(hicmi_ctx *ctx, const double *d_contacts, int64_t n, int64_t ld);
/* Device address, size and leading dimension of the context's contact matrix, so that further
 * contexts on the same GPU (e.g. one per chromosome worker thread in Part 2, chromosomes being
 * independent - OG:608-612) can adopt it with hicmi_set_contacts_device instead of copying it. */
int hicmi_contacts_device(hicmi_ctx *ctx, void **d_contacts_out, int64_t *n_out, int64_t *ld_out);

/* Host-side loader (no GPU involved): buildAdjacencyMatrix's triplet parse (S2C:70-98, OG:65-93).
 * `path`: HiC-Pro matrix file, lines "id1<TAB>id2<TAB>value"; bin_ids[0..n): the bin ID of every
 * row; out: n*n doubles (zero-filled here).  Triplets naming an unknown ID are skipped, each one sets
 * [i][j] and [j][i], a cell named twice keeps the later line's value, values are converted with
 * correct rounding (as Python's float()).  threads <= 0: all hardware threads.  A malformed line is
 * an error (the reference raises there).  edges_out (may be NULL): triplets used. */
int hicmi_load_hicpro_matrix(const char *path, const int64_t *bin_ids, int64_t n, double *out, int threads,
                             int64_t *edges_out);

/* Row sums, both flavours the reference uses:
 *   np_sum[i]  = row.sum() as NumPy reduces it (pairwise, 8192-element chunks)   S2C:112, S2C:147
 *   seq_sum[i] = builtin sum() left to right                                     S2C:134
 * Either output may be NULL.  Results are also kept on the device for the later stages. */
int hicmi_row_sums(hicmi_ctx *ctx, double *np_sum, double *seq_sum);

/* One map over several GPUs (first bullet): the row-independent stages - row sums (S2C:112,134), the
 * per-row argsort (S2C:1132) and the per-row counts of the cut and filter scans (S2C:455-459, 622-636) - are computed
 * only for the rows first, first + stride, first + 2 stride, ... of this context (a CYCLIC row partition: the scans
 * read a triangle of the rank matrix, contiguous blocks would leave the last rank with most of it).  Afterwards
 *   hicmi_row_sums      fills only the owned entries (0 elsewhere) until hicmi_set_row_sums installs the gathered vectors,
 *   hicmi_rank_matrix   sorts and inverts only the owned rows,
 *   hicmi_cut_scan / hicmi_filter_scan  return counts and flags of the owned rows, 0 for the others;
 * the caller all-gathers the owned entries across the ranks (hic_genome_assembler_amd/dist.py: RCCL through
 * torch.distributed).  first = 0, stride = 1 (the default) is the whole matrix. */
int hicmi_set_row_shard(hicmi_ctx *ctx, int64_t first, int64_t stride);
int hicmi_set_row_sums(hicmi_ctx *ctx, const double *np_sum, const double *seq_sum);

/* removeRows (S2C:100-136): keep only rows/columns keep[0..n_keep) (ascending); recomputes both
 * row sums on the compacted matrix.  The compacted copy is owned by the context (an adopted
 * device matrix is left untouched). */
int hicmi_compact(hicmi_ctx *ctx, const int32_t *keep, int64_t n_keep);

/* ---- Part 1: clustering ---------------------------------------------------------------------
 * convertMatrix(distance) + squareform + scipy average + dendrogram leaf order
 * (S2C:138-155, S2C:187-208).  Z_out: (n-1) x 4 doubles in SciPy's linkage convention (may be
 * NULL); leaves_out: n int32 (dendrogram(count_sort='ascending')['leaves']). */
int hicmi_upgma(hicmi_ctx *ctx, double *Z_out, int32_t *leaves_out);

/* reorderMatrix + convertMatrix(similarity) + numpy.argsort(axis=1)[:, ::-1]
 * (S2C:157-163, S2C:149, S2C:1132) for the row/column order `order` (n int32, normally the leaves).
 * Builds, on the device, R[a][k] = column with the k-th largest similarity in row a (ties:
 * larger column index first) and its inverse rank[a][b] = position of column b in row a. */
int hicmi_rank_matrix(hicmi_ctx *ctx, const int32_t *order);
/* How the last hicmi_rank_matrix got its rows.  hicmi_upgma starts, on a second stream beside the nn-chain, a sort of
 * every row in STORAGE numbering: a row without equal similarities has the same sorted sequence under any numbering,
 * so hicmi_rank_matrix only re-addresses it by `order`; in a row that does hold equal similarities the order inside each
 * run of equal values depends on the labels, and is made afterwards by a cheaper sort of (run, label) keys.
 * Under a row shard the pre-sort covers ALL rows (the GPU is idle during the replicated chain) and only the own rows are
 * re-addressed.  *state_out: 0 = all rows sorted by hicmi_rank_matrix itself (no hicmi_upgma before it, n < 2048 or
 * HICMI_NO_PRESORT=1), 1 = pre-sorted rows used, *tied_rows_out (may be NULL) of all rows held equal similarities,
 * 2 = pre-sort discarded (only with HICMI_PRESORT_TIES=resort, the A/B mode that re-sorts tied rows in full). */
int hicmi_presort_state(hicmi_ctx *ctx, int *state_out, int64_t *tied_rows_out);
/* Copy rows [row0, row0+nrows) of R (or of its inverse when inverse != 0) to the host as uint16. */
int hicmi_get_rank_rows(hicmi_ctx *ctx, int64_t row0, int64_t nrows, int inverse, uint16_t *out);
/* Similarity values of one reordered row (S2C:149), for tests. */
int hicmi_get_similarity_row(hicmi_ctx *ctx, int64_t row, double *out);

/* ---- Part 1: hypergeometric cut scan ---------------------------------------------------------
 * find_matrix_pvalue_breakpoints inner loop (S2C:449-469) for one `start`:
 *   x[i-start] = #{ v in R[i][0 : i-start] : start <= v <= i }           for i in (start, n)
 *   sig[i-start] = 0 if hyper_geom(x, M, i-start, i-start) >= psig else 1   (NaN counts as 1)
 * and x[0] = sig[0] = 0 (S2C:448-452).  x_out / sig_out hold n-start entries; either may be NULL.
 * The counts of the last `start` are cached, so the M-rescan of S2C:473-477 costs no matrix pass. */
int hicmi_cut_scan(hicmi_ctx *ctx, int64_t start, int64_t M, double psig, int32_t *x_out, uint8_t *sig_out);

/* filter_noisy_breakpoints row tests (S2C:622-636) for one (start, c):
 *   rows ii in [start, start+n_rows):  x = #{ v in R[ii][0 : c-start] : start <= v <= c }
 *   sig = 1 if hyper_geom(x, M, c-start, c-start) < psig else 0            (NaN counts as 0) */
int hicmi_filter_scan(hicmi_ctx *ctx, int64_t start, int64_t c, int64_t n_rows, int64_t M, double psig,
                      int32_t *x_out, uint8_t *sig_out);

/* The two scan loops as a whole, with their control flow on the device: every scan's arguments come out of the previous
 * scan's flags, so driven from the host (hicmi_cut_scan / hicmi_filter_scan in a Python loop) each of the ~230 scans of
 * a 16k map pays a launch, a download and a decision on top of its few tens of microseconds of device work.  Here the
 * decisions are kernels too (k_part1_scan.hip), the host enqueues scans in batches and reads one small record per batch.
 * Not available on a row shard (the flags of every scan would have to be gathered): the per-scan calls remain for that.
 *
 * hicmi_first_pass_cuts = pre_process_all_matrix_breakpoints (S2C:513-551) with find_matrix_pvalue_breakpoints
 * (S2C:413-511) inside: min_size >= 1; stop_ind = int(n - n * min_frac), computed by the caller; psig as the caller
 * passes it (the reference passes the literal .05, S2C:535).  cuts_out: the cut indices in the order found.  m_log_out:
 * pairs (M before, M after) of every "M value (world_size) changed" event (S2C:473-483), in order - the reference
 * prints them.
 * hicmi_filter_cuts = filter_noisy_breakpoints (S2C:553-727): cuts_in ascending; cuts_out = sorted(filtered);
 * *warned_out = how many times the "maximum number of rounds" warning (S2C:592-595) was reached. */
int hicmi_first_pass_cuts(hicmi_ctx *ctx, int64_t min_size, int64_t stop_ind, double psig, int32_t *cuts_out,
                          int64_t cuts_cap, int64_t *n_cuts_out, int32_t *m_log_out, int64_t log_cap, int64_t *n_log_out);
int hicmi_filter_cuts(hicmi_ctx *ctx, const int32_t *cuts_in, int64_t n_in, double psig, int32_t *cuts_out,
                      int64_t cuts_cap, int64_t *n_out, int64_t *warned_out);

/* hyper_geom (S2C:352-368) = scipy.stats.hypergeom.sf(x-1, M, n, N); NaN for invalid arguments.
 * Host-side scalar evaluation with the same code the kernels run. */
double hicmi_hypergeom_sf(int64_t x, int64_t M, int64_t n, int64_t N);
/* The comparison the scans make with it, as the kernels evaluate it (host build of the same routine, for tests):
 * 1 if hyper_geom(x, M, n, N) < psig, 0 if >= psig, -1 if it is NaN - the tail sum stops as soon as that is settled. */
int hicmi_hypergeom_decide(int64_t x, int64_t M, int64_t n, int64_t N, double psig);

/* Host helpers that finish scipy's linkage: stable sort of raw merges by height + union-find
 * relabel, and the count-sorted leaf walk.  Exposed for tests. */
int hicmi_label_linkage(const double *Zraw, int64_t n, double *Z_out);
int hicmi_leaf_order(const double *Z, int64_t n, int32_t *leaves_out);
/* Device self test: the Lance-Williams update divides by (nx+ny) with a 3-instruction exact sequence
 * (k_nnchain.hip: div_by_small_int); this compares it with the '/' operator on `samples` random
 * (numerator, integer divisor < 2^17) pairs and returns the number of mismatches (must be 0). */
int hicmi_selftest_division(hicmi_ctx *ctx, uint64_t seed, int64_t samples, uint64_t *mismatches_out);
/* Raw merges (x, y, height, size) in nn-chain merge order from the last hicmi_upgma. */
int hicmi_get_raw_merges(hicmi_ctx *ctx, double *Zraw_out);
/* Counters of the nn-chain kernels (scipy average, S2C:197) since the last hicmi_timing_reset, 6 doubles:
 * [0] merges, [1] row scans, [2] columns those scans visited (the "sum over scans of the live columns" of the
 * algorithmic-bytes definition), [3] chain steps answered by the neighbour cache instead of a scan,
 * [4] re-runs on one workgroup after a late peer of the column-sliced kernel, [5] reserved. */
int hicmi_nnchain_stats(hicmi_ctx *ctx, double *out6);

/* ---- Part 2: ordering objective --------------------------------------------------------------
 * giveNewAdjMat (OG:296-308): select the sub-matrix of the context's contact matrix for the bins
 * sel[0..n) (indices into the contact matrix); later candidates index into this selection. */
int hicmi_p2_select(hicmi_ctx *ctx, const int32_t *sel, int64_t n);
/* total = sum of all entries above the diagonal of the selected sub-matrix, with the reference's
 * own rounding: Python sum over offsets 1..n-1 of numpy.trace(adjMat, offset) (OG:343,448,506). */
int hicmi_p2_total(hicmi_ctx *ctx, double *total_out);
/* costFunction_numba (OG:184-191) of n_cand candidate orders at once.  perms: n_cand x n_used
 * int32 positions into the current selection (the reference's nOrder lists, OG:347,357,460,532);
 * n_used <= selection size.  scores_out: n_cand doubles.  Identical index lists give bit-identical
 * scores. */
int hicmi_p2_score(hicmi_ctx *ctx, const int32_t *perms, int64_t n_cand, int64_t n_used, double total,
                   double *scores_out);
/* The same objective in the reference's exact operation order (numpy.trace pairwise sums per
 * offset, then the sequential cum/total/i recurrence, OG:185-191): bit-identical to the reference's
 * NumPy path.  The search compares scores that can differ by one ulp (OG:349,359,464,535), so the
 * candidates within 1e-9 of a step's best hicmi_p2_score are re-scored with this entry point and
 * the decision is taken on these values. */
int hicmi_p2_score_exact(hicmi_ctx *ctx, const int32_t *perms, int64_t n_cand, int64_t n_used, double total,
                         double *scores_out);

/* ---- Part 2: search steps with the candidates enumerated on the device -------------------------
 * The reference builds a Python index list and a gathered matrix per candidate (OG:344-365,
 * 457-466, 519-538).  Here the scaffolds of a chromosome are contiguous ranges of the current
 * selection ("layout"), the order/orientation under test is a list of (scaffold, reversed) pairs
 * ("arrangement"), and a candidate is a couple of integers.
 *
 * hicmi_p2_layout: scaffold s occupies selection positions [scaf_start[s], scaf_start[s]+scaf_len[s]),
 * bins in ascending-ID ('+') order.  Reset by hicmi_p2_select. */
int hicmi_p2_layout(hicmi_ctx *ctx, const int32_t *scaf_start, const int32_t *scaf_len, int64_t n_scaf);
/* The arrangement = scaffolds ids[0..S) left to right, rev[j] != 0 when scaffold j is laid down
 * in '-' orientation (reorderScaffList, OG:310-321). */
int hicmi_p2_set_arrangement(hicmi_ctx *ctx, const int32_t *ids, const uint8_t *rev, int64_t S);
/* Literal total (see hicmi_p2_total) of the sub-matrix in the arrangement's order - what the
 * reference computes from giveNewAdjMat's matrix at OG:343 / OG:448 / OG:506. */
int hicmi_p2_arrangement_total(hicmi_ctx *ctx, double *total_out);
/* Closed-form objective of the arrangement itself. */
int hicmi_p2_arrangement_score(hicmi_ctx *ctx, double total, double *score_out);
/* checkAllScores (OG:332-372): objective of the arrangement with scaffold new_id inserted at every
 * gap g = 0..S in both orientations; scores_out[2*g + r], r = 1 for '-'.  2*(S+1) doubles. */
int hicmi_p2_score_insertions(hicmi_ctx *ctx, int32_t new_id, double total, double *scores_out);
/* Enumeration tables for windows of k scaffolds (permutations/removeReverseDuplicates/plusMinusPerms,
 * OG:381-430): orders is n_orders x k (window-local scaffold index per slot), orients is
 * n_orients x k (1 = '-').  Candidate c = order c / n_orients, orientation c % n_orients. */
int hicmi_p2_window_tables(hicmi_ctx *ctx, int64_t k, const int8_t *orders, int64_t n_orders,
                           const uint8_t *orients, int64_t n_orients);
/* bruteForceBestScore / scanOrdering inner loops (OG:457-466, 519-538) for the window of k
 * scaffolds starting at arrangement index `first`: delta_out[c] = (objective of candidate c) * total
 * minus a term common to all candidates of this window (the pairs that lie outside it), so
 * score(c) = score(c0) + (delta[c] - delta[c0]) / total for any candidate c0 whose score is known.
 * n_orders * n_orients doubles. */
int hicmi_p2_score_window(hicmi_ctx *ctx, int64_t first, int64_t k, double *delta_out);

/* Whole decision steps (what the drop-in Part 2 calls in its inner loops): fast scores of all
 * candidates, short list within 1e-9 of the best, literal re-scoring of the short list (cached by
 * bin order under `total`), then the reference's "first strict maximum above bestCost" rule
 * (OG:349,359,464,535) - identical decisions to doing the same with the entry points above.
 *
 * Window of k scaffolds at arrangement index `first` (k == S is bruteForceBestScore).  floor is the
 * incoming bestCost; cur_fast the fast score of the current arrangement (NaN: computed here).
 * pick_out = winning candidate index or -1; best_out = its literal score (or floor);
 * pick_fast_out = fast score of the arrangement that is current after applying the pick. */
int hicmi_p2_decide_window(hicmi_ctx *ctx, int64_t first, int64_t k, double total, double floor, double cur_fast,
                           int64_t *pick_out, double *best_out, double *pick_fast_out);
/* checkAllScores (OG:332-372) for scaffold new_id against the arrangement (ids, rev): computes the
 * literal total of "arrangement + new scaffold last" (OG:484-487, 343), scores the 2(S+1) candidates in
 * the reference's enumeration order (orientation tested first alternates with the gap, starting from
 * new_rev_now) and returns the winning gap / orientation (gap_out = -1: nothing scored above 0). */
int hicmi_p2_decide_insertion(hicmi_ctx *ctx, const int32_t *ids, const uint8_t *rev, int64_t S, int32_t new_id,
                              int32_t new_rev_now, int64_t *gap_out, int32_t *rev_out, double *best_out);

/* Whole loops, so that one chromosome costs a handful of host calls (chromosomes are independent and
 * are driven concurrently from host threads, one context each).
 * hicmi_p2_insert_all = orderRemainderScaffolds (OG:475-493): ids/rev hold the S0 ordered scaffolds on
 * entry and S0 + n_new on return (caller provides the capacity); new_ids are the remaining scaffolds
 * in pull order, each entering in '+' orientation.  best_out = bestCost of the last insertion.
 * hicmi_p2_scan_pass = one round of scanOrdering (OG:513-541) over windows of k scaffolds, every
 * winner applied before the next window; ids/rev updated in place, *best_io / *cur_fast_io carried,
 * *improved_out = 1 if any window improved (the reference's `stop`).
 * hicmi_p2_scan_all = the whole `while True` loop of scanOrdering (OG:509-547): rounds until one brings no
 * improvement, *rounds_out = how many ran ("Working on round i of final step..." is printed once per round by the caller). */
int hicmi_p2_insert_all(hicmi_ctx *ctx, int32_t *ids, uint8_t *rev, int64_t S0, const int32_t *new_ids, int64_t n_new,
                        double *best_out);
int hicmi_p2_scan_pass(hicmi_ctx *ctx, int32_t *ids, uint8_t *rev, int64_t S, int64_t k, double total, double *best_io,
                       double *cur_fast_io, int32_t *improved_out);
int hicmi_p2_scan_all(hicmi_ctx *ctx, int32_t *ids, uint8_t *rev, int64_t S, int64_t k, double total, double *best_io,
                      double *cur_fast_io, int64_t *rounds_out);
/* hicmi_p2_insert_all for n_jobs chromosomes at once (the loop over chromosomes of OG:608-612 turned
 * inside out): job j uses context ctxs[j] - its own selection and layout, all contexts on one device - and
 * the arrays ids[j] / rev[j] / new_ids[j] with S0[j] / n_new[j] entries as above.  The chromosomes advance
 * in lock step, every step of all of them decided on the device; best_out[j] as for hicmi_p2_insert_all.
 * Must not run concurrently with other calls on any of the contexts. */
int hicmi_p2_insert_all_multi(int64_t n_jobs, hicmi_ctx *const *ctxs, int32_t *const *ids, uint8_t *const *rev,
                              const int64_t *S0, const int32_t *const *new_ids, const int64_t *n_new, double *best_out);

/* ---- Part 3 input scan (host code, no GPU) -----------------------------------------------------
 * readValidPairFile (orientSmallScaffolds.py:159-177): of a HiC-Pro allValidPairs file
 * (read, scaffold1, pos1, strand1, scaffold2, pos2, ...) keep the lines whose (scaffold1, scaffold2) is one of the
 * registered ORDERED name pairs.  names_blob / name_off[n_names + 1]: the scaffold names, concatenated;
 * pair_a / pair_b: name indices of the n_pairs registered pairs.  hicmi_scan_valid_pairs parses the file with
 * `threads` host threads (0 = all) and returns the number of hits and of lines; hicmi_scan_fetch copies the hits
 * out in file order - (registered pair, pos1, pos2) - and releases the handle.  A line with fewer than six
 * columns or a non-integer position of a registered pair is an error, as in the reference. */
int hicmi_scan_valid_pairs(const char *path, const char *names_blob, const int64_t *name_off, int64_t n_names,
                           const int32_t *pair_a, const int32_t *pair_b, int64_t n_pairs, int threads,
                           int64_t *n_hits_out, int64_t *n_lines_out, void **handle_out);
int hicmi_scan_fetch(void *handle, int32_t *pair_idx, int64_t *pos1, int64_t *pos2);

/* ---- plot support -----------------------------------------------------------------------------
 * plotContactMap (plotContactMaps.py:15-91) colours every cell of an N x N matrix between two
 * numpy.percentile limits.  The matrix stays on the device: kind 0 = raw contacts (Part 2 plots,
 * OG:619,704), 1 = distance transform (S2C:147 -> S2C:1124), 2 = similarity transform (S2C:149 ->
 * S2C:1156); order = the n_sel matrix rows shown, in plot order (NULL: all rows as stored).
 * hicmi_plot_percentiles: out[i] = numpy.percentile(cells, q[i]) (method "linear"), exact.
 * hicmi_plot_downsample: out = px x px block means (row-major fp64), px <= n_sel. */
int hicmi_plot_percentiles(hicmi_ctx *ctx, int kind, const int32_t *order, int64_t n_sel, const double *q, int64_t n_q,
                           double *out);
int hicmi_plot_downsample(hicmi_ctx *ctx, int kind, const int32_t *order, int64_t n_sel, int64_t px, double *out);

/* ---- timing ----------------------------------------------------------------------------------
 * Accumulated device time (HIP events on the context stream) per kernel family since the last
 * reset, for bench.py's roofline object.  names_out: caller buffer receiving ';'-separated names;
 * ms_out / launches_out / bytes_out: one entry per name (algorithmic bytes as defined in DESIGN.md).
 * hicmi_timing_enable: 0 = off, 1 = every family, 2 = only the Part 1 families that are launched a few times
 * per map (row sums, distance build, nn-chain, row sort, rank inversion) - launches and bytes are counted in
 * every mode. */
int hicmi_timing_reset(hicmi_ctx *ctx);
int hicmi_timing_enable(hicmi_ctx *ctx, int on);
int hicmi_timing_get(hicmi_ctx *ctx, char *names_out, int64_t names_cap, double *ms_out,
                     int64_t *launches_out, double *bytes_out, int64_t cap, int64_t *count_out);

#ifdef __cplusplus
}
#endif
#endif /* HICMI_H */

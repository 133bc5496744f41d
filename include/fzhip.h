/* fzhip.h — C-ABI of libfzhip.so, the MI355X (gfx950) fuzzy substring search engine.
 *
 * This is the drop-in boundary for the hot path of taleinat/fuzzysearch 0.8.1.
 * The reference's native hooks are per-candidate CPython functions (GIL held, `y*` buffers):
 *
 *   search_exact_byteslike(sub, seq, start, end) -> list[int]        src/fuzzysearch/_common.c:5-112
 *   count_differences_with_maximum_byteslike(a, b, max) -> int       src/fuzzysearch/_common.c:115-173
 *   substitutions_only_find_near_matches_ngrams_byteslike(sub, seq, k) -> list[int]
 *                                   src/fuzzysearch/_substitutions_only_ngrams_template.h:10-138
 *   c_expand_short / c_expand_long(sub, seq, k) -> (dist, n) | (None, None)
 *                                   src/fuzzysearch/_levenshtein_ngrams.pyx:9-75, :78-154
 *   c_find_near_matches_generic_linear_programming(sub, seq, params) -> list[Match]
 *                                   src/fuzzysearch/_generic_search.pyx:25-233
 *
 * That granularity (one call per n-gram hit) cannot feed a GPU, so the replacement boundary sits
 * one level up, at the whole-search functions that *drive* those hooks:
 *
 *   fz_search_exact    <->  search_exact()                          search_exact.py:59-77
 *   fz_lev_ngrams      <->  find_near_matches_levenshtein_ngrams()  levenshtein_ngram.py:159-198
 *   fz_subs_ngrams     <->  substitutions_only_find_near_matches_ngrams_byteslike + the Match
 *                           construction of substitutions_only.py:258-278
 *   fz_generic_ngrams  <->  find_near_matches_generic_ngrams()      generic_search.py:198-237
 *   fz_consolidate     <->  consolidate_overlapping_matches()       common.py:185-189
 *   fz_group_best      <->  [get_best_match_in_group(g) for g in group_matches(ms)]
 *                                                                   substitutions_only.py:279-282
 *
 * Conventions: plain pointers and sizes only (no torch / Python types).  Every function returns
 * FZ_OK (0) or a negative FZ_E* code; fz_last_error() gives the text for the calling thread.
 * Inputs are borrowed for the duration of the call.  Outputs (`fz_match**`, `uint64_t**`) are
 * allocated by the library and released with fz_free().  A fz_ctx is not internally locked: use one
 * per host thread.  All device work happens on the ctx's own HIP streams; calls return after the
 * results are on the host.
 *
 * Error behaviour mirrors the reference: empty subsequence / n-gram length 0 -> FZ_EINVAL (the
 * reference raises ValueError: levenshtein_ngram.py:164-165, _common.c:50-53).
 */
#ifndef FZHIP_H
#define FZHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FZ_ABI_VERSION 1

#define FZ_OK          0
#define FZ_EINVAL     -1   /* invalid argument (maps to ValueError in the Python layer)     */
#define FZ_ENOMEM     -2   /* host or device allocation failed (MemoryError)               */
#define FZ_EDEVICE    -3   /* HIP runtime error / no usable gfx950 device (RuntimeError)   */
#define FZ_EUNSUPPORTED -4 /* parameters outside what the kernels support (m, k limits)    */
#define FZ_EHALO      -5   /* shard halo too small for this pattern (m + k bytes needed)   */
#define FZ_ETIMEOUT   -6   /* a collective did not complete within FZ_COMM_TIMEOUT_MS (a rank never arrived): the
                            * communicator is unusable from then on; searches without the collective still work  */

typedef struct fz_ctx fz_ctx;
typedef struct fz_seq fz_seq;

/* Raw-stream record.  `block` = n-gram block index g = ngram_start / ngram_len (the outer loop
 * variable of levenshtein_ngram.py:171); -1 where it does not apply.  Field order start, end,
 * dist matches fuzzysearch.Match (common.py:15-20). */
typedef struct {
    int64_t start, end;
    int32_t dist;
    int32_t block;
} fz_match;

typedef struct {
    uint64_t bytes_scanned;    /* sequence bytes streamed by the filter kernel(s), last call  */
    uint64_t ngram_hits;       /* exact n-gram hits handed to verification, last call        */
    uint64_t raw_matches;      /* raw-stream records produced, last call                     */
    double   filter_ms;        /* hipEvent span of the filter kernel(s), last call           */
    double   verify_ms;        /* hipEvent span of the verify kernel(s), last call           */
    double   device_ms;        /* hipEvent span first launch -> results on host, last call   */
    uint32_t filter_launches;  /* number of filter kernel launches in the last call          */
    uint32_t n_devices;
    uint32_t verify_form;      /* how the last call's n-gram hits were verified (FZ_FORM_*)   */
    uint32_t reserved_;
} fz_stats_t;

/* fz_stats_t.verify_form: chosen from the search's arguments alone (pattern, budget, sequence kind) */
#define FZ_FORM_NONE        0u   /* no verification (exact search, automaton searches)                          */
#define FZ_FORM_FUSED_BAND  1u   /* inside the scan: register band / Hamming count, one candidate per lane     */
#define FZ_FORM_FUSED_CELLS 2u   /* inside the scan: lane-per-DP-cell, 16 or 32 lanes per candidate            */
#define FZ_FORM_FUSED_BITS1 3u   /* inside the scan: bit-vector columns on one 64-bit word, candidate per lane */
#define FZ_FORM_FUSED_BITS2 4u   /* ... on two 64-bit words (patterns of 65 .. 128 characters)                 */
#define FZ_FORM_KERNEL      5u   /* a verification kernel of its own behind a hit list                         */
#define FZ_FORM_FUSED_BITS32 6u  /* inside the scan: bit-vector columns on one 32-bit word (patterns <= 32)     */

int         fz_abi_version(void);
const char *fz_last_error(void);
int         fz_device_count(int *n);

/* device_ids == NULL / n_devices == 0 -> device 0 only.  Requires gfx950.
 * fz_destroy also frees the sequences of the ctx that were not released (their handles die with it). */
int  fz_create(const int *device_ids, int n_devices, fz_ctx **out);
void fz_destroy(fz_ctx *ctx);

/* Make a sequence resident in HBM.  With several devices in the ctx it is split into contiguous
 * shards, each with a halo, and every later search runs on all devices concurrently
 * (SURVEY.md §8(e)). */
int  fz_seq_upload(fz_ctx *ctx, const uint8_t *host, uint64_t n, fz_seq **out);

/* One-process-per-GPU form (torchrun / RCCL jobs): this process holds bytes
 * [buf_global_off, buf_global_off + buf_len) of a global sequence of global_n bytes and OWNS the
 * n-gram hits whose index lies in [own_lo, own_hi).  The buffer must extend (m + k) bytes past
 * the owned range on both sides (clamped to the global sequence) for every later query, else the
 * query returns FZ_EHALO.  Results are in GLOBAL coordinates. */
int  fz_seq_upload_shard(fz_ctx *ctx, const uint8_t *host_buf, uint64_t buf_len,
                         uint64_t buf_global_off, uint64_t own_lo, uint64_t own_hi,
                         uint64_t global_n, fz_seq **out);
/* Several-devices-in-one-process form of the same thing, without a host copy of the whole sequence (32 GiB over
 * 8 GPUs is built shard by shard): fz_seq_new creates an empty sequence of global_n bytes, fz_seq_add_shard uploads
 * one shard — same meaning of the ranges as fz_seq_upload_shard — to device number dev_index OF THE CTX (one shard
 * per device; host_buf is borrowed for the call).  Every later search runs on all shards concurrently and the host
 * merges their streams (__init__.py:129-171 is the reference's only analogue: chunks with overlap). */
int  fz_seq_new(fz_ctx *ctx, uint64_t global_n, fz_seq **out);
int  fz_seq_add_shard(fz_seq *seq, int dev_index, const uint8_t *host_buf, uint64_t buf_len, uint64_t buf_global_off,
                      uint64_t own_lo, uint64_t own_hi);
uint64_t fz_seq_len(const fz_seq *seq);       /* global length */
void fz_seq_release(fz_seq *seq);

/* All (overlapping) occurrences of p in seq[lo:hi], ascending.  lo/hi are clamped like
 * search_exact.py:70-71; pass hi = UINT64_MAX for "to the end". */
int fz_search_exact(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m,
                    uint64_t lo, uint64_t hi, uint64_t **idx, uint64_t *n);

/* Raw, un-consolidated match stream of find_near_matches_levenshtein_ngrams, in the reference's
 * emission order (block ascending, then hit index ascending).  Requires m / (k+1) >= 1. */
int fz_lev_ngrams(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k,
                  fz_match **out, uint64_t *n);

/* The same search split in two so that the host (and other streams: a collective, a copy) can work
 * while the scan runs: _begin launches it and returns, _end waits and delivers exactly what
 * fz_lev_ngrams would.  Up to TWO searches may be in flight per ctx (a two-deep pipeline: the scan
 * of search i + 1 runs while the host orders and consumes the records of search i; every device has
 * two pinned result slots); _end always delivers the OLDEST one.  `p` is copied, `seq` must stay
 * alive; every other search call on the ctx fails with FZ_EINVAL until every _begin has had its _end. */
int fz_lev_ngrams_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k);
int fz_lev_ngrams_end(fz_ctx *ctx, fz_match **out, uint64_t *n);       /* = fz_search_end */
/* The same two-deep pipeline for the other n-gram searches; fz_search_end delivers the OLDEST search in flight, whatever
 * its kind, exactly as the synchronous call would (fz_subs_ngrams / fz_generic_ngrams, or — consolidated != 0 —
 * fz_generic_ngrams_consolidated).  Substitutions-only searches share the Levenshtein searches' result slots and may
 * be mixed with them.  A generic search keeps its hit list and records on the device until it is collected, so two of
 * them run on two LANES (a second set of per-device buffers and a second stream, created on first use): the scan of
 * the younger search runs next to the automaton kernel of the older one.  Generic searches cannot be in flight
 * together with the other kinds (FZ_EINVAL). */
int fz_subs_ngrams_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k);
int fz_generic_ngrams_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m,
                            uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l, int consolidated);
int fz_search_end(fz_ctx *ctx, fz_match **out, uint64_t *n);

/* Raw stream of the substitutions-only n-gram search: (i, i+m, min(Hamming, k+1), block) in the
 * reference's discovery order, cross-block duplicates preserved. */
int fz_subs_ngrams(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k,
                   fz_match **out, uint64_t *n);

/* Raw stream of find_near_matches_generic_ngrams (the greedy candidate-set automaton run on the
 * window around every n-gram hit), reference emission order: blocks in order, the hits of a block by index,
 * the matches of a hit as the automaton emits them, duplicates included.  The rows are ordered on the
 * device (one shard, at most 16 384 hits) or by the host; the bytes returned are the same either way. */
int fz_generic_ngrams(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m,
                      uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l,
                      fz_match **out, uint64_t *n);

/* find_near_matches_generic_ngrams + consolidate_overlapping_matches in one call (what GenericSearch.search followed by
 * GenericSearch.consolidate_matches computes: generic_search.py:198-237, :256-273, common.py:185-189) — the same rows
 * as fz_consolidate(fz_generic_ngrams(...)).  The first stage of the consolidation runs on the device: the automaton
 * kernel folds the matches of every n-gram hit into (hull, best match) pairs, a few thousand pairs cross PCIe instead
 * of every raw match (BASELINE configs[3b]: 6e3 instead of 2.1e5 rows), the host merges overlapping hulls.  In-memory
 * sequences only (no file segments). */
int fz_generic_ngrams_consolidated(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m,
                                   uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l,
                                   fz_match **out, uint64_t *n);

/* Search + the reduction the reference's strategy class applies to its result, in one call (round 4: find_near_matches
 * on a resident sequence paid a second C-ABI call and an array round trip through Python for it):
 *   fz_lev_ngrams_consolidated = fz_consolidate(fz_lev_ngrams(...))   LevenshteinSearch.search + consolidate_matches,
 *                                                                      levenshtein.py:151-164, common.py:185-189
 *   fz_subs_ngrams_best        = fz_group_best(fz_subs_ngrams(...))   what find_near_matches_substitutions_ngrams returns
 *                                                                      for bytes-like input, substitutions_only.py:258-282
 * (in a communicator: collective like the searches they start with). */
int fz_lev_ngrams_consolidated(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n);
int fz_subs_ngrams_best(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n);

/* has_near_match_* (substitutions_only.py:139-145, :218-233; generic_search.py:240-253): *found = 1 iff the
 * corresponding search would return at least one record.  Nothing is ordered or copied, and device work that starts
 * after the first record has been counted is skipped (workgroups of the scan, hits of the automaton kernel). */
int fz_subs_ngrams_any(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, int *found);
int fz_subs_lp_any(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, int *found);
int fz_generic_ngrams_any(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m,
                          uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l, int *found);

/* The reference's linear-programming fallbacks, used by its dispatchers when
 * len(subsequence) // (k + 1) < 3 (short patterns).  Whole-sequence candidate automata, tiled by
 * start position on the GPU; ordered emission lists exactly as the reference yields them.
 *   fz_lev_lp     <->  find_near_matches_levenshtein_linear_programming   levenshtein.py:52-148
 *   fz_subs_lp    <->  _find_near_matches_substitutions_lp                substitutions_only.py:82-136
 *   fz_generic_lp <->  _find_near_matches_generic_linear_programming      generic_search.py:57-177 */
int fz_lev_lp(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n);
int fz_subs_lp(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n);
int fz_generic_lp(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m,
                  uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l,
                  fz_match **out, uint64_t *n);

/* RCCL over xGMI without PyTorch (the reference's only scale-out analogue is the chunk loop of
 * __init__.py:129-171).  A context that joined a communicator searches COLLECTIVELY: fz_lev_ngrams and
 * fz_lev_ngrams_begin / _end must be called by every rank in the same order, every rank scans the shard(s) it
 * holds (fz_seq_upload_shard / fz_seq_add_shard), the ranks' record lists are exchanged by ONE ncclAllGather of
 * device buffers per search (capacity follows the counts) and every rank receives the merged stream of the whole
 * sequence in the reference's order.  With _begin / _end the all-gather of search i runs on its own stream next to
 * the scan of search i + 1.
 *   one process per GPU:   rank 0 calls fz_comm_unique_id and hands the FZ_COMM_ID_BYTES bytes to the other ranks
 *                          (file, socket, MPI: not this library's business), every rank calls fz_comm_init_rank on
 *                          its single-device context;
 *   one process, N GPUs:   fz_comm_init_all on the multi-device context (ncclCommInitAll; every device = one rank).
 * fz_comm_allgather / fz_comm_max_f64 / fz_comm_barrier serve load-time exchanges (halo bytes) and the job's clock.
 * fz_comm_set_collective(ctx, 0) makes the searches of the context local again (the communicator stays). */
#define FZ_COMM_ID_BYTES 128
int  fz_comm_unique_id(void *id, uint64_t id_bytes);
int  fz_comm_init_rank(fz_ctx *ctx, const void *id, int world, int rank);
int  fz_comm_init_all(fz_ctx *ctx);
int  fz_comm_info(fz_ctx *ctx, int *world, int *rank, int *collective);
int  fz_comm_set_collective(fz_ctx *ctx, int on);
int  fz_comm_allgather(fz_ctx *ctx, const void *send, uint64_t nbytes, void *recv);
int  fz_comm_max_f64(fz_ctx *ctx, double *value);
int  fz_comm_barrier(fz_ctx *ctx);
void fz_comm_destroy(fz_ctx *ctx);
/* Host time (ms) of the exchange step of the collective search collected last on this context: from "every local
 * shard has finished" to "every rank's records are on this host" (all-gather + D2H copy + parsing the ranks' blocks). */
int  fz_comm_gather_ms(fz_ctx *ctx, double *ms);
/* Which collective library fz_comm_* would use (loads it): 0 none found (fz_comm_* answer FZ_EUNSUPPORTED), 1 RCCL,
 * 2 the test suite's stand-in (tests/mock_rccl.cpp named by FZ_RCCL_LIB: the only one that accepts several ranks on
 * one device — fz_comm_init_all then accepts a context that lists a device more than once). */
int  fz_comm_backend(void);

/* consolidate_overlapping_matches: overlap groups -> best (dist, -len) per group -> sorted by
 * (start, end, dist).  Ties inside a group (the reference breaks them by set iteration order, i.e.
 * by PYTHONHASHSEED) are broken deterministically: smallest start. */
int fz_consolidate(const fz_match *in, uint64_t n, fz_match **out, uint64_t *n_out);

/* Best match of every overlap group in the reference's group-LIST order (group_matches appends new
 * groups and re-appends merged ones at the end, common.py:161-177) — the order that
 * substitutions_only.py:279-282 exposes.  Input order matters. */
int fz_group_best(const fz_match *in, uint64_t n, fz_match **out, uint64_t *n_out);

/* Multi-process jobs (one rank per GPU): merge the raw streams of `world` ranks,
 * each already in reference order (block-major, index ascending) and owning an ascending index range,
 * into the global reference order: for every block, the ranks' segments of that block back to back.
 * No sort, O(total) copies.  parts[r] / counts[r] = rank r's records; block_counts[r * nb + g] = how
 * many of them belong to block g.  `out` must hold sum(counts) records.  Needs no device. */
int fz_merge_ranks(const fz_match *const *parts, const uint64_t *counts, const uint64_t *block_counts,
                   uint32_t world, uint32_t nb, fz_match *out);

/* Wire format of one rank's contribution to the all-gather of match lists (16-byte rows so that the
 * collective and the D2H copy move 2/3 of the bytes of raw fz_match rows):
 *   header, FZ_WIRE_HEADER_ROWS rows: u64 count, u64 nblocks, u32 per_block_count[256]
 *   then `count` rows: i64 start, u32 end - start, u16 dist, u16 block   (at most cap_rows are stored)
 * fz_wire_pack fills `dst` (>= FZ_WIRE_HEADER_ROWS + cap_rows rows) from a stream in reference order.
 * fz_wire_merge reads `world` such blocks, each rows_per_rank rows apart, and writes the merged stream
 * (global reference order, like fz_merge_ranks) to `out`; *n_out = total.  If some rank stored fewer
 * rows than its count (count > cap), nothing is written and *max_count tells the capacity needed. */
#define FZ_WIRE_HEADER_ROWS 65
int fz_wire_pack(const fz_match *in, uint64_t n, uint64_t cap_rows, void *dst);
int fz_wire_merge(const void *recv, uint32_t world, uint64_t rows_per_rank, uint64_t cap_rows,
                  fz_match *out, uint64_t out_cap, uint64_t *n_out, uint64_t *max_count);

/* find_near_matches_in_file as a pipeline (replaces the chunk loops of __init__.py:129-171 and :174-200).
 * The reference searches every chunk of the file as an INDEPENDENT sequence, so results depend on the
 * chunk geometry; the stream reproduces it: chunk ("segment") j covers the items
 *     [j * seg_stride - seg_pre, (j + 1) * seg_stride + seg_post)   clipped to the file,
 * binary files: seg_stride = _chunk_size - keep, seg_pre = 0, seg_post = keep; text files: seg_stride =
 * _chunk_size, seg_pre = keep, seg_post = 0 (keep = len(subsequence) - 1 + extra_items_for_chunked_search).
 * Many chunks cross PCIe as one batch from pinned, double-buffered staging memory and are searched by one
 * launch with per-chunk clamps while the caller fills the next staging buffer.
 *   mode 0 exact (search_exact per chunk), 1 Levenshtein n-grams (k = max_l_dist), 2 substitutions-only
 *   n-grams (k = max_substitutions), 3 generic n-grams (k = max_l_dist + the three limits).
 * Protocol: fz_stream_buffer -> where to put the next bytes and how many fit; fz_stream_submit(n, last)
 * after writing n of them (a launch happens whenever whole chunks are available); or fz_stream_read_fd,
 * which drives both from a file descriptor until end of file: `threads` readers (0: half the cores, at most 32)
 * live for the whole call and take 1 MiB pieces of the current batch from a shared counter.
 * fz_stream_finish -> the raw match stream of the whole file in the reference's order (chunk by chunk,
 * each chunk in its in-memory order, file coordinates) and the chunk number of every record.
 * Requires seg_pre + seg_post <= seg_stride / 2 (FZ_EUNSUPPORTED otherwise: use per-chunk searches). */
typedef struct fz_stream fz_stream;
int  fz_stream_open(fz_ctx *ctx, uint32_t mode, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                    uint32_t max_dels, uint32_t k, uint64_t seg_stride, uint32_t seg_pre, uint32_t seg_post,
                    uint64_t batch_bytes, fz_stream **out);
int  fz_stream_buffer(fz_stream *st, uint8_t **host, uint64_t *capacity);
int  fz_stream_submit(fz_stream *st, uint64_t nbytes, int last);
int  fz_stream_read_fd(fz_stream *st, int fd, int64_t offset, int threads, uint64_t *total);
int  fz_stream_finish(fz_stream *st, fz_match **out, uint32_t **seg, uint64_t *n);
void fz_stream_close(fz_stream *st);

/* Test hook: the library reads its FZ_* environment switches (none is needed to use it) once,
 * on first use; this reads them again.  For tests that flip a switch inside one process; nothing may be in flight.
 * What a reload reaches: every switch that is consulted per search (routing, queue sizes, grids, the fused forms' LDS
 * budget, FZ_COMM_TIMEOUT_MS).  What it does NOT reach: the collective library (FZ_NO_RCCL / FZ_RCCL_LIB: loaded once per
 * process) and what a context took over when it was created (fz_create: timing, streams, the generic search's kernel
 * choice, the worker threads' spin time) — create a new context, or a new process, for those. */
void fz_debug_reload_switches(void);

/* Test hook (no device needed): how the scan would split the n-gram blocks of pattern p (block
 * length L, blocks at 0, L, 2L, ...) into launches.  out[4i .. 4i+3] = first block, number of blocks,
 * hash multiplier, slot shift of launch i (at most `cap` launches are written); *n_launches = total. */
int fz_debug_launch_plan(const uint8_t *p, uint32_t m, uint32_t L, uint32_t *out, uint32_t cap, uint32_t *n_launches);

/* Test hook (no device needed): the host step that turns the device's unordered 24-byte records
 * {u64 key = block << 48 | hit index, u32 left, u32 right, u32 dist (0xffffffff: empty slot), u32 aux} into fz_match
 * rows in the reference's emission order (block ascending, hit index ascending; n-gram length L). */
int fz_debug_order_records(const void *recs, uint64_t n, uint32_t L, fz_match **out, uint64_t *n_out);
/* ... the same with the key ranges a search knows beforehand (hit index < idx_bound, block < blk_bound): the form the
 * searches use — no pass over the records for their ranges, empty slots dropped while the sort keys are built. */
int fz_debug_order_records_bounded(const void *recs, uint64_t n, uint32_t L, uint64_t idx_bound, uint32_t blk_bound,
                                   fz_match **out, uint64_t *n_out);
/* ... and the sharded form: recs = the shards' records one shard after the other (seg_ends[i] = end of shard i), shards
 * owning ascending index ranges; every shard is ordered on its own and the rows are merged block by block. */
int fz_debug_order_segments(const void *recs, const uint64_t *seg_ends, uint32_t n_segments, uint32_t L, fz_match **out, uint64_t *n_out);

/* Test hook, no device needed: the regions a scan launch of `grid` workgroups over `ntiles` tiles is cut into on a GPU
 * with `n_cus` compute units (fzhip.hip: plan_scan_regions — the last `wg_per_cu * n_cus` workgroups take `steps` groups of
 * shrinking tile shares, down to `fmin` of a full one).  table[4 r .. 4 r + 3] = {first workgroup, workgroups, first tile,
 * end tile} of region r, `*n_regions` of them (0: one region, every workgroup strides over all tiles; at most 8 rows).
 * tests/test_host_logic.py checks that the regions partition both ranges. */
int fz_debug_scan_regions(uint64_t ntiles, uint64_t grid, uint32_t n_cus, int steps, double fmin, int wg_per_cu, uint32_t *n_regions,
                          uint64_t *table);

/* Test hook (no device needed): the host half of the exchange step of a collective search (what follows the
 * ncclAllGather).  `blocks` = `world` blocks of 1024 + cap * 24 bytes, block r = what rank r contributed: 128 64-bit
 * counters (word 1 = records the rank produced) followed by min(count, cap) device records (see fz_debug_order_records);
 * own_lo[r] = first hit index rank r owns (NULL: ranks own ascending ranges in rank order), L = n-gram length.
 * Some rank produced more than `cap` records: *need_cap = the capacity the re-gather uses (every rank computes the same
 * value from the same headers) and no stream.  Else *need_cap = 0 and *out = the merged stream in the reference's order:
 * every rank's records ordered by (block, index), then block by block the ranks' runs in ascending order of own_lo. */
int fz_debug_gather_merge(const void *blocks, uint32_t world, uint64_t cap, const uint64_t *own_lo, uint32_t L, fz_match **out,
                          uint64_t *n_out, uint64_t *need_cap);

/* Statistics of the search collected last.  The *_ms fields are hipEvent spans read from that search's events by THIS
 * call (and by fz_device_ms), not by the search: they describe it until the next search of the context is launched
 * (its launch re-records the events; the fields then read 0). */
int  fz_stats(fz_ctx *ctx, fz_stats_t *out);
/* Device memory of the context: the smallest free / total byte counts over its devices (hipMemGetInfo).  The host layer
 * sizes its residency cache with it (fuzzysearch_amd/engine.py: the reference itself holds nothing between calls,
 * __init__.py:35-57). */
int  fz_mem_info(fz_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* hipEvent timing of the kernels (filter_ms / verify_ms / device_ms of fz_stats, fz_device_ms): on by default.
 * Off, no events are recorded around the kernels and the *_ms fields read 0. */
int  fz_set_timing(fz_ctx *ctx, int on);
/* Streams the two-deep pipeline (fz_lev_ngrams_begin / fz_subs_ngrams_begin) uses per device: 1 (default) = both searches in
 * flight on one stream, the younger scan starts when the older one has finished; 2 = the younger scan — when it is fused
 * and writes its records straight to the host — runs on a stream and a counter block of its own and starts while the older
 * one drains (headline workload: 0.222 -> 0.205 ms per search; a kernel's own hipEvent span then includes the time it
 * shares the device, so filter_ms no longer measures the kernel alone). */
int  fz_set_streams(fz_ctx *ctx, int n);
/* hipEvent span of the filter kernel(s) of the search collected last, per device of the ctx (at most `cap` values
 * are written); returns the number of devices, or a negative FZ_E* code. */
int  fz_device_ms(fz_ctx *ctx, double *filter_ms, int cap);
void fz_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* FZHIP_H */

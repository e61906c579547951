/*
 * anrag.h -- C ABI of libanrag.so: the MI355X (gfx950) retrieval hot path of
 * A-NICE-RAG (dense dot-product top-k, BM25 term-at-a-time top-k, weighted
 * reciprocal-rank fusion) behind plain pointers and sizes.
 *
 * The reference has no FFI of its own: its boundary is the Python method surface
 * of `SearchEngine` (src/search_engine.py:14-293) and `DatabaseManager`
 * (src/database_manager.py:14-99).  Each entry point below names the reference
 * lines it replaces; the Python binding a maintainer would add is shown in
 * INTEGRATION.md and shipped in a-nice-rag_amd/_native.py.
 *
 * Conventions
 *   - every function returns ANRAG_OK (0) or a negative ANRAG_ERR_*; the message
 *     of the last failure on the calling thread is anrag_last_error().  Nothing
 *     throws or aborts across this boundary (the reference's own convention is
 *     "log and return empty", search_engine.py:94-98, :267-269 -- the Python
 *     shim maps error codes back to that).
 *   - "host" pointers are ordinary process memory; "device" pointers are HBM
 *     addresses of the index's GPU (hipMalloc / torch tensor .data_ptr()).
 *     The *_device entry points enqueue on the index's streams and return: a
 *     device INPUT must be complete before the call, and so must any pending
 *     write of the caller's to a device OUTPUT (a framework's zero fill of a
 *     fresh tensor runs on the framework's stream and can land after the answer):
 *     synchronise the producing stream first, or hand the index your streams
 *     (anrag_index_set_streams).
 *   - one anrag_index = one GPU's shard: a row block of the corpus matrix and
 *     the postings of the same documents.  Calls on one index are serialised by
 *     an internal mutex (the reference shares one SearchEngine between Streamlit
 *     session threads, src/app.py:17-27); different indexes are independent.
 *   - document identity inside the library is an int64 "doc id" chosen by the
 *     caller (a global id space shared by the dense rows and the BM25 rows, so
 *     that fusion can run on ids as the reference fuses on chunk-id strings,
 *     search_engine.py:27-32).  Strings never cross the boundary.
 *   - a NaN dense score (a corrupt row, a NaN in the query) ranks FIRST, as numpy's
 *     argpartition / argsort rank it (search_engine.py:83-87); it is reported as
 *     +inf.  BM25 scores cannot be NaN: anrag_bm25_load rejects non-finite idf / avgdl.
 *   - ordering rule everywhere: score descending, then ROW ascending (rows are
 *     what the reference's tie behaviour is stated in: its filtered BM25 path is
 *     a stable sort = low row first, :233; its numpy paths leave ties
 *     unspecified).  Across shards the tie-break is doc id ascending, which is
 *     the same thing under row sharding with doc id = doc_id_base + row.
 */
#ifndef ANRAG_H
#define ANRAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANRAG_ABI_VERSION 1

#define ANRAG_OK 0
#define ANRAG_ERR_INVALID (-1) /* bad argument                                   */
#define ANRAG_ERR_HIP (-2)     /* a HIP runtime call failed                      */
#define ANRAG_ERR_STATE (-3)   /* e.g. search before load                        */
#define ANRAG_ERR_NOMEM (-4)   /* host or device allocation failed               */
#define ANRAG_ERR_NODEVICE (-5) /* no usable gfx950 device                       */

/* Largest k served by the fused in-register per-wave top-k (one candidate per
 * lane of a 64-wide wavefront).  Larger k (retrieval_eval.py:142-143 uses 12000)
 * takes the score-array + radix-sort path; both are exact. */
#define ANRAG_FUSED_K_MAX 64
/* Most ranked lists one fusion call takes (the reference fuses at most 4 dense models + BM25). */
#define ANRAG_WRRF_MAX_LISTS 16

/* Candidate record exchanged between shards (RCCL all-gather payload) and
 * consumed by anrag_merge_candidates_device: 16 bytes, score widened to fp64
 * (exact for the dense fp32 scores). Unused slots: doc = -1, score = -inf. */
typedef struct anrag_candidate {
    double score;
    int64_t doc;
} anrag_candidate;

typedef struct anrag_index anrag_index;

/* Kernel ids for anrag_profile_read */
#define ANRAG_KERNEL_DENSE_SCAN 0    /* K1: N x D fp32 scan + per-wave top-k      */
#define ANRAG_KERNEL_DENSE_BATCHED 1 /* K2: Q x N fp32 MFMA GEMM + top-k epilogue */
#define ANRAG_KERNEL_BM25 2          /* K3: CSR postings scorer                   */
#define ANRAG_KERNEL_SELECT 3        /* K4: candidate merge / score-array select  */
#define ANRAG_KERNEL_WRRF 4          /* K5: weighted RRF + top-n                  */
#define ANRAG_KERNEL_COUNT 5

/* ------------------------------------------------------------------ library */
int anrag_abi_version(void);
const char *anrag_last_error(void);
int anrag_device_count(int *out_count);

/* ------------------------------------------------------------------ index lifetime
 * Replaces the process-lifetime DataFrame / pickle caches of DatabaseManager
 * (database_manager.py:17-18, :65-66, :92-93): the index owns the HBM copies. */
int anrag_index_create(int device, anrag_index **out);
int anrag_index_destroy(anrag_index *idx);
/* Run the index's kernels on caller-owned HIP streams (hipStream_t as void*).
 * NULL = the index's own streams.  Roles:
 *   primary    dense scans (and every non-hybrid entry point)
 *   secondary  the BM25 leg of a hybrid query and, behind it, the fusion kernel
 *              of anrag_hybrid_search_device (its results are complete in
 *              secondary-stream order; anrag_index_sync waits for everything)
 *   fusion     anrag_merge_candidates_device, anrag_wrrf_device and the
 *              copy-out of anrag_hybrid_candidates_device.  Pass the stream a
 *              torch.distributed (RCCL) collective will be issued on, e.g. a
 *              torch.cuda.Stream's .cuda_stream, and the collective orders
 *              after the per-shard candidates with no host sync. */
int anrag_index_set_streams(anrag_index *idx, void *primary, void *secondary, void *fusion);
/* Block until everything enqueued on the index's streams has finished. */
int anrag_index_sync(anrag_index *idx);
/* Stream ordering WITHOUT a host sync, for callers that keep their own streams (a framework's current stream):
 *   anrag_index_wait_stream    everything the index enqueues from now on runs after the work ALREADY enqueued on
 *                              `stream` (hipStream_t as void*, NULL = the null stream): call it after producing
 *                              device inputs AND after allocating / filling device outputs on that stream (a
 *                              framework's zero fill of a fresh tensor would otherwise race the library's answer),
 *                              before the *_device call;
 *   anrag_index_signal_stream  everything enqueued on `stream` from now on runs after the work already enqueued
 *                              on the index's three streams: call it after the *_device call(s), before the
 *                              framework reads the results on its stream.
 * The default recipe of INTEGRATION.md ("Streams"); anrag_index_set_streams is the alternative that removes the
 * index's own streams altogether. */
int anrag_index_wait_stream(anrag_index *idx, void *stream);
int anrag_index_signal_stream(anrag_index *idx, void *stream);

/* ------------------------------------------------------------------ dense: load
 * The upload point that replaces per-row np.frombuffer + DataFrame
 * (database_manager.py:39-66) and the per-query np.stack (search_engine.py:80):
 * `embeddings` is the row-major n_rows x dim fp32 matrix (host or device
 * pointer), copied once into HBM.  A device operand must be complete when the
 * call is made (synchronise the stream that produced it): the copy runs on the
 * index's own stream and the call returns when it has finished.
 *   source_id  nullable, n_rows x uint16: interned `source` string of each row
 *              (the column the filter of search_engine.py:36-55 looks at)
 *   doc_id     nullable, n_rows x int64: global doc id of each row; NULL means
 *              doc id = doc_id_base + row */
int anrag_dense_load(anrag_index *idx, const float *embeddings, int64_t n_rows, int32_t dim,
                     const uint16_t *source_id, const int64_t *doc_id, int64_t doc_id_base);

/* ------------------------------------------------------------------ dense: search
 * SearchEngine.similarity_search_with_embedding, search_engine.py:57-98 (and the
 * arithmetic of similarity_search, :100-146): raw dot product of each query with
 * every allowed row, top-k by (score desc, row asc).
 *   queries       host, n_queries x dim fp32 (the reference is batch=1,
 *                 search_engine.py:77-81; n_queries > 1 = that loop, batched)
 *   allow_source  nullable host, n_sources bytes: allow_source[s] != 0 keeps rows
 *                 whose source_id == s (the Python shim evaluates the reference's
 *                 prefix/regex rule once per distinct source string)
 *   out_doc / out_score   host, n_queries x k, rank order; unused tail: -1 / -inf
 *   out_count     host, n_queries: entries returned (min(k, allowed rows)) */
int anrag_dense_search(anrag_index *idx, const float *queries, int32_t n_queries, int32_t k,
                       const uint8_t *allow_source, int32_t n_sources, int64_t *out_doc,
                       float *out_score, int32_t *out_count);

/* The same for ONE fp64 query: the reference's text path gets float64 from the
 * embedding API and np.dot then promotes the fp32 matrix and scores in fp64
 * (search_engine.py:157, :129).  Rows stay fp32 in HBM; every dot product is
 * accumulated in fp64 (fp32 values are exact in fp64), selection by sorting the
 * fp64 score array: any k.  out_score: fp64.  (The fp32 entry points round an
 * fp64 query to fp32 first -- inside the 1e-4 bar, not the reference's bits;
 * the Python shim calls this one when it is handed a float64 query.) */
int anrag_dense_search_f64(anrag_index *idx, const double *query, int32_t k,
                           const uint8_t *allow_source, int32_t n_sources, int64_t *out_doc,
                           double *out_score, int32_t *out_count);

/* Same, all operands in HBM, no host sync: d_out is n_queries x k anrag_candidate.
 * n_queries > 1: up to 8 queries share a scan launch (each is still its own pass over the matrix; a workgroup starts
 * the next query when it has finished this one); every query's list merge runs on the fusion stream under the following
 * scans: results are complete in fusion-stream order.
 * n_queries == 1 on the index's own streams (no anrag_index_set_streams): consecutive calls rotate over up to 4 scan
 * streams whose kernels overlap, and a query's list merge rides in the NEXT scan launch of its stream -- no marker, no
 * merge launch per query (a launch boundary, ramp and drain are a fifth of a 100k-row pass).  Results are complete
 * after anrag_index_sync or behind anrag_index_signal_stream (which launch the merges still pending).
 * d_allow_bits: nullable device bitmap, bit s of word s/32 = source s allowed. */
int anrag_dense_search_device(anrag_index *idx, const float *d_queries, int32_t n_queries,
                              int32_t k, const uint32_t *d_allow_bits, anrag_candidate *d_out);

/* K2: up to 256 queries in ONE pass over the corpus on the fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32: exact f32 products and sums, no bf16 rounding), exact
 * top-k through a sampled-threshold filter (dense_batched.hip).  All operands in
 * HBM, primary stream, no host sync.  d_flag[q] = 0, or -1 when query q's
 * survivor list overflowed and the caller must redo it with
 * anrag_dense_search_device (anrag_dense_search does this by itself and picks
 * this path on its own for n_queries >= 16 on corpora of >= 65536 rows). */
int anrag_dense_search_batch_device(anrag_index *idx, const float *d_queries, int32_t n_queries,
                                    int32_t k, const uint32_t *d_allow_bits, anrag_candidate *d_out,
                                    int32_t *d_flag);

/* Arithmetic of the batched path.  0 (default): exact f32 products and sums on the f32 matrix cores.
 * 1: split-precision -- every operand as hi + lo bf16, three bf16 MFMAs per product (hi.hi + hi.lo + lo.hi),
 * f32 accumulation: scores within ~3e-5 of the f32 result for unit-norm vectors (inside the 1e-4 bar, not
 * bit-equal), ~2.7x the throughput of mode 0.  Applies to passes of any size (the query block is padded to
 * 256).  MEMORY: the first mode-1 pass after a dense load builds a second copy of the corpus as bf16 hi / lo
 * images (4 bytes per element, i.e. the corpus size again; counted in anrag_index_stats' hbm_bytes); it is
 * dropped by the next anrag_dense_load / anrag_index_destroy, not by switching back to mode 0. */
int anrag_set_batched_precision(anrag_index *idx, int32_t mode);

/* All N scores of one query (what search_engine.py:81 materialises), for tests
 * and for callers that post-process scores themselves.  out: host, n_rows fp32. */
int anrag_dense_scores(anrag_index *idx, const float *query, float *out_scores);

/* ------------------------------------------------------------------ BM25: load
 * Replaces the pickled rank_bm25.BM25Okapi object (database_manager.py:77-99;
 * built at processing/bm25_search.py:77): term-major CSR postings plus the
 * statistics BM25Okapi.__init__ derives.  All host pointers.
 *   indptr    n_terms+1 offsets into post_doc/post_tf; documents strictly
 *             ascending inside each term
 *   post_doc  document ROW (0..n_docs-1) of each posting;  post_tf  term frequency
 *   idf       n_terms, already epsilon-floored (Python floats from math.log)
 *   doc_len   n_docs token counts;  avgdl, k1, b as BM25Okapi holds them
 *   source_id / doc_id / doc_id_base   as for anrag_dense_load (BM25 rows are a
 *             different row space from the dense rows: bm25_search.py:67-68) */
int anrag_bm25_load(anrag_index *idx, const int64_t *indptr, int64_t n_terms,
                    const int32_t *post_doc, const int32_t *post_tf, const double *idf,
                    const int32_t *doc_len, int64_t n_docs, double avgdl, double k1, double b,
                    const uint16_t *source_id, const int64_t *doc_id, int64_t doc_id_base);

/* ------------------------------------------------------------------ BM25: search
 * SearchEngine._core_bm25_search, search_engine.py:205-243, including the
 * bm25.get_scores call at :219 (rank_bm25 arithmetic: fp64, term-at-a-time in
 * query order, duplicates counted again).  term_ids: host, query tokens mapped
 * to term ids IN QUERY ORDER; a negative id = token not in the vocabulary
 * (contributes nothing).  Zero-score documents are ranked, not dropped.
 * Bit-exact fp64 scores; order (score desc, row asc). */
int anrag_bm25_search(anrag_index *idx, const int32_t *term_ids, int32_t n_terms, int32_t k,
                      const uint8_t *allow_source, int32_t n_sources, int64_t *out_doc,
                      double *out_score, int32_t *out_count);
/* Same, operands in HBM, no host sync: the BM25-only member of the query pipeline (it
 * takes a pipeline slot like anrag_hybrid_search_device does, so it may be mixed
 * freely with hybrid and dense queries in flight).  K3 runs on the secondary
 * stream, the list merge on the fusion stream: d_out (k records) is complete in
 * fusion-stream order (anrag_index_sync waits for everything).  n_terms == 0
 * ranks the all-zero score array, as anrag_bm25_search does. */
int anrag_bm25_search_device(anrag_index *idx, const int32_t *d_term_ids, int32_t n_terms,
                             int32_t k, const uint32_t *d_allow_bits, anrag_candidate *d_out);
/* The same for n_queries queries in one call: d_term_ids / d_out are HOST arrays of device pointers, n_terms a host
 * array.  The queries run in groups of 16 per K3 launch (every query its own workgroups; launch, ramp and drain are paid
 * once per group: 6.5 us per query in the kernel at 1M documents against 7.8 at 8 per launch and 13.5 alone); each takes a
 * pipeline slot like a single call. */
int anrag_bm25_search_group_device(anrag_index *idx, const int32_t *const *d_term_ids, const int32_t *n_terms,
                                   int32_t n_queries, int32_t k, const uint32_t *d_allow_bits,
                                   anrag_candidate *const *d_out);
/* BM25Okapi.get_scores(query) itself: out host, n_docs fp64. */
int anrag_bm25_scores(anrag_index *idx, const int32_t *term_ids, int32_t n_terms,
                      double *out_scores);

/* ------------------------------------------------------------------ fusion
 * SearchEngine.weighted_reciprocal_rank_fusion, search_engine.py:21-34, followed
 * by the caller's truncation to common_sections_n (query_rag_retrieval.py:360-362):
 *   score[id] += weight[l] * (1 / (k + rank)), rank from 1, lists in order;
 *   stable sort by score descending (ties keep first-insertion order).
 * ids: host, the lists concatenated; list_len[l] entries each. fp64, bit-exact. */
int anrag_wrrf(anrag_index *idx, const int64_t *ids, const int32_t *list_len,
               const double *weight, int32_t n_lists, double k, int32_t top_n, int64_t *out_id,
               double *out_score, int32_t *out_count);

/* ------------------------------------------------------------------ fused hybrid query
 * The body of retrieve_documents for one dense model + BM25
 * (query_rag_retrieval.py:197-220, :304-335, :356-378) in one call: dense scan on
 * the primary stream, BM25 on the secondary, WRRF + top-n on the device.
 * n_terms == 0 or w_bm25 <= 0 skips BM25 (then the dense list is returned, :363-366).
 * allow_* as above, one per row space.  Callers on several threads overlap: the
 * index lock is held while a call enqueues, not while it waits for its result. */
int anrag_hybrid_search(anrag_index *idx, const float *query, const int32_t *term_ids,
                        int32_t n_terms, int32_t similarity_k, double w_dense, double w_bm25,
                        double wrrf_k, int32_t top_n, const uint8_t *allow_dense,
                        int32_t n_dense_sources, const uint8_t *allow_bm25,
                        int32_t n_bm25_sources, int64_t *out_id, double *out_score,
                        int32_t *out_count);

/* Same, operands in HBM, nothing syncs the host: d_out receives min(top_n,
 * distinct ids) records in fused order, *d_count that number, complete in
 * secondary-stream order (anrag_index_sync waits for all streams).
 * Back-to-back queries pipeline: the scans stay adjacent on the primary stream;
 * BM25 and the tail of query i run under the scans of the following queries.
 * Three launches per query: K1, K3 and one tail kernel (list merges + WRRF).
 * At most 32 queries are in flight per index: the call blocks on the HOST (never
 * on the device) until the query 32 back has finished with its buffers.
 * A corpus of at most 1 GiB on the index's own streams takes another route: the query's three launches go back to back on
 * ONE of four lane streams (no events between them) and consecutive queries rotate over the lanes (the reference's own corpus,
 * 9,609 x 384: 17 instead of 43 us per query); results are then complete behind anrag_index_sync / anrag_index_signal_stream. */
int anrag_hybrid_search_device(anrag_index *idx, const float *d_query, const int32_t *d_term_ids,
                               int32_t n_terms, int32_t similarity_k, double w_dense,
                               double w_bm25, double wrrf_k, int32_t top_n,
                               const uint32_t *d_allow_dense_bits,
                               const uint32_t *d_allow_bm25_bits, anrag_candidate *d_out,
                               int32_t *d_count);

/* n_queries hybrid queries from host memory in ONE call (no reference counterpart: the
 * reference's retrieval_eval.py:51-84 asks one query at a time; an evaluation run
 * holds thousands).  queries [n_queries][dim]; query q's term ids are
 * term_ids[term_offsets[q] .. term_offsets[q+1]) (term_offsets has n_queries+1
 * entries, term_offsets[0] == 0).  Operands go up once, one host sync per chunk.
 * Lists of 16 queries and more, both legs active and row numbers as document ids
 * (no doc_id arrays, bases 0), are ranked the way anrag_rank_batch ranks them: score
 * tiles that read the corpus once per 16 / 32 queries, a sort per list, the fusion in
 * LDS (1M x 768 top-10 lists: 43 us per query).  Everything else runs back to back
 * through the stream pipeline of anrag_hybrid_search_device -- the batch = 1 kernels,
 * one pass over the corpus per query (434 us).  Either way
 * out_id / out_score [n_queries][top_n] (tail -1 / -inf), out_count [n_queries];
 * each row equals what anrag_hybrid_search returns for that query, bit for bit. */
int anrag_hybrid_search_batch(anrag_index *idx, const float *queries, const int32_t *term_ids,
                              const int64_t *term_offsets, int32_t n_queries,
                              int32_t similarity_k, double w_dense, double w_bm25, double wrrf_k,
                              int32_t top_n, const uint8_t *allow_dense, int32_t n_dense_sources,
                              const uint8_t *allow_bm25, int32_t n_bm25_sources, int64_t *out_id,
                              double *out_score, int32_t *out_count);

/* ------------------------------------------------------------------ full ranking for lists of queries
 * retrieval_eval.py runs 7 of its 9 configurations with similarity_k = common_sections_n = 12000 -- "rank
 * everything" -- over ~8,000 queries each (src/retrieval_eval.py:142-143, :155-156, :366-378), one
 * retrieve_documents call per query.  This is the body of that call (src/query_rag_retrieval.py:197-370: every
 * active dense model's similarity search, the BM25 search, weighted RRF, truncation to common_sections_n) for a
 * LIST of queries, entirely on the device: score tiles, a per-(query, leg) radix select + sort, fusion over row
 * lists, one host sync per chunk of queries (rank_batch.hip).
 *
 * A LEG is one ranked list per query (the reference's `ranked_lists` entries, in the reference's order: dense models
 * first, BM25 last):
 *   kind ANRAG_LEG_DENSE  idx's dense rows, ranked by fp32 dot product with queries[q] (host, [n_queries][dim]) --
 *                         the arithmetic of anrag_dense_search, bit for bit
 *   kind ANRAG_LEG_BM25   idx's BM25 rows, ranked by BM25 score of query q's term ids
 *                         term_ids[term_offsets[q] .. term_offsets[q+1]) (host); a query WITHOUT term ids has no such
 *                         leg, as the reference skips BM25 for it (src/search_engine.py:216-217)
 *   allow_source / n_sources   as for anrag_dense_search (NULL = no filter); a leg the filter empties is dropped for
 *                         that query like any empty list (query_rag_retrieval.py:213, :326)
 *   doc_of_row            host, one int64 per row of the leg: the document id of the row in an id space shared by the
 *                         legs (what the chunk-id strings are to the reference's fusion); NULL = the row number.
 *                         With two or more legs every id must lie in [0, id_space) and a leg must not name a
 *                         document twice (ANRAG_ERR_INVALID otherwise)
 *   weight                model_weights[name] (> 0; the caller leaves out legs of weight 0 as the reference does)
 * Each leg keeps its best min(similarity_k, allowed rows) rows by (score desc, row asc).  One leg: out = its first
 * top_n documents (query_rag_retrieval.py:363-366), out_score = their leg scores.  Several: weighted RRF with wrrf_k,
 * stable by first insertion (search_engine.py:21-34, fp64, bit-exact), first top_n; out_score = fused scores.
 * out_id (nullable) / out_score (nullable, needs out_id) [n_queries][top_n], tail -1 / -inf; out_count [n_queries].
 * expect_id / out_rank (nullable, together) [n_queries]: out_rank[q] = 1-based position of document expect_id[q] in
 * query q's answer, -1 if it is not there -- what retrieval_eval.py:75-82 looks for in the returned list; with
 * out_id == NULL only counts and ranks come back (an evaluation run over thousands of queries needs nothing else).
 * Envelope: min(similarity_k, rows) <= 16384 for a dense leg, <= 13312 for a BM25 leg, min(top_n, entries) <= 13312
 * when fusing (anrag_rank_caps); all legs on one device.  Outside it: ANRAG_ERR_INVALID -- use the per-query entry
 * points (any k). */
#define ANRAG_LEG_DENSE 0
#define ANRAG_LEG_BM25 1
typedef struct anrag_rank_leg {
    anrag_index *idx;
    int32_t kind;
    const float *queries;
    const int32_t *term_ids;
    const int64_t *term_offsets;
    const uint8_t *allow_source;
    int32_t n_sources;
    const int64_t *doc_of_row;
    double weight;
} anrag_rank_leg;
int anrag_rank_batch(const anrag_rank_leg *legs, int32_t n_legs, int32_t n_queries, int32_t similarity_k,
                     double wrrf_k, int32_t top_n, int64_t id_space, int64_t *out_id, double *out_score,
                     int32_t *out_count, const int64_t *expect_id, int32_t *out_rank);
/* Largest list a dense (fp32 scores) / BM25 or fused (fp64 scores) ranking of anrag_rank_batch can return. */
int anrag_rank_caps(int32_t *out_k_max_fp32, int32_t *out_k_max_fp64);

/* ------------------------------------------------------------------ sharded merge
 * After an all-gather of every shard's k candidates: merge n_lists sorted lists of
 * k records each into the global top-k (score desc, doc asc -- row order is
 * shard-local, doc ids are global and ascend with rows under row sharding).
 * List l starts at d_lists + l * list_stride records (list_stride >= k).
 * All device pointers; enqueued on the fusion stream. */
int anrag_merge_candidates_device(anrag_index *idx, const anrag_candidate *d_lists,
                                  int32_t n_lists, int32_t k, int64_t list_stride,
                                  anrag_candidate *d_out);
/* Both legs of a hybrid query on this shard, no fusion: d_out[0..k) = dense
 * candidates, d_out[k..2k) = BM25 candidates -- one rank's all-gather payload.
 * Dense runs on the primary stream, BM25 on the secondary; d_out is written
 * on the fusion stream, i.e. it is ready for whatever is enqueued on that
 * stream next (the all-gather).  n_terms == 0: BM25 is skipped as the reference
 * skips it for a query without tokens (search_engine.py:216-217) and the BM25 half
 * of d_out is padding (doc -1, score -inf) -- the payload keeps its 2k shape. */
int anrag_hybrid_candidates_device(anrag_index *idx, const float *d_query,
                                   const int32_t *d_term_ids, int32_t n_terms, int32_t k,
                                   const uint32_t *d_allow_dense_bits,
                                   const uint32_t *d_allow_bm25_bits, anrag_candidate *d_out);

/* The same for n_queries queries in one call: d_queries / d_term_ids / d_out are HOST
 * arrays of device pointers, n_terms a host array.  The library scans the queries in
 * groups of 8 per launch (each query still its own pass over the shard) -- at the
 * shard sizes of 8 GPUs that is worth 10-14 % over n single calls.  What the sharded
 * searcher calls once per exchange group. */
int anrag_hybrid_candidates_group_device(anrag_index *idx, const float *const *d_queries,
                                         const int32_t *const *d_term_ids,
                                         const int32_t *n_terms, int32_t n_queries, int32_t k,
                                         const uint32_t *d_allow_dense_bits,
                                         const uint32_t *d_allow_bm25_bits,
                                         anrag_candidate *const *d_out);
/* The whole global tail of a GROUP of sharded queries in ONE launch (fusion
 * stream).  d_lists is the all-gather receive buffer: shard l's slab starts at
 * d_lists + l*list_stride and holds n_queries blocks of 2k records, block q as
 * anrag_hybrid_candidates_device wrote it for query q ([0,k) dense, [k,2k)
 * BM25).  Per query and modality: merge the n_lists lists into the global
 * top-k, then weighted RRF on those global ranks + top-n
 * -> d_out + q*top_n, d_count[q].  Exchanging several in-flight queries per
 * all-gather amortises the collective's fixed cost (each query is still
 * scanned on its own: batch = 1 kernels). */
int anrag_merge_fuse_device(anrag_index *idx, const anrag_candidate *d_lists, int32_t n_lists,
                            int32_t k, int64_t list_stride, double w_dense, double w_bm25,
                            double wrrf_k, int32_t top_n, int32_t n_queries,
                            anrag_candidate *d_out, int32_t *d_count);
/* WRRF over two device candidate lists (dense, bm25) -> top_n on the device
 * (fusion stream). */
int anrag_wrrf_device(anrag_index *idx, const anrag_candidate *d_dense, int32_t n_dense,
                      const anrag_candidate *d_bm25, int32_t n_bm25, double w_dense,
                      double w_bm25, double k, int32_t top_n, anrag_candidate *d_out,
                      int32_t *d_count);

/* ------------------------------------------------------------------ device memory helpers
 * (so a pure-ctypes caller can stage operands without torch) */
int anrag_device_alloc(anrag_index *idx, int64_t bytes, void **out_ptr);
int anrag_device_free(anrag_index *idx, void *ptr);
int anrag_copy_to_device(anrag_index *idx, void *d_dst, const void *h_src, int64_t bytes);
int anrag_copy_to_host(anrag_index *idx, void *h_dst, const void *d_src, int64_t bytes);

/* ------------------------------------------------------------------ measurement
 * kernel_mask bit i set: every launch of kernel id i is bracketed by HIP events
 * on the stream it runs on (0 = profiling off).  anrag_profile_read drains them
 * (waits for the launches) and returns the summed device time and launch count
 * of one kernel id since the last reset. */
int anrag_profile_enable(anrag_index *idx, uint32_t kernel_mask);
/* Bracket only every n-th launch of each selected kernel id (default 1 = all).
 * A bracketed launch costs ~10 us of stream time (it cannot overlap the marker
 * before it), so a throughput measurement samples. */
int anrag_profile_set_sampling(anrag_index *idx, int32_t every_n);
int anrag_profile_reset(anrag_index *idx);
int anrag_profile_read(anrag_index *idx, int kernel_id, double *out_total_ms, int64_t *out_launches);
/* Queries the timed launches of a kernel carried (a K1 launch carries up to 8 when the
 * caller submits query groups): algorithmic bytes per launch = units / launches x N*D*4. */
int anrag_profile_read_units(anrag_index *idx, int kernel_id, int64_t *out_units);
/* Shape facts a caller needs for roofline arithmetic. */
int anrag_index_info(anrag_index *idx, int64_t *dense_rows, int32_t *dense_dim,
                     int64_t *bm25_docs, int64_t *bm25_postings, int64_t *hbm_bytes);
/* Device / pinned-host allocations and releases the library has made so far (process-wide; each one
 * synchronises the device).  A steady-state query loop must not move it: tests/test_gpu_no_alloc.py. */
int anrag_debug_alloc_calls(int64_t *out_calls);

#ifdef __cplusplus
}
#endif
#endif /* ANRAG_H */

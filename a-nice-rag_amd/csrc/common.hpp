// Shared host/device definitions of libanrag.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "anrag.h"
#include "host_slots.hpp"

namespace anrag {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kListLen = 64;         // one top-k slot per lane (ANRAG_FUSED_K_MAX)
constexpr int kScanThreads = 256;    // 4 waves: one scan workgroup per CU (sweep: profiles/r01_scan_config_sweep.txt)
constexpr int kScanWaves = kScanThreads / kWave;
constexpr int kScanGroupMax = 16;  // queries one scan / K3 / tail launch can carry (dense_scan.hip, bm25.hip, tail.hip)
constexpr int kPipeSlots = 32;     // queries in flight in the hybrid pipeline (list sets, events): four exchange
                                   // groups of 8, so a slow collective on the communication stream (which the tails
                                   // queue behind) does not stall the scans two groups later
constexpr int kScanLanesMax = 4;     // scan streams that single dense queries rotate over (api.hip)
constexpr int kMaxScanBlocks = 256;  // one per CU
constexpr int kMaxScanLists = kMaxScanBlocks * kScanWaves;  // K1 leaves one sorted list per WAVE (the tail merges them)
constexpr uint32_t kNoRow = 0xFFFFFFFFu;

void set_error(const char *fmt, ...);

#define ANRAG_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::anrag::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                               __FILE__, __LINE__);                                        \
            return ANRAG_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define ANRAG_REQUIRE(cond, ...)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ::anrag::set_error(__VA_ARGS__);                                               \
            return ANRAG_ERR_INVALID;                                                      \
        }                                                                                  \
    } while (0)

struct ProfSpan {
    int kernel;
    hipEvent_t start, stop;
    int units;  // queries the bracketed launch carried
};

// Every device / pinned-host allocation and release of the library goes through these four: they count (process-wide)
// so that a test can assert that a steady-state query loop performs none (anrag_debug_alloc_calls) -- hipMalloc and
// hipFree synchronise the whole device.
int64_t alloc_calls();
hipError_t counted_malloc(void **p, size_t bytes);
hipError_t counted_free(void *p);
hipError_t counted_host_malloc(void **p, size_t bytes, unsigned flags);
hipError_t counted_host_free(void *p);
template <typename T>
inline hipError_t counted_malloc(T **p, size_t bytes) { return counted_malloc(reinterpret_cast<void **>(p), bytes); }

// A kernel that asks for more than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize set -- per
// DEVICE: remembered per (device, function) under a mutex, so that indexes on several GPUs of one process are
// independent (include/anrag.h) and concurrent first calls do not race.
int ensure_dynamic_lds(int device, const void *func, int bytes);

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// grow-only device block, freed at anrag_index_destroy
struct DevicePool {
    char *p = nullptr;
    int64_t bytes = 0;
};

// a single query's list merge that rides in the NEXT scan launch of its stream (dense_scan.hip: DeferredTail)
struct PendingTail {
    int set;              // list set the query's scan wrote
    int32_t k;
    anrag_candidate *out;
};

// carve pieces of 256-byte granularity out of one block (base == nullptr: only measure)
struct Carver {
    char *base;
    int64_t at = 0;
    explicit Carver(char *b) : base(b) {}
    template <typename T>
    T *take(int64_t count) {
        T *p = base ? reinterpret_cast<T *>(base + at) : nullptr;
        at += (count * (int64_t)sizeof(T) + 255) / 256 * 256;
        return p;
    }
};

}  // namespace anrag

// One GPU's shard.  Everything the kernels touch lives in HBM for the index's lifetime.
struct anrag_index {
    int device = 0;
    int n_cus = 256;
    std::mutex mu;
    // primary: dense scans.  secondary: BM25 + its list merge.  fusion: dense list merge + WRRF (and, in the
    // sharded path, the caller's collectives).  A hybrid query touches all three and never syncs the host.
    hipStream_t own_primary = nullptr, own_secondary = nullptr, own_fusion = nullptr;
    hipStream_t primary = nullptr, secondary = nullptr, fusion = nullptr;
    // hybrid pipeline: slot = query sequence number % kPipeSlots owns one set of dense block lists, one set
    // of BM25 partition lists and three events
    hipEvent_t ev_scan[anrag::kPipeSlots] = {};    // scan of the slot finished (primary)
    hipEvent_t ev_bm25[anrag::kPipeSlots] = {};    // BM25 lists of the slot ready (secondary)
    hipEvent_t ev_fused[anrag::kPipeSlots] = {};   // tail of the slot finished: its lists may be overwritten
    uint64_t hyb_seq = 0;
    bool hyb_outstanding = false;
    // Dense-only queries submitted ONE per call (anrag_dense_search_device, n_queries == 1, the index's own streams):
    // lanes of scans, each lane its own stream, consecutive queries on consecutive lanes.  A query's list merge rides in
    // the next scan launch of its lane (no marker between the scans, no second stream to wake), the lanes' kernels
    // overlap, so one query's launch boundary, ramp and drain are covered by the other lane's streaming.  api.hip.
    struct ScanLane {
        hipStream_t st = nullptr;
        bool pending = false;         // a scan whose lists have not been merged yet
        anrag::PendingTail tail{};
        uint64_t count = 0;           // queries this lane has taken since the last drain
        hipEvent_t ev[4] = {};        // backpressure: recorded every (slots per lane / 4) queries
    };
    ScanLane lane[anrag::kScanLanesMax];
    int n_lanes = 1;                  // configured (ANRAG_SCAN_LANES)
    int lanes_in_use = 1;             // ... and what the current run of single queries uses (1 for big corpora)
    uint64_t lane_rr = 0;
    bool lanes_active = false;
    hipEvent_t ev_order = nullptr;  // anrag_index_wait_stream / anrag_index_signal_stream
    // host-pointer hybrid queries (anrag_hybrid_search): per-slot staging, so that callers on several threads
    // overlap -- a caller holds the index lock while it enqueues, not while it waits for its result
    struct HostSlot {
        char *h = nullptr;            // pinned: query | terms | allow bitmaps | result records | count
        float *d_query = nullptr;
        int32_t *d_terms = nullptr;
        uint32_t *d_allow_a = nullptr, *d_allow_b = nullptr;
        anrag_candidate *d_out = nullptr;
        int32_t *d_count = nullptr;
        hipEvent_t done = nullptr;
    };
    HostSlot host_slot[anrag::kPipeSlots];
    char *host_slots_h = nullptr, *host_slots_d = nullptr;  // the one pinned / one device block the slots carve
    anrag::HostSlotRing<anrag::kPipeSlots> host_ring;       // busy flags, waiters, the dimension the slots are sized for

    // ---- dense shard: row-major fp32, rows 16-byte aligned when dim % 4 == 0
    float *d_emb = nullptr;
    int64_t n_rows = 0;
    int32_t dim = 0;
    uint16_t *d_dense_src = nullptr;
    int64_t *d_dense_doc = nullptr;
    int64_t dense_doc_base = 0;

    // ---- BM25 shard (see bm25.hip for the layout)
    int64_t n_docs = 0, n_terms = 0, n_postings = 0;
    int64_t *d_indptr = nullptr;       // n_terms + 1
    int32_t *d_post_doc = nullptr;     // n_postings
    double *d_post_impact = nullptr;   // n_postings: tf*(k1+1)/(tf + k1*(1-b+b*dl/avgdl)), fp64
    double *d_idf = nullptr;           // n_terms
    int32_t n_parts = 0, part_docs = 0;  // doc-range partitions (one workgroup each)
    int32_t *d_part_ptr = nullptr;     // per "frequent" term: n_parts+1 offsets (relative to indptr[t])
    int32_t *d_part_slot = nullptr;    // n_terms: row into d_part_ptr, or -1 (rare term: scan whole list)
    uint16_t *d_bm25_src = nullptr;
    int64_t *d_bm25_doc = nullptr;
    int64_t bm25_doc_base = 0;
    double bm25_k1 = 0, bm25_b = 0, bm25_avgdl = 0;

    // ---- workspaces (sized at load; reused by every query on the stream that owns them)
    float *d_blk_score_f32 = nullptr;  // [kPipeSlots][kMaxScanLists][kListLen]
    uint32_t *d_blk_row_a = nullptr;
    double *d_blk_score_f64 = nullptr;  // BM25 per-partition lists [kPipeSlots][n_parts][kListLen]
    uint32_t *d_blk_row_b = nullptr;
    float *d_query = nullptr;           // staged queries
    int64_t query_cap = 0;
    uint32_t *d_allow_a = nullptr, *d_allow_b = nullptr;  // staged allow bitmaps (2048 words each)
    int32_t *d_terms = nullptr;         // staged term ids
    anrag_candidate *d_cand_a = nullptr, *d_cand_out = nullptr;
    int64_t cand_cap = 0;
    void *h_pinned = nullptr;           // result staging
    int64_t pinned_bytes = 0;
    float *d_scores_f32 = nullptr;      // full score arrays for k > ANRAG_FUSED_K_MAX
    double *d_scores_f64 = nullptr;
    double *d_dense_scores_f64 = nullptr;  // fp64-query dense path: n_rows scores
    double *d_query_f64 = nullptr;
    void *d_sort_tmp = nullptr;
    int64_t sort_tmp_bytes = 0;
    void *d_sort_buf = nullptr;
    int64_t sort_buf_bytes = 0;
    // K2 (batched queries) workspace
    float *d_bq = nullptr, *d_btau = nullptr, *d_bsample = nullptr;
    int32_t *d_bcnt = nullptr, *d_bflag = nullptr;
    void *d_bcand = nullptr;
    int64_t bsample_cap = 0;
    bool batched_split = false;         // K2 arithmetic: false = exact f32 MFMA, true = bf16 x 3 split products
    void *d_bq_hi = nullptr, *d_bq_lo = nullptr;  // bf16 halves of the padded query block
    void *d_split_img = nullptr;        // split-precision K2: the corpus as ready-to-copy LDS images (built on first use)
    int64_t split_img_bytes = 0;
    void *d_bq_img = nullptr;           // ... and the query block in the same form
    // WRRF scratch
    int64_t *d_w_ids = nullptr, *d_w_in = nullptr;
    double *d_w_contrib = nullptr, *d_w_score = nullptr;
    void *d_w_blob = nullptr;           // sort-based long form (sort_select.hip)
    int64_t w_blob_bytes = 0;
    int32_t *d_w_first = nullptr, *d_w_count = nullptr;
    anrag_candidate *d_w_out = nullptr;
    int64_t wrrf_cap = 0;

    // full-ranking batches (rank_batch.hip): score tiles, ranked row lists, fusion arrays -- one grow-only block
    anrag::DevicePool rank_pool;
    // pooled per-call buffers of the host-pointer list entry points (anrag_dense_search with many queries,
    // anrag_hybrid_search_batch): grow-only, so a steady-state loop allocates nothing
    anrag::DevicePool call_pool;
    void *call_pin = nullptr;           // pinned staging of the same calls
    int64_t call_pin_bytes = 0;

    int64_t hbm_bytes = 0;
    int64_t bm25_hbm_bytes = 0;

    // ---- measurement
    bool profiling = false;
    uint32_t profile_mask = 0xFFFFFFFFu;  // bit i: time kernel id i
    uint32_t profile_every = 1;           // bracket every n-th launch of a kernel id
    uint32_t profile_seen[ANRAG_KERNEL_COUNT] = {0};
    std::vector<anrag::ProfSpan> spans;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[ANRAG_KERNEL_COUNT] = {0};
    int64_t prof_launches[ANRAG_KERNEL_COUNT] = {0};
    int64_t prof_units[ANRAG_KERNEL_COUNT] = {0};
};

namespace anrag {

// Upload helper for operands that may already live in HBM (torch tensors): a device-to-device hipMemcpy is NOT
// ordered against the index's non-blocking streams and may return before it has finished, so copy on the
// primary stream and wait for it.  (The caller must have synchronised whatever produced a device operand.)
inline hipError_t copy_in(anrag_index *idx, void *dst, const void *src, size_t bytes) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, idx->primary);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(idx->primary);
}

// Bracket a launch with events when profiling is on.
struct LaunchTimer {
    anrag_index *idx;
    hipStream_t stream;
    int kernel;
    hipEvent_t start = nullptr, stop = nullptr;
    int units;
    LaunchTimer(anrag_index *i, int k, hipStream_t s, int units = 1);
    ~LaunchTimer();
};

int drain_profile(anrag_index *idx);
int sync_all(anrag_index *idx);         // wait for the index's three streams
int settle_pipeline(anrag_index *idx);  // ... only when pipeline queries may be outstanding
int ensure_pool(anrag_index *idx, DevicePool &pool, int64_t bytes);  // grow-only (contents are NOT kept)

// ---- kernel launchers (each enqueues on `stream`, never syncs)
int dense_scan_grid(const anrag_index *idx);
inline int dense_scan_lists(const anrag_index *idx) { return dense_scan_grid(idx) * kScanWaves; }
int dense_scan_vgprs(const anrag_index *idx);  // registers of the scan kernel this index's dimension runs
// K1 alone: one sorted list per workgroup into block-list set `set` (or every score into d_scores_out, k = 0)
int launch_dense_scan(anrag_index *idx, hipStream_t stream, const float *d_query, int32_t k,
                      const uint32_t *d_allow_bits, float *d_scores_out, int set);
constexpr int kTileLaunchMax = 128;  // queries one K1T launch carries (dense_tile.hip)
int dense_tile_group_max(const anrag_index *idx);  // queries K1T holds in LDS at a time at this dimension (0: no tile kernel)
int launch_dense_tile(anrag_index *idx, hipStream_t stream, const float *d_queries, int64_t q_stride, int32_t n_queries,
                      const uint32_t *d_allow_bits, float *d_scores_out, int64_t scores_stride);
bool dense_tile_mfma_has_shape(const anrag_index *idx);  // ... on the matrix cores (dense_tile_mfma.hip)
int dense_tile_mfma_block_rows(const anrag_index *idx);   // rows per block of that kernel (score rows must hold whole blocks)
int launch_dense_tile_mfma(anrag_index *idx, hipStream_t stream, const float *d_queries, int64_t q_stride, int32_t n_queries,
                           const uint32_t *d_allow_bits, float *d_scores_out, int64_t scores_stride);
bool dense_scan_has_shape(const anrag_index *idx);  // a shaped (not the generic) scan kernel serves this dimension
int launch_dense_scan_group(anrag_index *idx, hipStream_t stream, const float *const *d_queries, int32_t n_queries,
                            int32_t k, const uint32_t *d_allow_bits, float *d_scores_out, const int *sets,
                            int64_t scores_stride = 0, const PendingTail *pending = nullptr);
int launch_bm25_lists(anrag_index *idx, hipStream_t st, const int32_t *d_terms, int32_t n_terms, int32_t k,
                      const uint32_t *d_allow_bits, double *d_scores_out, int set);
int launch_bm25_lists_group(anrag_index *idx, hipStream_t st, const int32_t *const *d_terms, const int32_t *n_terms,
                            int32_t n_queries, int32_t k, const uint32_t *d_allow_bits, double *const *d_scores_out,
                            const int *sets);
// ... every score of any number of queries in one launch, the queries described in HBM (bm25.hip)
int launch_bm25_scores_table(anrag_index *idx, hipStream_t stream, const int32_t *d_terms_base, const int64_t *d_term_off,
                             int32_t n_queries, const uint32_t *d_allow_bits, double *d_scores_base, int64_t scores_stride);
// Tail of a query, ONE launch (tail.hip): merge the dense block lists of set `set` and/or the BM25 partition
// lists into per-modality top-k, then either write both lists (kTailCandidates: d_out[0..k) dense,
// [k..2k) BM25; with one modality only its k records at d_out[0..k)) or fuse them (kTailFuse: WRRF + top_n).
// kTailCandidates2k: always the two-list layout ([0,k) dense, [k,2k) BM25), an absent leg all padding (doc -1,
// score -inf) -- the shape of one rank's all-gather payload, whatever the query skipped.
enum TailMode { kTailFuse = 0, kTailCandidates = 1, kTailCandidates2k = 2 };
int launch_tail(anrag_index *idx, hipStream_t st, int set, bool use_dense, bool use_bm25, int32_t k, TailMode mode,
                double w_dense, double w_bm25, double wrrf_k, int32_t top_n, anrag_candidate *d_out,
                int32_t *d_count);
// ... for the n <= kScanGroupMax queries of a group in ONE launch (a workgroup per query)
int launch_tail_group(anrag_index *idx, hipStream_t st, const int *sets, bool use_dense, const bool *use_bm25, int32_t n,
                      int32_t k, TailMode mode, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                      anrag_candidate *const *d_out, int32_t *const *d_count);
// K1 + tail on one stream (set 0)
int launch_dense_topk(anrag_index *idx, hipStream_t stream, const float *d_query, int32_t k,
                      const uint32_t *d_allow_bits, anrag_candidate *d_out, float *d_scores_out);
// fp64 query: every row's fp64 dot product (dense_scan.hip); selection by dense_search_f64 (sort_select.hip)
int launch_dense_scores_f64(anrag_index *idx, hipStream_t st, const double *d_query, const uint32_t *d_allow_bits,
                            double *d_scores_out);
int dense_search_f64(anrag_index *idx, hipStream_t st, const double *h_query, int32_t k, const uint32_t *d_allow_bits,
                     int64_t *out_doc, double *out_score, int32_t *out_count);
// K2: up to 256 queries per pass on the fp32 matrix cores (dense_batched.hip)
bool batched_path_applies(const anrag_index *idx, int32_t n_queries, int32_t k);
int launch_dense_batched(anrag_index *idx, hipStream_t st, const float *d_queries, int32_t nq, int32_t k,
                         const uint32_t *d_allow_bits, anrag_candidate *d_out, int32_t *d_flag);
void free_batched(anrag_index *idx);
// k > ANRAG_FUSED_K_MAX: score array + radix sort (select.hip); host operands, syncs.
int dense_search_large_k(anrag_index *idx, hipStream_t stream, const float *h_queries, int32_t n_queries, int32_t k,
                         const uint32_t *d_allow_bits, int64_t *out_doc, float *out_score, int32_t *out_count);
void free_bm25(anrag_index *idx);
int bm25_load(anrag_index *idx, const int64_t *indptr, int64_t n_terms, const int32_t *post_doc,
              const int32_t *post_tf, const double *idf, const int32_t *doc_len, int64_t n_docs, double avgdl,
              double k1, double b, const uint16_t *source_id, const int64_t *doc_id, int64_t doc_id_base);
// d_out: k candidates (fused path, k <= 64) -- or d_scores_out: all n_docs scores; exactly one non-null
int launch_bm25(anrag_index *idx, hipStream_t stream, const int32_t *d_terms, int32_t n_terms, int32_t k,
                const uint32_t *d_allow_bits, anrag_candidate *d_out, double *d_scores_out);
int bm25_search_large_k(anrag_index *idx, hipStream_t stream, const int32_t *d_terms, int32_t n_terms, int32_t k,
                        const uint32_t *d_allow_bits, int64_t *out_doc, double *out_score, int32_t *out_count);
// WRRF over `n_lists` id lists laid out back to back in d_ids or d_cands (list l = [h_off[l], h_off[l+1])); entries
// with id < 0 are padding.  Writes min(top_n, distinct) records to d_out and the count to *d_count.
int ensure_wrrf_scratch(anrag_index *idx, int64_t n_entries);
void free_wrrf_scratch(anrag_index *idx);
int wrrf_sorted(anrag_index *idx, hipStream_t st, const int64_t *d_ids, const double *d_contrib, int32_t m,
                int32_t top_n, anrag_candidate *d_out, int32_t *d_count);
int launch_wrrf(anrag_index *idx, hipStream_t st, const int64_t *d_ids, const anrag_candidate *d_cands,
                const int32_t *h_off, const double *h_weight, int32_t n_lists, double k, int32_t top_n,
                anrag_candidate *d_out, int32_t *d_count);
int launch_merge_fuse(anrag_index *idx, hipStream_t st, const anrag_candidate *d_lists, int32_t n_lists, int32_t k,
                      int64_t list_stride, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                      int32_t n_queries, anrag_candidate *d_out, int32_t *d_count);
int launch_merge_candidates(anrag_index *idx, hipStream_t stream, const anrag_candidate *d_lists,
                            int32_t n_lists, int32_t k, int64_t list_stride, anrag_candidate *d_out);

}  // namespace anrag

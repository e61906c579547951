// C ABI of libanrag.so (include/anrag.h): index lifetime, uploads, query entry points.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.hpp"

namespace anrag {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------ counted allocation, per-device attributes
static std::atomic<int64_t> g_alloc_calls{0};
int64_t alloc_calls() { return g_alloc_calls.load(); }
hipError_t counted_malloc(void **p, size_t bytes) {
    g_alloc_calls.fetch_add(1);
    return hipMalloc(p, bytes);
}
hipError_t counted_free(void *p) {
    g_alloc_calls.fetch_add(1);
    return hipFree(p);
}
hipError_t counted_host_malloc(void **p, size_t bytes, unsigned flags) {
    g_alloc_calls.fetch_add(1);
    return hipHostMalloc(p, bytes, flags);
}
hipError_t counted_host_free(void *p) {
    g_alloc_calls.fetch_add(1);
    return hipHostFree(p);
}

int ensure_dynamic_lds(int device, const void *func, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> done;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({device, func});
    if (it != done.end() && it->second >= bytes) return ANRAG_OK;
    ANRAG_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));  // on the current device
    done[{device, func}] = bytes;
    return ANRAG_OK;
}

int ensure_pool(anrag_index *idx, DevicePool &pool, int64_t bytes) {
    if (pool.bytes >= bytes) return ANRAG_OK;
    if (pool.p) {
        (void)counted_free(pool.p);
        idx->hbm_bytes -= pool.bytes;
    }
    pool.p = nullptr;
    pool.bytes = 0;
    const int64_t want = (bytes + (1 << 20) - 1) / (1 << 20) * (1 << 20);
    hipError_t e = counted_malloc(reinterpret_cast<void **>(&pool.p), (size_t)want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%lld bytes) failed: %s", (long long)want, hipGetErrorString(e));
        return ANRAG_ERR_NOMEM;
    }
    pool.bytes = want;
    idx->hbm_bytes += want;
    return ANRAG_OK;
}

// ------------------------------------------------------------------ profiling
static hipEvent_t take_event(anrag_index *idx) {
    if (!idx->event_pool.empty()) {
        hipEvent_t e = idx->event_pool.back();
        idx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

LaunchTimer::LaunchTimer(anrag_index *i, int k, hipStream_t s, int u) : idx(i), stream(s), kernel(k), units(u) {
    if (!idx->profiling || !((idx->profile_mask >> k) & 1u)) return;
    // timing events are not free: a bracketed launch cannot start before the marker in front of it retires and
    // holds back the launch behind it (~10 us per bracket on the stream).  Sample every n-th launch.
    if (idx->profile_every > 1 && (idx->profile_seen[k]++ % idx->profile_every) != 0) return;
    start = take_event(idx);
    stop = take_event(idx);
    if (start) (void)hipEventRecord(start, stream);
}

LaunchTimer::~LaunchTimer() {
    if (!start || !stop) return;
    (void)hipEventRecord(stop, stream);
    idx->spans.push_back(ProfSpan{kernel, start, stop, units});
}

int drain_profile(anrag_index *idx) {
    for (ProfSpan &sp : idx->spans) {
        ANRAG_HIP(hipEventSynchronize(sp.stop));
        float ms = 0.f;
        ANRAG_HIP(hipEventElapsedTime(&ms, sp.start, sp.stop));
        idx->prof_ms[sp.kernel] += ms;
        idx->prof_launches[sp.kernel] += 1;
        idx->prof_units[sp.kernel] += sp.units;
        idx->event_pool.push_back(sp.start);
        idx->event_pool.push_back(sp.stop);
    }
    idx->spans.clear();
    return ANRAG_OK;
}

// ------------------------------------------------------------------ small helpers
template <typename T>
static int dev_alloc(anrag_index *idx, T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) return ANRAG_OK;
    hipError_t e = counted_malloc(reinterpret_cast<void **>(p), (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%lld bytes) failed: %s", (long long)(count * (int64_t)sizeof(T)), hipGetErrorString(e));
        return ANRAG_ERR_NOMEM;
    }
    idx->hbm_bytes += count * (int64_t)sizeof(T);
    return ANRAG_OK;
}

template <typename T>
static void dev_free(anrag_index *idx, T *&p, int64_t count) {
    if (p) {
        (void)counted_free(p);
        idx->hbm_bytes -= count * (int64_t)sizeof(T);
        p = nullptr;
    }
}

static int ensure_common_workspace(anrag_index *idx) {
    int rc;
    if (!idx->d_allow_a) {
        if ((rc = dev_alloc(idx, &idx->d_allow_a, 2048))) return rc;
        if ((rc = dev_alloc(idx, &idx->d_allow_b, 2048))) return rc;
        if ((rc = dev_alloc(idx, &idx->d_terms, 4096))) return rc;
    }
    if (!idx->d_cand_a) {
        idx->cand_cap = 4096;  // records per buffer (n_queries * k per call is chunked to this)
        if ((rc = dev_alloc(idx, &idx->d_cand_a, idx->cand_cap))) return rc;
        if ((rc = dev_alloc(idx, &idx->d_cand_out, idx->cand_cap))) return rc;
    }
    if (!idx->h_pinned) {
        idx->pinned_bytes = 1 << 20;
        ANRAG_HIP(counted_host_malloc(&idx->h_pinned, idx->pinned_bytes, hipHostMallocDefault));
    }
    return ANRAG_OK;
}

static int ensure_query_buffer(anrag_index *idx, int64_t floats) {
    if (idx->query_cap >= floats) return ANRAG_OK;
    dev_free(idx, idx->d_query, idx->query_cap);
    idx->query_cap = 0;
    int rc = dev_alloc(idx, &idx->d_query, floats);
    if (rc) return rc;
    idx->query_cap = floats;
    return ANRAG_OK;
}

// bytes-per-source allow list (host) -> 2048-word bitmap staged in HBM; NULL allow -> nullptr (no filter)
static int stage_allow(anrag_index *idx, hipStream_t st, const uint8_t *allow, int32_t n_sources, uint32_t *d_bits,
                       uint32_t *h_bits, const uint32_t **out) {
    *out = nullptr;
    if (!allow) return ANRAG_OK;
    ANRAG_REQUIRE(n_sources >= 0 && n_sources <= 65536, "n_sources %d out of range [0, 65536]", n_sources);
    memset(h_bits, 0, 2048 * sizeof(uint32_t));
    for (int32_t s = 0; s < n_sources; ++s)
        if (allow[s]) h_bits[s >> 5] |= 1u << (s & 31);
    ANRAG_HIP(hipMemcpyAsync(d_bits, h_bits, 2048 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    *out = d_bits;
    return ANRAG_OK;
}

// the merges still riding behind the lanes' last scans: launched on their own, each on its lane's stream
static int flush_lanes(anrag_index *idx) {
    if (!idx->lanes_active) return ANRAG_OK;
    for (int l = 0; l < kScanLanesMax; ++l) {
        anrag_index::ScanLane &ln = idx->lane[l];
        if (!ln.pending) continue;
        ln.pending = false;
        int rc = launch_tail(idx, ln.st, ln.tail.set, /*dense*/ true, /*bm25*/ false, ln.tail.k, kTailCandidates, 0, 0, 0, 0,
                             ln.tail.out, nullptr);
        if (rc) return rc;
    }
    return ANRAG_OK;
}

int sync_all(anrag_index *idx) {
    if (idx->lanes_active) {
        int rc = flush_lanes(idx);
        for (int l = 0; l < kScanLanesMax; ++l) {
            ANRAG_HIP(hipStreamSynchronize(idx->lane[l].st));
            idx->lane[l].count = 0;
        }
        idx->lanes_active = false;
        if (rc) return rc;
    }
    ANRAG_HIP(hipStreamSynchronize(idx->primary));
    ANRAG_HIP(hipStreamSynchronize(idx->secondary));
    ANRAG_HIP(hipStreamSynchronize(idx->fusion));
    idx->hyb_outstanding = false;
    return ANRAG_OK;
}

// Entry points outside the hybrid pipeline reuse its buffers: let an outstanding pipeline drain first.
int settle_pipeline(anrag_index *idx) {
    if (!idx->hyb_outstanding) return ANRAG_OK;
    return sync_all(idx);
}

// lanes single dense queries rotate over (ANRAG_SCAN_LANES = 0, 1, 2 or 4 overrides; 0 = off).  100k x 768 at one query
// per call: off 54.0, 1 lane 49.3, 2 lanes 46.4, 4 lanes 44.4 us per query (71 / 78 / 83 / 87 % of 8 TB/s); 9,609 x 384:
// 21.2 -> 7.7 us per query.  A corpus whose pass dwarfs a launch boundary (> 1 GiB) takes ONE lane: several scans in
// flight would only stretch each query's latency.
constexpr int kScanLanesDefault = 4;
constexpr int64_t kScanLanesMaxBytes = 1ll << 30;

#define ANRAG_ENTER(idx)                                                   \
    ANRAG_REQUIRE((idx) != nullptr, "index handle is NULL");              \
    std::lock_guard<std::mutex> lock__((idx)->mu);                        \
    ::anrag::DeviceGuard guard__((idx)->device);                          \
    if (!guard__.ok) {                                                    \
        ::anrag::set_error("hipSetDevice(%d) failed", (idx)->device);     \
        return ANRAG_ERR_HIP;                                             \
    }

}  // namespace anrag

using namespace anrag;

extern "C" {

int anrag_abi_version(void) { return ANRAG_ABI_VERSION; }

const char *anrag_last_error(void) { return g_err; }

int anrag_device_count(int *out_count) {
    ANRAG_REQUIRE(out_count != nullptr, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return ANRAG_ERR_NODEVICE;
    }
    *out_count = n;
    return ANRAG_OK;
}

int anrag_index_create(int device, anrag_index **out) {
    ANRAG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible: libanrag has no CPU fallback");
        return ANRAG_ERR_NODEVICE;
    }
    ANRAG_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    ANRAG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libanrag carries gfx950 code objects only", device, prop.gcnArchName);
        return ANRAG_ERR_NODEVICE;
    }
    anrag_index *idx = new (std::nothrow) anrag_index();
    if (!idx) {
        set_error("out of host memory");
        return ANRAG_ERR_NOMEM;
    }
    idx->device = device;
    idx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    DeviceGuard g(device);
    hipError_t e = hipStreamCreateWithFlags(&idx->own_primary, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&idx->own_secondary, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&idx->own_fusion, hipStreamNonBlocking);
    for (auto &ln : idx->lane) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking);
        for (auto &ev : ln.ev)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    }
    idx->n_lanes = kScanLanesDefault;
    if (const char *env = getenv("ANRAG_SCAN_LANES")) { const int v = atoi(env); idx->n_lanes = v <= 0 ? 0 : (v >= 4 ? 4 : (v == 3 ? 2 : v)); }
    for (int b = 0; b < kPipeSlots && e == hipSuccess; ++b) {
        e = hipEventCreateWithFlags(&idx->ev_scan[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&idx->ev_bm25[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&idx->ev_fused[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        set_error("stream/event creation failed: %s", hipGetErrorString(e));
        delete idx;
        return ANRAG_ERR_HIP;
    }
    idx->primary = idx->own_primary;
    idx->secondary = idx->own_secondary;
    idx->fusion = idx->own_fusion;
    *out = idx;
    return ANRAG_OK;
}

static void free_host_slots(anrag_index *idx);

static void free_dense(anrag_index *idx) {
    free_batched(idx);
    dev_free(idx, idx->d_dense_scores_f64, idx->n_rows);
    dev_free(idx, idx->d_emb, idx->n_rows * idx->dim);
    dev_free(idx, idx->d_dense_src, idx->n_rows);
    dev_free(idx, idx->d_dense_doc, idx->n_rows);
    dev_free(idx, idx->d_scores_f32, idx->n_rows);
    idx->n_rows = 0;
    idx->dim = 0;
}

int anrag_index_destroy(anrag_index *idx) {
    if (!idx) return ANRAG_OK;
    {
        DeviceGuard g(idx->device);
        (void)sync_all(idx);
        (void)drain_profile(idx);
        free_dense(idx);
        free_bm25(idx);
        void *ptrs[] = {idx->d_query_f64,
                        idx->d_blk_score_f32, idx->d_blk_row_a, idx->d_blk_score_f64, idx->d_blk_row_b, idx->d_query,
                        idx->d_allow_a,       idx->d_allow_b,   idx->d_terms,         idx->d_cand_a,
                        idx->d_cand_out,      idx->d_scores_f64, idx->d_sort_tmp,     idx->d_sort_buf};
        for (void *p : ptrs)
            if (p) (void)counted_free(p);
        free_wrrf_scratch(idx);
        free_batched(idx);
        free_host_slots(idx);
        if (idx->rank_pool.p) (void)counted_free(idx->rank_pool.p);
        if (idx->call_pool.p) (void)counted_free(idx->call_pool.p);
        if (idx->call_pin) (void)counted_host_free(idx->call_pin);
        if (idx->h_pinned) (void)counted_host_free(idx->h_pinned);
        for (hipEvent_t e : idx->event_pool) (void)hipEventDestroy(e);
        if (idx->ev_order) (void)hipEventDestroy(idx->ev_order);
        for (int b = 0; b < kPipeSlots; ++b) {
            hipEvent_t evs[] = {idx->ev_scan[b], idx->ev_bm25[b], idx->ev_fused[b]};
            for (hipEvent_t ev : evs)
                if (ev) (void)hipEventDestroy(ev);
        }
        for (auto &ln : idx->lane) {
            for (auto &ev : ln.ev)
                if (ev) (void)hipEventDestroy(ev);
            if (ln.st) (void)hipStreamDestroy(ln.st);
        }
        if (idx->own_fusion) (void)hipStreamDestroy(idx->own_fusion);
        if (idx->own_primary) (void)hipStreamDestroy(idx->own_primary);
        if (idx->own_secondary) (void)hipStreamDestroy(idx->own_secondary);
    }
    delete idx;
    return ANRAG_OK;
}

int anrag_index_set_streams(anrag_index *idx, void *primary, void *secondary, void *fusion) {
    ANRAG_ENTER(idx);
    int rc = sync_all(idx);
    if (rc) return rc;
    idx->primary = primary ? (hipStream_t)primary : idx->own_primary;
    idx->secondary = secondary ? (hipStream_t)secondary : idx->own_secondary;
    idx->fusion = fusion ? (hipStream_t)fusion : idx->own_fusion;
    idx->hyb_seq = 0;
    return ANRAG_OK;
}

int anrag_index_sync(anrag_index *idx) {
    ANRAG_ENTER(idx);
    return sync_all(idx);
}

int anrag_index_wait_stream(anrag_index *idx, void *stream) {
    ANRAG_ENTER(idx);
    if (!idx->ev_order) ANRAG_HIP(hipEventCreateWithFlags(&idx->ev_order, hipEventDisableTiming));
    ANRAG_HIP(hipEventRecord(idx->ev_order, (hipStream_t)stream));
    for (hipStream_t s : {idx->primary, idx->secondary, idx->fusion})
        if (s != (hipStream_t)stream) ANRAG_HIP(hipStreamWaitEvent(s, idx->ev_order, 0));
    for (int l = 0; l < kScanLanesMax; ++l) ANRAG_HIP(hipStreamWaitEvent(idx->lane[l].st, idx->ev_order, 0));
    return ANRAG_OK;
}

int anrag_index_signal_stream(anrag_index *idx, void *stream) {
    ANRAG_ENTER(idx);
    if (!idx->ev_order) ANRAG_HIP(hipEventCreateWithFlags(&idx->ev_order, hipEventDisableTiming));
    if (idx->lanes_active) {  // single dense queries: launch the merges still pending, then the lanes' streams count too
        if (int rc = flush_lanes(idx)) return rc;
        for (int l = 0; l < kScanLanesMax; ++l) {
            ANRAG_HIP(hipEventRecord(idx->ev_order, idx->lane[l].st));
            ANRAG_HIP(hipStreamWaitEvent((hipStream_t)stream, idx->ev_order, 0));
        }
    }
    for (hipStream_t s : {idx->primary, idx->secondary, idx->fusion}) {
        if (s == (hipStream_t)stream) continue;
        ANRAG_HIP(hipEventRecord(idx->ev_order, s));  // a wait captures the event's state when it is enqueued:
        ANRAG_HIP(hipStreamWaitEvent((hipStream_t)stream, idx->ev_order, 0));  // one event serves the three in turn
    }
    return ANRAG_OK;
}

// ------------------------------------------------------------------ dense
int anrag_dense_load(anrag_index *idx, const float *embeddings, int64_t n_rows, int32_t dim,
                     const uint16_t *source_id, const int64_t *doc_id, int64_t doc_id_base) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(embeddings != nullptr, "embeddings is NULL");
    ANRAG_REQUIRE(n_rows > 0 && n_rows < 0xFFFFFFFFll, "n_rows %lld out of range (1 .. 2^32-2 per shard)",
                  (long long)n_rows);
    ANRAG_REQUIRE(dim > 0 && dim <= 65536, "dim %d out of range", dim);
    ANRAG_HIP(hipStreamSynchronize(idx->primary));
    free_dense(idx);
    int rc;
    if ((rc = ensure_common_workspace(idx))) return rc;
    if ((rc = dev_alloc(idx, &idx->d_emb, n_rows * dim))) return rc;
    idx->n_rows = n_rows;
    idx->dim = dim;
    ANRAG_HIP(copy_in(idx, idx->d_emb, embeddings, (size_t)n_rows * dim * sizeof(float)));
    if (source_id) {
        if ((rc = dev_alloc(idx, &idx->d_dense_src, n_rows))) return rc;
        ANRAG_HIP(copy_in(idx, idx->d_dense_src, source_id, (size_t)n_rows * sizeof(uint16_t)));
    }
    if (doc_id) {
        if ((rc = dev_alloc(idx, &idx->d_dense_doc, n_rows))) return rc;
        ANRAG_HIP(copy_in(idx, idx->d_dense_doc, doc_id, (size_t)n_rows * sizeof(int64_t)));
    }
    idx->dense_doc_base = doc_id_base;
    if (!idx->d_blk_score_f32) {
        if ((rc = dev_alloc(idx, &idx->d_blk_score_f32, (int64_t)kPipeSlots * kMaxScanLists * kListLen))) return rc;
        if ((rc = dev_alloc(idx, &idx->d_blk_row_a, (int64_t)kPipeSlots * kMaxScanLists * kListLen))) return rc;
    }
    if ((rc = ensure_query_buffer(idx, (int64_t)64 * dim))) return rc;
    return ANRAG_OK;
}

struct GroupQuery {
    const float *d_query;
    const int32_t *d_terms;
    int32_t n_terms;
    anrag_candidate *d_out;
    int32_t *d_count;
};
static int hybrid_enqueue_group(anrag_index *idx, TailMode tail, const GroupQuery *q, int32_t n, int32_t k,
                                double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                                const uint32_t *d_allow_dense, const uint32_t *d_allow_bm25);
// One dense-only query per call on the index's own streams: lane = query number mod n_lanes.  A lane owns
// kPipeSlots / n_lanes list sets; the merge of its previous query rides in the scan launch of this one (DeferredTail),
// so a lane's stream carries scans only -- no marker, no second stream -- and the lanes overlap.  Backpressure: an event
// every quarter of a lane's sets; before a set is reused the call waits (on the HOST) for the launch that carried the
// merge of its previous user.  The last merges are launched by whatever drains the index (sync_all).
static int dense_single_enqueue(anrag_index *idx, const float *d_query, int32_t k, const uint32_t *d_allow,
                                anrag_candidate *d_out) {
    int rc;
    if (!idx->lanes_active) {
        if ((rc = settle_pipeline(idx))) return rc;  // the pipeline's queries own the same list sets
        idx->lanes_active = true;
        idx->lane_rr = 0;
        idx->lanes_in_use = (int64_t)idx->n_rows * idx->dim * 4 > kScanLanesMaxBytes ? 1 : idx->n_lanes;
    }
    const int L = idx->lanes_in_use;
    const int l = (int)(idx->lane_rr++ % (uint64_t)L);
    anrag_index::ScanLane &ln = idx->lane[l];
    const int per_lane = kPipeSlots / L, every = per_lane / 4;
    const uint64_t ls = ln.count;
    const int set = l * per_lane + (int)(ls % (uint64_t)per_lane);
    if (ls >= (uint64_t)per_lane) {
        // the set's previous user is lane query ls - per_lane; its merge rode in launch ls - per_lane + 1; event m
        // was recorded behind launch (m + 1) * every - 1
        const int64_t m = ((int64_t)ls - per_lane + 2 + every - 1) / every - 1;
        hipEvent_t ev = ln.ev[m % 4];
        if (hipEventQuery(ev) != hipSuccess) ANRAG_HIP(hipEventSynchronize(ev));
    }
    if ((rc = launch_dense_scan_group(idx, ln.st, &d_query, 1, k, d_allow, nullptr, &set, 0, ln.pending ? &ln.tail : nullptr)))
        return rc;
    ln.pending = true;
    ln.tail = PendingTail{set, k, d_out};
    ln.count = ls + 1;
    if (ln.count % (uint64_t)every == 0) ANRAG_HIP(hipEventRecord(ln.ev[(ln.count / every - 1) % 4], ln.st));
    idx->hyb_outstanding = true;
    return ANRAG_OK;
}

constexpr int kRankRouteMin = 16;  // queries from which anrag_hybrid_search_batch takes the ranking route
constexpr int kBm25Group = 16;  // queries per K3 launch of a BM25-only list (anrag_bm25_search_group_device)
constexpr int kScanGroup = 8;  // queries per scan launch when a call brings several (measured, queries per launch
                               // 1 -> 4 -> 8: 53.6 -> 47.1 -> 45.7 us per query at 100k rows, 64.0 -> 57.9 us at 125k
                               // rows for 4, nothing at 1M rows)

int anrag_dense_search_device(anrag_index *idx, const float *d_queries, int32_t n_queries, int32_t k,
                              const uint32_t *d_allow_bits, anrag_candidate *d_out) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(idx->d_emb != nullptr, "dense search before anrag_dense_load");
    ANRAG_REQUIRE(d_queries && d_out, "NULL operand");
    ANRAG_REQUIRE(n_queries > 0, "n_queries must be positive");
    ANRAG_REQUIRE(k > 0 && k <= ANRAG_FUSED_K_MAX, "device path serves 1 <= k <= %d (got %d)", ANRAG_FUSED_K_MAX, k);
    // one query per call on the index's own streams: the lane path (no marker and no merge launch per query)
    if (n_queries == 1 && idx->n_lanes >= 1 && idx->primary == idx->own_primary && idx->fusion == idx->own_fusion &&
        dense_scan_has_shape(idx) && (!d_allow_bits || idx->d_dense_src))
        return dense_single_enqueue(idx, d_queries, k, d_allow_bits, d_out);
    // the dense-only member of the query pipeline: the queries of a call are scanned in groups -- one launch per
    // group, each query still its own pass over the matrix -- and every query's list merge runs on the fusion
    // stream under the following scans (results complete in fusion-stream order)
    for (int32_t q0 = 0; q0 < n_queries; q0 += kScanGroup) {
        const int n = n_queries - q0 < kScanGroup ? n_queries - q0 : kScanGroup;
        GroupQuery g[kScanGroup];
        for (int i = 0; i < n; ++i)
            g[i] = GroupQuery{d_queries + (int64_t)(q0 + i) * idx->dim, nullptr, 0, d_out + (int64_t)(q0 + i) * k, nullptr};
        int rc = hybrid_enqueue_group(idx, kTailCandidates, g, n, k, 1.0, 0.0, 0.0, 0, d_allow_bits, nullptr);
        if (rc) return rc;
    }
    return ANRAG_OK;
}

int anrag_dense_search_f64(anrag_index *idx, const double *query, int32_t k, const uint8_t *allow_source,
                           int32_t n_sources, int64_t *out_doc, double *out_score, int32_t *out_count) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(idx->d_emb != nullptr, "dense search before anrag_dense_load");
    ANRAG_REQUIRE(query && out_doc && out_score && out_count, "NULL operand");
    ANRAG_REQUIRE(k > 0, "k must be positive");
    ANRAG_REQUIRE(!(allow_source && !idx->d_dense_src), "a source filter needs source ids (anrag_dense_load)");
    hipStream_t st = idx->primary;
    int rc;
    if ((rc = settle_pipeline(idx))) return rc;
    const uint32_t *d_allow = nullptr;
    uint32_t *h_bits = reinterpret_cast<uint32_t *>(idx->h_pinned);
    if ((rc = stage_allow(idx, st, allow_source, n_sources, idx->d_allow_a, h_bits, &d_allow))) return rc;
    return dense_search_f64(idx, st, query, k, d_allow, out_doc, out_score, out_count);
}

int anrag_dense_scores(anrag_index *idx, const float *query, float *out_scores) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(idx->d_emb != nullptr, "dense scores before anrag_dense_load");
    ANRAG_REQUIRE(query && out_scores, "NULL operand");
    int rc;
    if (!idx->d_scores_f32 && (rc = dev_alloc(idx, &idx->d_scores_f32, idx->n_rows))) return rc;
    hipStream_t st = idx->primary;
    ANRAG_HIP(hipMemcpyAsync(idx->d_query, query, (size_t)idx->dim * sizeof(float), hipMemcpyHostToDevice, st));
    if ((rc = launch_dense_topk(idx, st, idx->d_query, 0, nullptr, nullptr, idx->d_scores_f32))) return rc;
    ANRAG_HIP(hipMemcpyAsync(out_scores, idx->d_scores_f32, (size_t)idx->n_rows * sizeof(float),
                             hipMemcpyDeviceToHost, st));
    ANRAG_HIP(hipStreamSynchronize(st));
    return ANRAG_OK;
}

// n_queries >= 16 on a large corpus: K2 passes of up to 256 queries; a query whose survivor list overflowed
// (flag -1: adversarial threshold sample) is redone by the batch = 1 scan.
static int dense_search_batched_host(anrag_index *idx, hipStream_t st, const float *queries, int32_t n_queries,
                                     int32_t k, const uint32_t *d_allow, int64_t *out_doc, float *out_score,
                                     int32_t *out_count) {
    int rc;
    if ((rc = ensure_query_buffer(idx, (int64_t)256 * idx->dim))) return rc;
    std::vector<anrag_candidate> h((size_t)256 * k);
    std::vector<int32_t> flag(256);
    // the pass's result records and flags: the index's grow-only call pool (no allocation in a steady-state loop)
    if ((rc = ensure_pool(idx, idx->call_pool, (int64_t)256 * k * (int64_t)sizeof(anrag_candidate) + 256 * 4 + 1024)))
        return rc;
    Carver cv(idx->call_pool.p);
    anrag_candidate *d_out = cv.take<anrag_candidate>((int64_t)256 * k);
    int32_t *d_flag = cv.take<int32_t>(256);
    hipError_t e = hipSuccess;
    rc = ANRAG_OK;
    for (int32_t q0 = 0; q0 < n_queries && rc == ANRAG_OK; q0 += 256) {
        const int32_t nq = std::min(256, n_queries - q0);
        e = hipMemcpyAsync(idx->d_query, queries + (int64_t)q0 * idx->dim, (size_t)nq * idx->dim * sizeof(float),
                           hipMemcpyHostToDevice, st);
        if (e == hipSuccess) rc = launch_dense_batched(idx, st, idx->d_query, nq, k, d_allow, d_out, d_flag);
        if (e == hipSuccess && rc == ANRAG_OK)
            e = hipMemcpyAsync(h.data(), d_out, (size_t)nq * k * sizeof(anrag_candidate), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && rc == ANRAG_OK)
            e = hipMemcpyAsync(flag.data(), d_flag, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && rc == ANRAG_OK) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error("batched dense search failed: %s", hipGetErrorString(e));
            rc = ANRAG_ERR_HIP;
            break;
        }
        for (int32_t qi = 0; qi < nq && rc == ANRAG_OK; ++qi) {
            if (flag[qi] != 0) {  // redo with K1
                rc = launch_dense_topk(idx, st, idx->d_query + (int64_t)qi * idx->dim, k, d_allow, idx->d_cand_out, nullptr);
                if (rc == ANRAG_OK) {
                    e = hipMemcpyAsync(h.data() + (size_t)qi * k, idx->d_cand_out, (size_t)k * sizeof(anrag_candidate),
                                       hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    if (e != hipSuccess) {
                        set_error("dense redo failed: %s", hipGetErrorString(e));
                        rc = ANRAG_ERR_HIP;
                    }
                }
            }
            int32_t cnt = 0;
            for (int32_t i = 0; i < k; ++i) {
                const anrag_candidate &c = h[(size_t)qi * k + i];
                out_doc[(int64_t)(q0 + qi) * k + i] = c.doc;
                out_score[(int64_t)(q0 + qi) * k + i] = (float)c.score;
                if (c.doc >= 0) ++cnt;
            }
            out_count[q0 + qi] = cnt;
        }
    }
    return rc;
}

int anrag_dense_search_batch_device(anrag_index *idx, const float *d_queries, int32_t n_queries, int32_t k,
                                    const uint32_t *d_allow_bits, anrag_candidate *d_out, int32_t *d_flag) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(idx->d_emb != nullptr, "dense search before anrag_dense_load");
    ANRAG_REQUIRE(d_queries && d_out && d_flag, "NULL operand");
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= 256, "a batched pass takes 1..256 queries");
    ANRAG_REQUIRE(k > 0 && k <= ANRAG_FUSED_K_MAX, "1 <= k <= %d", ANRAG_FUSED_K_MAX);
    ANRAG_REQUIRE(idx->dim % 32 == 0, "the MFMA path needs dim %% 32 == 0 (dim = %d)", idx->dim);
    return launch_dense_batched(idx, idx->primary, d_queries, n_queries, k, d_allow_bits, d_out, d_flag);
}

int anrag_set_batched_precision(anrag_index *idx, int32_t mode) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (exact f32 MFMA) or 1 (bf16 x 3 split products)");
    idx->batched_split = mode == 1;
    return ANRAG_OK;
}

int anrag_dense_search(anrag_index *idx, const float *queries, int32_t n_queries, int32_t k,
                       const uint8_t *allow_source, int32_t n_sources, int64_t *out_doc, float *out_score,
                       int32_t *out_count) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(idx->d_emb != nullptr, "dense search before anrag_dense_load");
    ANRAG_REQUIRE(queries && out_doc && out_score && out_count, "NULL operand");
    ANRAG_REQUIRE(n_queries > 0 && k > 0, "n_queries and k must be positive");
    hipStream_t st = idx->primary;
    int rc;
    if ((rc = settle_pipeline(idx))) return rc;  // list set 0 and the staging buffers below belong to the pipeline too
    const uint32_t *d_allow = nullptr;
    uint32_t *h_bits = reinterpret_cast<uint32_t *>(idx->h_pinned);
    if ((rc = stage_allow(idx, st, allow_source, n_sources, idx->d_allow_a, h_bits, &d_allow))) return rc;
    ANRAG_REQUIRE(!(allow_source && !idx->d_dense_src), "a source filter needs source ids (anrag_dense_load)");
    if (k > ANRAG_FUSED_K_MAX) {
        return dense_search_large_k(idx, st, queries, n_queries, k, d_allow, out_doc, out_score, out_count);
    }
    if (batched_path_applies(idx, n_queries, k))
        return dense_search_batched_host(idx, st, queries, n_queries, k, d_allow, out_doc, out_score, out_count);
    anrag_candidate *h_cand = reinterpret_cast<anrag_candidate *>(static_cast<char *>(idx->h_pinned) + 8192);
    const int32_t chunk = (int32_t)std::min<int64_t>(64, idx->cand_cap / k);
    for (int32_t q0 = 0; q0 < n_queries; q0 += chunk) {
        const int32_t nq = std::min(chunk, n_queries - q0);
        ANRAG_HIP(hipMemcpyAsync(idx->d_query, queries + (int64_t)q0 * idx->dim, (size_t)nq * idx->dim * sizeof(float),
                                 hipMemcpyHostToDevice, st));
        for (int32_t qi = 0; qi < nq; ++qi)
            if ((rc = launch_dense_topk(idx, st, idx->d_query + (int64_t)qi * idx->dim, k, d_allow,
                                        idx->d_cand_out + (int64_t)qi * k, nullptr)))
                return rc;
        ANRAG_HIP(hipMemcpyAsync(h_cand, idx->d_cand_out, (size_t)nq * k * sizeof(anrag_candidate),
                                 hipMemcpyDeviceToHost, st));
        ANRAG_HIP(hipStreamSynchronize(st));
        for (int32_t qi = 0; qi < nq; ++qi) {
            int32_t cnt = 0;
            for (int32_t i = 0; i < k; ++i) {
                const anrag_candidate &c = h_cand[(int64_t)qi * k + i];
                out_doc[(int64_t)(q0 + qi) * k + i] = c.doc;
                out_score[(int64_t)(q0 + qi) * k + i] = (float)c.score;
                if (c.doc >= 0) ++cnt;
            }
            out_count[q0 + qi] = cnt;
        }
    }
    return ANRAG_OK;
}

// ------------------------------------------------------------------ BM25
int anrag_bm25_load(anrag_index *idx, const int64_t *indptr, int64_t n_terms, const int32_t *post_doc,
                    const int32_t *post_tf, const double *idf, const int32_t *doc_len, int64_t n_docs, double avgdl,
                    double k1, double b, const uint16_t *source_id, const int64_t *doc_id, int64_t doc_id_base) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    int rc = ensure_common_workspace(idx);
    if (rc) return rc;
    return bm25_load(idx, indptr, n_terms, post_doc, post_tf, idf, doc_len, n_docs, avgdl, k1, b, source_id, doc_id,
                     doc_id_base);
}

static int stage_terms(anrag_index *idx, hipStream_t st, const int32_t *term_ids, int32_t n_terms) {
    ANRAG_REQUIRE(n_terms >= 0 && n_terms <= 4096, "n_terms %d out of range [0, 4096]", n_terms);
    ANRAG_REQUIRE(n_terms == 0 || term_ids != nullptr, "term_ids is NULL");
    if (n_terms > 0)
        ANRAG_HIP(hipMemcpyAsync(idx->d_terms, term_ids, (size_t)n_terms * sizeof(int32_t), hipMemcpyHostToDevice, st));
    return ANRAG_OK;
}

int anrag_bm25_search_device(anrag_index *idx, const int32_t *d_term_ids, int32_t n_terms, int32_t k,
                             const uint32_t *d_allow_bits, anrag_candidate *d_out) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(idx->d_post_doc != nullptr, "BM25 search before anrag_bm25_load");
    ANRAG_REQUIRE(d_out && (n_terms == 0 || d_term_ids), "NULL operand");
    ANRAG_REQUIRE(n_terms >= 0, "n_terms must be >= 0");
    ANRAG_REQUIRE(k > 0 && k <= ANRAG_FUSED_K_MAX, "device path serves 1 <= k <= %d (got %d)", ANRAG_FUSED_K_MAX, k);
    // the BM25-only member of the query pipeline: it takes a slot like every other query (K3 on the secondary
    // stream, the list merge on the fusion stream), so that its list set is never shared with a query in flight
    const GroupQuery q{nullptr, d_term_ids, n_terms, d_out, nullptr};
    return hybrid_enqueue_group(idx, kTailCandidates, &q, 1, k, 0.0, 1.0, 0.0, 0, nullptr, d_allow_bits);
}

int anrag_bm25_search_group_device(anrag_index *idx, const int32_t *const *d_term_ids, const int32_t *n_terms,
                                   int32_t n_queries, int32_t k, const uint32_t *d_allow_bits,
                                   anrag_candidate *const *d_out) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(idx->d_post_doc != nullptr, "BM25 search before anrag_bm25_load");
    ANRAG_REQUIRE(d_term_ids && n_terms && d_out, "NULL operand");
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= 4096, "n_queries %d out of range", n_queries);
    ANRAG_REQUIRE(k > 0 && k <= ANRAG_FUSED_K_MAX, "device path serves 1 <= k <= %d (got %d)", ANRAG_FUSED_K_MAX, k);
    for (int32_t i = 0; i < n_queries; ++i)
        ANRAG_REQUIRE(d_out[i] && n_terms[i] >= 0 && (n_terms[i] == 0 || d_term_ids[i]),
                      "query %d: needs an output block and term ids for its n_terms", i);
    // BM25 alone: kBm25Group queries per launch.  A launch of 8 is 1,960 workgroups on 768 slots at 1M documents: 2.6
    // "rounds" of 18.7 us workgroups plus the ramp and the drain; 16 per launch halve the share of ramp and drain.
    static const int group = [] {
        const char *e = getenv("ANRAG_BM25_GROUP");  // measurements only
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= kScanGroupMax ? v : kBm25Group;
    }();
    for (int32_t q0 = 0; q0 < n_queries; q0 += group) {
        const int n = n_queries - q0 < group ? n_queries - q0 : group;
        GroupQuery g[kScanGroupMax];
        for (int i = 0; i < n; ++i) g[i] = GroupQuery{nullptr, d_term_ids[q0 + i], n_terms[q0 + i], d_out[q0 + i], nullptr};
        int rc = hybrid_enqueue_group(idx, kTailCandidates, g, n, k, 0.0, 1.0, 0.0, 0, nullptr, d_allow_bits);
        if (rc) return rc;
    }
    return ANRAG_OK;
}

int anrag_bm25_search(anrag_index *idx, const int32_t *term_ids, int32_t n_terms, int32_t k,
                      const uint8_t *allow_source, int32_t n_sources, int64_t *out_doc, double *out_score,
                      int32_t *out_count) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(idx->d_post_doc != nullptr, "BM25 search before anrag_bm25_load");
    ANRAG_REQUIRE(out_doc && out_score && out_count, "NULL operand");
    ANRAG_REQUIRE(k > 0, "k must be positive");
    ANRAG_REQUIRE(!(allow_source && !idx->d_bm25_src), "a source filter needs source ids (anrag_bm25_load)");
    hipStream_t st = idx->primary;
    int rc;
    if ((rc = settle_pipeline(idx))) return rc;  // list set 0 and the staging buffers below belong to the pipeline too
    const uint32_t *d_allow = nullptr;
    uint32_t *h_bits = reinterpret_cast<uint32_t *>(static_cast<char *>(idx->h_pinned) + 8192);
    if ((rc = stage_allow(idx, st, allow_source, n_sources, idx->d_allow_b, h_bits, &d_allow))) return rc;
    if ((rc = stage_terms(idx, st, term_ids, n_terms))) return rc;
    if (k > ANRAG_FUSED_K_MAX)
        return bm25_search_large_k(idx, st, idx->d_terms, n_terms, k, d_allow, out_doc, out_score, out_count);
    if ((rc = launch_bm25(idx, st, idx->d_terms, n_terms, k, d_allow, idx->d_cand_out, nullptr))) return rc;
    anrag_candidate *h_cand = reinterpret_cast<anrag_candidate *>(static_cast<char *>(idx->h_pinned) + 16384);
    ANRAG_HIP(hipMemcpyAsync(h_cand, idx->d_cand_out, (size_t)k * sizeof(anrag_candidate), hipMemcpyDeviceToHost, st));
    ANRAG_HIP(hipStreamSynchronize(st));
    int32_t cnt = 0;
    for (int32_t i = 0; i < k; ++i) {
        out_doc[i] = h_cand[i].doc;
        out_score[i] = h_cand[i].score;
        if (h_cand[i].doc >= 0) ++cnt;
    }
    *out_count = cnt;
    return ANRAG_OK;
}

int anrag_bm25_scores(anrag_index *idx, const int32_t *term_ids, int32_t n_terms, double *out_scores) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(idx->d_post_doc != nullptr, "BM25 scores before anrag_bm25_load");
    ANRAG_REQUIRE(out_scores != nullptr, "NULL operand");
    hipStream_t st = idx->primary;
    int rc;
    if ((rc = stage_terms(idx, st, term_ids, n_terms))) return rc;
    if (!idx->d_scores_f64) {
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_scores_f64), (size_t)idx->n_docs * sizeof(double)));
        idx->hbm_bytes += idx->n_docs * 8;
        idx->bm25_hbm_bytes += idx->n_docs * 8;
    }
    if ((rc = launch_bm25(idx, st, idx->d_terms, n_terms, 0, nullptr, nullptr, idx->d_scores_f64))) return rc;
    ANRAG_HIP(hipMemcpyAsync(out_scores, idx->d_scores_f64, (size_t)idx->n_docs * sizeof(double),
                             hipMemcpyDeviceToHost, st));
    ANRAG_HIP(hipStreamSynchronize(st));
    return ANRAG_OK;
}

// ------------------------------------------------------------------ fusion
int anrag_wrrf(anrag_index *idx, const int64_t *ids, const int32_t *list_len, const double *weight, int32_t n_lists,
               double k, int32_t top_n, int64_t *out_id, double *out_score, int32_t *out_count) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(list_len && weight && out_id && out_score && out_count, "NULL operand");
    ANRAG_REQUIRE(n_lists >= 1 && n_lists <= ANRAG_WRRF_MAX_LISTS, "n_lists %d out of range [1, %d]", n_lists,
                  ANRAG_WRRF_MAX_LISTS);
    ANRAG_REQUIRE(top_n > 0, "top_n must be positive");
    int32_t off[ANRAG_WRRF_MAX_LISTS + 1];
    off[0] = 0;
    for (int l = 0; l < n_lists; ++l) {
        ANRAG_REQUIRE(list_len[l] >= 0, "negative list length");
        ANRAG_REQUIRE((int64_t)off[l] + list_len[l] < 0x7FFFFFFF, "lists too long");
        off[l + 1] = off[l] + list_len[l];
    }
    const int32_t m = off[n_lists];
    *out_count = 0;
    if (m == 0) return ANRAG_OK;
    ANRAG_REQUIRE(ids != nullptr, "ids is NULL");
    hipStream_t st = idx->primary;
    int rc;
    if ((rc = ensure_wrrf_scratch(idx, m))) return rc;
    ANRAG_HIP(hipMemcpyAsync(idx->d_w_in, ids, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if ((rc = launch_wrrf(idx, st, idx->d_w_in, nullptr, off, weight, n_lists, k, top_n, idx->d_w_out, idx->d_w_count)))
        return rc;
    int32_t cnt = 0;
    ANRAG_HIP(hipMemcpyAsync(&cnt, idx->d_w_count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ANRAG_HIP(hipStreamSynchronize(st));
    cnt = std::min(cnt, std::min(top_n, m));
    std::vector<anrag_candidate> h((size_t)cnt);
    if (cnt > 0) {
        ANRAG_HIP(hipMemcpyAsync(h.data(), idx->d_w_out, (size_t)cnt * sizeof(anrag_candidate), hipMemcpyDeviceToHost, st));
        ANRAG_HIP(hipStreamSynchronize(st));
    }
    for (int32_t i = 0; i < cnt; ++i) {
        out_id[i] = h[i].doc;
        out_score[i] = h[i].score;
    }
    *out_count = cnt;
    return ANRAG_OK;
}

int anrag_wrrf_device(anrag_index *idx, const anrag_candidate *d_dense, int32_t n_dense,
                      const anrag_candidate *d_bm25, int32_t n_bm25, double w_dense, double w_bm25, double k,
                      int32_t top_n, anrag_candidate *d_out, int32_t *d_count) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(d_dense && d_out && d_count, "NULL operand");
    ANRAG_REQUIRE(n_dense >= 0 && n_bm25 >= 0 && n_dense + n_bm25 > 0 && n_dense + n_bm25 <= 1024,
                  "need 0 < n_dense + n_bm25 <= 1024");
    ANRAG_REQUIRE(top_n > 0, "top_n must be positive");
    hipStream_t st = idx->fusion;
    int rc;
    if ((rc = ensure_common_workspace(idx))) return rc;
    // the two lists have to be contiguous for the kernel: stage them in the index's candidate buffer
    ANRAG_HIP(hipMemcpyAsync(idx->d_cand_a, d_dense, (size_t)n_dense * sizeof(anrag_candidate), hipMemcpyDeviceToDevice, st));
    if (n_bm25 > 0)
        ANRAG_HIP(hipMemcpyAsync(idx->d_cand_a + n_dense, d_bm25, (size_t)n_bm25 * sizeof(anrag_candidate),
                                 hipMemcpyDeviceToDevice, st));
    const int32_t off[3] = {0, n_dense, n_dense + n_bm25};
    const double w[2] = {w_dense, w_bm25};
    return launch_wrrf(idx, st, nullptr, idx->d_cand_a, off, w, n_bm25 > 0 ? 2 : 1, k, top_n, d_out, d_count);
}

// ------------------------------------------------------------------ fused hybrid query
// One hybrid query through the stream pipeline, THREE launches:
//   primary    K1 scan                      -> dense block lists of the query's slot
//   secondary  K3 BM25                      -> BM25 partition lists of the slot
//   tail       (fused: secondary; candidates: fusion stream) waits for the scan, then ONE kernel merges both
//              list sets and either fuses (WRRF + top-n -> d_out) or writes both candidate lists to d_out
// slot = sequence number % kPipeSlots.  Back-to-back queries keep the scans adjacent on `primary` with NO
// event wait between them (a barrier packet in front of every scan cost 5-6 us per query on MI355X); everything
// else of query i runs under the scans of the following queries.  The only backpressure is on the host: before
// a slot is reused the call waits until the tail that last read it has finished, so at most kPipeSlots queries
// are in flight and the device never has to be told to wait for a buffer.
// A GROUP of n <= kScanGroupMax queries takes n consecutive slots and ONE scan launch (dense_scan.hip: each query
// is still its own pass over the matrix; a workgroup moves on to the next query without a grid-wide step, so the
// launch gap, the ramp and the spread of finishing times are paid once per launch): worth +10-14 % at the shard
// sizes of 8 GPUs, nothing at 1M rows.  K3 and the tail stay per query.
static int hybrid_enqueue_group(anrag_index *idx, TailMode tail, const GroupQuery *q, int32_t n, int32_t k,
                                double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                                const uint32_t *d_allow_dense, const uint32_t *d_allow_bm25) {
    ANRAG_REQUIRE(n >= 1 && n <= kScanGroupMax && n <= kPipeSlots, "group of %d queries", n);
    if (idx->lanes_active)  // single dense queries were using the list sets: drain them first (a rare change of mode)
        if (int rc0 = sync_all(idx)) return rc0;
    // legs: a query vector asks for the dense leg (one scan launch serves the whole group: all or none); term ids
    // ask for the BM25 leg.  No term ids: BM25 is skipped as the reference skips it (search_engine.py:216-217) --
    // except for a BM25-only candidate query, which then ranks all-zero scores like anrag_bm25_search does.
    const bool use_dense = idx->d_emb != nullptr && q[0].d_query != nullptr && (tail != kTailFuse || w_dense > 0.0);
    bool use_bm25[kScanGroupMax];
    for (int i = 0; i < n; ++i) {
        ANRAG_REQUIRE(!use_dense || q[i].d_query != nullptr, "query %d of the group has no query vector", i);
        use_bm25[i] = idx->d_post_doc != nullptr && (tail != kTailFuse || w_bm25 > 0.0) &&
                      (q[i].n_terms > 0 || (!use_dense && tail != kTailFuse));
        ANRAG_REQUIRE(use_dense || use_bm25[i], "hybrid search with neither a dense nor a BM25 leg");
    }
    hipStream_t P = idx->primary, S = idx->secondary, F = idx->fusion;
    int slot[kScanGroupMax];
    const float *qs[kScanGroupMax];
    for (int i = 0; i < n; ++i) {
        const uint64_t seq = idx->hyb_seq + i;
        slot[i] = (int)(seq % kPipeSlots);
        if (seq >= (uint64_t)kPipeSlots && hipEventQuery(idx->ev_fused[slot[i]]) != hipSuccess)
            ANRAG_HIP(hipEventSynchronize(idx->ev_fused[slot[i]]));
        qs[i] = q[i].d_query;
    }
    idx->hyb_seq += n;
    idx->hyb_outstanding = true;
    int rc;
    if (use_dense) {
        if ((rc = launch_dense_scan_group(idx, P, qs, n, k, d_allow_dense, nullptr, slot))) return rc;
        ANRAG_HIP(hipEventRecord(idx->ev_scan[slot[0]], P));
    }
    hipStream_t T = tail == kTailFuse ? S : F;
    {  // the BM25 legs of the group: one launch, a workgroup set per query
        const int32_t *bt[kScanGroupMax];
        int32_t bn[kScanGroupMax];
        int bs[kScanGroupMax], nb = 0;
        for (int i = 0; i < n; ++i) {
            if (!use_bm25[i]) continue;
            bt[nb] = q[i].d_terms;
            bn[nb] = q[i].n_terms;
            bs[nb++] = slot[i];
        }
        if (nb > 0 && (rc = launch_bm25_lists_group(idx, S, bt, bn, nb, k, d_allow_bm25, nullptr, bs))) return rc;
    }
    if (T != S) {
        bool any = false;
        for (int i = 0; i < n; ++i) any = any || use_bm25[i];
        if (any) {
            ANRAG_HIP(hipEventRecord(idx->ev_bm25[slot[0]], S));
            ANRAG_HIP(hipStreamWaitEvent(T, idx->ev_bm25[slot[0]], 0));
        }
    }
    if (use_dense) ANRAG_HIP(hipStreamWaitEvent(T, idx->ev_scan[slot[0]], 0));
    {  // the group's tails: one launch, a workgroup per query
        anrag_candidate *outs[kScanGroupMax];
        int32_t *counts[kScanGroupMax];
        bool any_count = false;
        for (int i = 0; i < n; ++i) {
            outs[i] = q[i].d_out;
            counts[i] = q[i].d_count;
            any_count = any_count || q[i].d_count != nullptr;
        }
        if ((rc = launch_tail_group(idx, T, slot, use_dense, use_bm25, n, k, tail, w_dense, w_bm25, wrrf_k, top_n, outs,
                                    any_count ? counts : nullptr)))
            return rc;
        for (int i = 0; i < n; ++i) ANRAG_HIP(hipEventRecord(idx->ev_fused[slot[i]], T));
    }
    return ANRAG_OK;
}

// One hybrid query per call on a SMALL corpus (the reference's own: 9,609 x 384): the pipeline above spends it on
// plumbing -- three streams, two or three event records at 4-5 us of stream time each, waits between them: 44.8 us per
// query for kernels of 4 + 8 + 9 us.  Here the query's scan, BM25 kernel and tail are launched back to back on ONE lane
// stream (stream order is the only dependency; no events), and consecutive queries rotate over the lanes, whose streams
// overlap on the device.  The lane's list sets and its back-pressure events are the single dense queries' (a dense
// query's pending merge rides in this scan launch like in any other of the lane).  Large corpora keep the pipeline: there
// K3 must run UNDER the scan, on its own stream.
constexpr int64_t kHybridLanesMaxBytes = kScanLanesMaxBytes;  // 1 GiB (measured: 9,609 x 384 43.1 -> 17.3 us per query, 100k x 768
                                                              // 56.9 -> 49.2, 250k x 768 119.6 -> 110.6; 1M x 768 keeps the pipeline)

static bool hybrid_lane_route(const anrag_index *idx) {
    static const int64_t max_bytes = [] {
        const char *e = getenv("ANRAG_HYBRID_LANES_MAX_MB");  // measurements only (0 = the pipeline for every corpus)
        return e ? (int64_t)atoll(e) << 20 : kHybridLanesMaxBytes;
    }();
    return idx->n_lanes >= 1 && idx->primary == idx->own_primary && idx->secondary == idx->own_secondary &&
           idx->fusion == idx->own_fusion && idx->d_emb != nullptr && dense_scan_has_shape(idx) &&
           (int64_t)idx->n_rows * idx->dim * 4 <= max_bytes;
}

static int lanes_activate(anrag_index *idx) {
    if (idx->lanes_active) return ANRAG_OK;
    if (int rc = settle_pipeline(idx)) return rc;  // the pipeline's queries own the same list sets
    idx->lanes_active = true;
    idx->lane_rr = 0;
    idx->lanes_in_use = (int64_t)idx->n_rows * idx->dim * 4 > kScanLanesMaxBytes ? 1 : idx->n_lanes;
    return ANRAG_OK;
}

// the stream the NEXT lane query of this index will be launched on (the caller holds the index's lock): operands staged on
// it are in place when the query's kernels start, results copied on it follow the tail
static int hybrid_lane_peek(anrag_index *idx, hipStream_t *st) {
    if (int rc = lanes_activate(idx)) return rc;
    *st = idx->lane[idx->lane_rr % (uint64_t)idx->lanes_in_use].st;
    return ANRAG_OK;
}

static int hybrid_lane_enqueue(anrag_index *idx, TailMode tail, const GroupQuery &q, int32_t k, double w_dense,
                               double w_bm25, double wrrf_k, int32_t top_n, const uint32_t *d_allow_dense,
                               const uint32_t *d_allow_bm25) {
    const bool use_dense = q.d_query != nullptr && (tail != kTailFuse || w_dense > 0.0);
    bool use_bm25 = idx->d_post_doc != nullptr && (tail != kTailFuse || w_bm25 > 0.0) &&
                    (q.n_terms > 0 || (!use_dense && tail != kTailFuse));
    ANRAG_REQUIRE(use_dense || use_bm25, "hybrid search with neither a dense nor a BM25 leg");
    int rc;
    if ((rc = lanes_activate(idx))) return rc;
    const int L = idx->lanes_in_use;
    const int l = (int)(idx->lane_rr++ % (uint64_t)L);
    anrag_index::ScanLane &ln = idx->lane[l];
    const int per_lane = kPipeSlots / L, every = per_lane / 4;
    const uint64_t ls = ln.count;
    const int set = l * per_lane + (int)(ls % (uint64_t)per_lane);
    if (ls >= (uint64_t)per_lane) {  // the set's previous user: as in dense_single_enqueue (its event also covers a tail
                                     // that ran right behind its own launches)
        const int64_t m = ((int64_t)ls - per_lane + 2 + every - 1) / every - 1;
        hipEvent_t ev = ln.ev[m % 4];
        if (hipEventQuery(ev) != hipSuccess) ANRAG_HIP(hipEventSynchronize(ev));
    }
    if (use_dense) {
        if ((rc = launch_dense_scan_group(idx, ln.st, &q.d_query, 1, k, d_allow_dense, nullptr, &set, 0,
                                          ln.pending ? &ln.tail : nullptr)))
            return rc;
        ln.pending = false;
    } else if (ln.pending) {  // no scan launch to carry a dense query's merge: launch it
        ln.pending = false;
        if ((rc = launch_tail(idx, ln.st, ln.tail.set, true, false, ln.tail.k, kTailCandidates, 0, 0, 0, 0, ln.tail.out, nullptr)))
            return rc;
    }
    if (use_bm25 && (rc = launch_bm25_lists_group(idx, ln.st, &q.d_terms, &q.n_terms, 1, k, d_allow_bm25, nullptr, &set)))
        return rc;
    anrag_candidate *out = q.d_out;
    int32_t *cnt = q.d_count;
    if ((rc = launch_tail_group(idx, ln.st, &set, use_dense, &use_bm25, 1, k, tail, w_dense, w_bm25, wrrf_k, top_n, &out,
                                cnt ? &cnt : nullptr)))
        return rc;
    ln.count = ls + 1;
    if (ln.count % (uint64_t)every == 0) ANRAG_HIP(hipEventRecord(ln.ev[(ln.count / every - 1) % 4], ln.st));
    idx->hyb_outstanding = true;
    return ANRAG_OK;
}

static int hybrid_enqueue(anrag_index *idx, TailMode tail, const float *d_query, const int32_t *d_terms,
                          int32_t n_terms, int32_t k, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                          const uint32_t *d_allow_dense, const uint32_t *d_allow_bm25, anrag_candidate *d_out,
                          int32_t *d_count) {
    const GroupQuery q{d_query, d_terms, n_terms, d_out, d_count};
    return hybrid_enqueue_group(idx, tail, &q, 1, k, w_dense, w_bm25, wrrf_k, top_n, d_allow_dense, d_allow_bm25);
}

int anrag_hybrid_candidates_device(anrag_index *idx, const float *d_query, const int32_t *d_term_ids, int32_t n_terms,
                                   int32_t k, const uint32_t *d_allow_dense_bits, const uint32_t *d_allow_bm25_bits,
                                   anrag_candidate *d_out) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(d_query && d_out, "NULL operand");
    ANRAG_REQUIRE(idx->d_emb != nullptr && idx->d_post_doc != nullptr, "needs both a dense and a BM25 shard");
    ANRAG_REQUIRE(n_terms >= 0 && (n_terms == 0 || d_term_ids), "n_terms %d with NULL term ids", n_terms);
    ANRAG_REQUIRE(k > 0 && k <= ANRAG_FUSED_K_MAX, "1 <= k <= %d", ANRAG_FUSED_K_MAX);
    return hybrid_enqueue(idx, kTailCandidates2k, d_query, d_term_ids, n_terms, k, 1.0, 1.0, 0.0, 0, d_allow_dense_bits,
                          d_allow_bm25_bits, d_out, nullptr);
}

int anrag_hybrid_candidates_group_device(anrag_index *idx, const float *const *d_queries,
                                         const int32_t *const *d_term_ids, const int32_t *n_terms, int32_t n_queries,
                                         int32_t k, const uint32_t *d_allow_dense_bits,
                                         const uint32_t *d_allow_bm25_bits, anrag_candidate *const *d_out) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(d_queries && d_term_ids && n_terms && d_out, "NULL operand");
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= 4096, "n_queries %d out of range", n_queries);
    ANRAG_REQUIRE(idx->d_emb != nullptr && idx->d_post_doc != nullptr, "needs both a dense and a BM25 shard");
    ANRAG_REQUIRE(k > 0 && k <= ANRAG_FUSED_K_MAX, "1 <= k <= %d", ANRAG_FUSED_K_MAX);
    for (int32_t i = 0; i < n_queries; ++i)
        ANRAG_REQUIRE(d_queries[i] && d_out[i] && n_terms[i] >= 0 && (n_terms[i] == 0 || d_term_ids[i]),
                      "query %d: needs a query vector, an output block and term ids for its n_terms", i);
    for (int32_t q0 = 0; q0 < n_queries; q0 += kScanGroup) {
        const int n = n_queries - q0 < kScanGroup ? n_queries - q0 : kScanGroup;
        GroupQuery g[kScanGroup];
        for (int i = 0; i < n; ++i)
            g[i] = GroupQuery{d_queries[q0 + i], d_term_ids[q0 + i], n_terms[q0 + i], d_out[q0 + i], nullptr};
        int rc = hybrid_enqueue_group(idx, kTailCandidates2k, g, n, k, 1.0, 1.0, 0.0, 0, d_allow_dense_bits,
                                      d_allow_bm25_bits);
        if (rc) return rc;
    }
    return ANRAG_OK;
}

int anrag_hybrid_search_device(anrag_index *idx, const float *d_query, const int32_t *d_term_ids, int32_t n_terms,
                               int32_t similarity_k, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                               const uint32_t *d_allow_dense_bits, const uint32_t *d_allow_bm25_bits,
                               anrag_candidate *d_out, int32_t *d_count) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(d_out && d_count, "NULL operand");
    ANRAG_REQUIRE(similarity_k > 0 && similarity_k <= ANRAG_FUSED_K_MAX, "fused hybrid serves 1 <= similarity_k <= %d",
                  ANRAG_FUSED_K_MAX);
    ANRAG_REQUIRE(top_n > 0 && top_n <= 2 * ANRAG_FUSED_K_MAX, "top_n %d out of range", top_n);
    if (hybrid_lane_route(idx) && d_query != nullptr && (!d_allow_dense_bits || idx->d_dense_src))
        return hybrid_lane_enqueue(idx, kTailFuse, GroupQuery{d_query, d_term_ids, n_terms, d_out, d_count}, similarity_k,
                                   w_dense, w_bm25, wrrf_k, top_n, d_allow_dense_bits, d_allow_bm25_bits);
    return hybrid_enqueue(idx, kTailFuse, d_query, d_term_ids, n_terms, similarity_k, w_dense, w_bm25, wrrf_k, top_n,
                          d_allow_dense_bits, d_allow_bm25_bits, d_out, d_count);
}

// ---- host-pointer hybrid query.  Staging is per pipeline slot (pinned host block + device operands + result
// records), and the caller holds the index lock only while it ENQUEUES: it waits for its own result outside the
// lock, so callers on several threads (the reference shares one SearchEngine across Streamlit session threads,
// src/app.py:17-27) overlap like back-to-back device calls do -- scans adjacent on the primary stream.
static constexpr size_t kSlotTerms = 4096 * sizeof(int32_t), kSlotAllow = 2048 * sizeof(uint32_t),
                        kSlotOut = 2 * ANRAG_FUSED_K_MAX * sizeof(anrag_candidate);

static void free_host_slots(anrag_index *idx) {
    if (idx->host_slots_h) (void)counted_host_free(idx->host_slots_h);
    if (idx->host_slots_d) (void)counted_free(idx->host_slots_d);
    idx->host_slots_h = nullptr;
    idx->host_slots_d = nullptr;
    for (auto &hs : idx->host_slot) {
        if (hs.done) (void)hipEventDestroy(hs.done);
        hs = anrag_index::HostSlot();
    }
    idx->host_ring.sized_for = 0;
}

// (re)allocate the slots' staging for rows of `dim` floats: ONE pinned block and ONE device block, carved per slot:
// query | terms | allow (dense) | allow (BM25) | result records | count.  Called by the ring when no slot is busy.
static int alloc_host_slots(anrag_index *idx, int32_t dim) {
    free_host_slots(idx);
    const size_t qbytes = ((size_t)dim * sizeof(float) + 255) / 256 * 256;
    const size_t slot_bytes = qbytes + kSlotTerms + 2 * kSlotAllow + kSlotOut + 256;
    ANRAG_HIP(counted_host_malloc(reinterpret_cast<void **>(&idx->host_slots_h), slot_bytes * kPipeSlots, hipHostMallocDefault));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->host_slots_d), slot_bytes * kPipeSlots));
    for (int i = 0; i < kPipeSlots; ++i) {
        anrag_index::HostSlot &hs = idx->host_slot[i];
        hs.h = idx->host_slots_h + (size_t)i * slot_bytes;
        char *d = idx->host_slots_d + (size_t)i * slot_bytes;
        hs.d_query = reinterpret_cast<float *>(d);
        hs.d_terms = reinterpret_cast<int32_t *>(d + qbytes);
        hs.d_allow_a = reinterpret_cast<uint32_t *>(d + qbytes + kSlotTerms);
        hs.d_allow_b = reinterpret_cast<uint32_t *>(d + qbytes + kSlotTerms + kSlotAllow);
        hs.d_out = reinterpret_cast<anrag_candidate *>(d + qbytes + kSlotTerms + 2 * kSlotAllow);
        hs.d_count = reinterpret_cast<int32_t *>(d + qbytes + kSlotTerms + 2 * kSlotAllow + kSlotOut);
        ANRAG_HIP(hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
    }
    return ANRAG_OK;
}

// allow list (one byte per source) -> bitmap in the slot's pinned block -> device, asynchronously
static int stage_allow_slot(hipStream_t st, const uint8_t *allow, int32_t n_sources, uint32_t *h_bits, uint32_t *d_bits,
                            const uint32_t **out) {
    *out = nullptr;
    if (!allow) return ANRAG_OK;
    ANRAG_REQUIRE(n_sources >= 0 && n_sources <= 65536, "n_sources %d out of range [0, 65536]", n_sources);
    memset(h_bits, 0, kSlotAllow);
    for (int32_t s = 0; s < n_sources; ++s)
        if (allow[s]) h_bits[s >> 5] |= 1u << (s & 31);
    ANRAG_HIP(hipMemcpyAsync(d_bits, h_bits, kSlotAllow, hipMemcpyHostToDevice, st));
    *out = d_bits;
    return ANRAG_OK;
}

// The device side of one host-pointer hybrid query, as host_slots.hpp's Backend: the slot / sequence / waiter
// bookkeeping itself lives there (GPU-free, exercised under ThreadSanitizer by tests/test_host_slots_tsan.py).
namespace {
struct HybridHostQuery {
    anrag_index *idx;
    const float *query;
    const int32_t *term_ids;
    int32_t n_terms, similarity_k;
    double w_dense, w_bm25, wrrf_k;
    int32_t top_n;
    const uint8_t *allow_dense;
    int32_t n_dense_sources;
    const uint8_t *allow_bm25;
    int32_t n_bm25_sources;
    int64_t *out_id;
    double *out_score;
    int32_t *out_count;
    bool dense = false;

    size_t qbytes() const { return ((size_t)idx->host_ring.sized_for * sizeof(float) + 255) / 256 * 256; }

    int prepare(std::unique_lock<std::mutex> &lock) {
        ANRAG_REQUIRE(out_id && out_score && out_count, "NULL operand");
        ANRAG_REQUIRE(similarity_k > 0 && similarity_k <= ANRAG_FUSED_K_MAX, "fused hybrid serves 1 <= similarity_k <= %d",
                      ANRAG_FUSED_K_MAX);
        ANRAG_REQUIRE(top_n > 0 && top_n <= 2 * ANRAG_FUSED_K_MAX, "top_n %d out of range", top_n);
        ANRAG_REQUIRE(!(allow_dense && !idx->d_dense_src) && !(allow_bm25 && !idx->d_bm25_src),
                      "a source filter needs source ids");
        ANRAG_REQUIRE(n_terms >= 0 && n_terms <= 4096, "n_terms %d out of range [0, 4096]", n_terms);
        ANRAG_REQUIRE(n_terms == 0 || term_ids != nullptr, "term_ids is NULL");
        dense = idx->d_emb && w_dense > 0.0;
        ANRAG_REQUIRE(!dense || query != nullptr, "query is NULL");
        return idx->host_ring.ensure_size(lock, idx->dim > 0 ? idx->dim : 1,
                                          [&](int32_t dim) { return alloc_host_slots(idx, dim); });
    }
    uint64_t next_seq() const { return idx->hyb_seq; }
    int enqueue(int s) {
        anrag_index::HostSlot &hs = idx->host_slot[s];
        hipStream_t P = idx->primary, S = idx->secondary;
        // a corpus of at most 1 GiB: everything of the query -- operands up, three kernels, results down -- on one lane
        // stream (hybrid_lane_enqueue); the host slot is still the ring's (the sequence number advances as in the pipeline)
        const bool lanes = dense && hybrid_lane_route(idx);
        if (lanes) {
            int r0;
            if ((r0 = hybrid_lane_peek(idx, &P))) return r0;
            S = P;
        }
        char *h_terms = hs.h + qbytes(), *h_aa = h_terms + kSlotTerms, *h_ab = h_aa + kSlotAllow, *h_out = h_ab + kSlotAllow;
        char *h_cnt = h_out + kSlotOut;
        const uint32_t *d_ad = nullptr, *d_ab = nullptr;
        int r;
        if ((r = stage_allow_slot(P, allow_dense, n_dense_sources, reinterpret_cast<uint32_t *>(h_aa), hs.d_allow_a, &d_ad)))
            return r;
        if ((r = stage_allow_slot(S, allow_bm25, n_bm25_sources, reinterpret_cast<uint32_t *>(h_ab), hs.d_allow_b, &d_ab)))
            return r;
        if (dense) {
            memcpy(hs.h, query, (size_t)idx->dim * sizeof(float));
            ANRAG_HIP(hipMemcpyAsync(hs.d_query, hs.h, (size_t)idx->dim * sizeof(float), hipMemcpyHostToDevice, P));
        }
        if (n_terms > 0) {
            memcpy(h_terms, term_ids, (size_t)n_terms * sizeof(int32_t));
            ANRAG_HIP(hipMemcpyAsync(hs.d_terms, h_terms, (size_t)n_terms * sizeof(int32_t), hipMemcpyHostToDevice, S));
        }
        if (lanes) {
            if ((r = hybrid_lane_enqueue(idx, kTailFuse, GroupQuery{hs.d_query, hs.d_terms, n_terms, hs.d_out, hs.d_count},
                                         similarity_k, w_dense, w_bm25, wrrf_k, top_n, d_ad, d_ab)))
                return r;
            idx->hyb_seq += 1;  // the ring's slot of the next caller
        } else if ((r = hybrid_enqueue(idx, kTailFuse, hs.d_query, hs.d_terms, n_terms, similarity_k, w_dense, w_bm25, wrrf_k,
                                       top_n, d_ad, d_ab, hs.d_out, hs.d_count))) {
            // (takes pipeline slot hyb_seq % kPipeSlots == s and advances the sequence number)
            return r;
        }
        // the fused tail runs on the secondary stream: the results follow it down
        ANRAG_HIP(hipMemcpyAsync(h_out, hs.d_out, (size_t)top_n * sizeof(anrag_candidate), hipMemcpyDeviceToHost, S));
        ANRAG_HIP(hipMemcpyAsync(h_cnt, hs.d_count, sizeof(int32_t), hipMemcpyDeviceToHost, S));
        ANRAG_HIP(hipEventRecord(hs.done, S));
        return ANRAG_OK;
    }
    void drain() { (void)sync_all(idx); }
    int wait(int s) {
        const hipError_t e = hipEventSynchronize(idx->host_slot[s].done);
        if (e != hipSuccess) {
            set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
            return ANRAG_ERR_HIP;
        }
        return ANRAG_OK;
    }
    void fetch(int s) {
        const char *h_out = idx->host_slot[s].h + qbytes() + kSlotTerms + 2 * kSlotAllow;
        const anrag_candidate *h_cand = reinterpret_cast<const anrag_candidate *>(h_out);
        const int32_t cnt = std::min(*reinterpret_cast<const int32_t *>(h_out + kSlotOut), top_n);
        for (int32_t i = 0; i < cnt; ++i) {
            out_id[i] = h_cand[i].doc;
            out_score[i] = h_cand[i].score;
        }
        *out_count = cnt;
    }
};
}  // namespace

int anrag_hybrid_search(anrag_index *idx, const float *query, const int32_t *term_ids, int32_t n_terms,
                        int32_t similarity_k, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                        const uint8_t *allow_dense, int32_t n_dense_sources, const uint8_t *allow_bm25,
                        int32_t n_bm25_sources, int64_t *out_id, double *out_score, int32_t *out_count) {
    ANRAG_REQUIRE(idx != nullptr, "index handle is NULL");
    DeviceGuard guard(idx->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", idx->device);
        return ANRAG_ERR_HIP;
    }
    HybridHostQuery q{idx, query, term_ids, n_terms, similarity_k, w_dense, w_bm25, wrrf_k, top_n, allow_dense,
                      n_dense_sources, allow_bm25, n_bm25_sources, out_id, out_score, out_count};
    return host_slot_query(idx->mu, idx->host_ring, q);
}

// Many hybrid queries from host memory through the device pipeline: operands go up once, the queries are
// enqueued back to back (scans adjacent on the primary stream, everything else underneath), one host sync, results
// come down once.  What a host caller with a LIST of queries (an evaluation run: retrieval_eval.py loops over
// 8,168 of them) should use instead of n host-synchronous anrag_hybrid_search calls.
int anrag_hybrid_search_batch(anrag_index *idx, const float *queries, const int32_t *term_ids,
                              const int64_t *term_offsets, int32_t n_queries, int32_t similarity_k, double w_dense,
                              double w_bm25, double wrrf_k, int32_t top_n, const uint8_t *allow_dense,
                              int32_t n_dense_sources, const uint8_t *allow_bm25, int32_t n_bm25_sources,
                              int64_t *out_id, double *out_score, int32_t *out_count) {
    // A LIST of 16 queries or more with both legs active and row numbers as document ids is served by the ranking route
    // (rank_batch.hip): its score tiles read the corpus once per 16 / 32 queries where the pipeline below -- the batch = 1
    // kernels back to back -- reads it once per query (1M x 768 top-10 lists: 41-45 against 434 us per query); same lists,
    // same fused scores, same padding.  ANRAG_BATCH_PIPELINE=1 keeps the pipeline (measurements, tests of both).
    if (idx && queries && term_offsets && out_id && out_score && out_count && n_queries >= kRankRouteMin &&
        similarity_k > 0 && similarity_k <= ANRAG_FUSED_K_MAX && top_n > 0 && top_n <= 2 * ANRAG_FUSED_K_MAX &&
        w_dense > 0.0 && w_bm25 > 0.0 && idx->d_emb && idx->d_post_doc && !idx->d_dense_doc && !idx->d_bm25_doc &&
        idx->dense_doc_base == 0 && idx->bm25_doc_base == 0 && !(allow_dense && !idx->d_dense_src) &&
        !(allow_bm25 && !idx->d_bm25_src) && !getenv("ANRAG_BATCH_PIPELINE")) {
        anrag_rank_leg legs[2] = {};
        legs[0].idx = idx;
        legs[0].kind = ANRAG_LEG_DENSE;
        legs[0].queries = queries;
        legs[0].allow_source = allow_dense;
        legs[0].n_sources = n_dense_sources;
        legs[0].weight = w_dense;
        legs[1].idx = idx;
        legs[1].kind = ANRAG_LEG_BM25;
        legs[1].term_ids = term_ids;
        legs[1].term_offsets = term_offsets;
        legs[1].allow_source = allow_bm25;
        legs[1].n_sources = n_bm25_sources;
        legs[1].weight = w_bm25;
        const int64_t space = idx->n_rows > idx->n_docs ? idx->n_rows : idx->n_docs;
        return anrag_rank_batch(legs, 2, n_queries, similarity_k, wrrf_k, top_n, space, out_id, out_score, out_count, nullptr,
                                nullptr);
    }
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(out_id && out_score && out_count && term_offsets, "NULL operand");
    ANRAG_REQUIRE(n_queries >= 0 && n_queries <= (1 << 20), "n_queries %d out of range", n_queries);
    ANRAG_REQUIRE(similarity_k > 0 && similarity_k <= ANRAG_FUSED_K_MAX, "fused hybrid serves 1 <= similarity_k <= %d",
                  ANRAG_FUSED_K_MAX);
    ANRAG_REQUIRE(top_n > 0 && top_n <= 2 * ANRAG_FUSED_K_MAX, "top_n %d out of range", top_n);
    ANRAG_REQUIRE(!(allow_dense && !idx->d_dense_src) && !(allow_bm25 && !idx->d_bm25_src),
                  "a source filter needs source ids");
    if (n_queries == 0) return ANRAG_OK;
    const bool dense = idx->d_emb && w_dense > 0.0;
    ANRAG_REQUIRE(!dense || queries != nullptr, "queries is NULL");
    const int64_t total_terms = term_offsets[n_queries];
    ANRAG_REQUIRE(term_offsets[0] == 0 && total_terms >= 0 && (total_terms == 0 || term_ids), "bad term offsets");
    for (int32_t q = 0; q < n_queries; ++q)
        ANRAG_REQUIRE(term_offsets[q + 1] >= term_offsets[q] && term_offsets[q + 1] - term_offsets[q] <= 4096,
                      "query %d: term count out of range [0, 4096]", q);
    int rc;
    if ((rc = settle_pipeline(idx))) return rc;
    if ((rc = ensure_common_workspace(idx))) return rc;
    hipStream_t P = idx->primary;
    char *pin = static_cast<char *>(idx->h_pinned);
    const uint32_t *d_ad = nullptr, *d_ab = nullptr;
    if ((rc = stage_allow(idx, P, allow_dense, n_dense_sources, idx->d_allow_a, reinterpret_cast<uint32_t *>(pin), &d_ad)))
        return rc;
    if ((rc = stage_allow(idx, P, allow_bm25, n_bm25_sources, idx->d_allow_b, reinterpret_cast<uint32_t *>(pin + 8192),
                          &d_ab)))
        return rc;
    // one call's operands and results: carved from the index's grow-only call pool, so a steady-state loop of list
    // calls allocates nothing (hipMalloc / hipFree synchronise the whole device)
    Carver measure(nullptr);
    auto carve = [&](Carver &c, float *&q, int32_t *&t, anrag_candidate *&o, int32_t *&n) {
        q = dense ? c.take<float>((int64_t)n_queries * idx->dim) : nullptr;
        t = total_terms > 0 ? c.take<int32_t>(total_terms) : nullptr;
        o = c.take<anrag_candidate>((int64_t)n_queries * top_n);
        n = c.take<int32_t>(n_queries);
    };
    float *d_q = nullptr;
    int32_t *d_t = nullptr, *d_cnt = nullptr;
    anrag_candidate *d_out = nullptr;
    carve(measure, d_q, d_t, d_out, d_cnt);
    if ((rc = ensure_pool(idx, idx->call_pool, measure.at))) return rc;
    Carver cv(idx->call_pool.p);
    carve(cv, d_q, d_t, d_out, d_cnt);
    std::vector<anrag_candidate> h_out;
    std::vector<int32_t> h_cnt;
    auto body = [&]() -> int {
        if (dense)
            ANRAG_HIP(hipMemcpyAsync(d_q, queries, (size_t)n_queries * idx->dim * sizeof(float), hipMemcpyHostToDevice, P));
        if (total_terms > 0)
            ANRAG_HIP(hipMemcpyAsync(d_t, term_ids, (size_t)total_terms * sizeof(int32_t), hipMemcpyHostToDevice, P));
        ANRAG_HIP(hipMemsetAsync(d_cnt, 0, (size_t)n_queries * sizeof(int32_t), P));
        ANRAG_HIP(hipStreamSynchronize(P));  // the other streams read the staged operands
        for (int32_t q0 = 0; q0 < n_queries; q0 += kScanGroup) {  // one scan launch per group of queries
            const int n = n_queries - q0 < kScanGroup ? n_queries - q0 : kScanGroup;
            GroupQuery g[kScanGroup];
            for (int i = 0; i < n; ++i) {
                const int32_t q = q0 + i;
                g[i] = GroupQuery{dense ? d_q + (int64_t)q * idx->dim : nullptr, d_t ? d_t + term_offsets[q] : nullptr,
                                  (int32_t)(term_offsets[q + 1] - term_offsets[q]), d_out + (int64_t)q * top_n, d_cnt + q};
            }
            int r = hybrid_enqueue_group(idx, kTailFuse, g, n, similarity_k, w_dense, w_bm25, wrrf_k, top_n, d_ad, d_ab);
            if (r) return r;
        }
        if (int r = sync_all(idx)) return r;
        h_out.resize((size_t)n_queries * top_n);
        h_cnt.resize(n_queries);
        ANRAG_HIP(hipMemcpy(h_out.data(), d_out, h_out.size() * sizeof(anrag_candidate), hipMemcpyDeviceToHost));
        ANRAG_HIP(hipMemcpy(h_cnt.data(), d_cnt, h_cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        return ANRAG_OK;
    };
    rc = body();
    if (rc) {
        (void)sync_all(idx);  // nothing may still read the pool when the next call carves it again
        return rc;
    }
    for (int32_t q = 0; q < n_queries; ++q) {
        const int32_t cnt = std::min(h_cnt[q], top_n);
        for (int32_t i = 0; i < top_n; ++i) {
            out_id[(int64_t)q * top_n + i] = i < cnt ? h_out[(int64_t)q * top_n + i].doc : -1;
            out_score[(int64_t)q * top_n + i] = i < cnt ? h_out[(int64_t)q * top_n + i].score : -__builtin_huge_val();
        }
        out_count[q] = cnt;
    }
    return ANRAG_OK;
}

// ------------------------------------------------------------------ sharded merge
int anrag_merge_candidates_device(anrag_index *idx, const anrag_candidate *d_lists, int32_t n_lists, int32_t k,
                                  int64_t list_stride, anrag_candidate *d_out) {
    ANRAG_ENTER(idx);
    if (int rc0 = settle_pipeline(idx)) return rc0;
    ANRAG_REQUIRE(d_lists && d_out, "NULL operand");
    ANRAG_REQUIRE(n_lists > 0 && k > 0 && k <= ANRAG_FUSED_K_MAX, "need n_lists > 0 and 1 <= k <= %d",
                  ANRAG_FUSED_K_MAX);
    ANRAG_REQUIRE(list_stride >= k, "list_stride %lld < k", (long long)list_stride);
    return launch_merge_candidates(idx, idx->fusion, d_lists, n_lists, k, list_stride, d_out);
}

int anrag_merge_fuse_device(anrag_index *idx, const anrag_candidate *d_lists, int32_t n_lists, int32_t k,
                            int64_t list_stride, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                            int32_t n_queries, anrag_candidate *d_out, int32_t *d_count) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(d_lists && d_out && d_count, "NULL operand");
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= 65535, "n_queries %d out of range", n_queries);
    ANRAG_REQUIRE(n_lists > 0 && k > 0 && k <= ANRAG_FUSED_K_MAX, "need n_lists > 0 and 1 <= k <= %d",
                  ANRAG_FUSED_K_MAX);
    ANRAG_REQUIRE(list_stride >= 2 * (int64_t)k * n_queries, "list_stride %lld < 2k * n_queries", (long long)list_stride);
    ANRAG_REQUIRE(top_n > 0 && top_n <= 2 * ANRAG_FUSED_K_MAX, "top_n %d out of range", top_n);
    return launch_merge_fuse(idx, idx->fusion, d_lists, n_lists, k, list_stride, w_dense, w_bm25, wrrf_k, top_n,
                             n_queries, d_out, d_count);
}

// ------------------------------------------------------------------ device memory helpers
int anrag_device_alloc(anrag_index *idx, int64_t bytes, void **out_ptr) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(out_ptr && bytes > 0, "bad arguments");
    ANRAG_HIP(counted_malloc(out_ptr, (size_t)bytes));
    return ANRAG_OK;
}

int anrag_device_free(anrag_index *idx, void *ptr) {
    ANRAG_ENTER(idx);
    if (ptr) ANRAG_HIP(counted_free(ptr));
    return ANRAG_OK;
}

int anrag_copy_to_device(anrag_index *idx, void *d_dst, const void *h_src, int64_t bytes) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(d_dst && h_src && bytes >= 0, "bad arguments");
    ANRAG_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, idx->primary));
    ANRAG_HIP(hipStreamSynchronize(idx->primary));
    return ANRAG_OK;
}

int anrag_copy_to_host(anrag_index *idx, void *h_dst, const void *d_src, int64_t bytes) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(h_dst && d_src && bytes >= 0, "bad arguments");
    ANRAG_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, idx->primary));
    ANRAG_HIP(hipStreamSynchronize(idx->primary));
    return ANRAG_OK;
}

// ------------------------------------------------------------------ measurement
int anrag_profile_enable(anrag_index *idx, uint32_t kernel_mask) {
    ANRAG_ENTER(idx);
    if (!kernel_mask) {
        int rc = drain_profile(idx);
        if (rc) return rc;
    }
    idx->profiling = kernel_mask != 0;
    idx->profile_mask = kernel_mask;
    return ANRAG_OK;
}

int anrag_profile_set_sampling(anrag_index *idx, int32_t every_n) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(every_n >= 1, "every_n must be >= 1");
    idx->profile_every = (uint32_t)every_n;
    return ANRAG_OK;
}

int anrag_profile_reset(anrag_index *idx) {
    ANRAG_ENTER(idx);
    int rc = drain_profile(idx);
    if (rc) return rc;
    for (int i = 0; i < ANRAG_KERNEL_COUNT; ++i) {
        idx->prof_ms[i] = 0;
        idx->prof_launches[i] = 0;
        idx->prof_units[i] = 0;
    }
    return ANRAG_OK;
}

int anrag_profile_read(anrag_index *idx, int kernel_id, double *out_total_ms, int64_t *out_launches) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(kernel_id >= 0 && kernel_id < ANRAG_KERNEL_COUNT, "kernel id %d out of range", kernel_id);
    ANRAG_REQUIRE(out_total_ms && out_launches, "NULL operand");
    int rc = drain_profile(idx);
    if (rc) return rc;
    *out_total_ms = idx->prof_ms[kernel_id];
    *out_launches = idx->prof_launches[kernel_id];
    return ANRAG_OK;
}

int anrag_profile_read_units(anrag_index *idx, int kernel_id, int64_t *out_units) {
    ANRAG_ENTER(idx);
    ANRAG_REQUIRE(kernel_id >= 0 && kernel_id < ANRAG_KERNEL_COUNT && out_units, "bad arguments");
    int rc = drain_profile(idx);
    if (rc) return rc;
    *out_units = idx->prof_units[kernel_id];
    return ANRAG_OK;
}

int anrag_debug_alloc_calls(int64_t *out_calls) {
    ANRAG_REQUIRE(out_calls != nullptr, "out_calls is NULL");
    *out_calls = alloc_calls();
    return ANRAG_OK;
}

int anrag_index_info(anrag_index *idx, int64_t *dense_rows, int32_t *dense_dim, int64_t *bm25_docs,
                     int64_t *bm25_postings, int64_t *hbm_bytes) {
    ANRAG_ENTER(idx);
    if (dense_rows) *dense_rows = idx->n_rows;
    if (dense_dim) *dense_dim = idx->dim;
    if (bm25_docs) *bm25_docs = idx->n_docs;
    if (bm25_postings) *bm25_postings = idx->n_postings;
    if (hbm_bytes) *hbm_bytes = idx->hbm_bytes;
    return ANRAG_OK;
}

}  // extern "C"

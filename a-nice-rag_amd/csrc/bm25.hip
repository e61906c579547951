// K3 -- BM25 term-at-a-time scorer over CSR postings + fused top-k.
//
// Replaces `bm25.get_scores(tokens)` (rank_bm25.BM25Okapi, called at src/search_engine.py:219) and the
// selection that follows it (:221-243).  Bit-exact fp64: the reference evaluates, per query token q in
// query order (duplicates again),
//     score += idf[q] * ( tf*(k1+1) / ( tf + k1*(1 - b + b*dl/avgdl) ) )
// as numpy fp64 vector expressions, i.e. separately rounded *, /, + in exactly that association.
// The library is built with -ffp-contract=off, fp64 division on gfx950 is correctly rounded, and the
// per-document accumulation order below is the query order, so every score has the same bits.
//
// MI355X design (not the reference's shape):
//   * load time: the bracketed factor depends only on (tf, document), so it is evaluated ONCE per
//     posting on the device (same association) and stored as an fp64 "impact" next to the doc id:
//     a posting is 12 bytes in HBM and a query touches no doc_len / tf at all.
//   * documents are cut into partitions of P <= 4096 consecutive rows; one workgroup owns one
//     partition and keeps its fp64 score slice in LDS (32 KB), so there is no N-sized score array to
//     clear, scatter into and re-read (the reference's np.zeros(N) + N-long adds per token):
//     HBM traffic per query is the postings themselves.
//   * terms are walked in query order inside the workgroup with a barrier between terms: two
//     postings of one term never hit the same document, postings of different terms are ordered.
//   * frequent terms (df >= kFrequentDf) carry a per-partition offset table built at load time, so a
//     workgroup reads only its own slice of the posting list; rare terms are scanned whole (<= 8 KB
//     of doc ids, L2-resident across the workgroups) and range-checked.
//   * selection runs on the LDS slice (wave_topk.hpp), zero-score documents included, as the
//     reference ranks them (:236-243); one sorted list per partition -> tail kernel (tail.hip).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "wave_topk.hpp"

namespace anrag {

// Workgroup size: 1,024 threads for partitions of up to 4,096 documents, 256 for partitions of up to 1,024 (the
// shards of a multi-GPU run: 125k documents over 256 CUs = 512 per partition).  What the kernel costs is
// instructions per wave x waves (it moves a few tens of KB per workgroup): a quarter of the waves for a quarter of
// the documents.
constexpr int kBm25Threads = 1024;
constexpr int kBm25ThreadsSmall = 256;
constexpr int kPostPerThread = 4;  // documents per thread in the selection
constexpr int kMaxPartDocs = kPostPerThread * kBm25Threads;
constexpr int kDocsPerThreadTall = kMaxPartDocs / kBm25ThreadsSmall;  // 16: the 256-thread form over 4,096 documents
constexpr int kFrequentDf = 1024;  // rarer terms carry no partition table: the whole list is gathered, range-checked
constexpr int kTermBatch = 128;

// ------------------------------------------------------------------ load-time kernels
__global__ void bm25_impact_kernel(const int32_t *__restrict__ post_doc, const int32_t *__restrict__ post_tf,
                                   const int32_t *__restrict__ doc_len, int64_t n_postings, double k1, double b,
                                   double avgdl, double *__restrict__ impact) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_postings) return;
    const double f = (double)post_tf[i];
    const double dl = (double)doc_len[post_doc[i]];
    // q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * doc_len / avgdl))   -- rank_bm25, numpy fp64
    const double num = f * (k1 + 1.0);
    const double den = f + k1 * ((1.0 - b) + (b * dl) / avgdl);
    impact[i] = num / den;
}

// part_ptr[slot][p] = first posting of the term whose doc >= p * part_docs (relative to indptr[term])
__global__ void bm25_part_ptr_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ post_doc,
                                     const int32_t *__restrict__ slot_term, int32_t n_slots, int32_t n_parts,
                                     int32_t part_docs, int32_t *__restrict__ part_ptr) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n_slots * (n_parts + 1);
    if (gid >= total) return;
    const int32_t slot = (int32_t)(gid / (n_parts + 1)), p = (int32_t)(gid % (n_parts + 1));
    const int32_t t = slot_term[slot];
    const int64_t base = indptr[t];
    const int32_t df = (int32_t)(indptr[t + 1] - base);
    const int64_t target = (int64_t)p * part_docs;
    int32_t lo = 0, hi = df;  // lower_bound
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if ((int64_t)post_doc[base + mid] < target) lo = mid + 1; else hi = mid;
    }
    part_ptr[gid] = lo;
}

// ------------------------------------------------------------------ diagnostic build only (make dbg):
// wall-clock stamps (100 MHz) at the kernel's phase boundaries, one row per workgroup, in a buffer nothing else reads
#ifdef ANRAG_K3_STAMPS
__device__ unsigned long long g_k3_stamps[4096 * 12];
// stamps go to LDS (a global store in front of a barrier would make the barrier wait ~1 us for its acknowledgement and
// bill it to the phase) and to the buffer at the very end
#define K3_STAMP(i)                                                              \
    do {                                                                         \
        if (threadIdx.x == 0) k3_lds_stamps[(i)] = wall_clock64();               \
    } while (0)
#else
#define K3_STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------ query kernel
// LDS plan of one workgroup (dynamic, ~60 KB):
//   slice[4096] fp64 scores | work area (the selection's survivor / merge lists) | term table | allow bitmap |
//   the gather round's slot table.  Postings are NOT staged in LDS: they go from HBM to registers and from there
//   into the slice.
constexpr int kRoundTerms = 16;                    // query terms gathered per round at most
// per form: work area (the selection's survivor / merge lists), survivors kept in LDS, slots of a gather round
constexpr int work_doubles(int threads) { return threads == kBm25Threads ? 2048 : 1024; }   // 16 KB / 8 KB
constexpr int surv_cap(int threads) { return threads == kBm25Threads ? 1024 : 512; }
// (term, THREADS-posting chunk) pairs per round = loads per thread.  The 256-thread x 16-document form takes 32: its
// 4,096-document partitions hold ~27 chunks of 256 postings for a 9-term query, and every extra round is a slot table, a
// barrier and a memory round trip (7-8 us of its 17 us were two to three rounds); at one wave per SIMD and three
// workgroups per CU it can afford the registers (<= 168).
constexpr int round_slots(int threads, int dpt) { return threads == kBm25ThreadsSmall && dpt == kDocsPerThreadTall ? 24 : 16; }
constexpr int bm25_lds_bytes(int threads, int dpt, bool filter) {
    return dpt * threads * 8 + work_doubles(threads) * 8 + kTermBatch * (8 + 8 + 4 + 4) + (filter ? 2048 * 4 : 0) +
           round_slots(threads, dpt) * (8 + 8 + 4) + 16 + 12 * 8;
}
static_assert(kBm25Threads / kWave * kListLen * (8 + 4) <= work_doubles(kBm25Threads) * 8 &&
                  kBm25ThreadsSmall / kWave * kListLen * (8 + 4) <= work_doubles(kBm25ThreadsSmall) * 8,
              "merge lists must fit in the work area");
static_assert((surv_cap(kBm25Threads) + surv_cap(kBm25Threads) / 2 + kWave + kWave / 2 + 4 + kWave / 2) * 8 <=
                      work_doubles(kBm25Threads) * 8 &&
                  (surv_cap(kBm25ThreadsSmall) + surv_cap(kBm25ThreadsSmall) / 2 + kWave + kWave / 2 + 4 + kWave / 2) * 8 <=
                      work_doubles(kBm25ThreadsSmall) * 8,
              "survivor lists must fit too");
static_assert(kMaxPartDocs == 64 * 64 && kBm25Threads % (64 * 4) == 0 && kBm25ThreadsSmall % (64 * 4) == 0,
              "the selection deals the documents to 64 groups of THREADS / 64 lanes");
static_assert((kFrequentDf + kBm25ThreadsSmall - 1) / kBm25ThreadsSmall <= 16 && kRoundTerms <= kWave &&
                  kMaxPartDocs / kBm25ThreadsSmall <= round_slots(kBm25ThreadsSmall, kDocsPerThreadTall),
              "a term alone must fit a round");

// Register budget of the 1,024-thread form: its 16 waves (4 per SIMD) must fit NEXT TO a scan workgroup (one wave of up
// to 104 VGPRs per SIMD, dense_scan.hip), or K3 can only run between scans instead of under them: 5 waves per SIMD
// asked for = at most 96 VGPRs each.
// The queries of one launch (blockIdx.y): every query has its own workgroups, one per partition, so a launch of a
// GROUP pays the launch once and a CU goes from a workgroup of query i straight to one of query i+1 (the hybrid
// pipeline's exchange groups, anrag_hybrid_search_batch, the full-ranking tiles of rank_batch.hip).
struct Bm25Queries {
    const int32_t *terms[kScanGroupMax];
    int32_t n_terms[kScanGroupMax];
    double *blk_score[kScanGroupMax];   // per-partition lists of the query's slot (selection form)
    uint32_t *blk_row[kScanGroupMax];
    double *scores[kScanGroupMax];      // n_docs scores (SCORES form)
    // SCORES form for LISTS of queries (rank_batch.hip): any number of queries per launch, described in HBM instead of
    // the kernel arguments -- query y's terms are terms_base[term_off[y] .. term_off[y + 1]), its scores go to
    // scores_base + y * scores_stride; a query without terms writes nothing.  term_off == nullptr: the arrays above.
    const int64_t *term_off;
    const int32_t *terms_base;
    double *scores_base;
    int64_t scores_stride;
};

// DPT: documents per thread (THREADS * DPT = the largest partition the form takes).  Three forms:
//   <1024, 4>  4,096-document partitions, 16 waves: round 2's kernel.  VALU-issue bound: ~1,500 instructions per wave
//              whatever the query touches, 16 waves on one CU's four SIMDs = 9-10 us
//   < 256, 4>  partitions of <= 1,024 documents (the shards of a multi-GPU run, small corpora)
//   < 256,16>  4,096-document partitions with a QUARTER of the waves: the per-wave overhead (term table, slot rounds,
//              barriers) is paid by 4 waves instead of 16, the document-proportional part of the selection grows 4x
//              per thread; two workgroups share a CU (LDS), so in query groups one's latency hides under the other
// GROUPED: the launch carries several queries (blockIdx.y picks one: a scalar load at a computed offset in front of
// everything else); a launch of ONE query reads its operands at fixed kernarg offsets with the rest of the arguments.
template <bool FILTER, bool SCORES, int THREADS, int DPT = kPostPerThread, bool GROUPED = true>
__global__ __launch_bounds__(THREADS, THREADS == kBm25Threads ? 5 : (DPT == kDocsPerThreadTall ? 3 : 1)) void bm25_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ post_doc, const double *__restrict__ impact,
    const double *__restrict__ idf, const int32_t *__restrict__ part_slot, const int32_t *__restrict__ part_ptr,
    int32_t n_parts, int32_t part_docs, int64_t n_docs, int64_t n_vocab, Bm25Queries Q, int32_t k,
    const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits, int64_t sentinel) {
    const int qy = GROUPED ? (int)blockIdx.y : 0;
    const bool table = SCORES && GROUPED && Q.term_off != nullptr;
    const int qa = table ? 0 : qy;  // (the arrays hold kScanGroupMax entries; a table launch has more queries than that)
    const int32_t *__restrict__ terms = table ? Q.terms_base + Q.term_off[qy] : Q.terms[qa];
    const int32_t n_terms = table ? (int32_t)(Q.term_off[qy + 1] - Q.term_off[qy]) : Q.n_terms[qa];
    double *__restrict__ blk_score = Q.blk_score[qa];
    uint32_t *__restrict__ blk_row = Q.blk_row[qa];
    double *__restrict__ scores_out = table ? Q.scores_base + (int64_t)qy * Q.scores_stride : Q.scores[qa];
    if (table && n_terms == 0) return;  // (the whole workgroup: nothing has been synchronised yet)
    extern __shared__ __attribute__((aligned(16))) unsigned char bm25_lds[];
    double *slice = reinterpret_cast<double *>(bm25_lds);
    constexpr int WAVES = THREADS / kWave;
    constexpr int kWorkDoubles = work_doubles(THREADS), kSurvCap = surv_cap(THREADS), kRoundSlots = round_slots(THREADS, DPT);
    double *st_val = slice + DPT * THREADS;  // work area
    double *t_w = st_val + kWorkDoubles;
    int64_t *t_base = reinterpret_cast<int64_t *>(t_w + kTermBatch);
    int32_t *t_begin = reinterpret_cast<int32_t *>(t_base + kTermBatch);
    int32_t *t_cnt = t_begin + kTermBatch;
    uint32_t *lds_allow = reinterpret_cast<uint32_t *>(t_cnt + kTermBatch);
    int64_t *s_at = reinterpret_cast<int64_t *>(lds_allow + (FILTER ? 2048 : 0));  // the gather round's slots (see there)
    double *s_w = reinterpret_cast<double *>(s_at + kRoundSlots);
    int32_t *s_cnt = reinterpret_cast<int32_t *>(s_w + kRoundSlots);
    int32_t *r_info = s_cnt + kRoundSlots;
    double *lds_s = st_val;                                              // after the last term
    uint32_t *lds_r = reinterpret_cast<uint32_t *>(st_val + WAVES * kListLen);

    const int tid = threadIdx.x;
#ifdef ANRAG_K3_STAMPS
    unsigned long long *k3_lds_stamps = reinterpret_cast<unsigned long long *>(r_info + 4);
    const unsigned long long k3_cycles0 = __builtin_amdgcn_s_memtime();
#endif
    K3_STAMP(0);
    int lane_zero;  // 0, opaque to the compiler: keeps table reads out of the scalarised (serialised) form
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const int32_t part = blockIdx.x;
    const int64_t lo = (int64_t)part * part_docs;
    const int64_t hi = lo + part_docs < n_docs ? lo + part_docs : n_docs;
    const int32_t len = (int32_t)(hi - lo);

    for (int i = tid; i < part_docs; i += THREADS) slice[i] = 0.0;
    if constexpr (FILTER)
        for (int i = tid; i < 2048; i += THREADS) lds_allow[i] = allow_bits[i];

    // Term-at-a-time, but the HBM round trips are not: the postings of as many consecutive query terms as fit a ROUND
    // are gathered together -- every load of the round independent and in flight at once -- into registers, then
    // APPLIED to the score slice term by term in query order with a barrier between terms: two postings of one term
    // never name the same document, postings of different terms stay ordered, so every fp64 sum is formed in exactly
    // the reference's order.  A round is kRoundSlots slots; slot = 1,024 consecutive postings of one term's run in
    // this partition, thread t <-> posting t of the slot, so nobody searches for "its" term: the slot table (16
    // entries) is built by 16 lanes and read once by everybody.  Idle lanes and idle slots load the SENTINEL posting
    // behind the last real one (document INT32_MAX: outside every partition), so the apply pass needs no counts.
    // (History: one dependent HBM round trip per term took 34 us for 9 terms at 1M documents; packing the round's
    // postings into an LDS staging area with a per-entry term search 24 us, 4.5 us of it the search's chain of
    // dependent LDS reads -- LDS latency is ~300 cycles with 16 waves on the CU.)
    for (int32_t b0 = 0; b0 < n_terms; b0 += kTermBatch) {
        const int32_t nb = n_terms - b0 < kTermBatch ? n_terms - b0 : kTermBatch;
        __syncthreads();  // slice zeroed / previous batch's table no longer read
        if (tid < nb) {
            const int32_t t = terms[b0 + tid];
            double w = 0.0;
            int64_t base = 0;
            int32_t a = 0, e = 0;
            if (t >= 0 && (int64_t)t < n_vocab) {
                w = idf[t];  // `(idf.get(q) or 0)`: an idf of exactly 0 contributes nothing
                base = indptr[t];
                const int32_t slot = part_slot[t];
                if (slot >= 0) {
                    a = part_ptr[(int64_t)slot * (n_parts + 1) + part];
                    e = part_ptr[(int64_t)slot * (n_parts + 1) + part + 1];
                } else {
                    e = (int32_t)(indptr[t + 1] - base);  // < kFrequentDf: scanned whole, range-checked
                }
                if (w == 0.0) e = a;
            }
            t_w[tid] = w;
            t_base[tid] = base;
            t_begin[tid] = a;
            t_cnt[tid] = e - a;  // <= part_docs (one posting per document and term) or < kFrequentDf
        }
        __syncthreads();
        K3_STAMP(1);
        int32_t j0 = 0;
        while (j0 < nb) {
            // the round's terms [j0, j1): greedy, at least one (a term alone always fits), at most kRoundTerms.
            // Lane t < kRoundTerms prices term j0 + t and writes its slots.
            if (tid < kRoundSlots) {  // (wave 0; the term lanes' writes below follow in program order)
                s_at[tid] = sentinel;  // every slot starts idle
                s_w[tid] = 0.0;
                s_cnt[tid] = 0;
            }
            if (tid < kRoundTerms) {
                // lane t's own count; the slots ahead of it by a scan over the 16 lanes (DPP row shifts, no LDS)
                const int32_t mine = j0 + tid < nb ? t_cnt[j0 + tid] : 0;
                const int32_t ns = (mine + THREADS - 1) / THREADS;
                int32_t incl = ns;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);  // row_shr:1, 0 shifted in
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);  // row_shr:2
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);  // row_shr:4
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);  // row_shr:8
                const int32_t before = incl - ns;
                const bool in = j0 + tid < nb && incl <= kRoundSlots;
                if (tid == 0) r_info[1] = 0;
                if (in && ns > 0) {  // same wave: these LDS writes follow the initialisation above in program order
                    const int64_t at = t_base[j0 + tid] + t_begin[j0 + tid];
                    const double w = t_w[j0 + tid];
                    for (int v = 0; v < ns; ++v) {
                        s_at[before + v] = at + (int64_t)v * THREADS;
                        s_w[before + v] = w;
                        s_cnt[before + v] = mine - v * THREADS < THREADS ? mine - v * THREADS : THREADS;
                    }
                    atomicOr(reinterpret_cast<unsigned int *>(&r_info[1]), 1u << (before + ns - 1));  // the term's last slot
                }
                const unsigned long long fits = __ballot(in);  // a prefix of the lanes: slot sums only grow
                if (tid == 0) r_info[0] = __builtin_popcountll(fits);  // terms in the round
            }
            K3_STAMP(11);
            __syncthreads();
            K3_STAMP(10);
            const int32_t j1 = j0 + r_info[0];
            const uint32_t last_of_term = (uint32_t)r_info[1];
            // gather: all of a thread's loads in flight together, one round trip.  The slot table is read through a
            // lane offset the compiler cannot see through (it is 0): a provably uniform LDS read becomes
            // ds_read + s_waitcnt + v_readfirstlane, i.e. one exposed LDS latency (~300 cycles here) PER ENTRY --
            // 48 of them in a row cost 4 us; as ordinary per-lane reads they are issued together and waited for once.
            int32_t g_doc[kRoundSlots];
            double g_val[kRoundSlots];
#pragma unroll
            for (int h = 0; h < kRoundSlots; h += 8) {  // table entries eight at a time: registers
                int32_t c[8];
                int64_t a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c[u] = s_cnt[h + u + lane_zero];
                    a[u] = s_at[h + u + lane_zero];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t at = tid < c[u] ? a[u] + tid : sentinel;
                    g_doc[h + u] = post_doc[at];
                    g_val[h + u] = impact[at];
                }
            }
            K3_STAMP(8);
#ifdef ANRAG_K3_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            K3_STAMP(9);
            K3_STAMP(2);
            // apply, in query order: a slot's postings name distinct documents; a barrier closes every term.  The
            // terms' weights are read half a round at a time: with all 16 live next to the 48 registers of postings
            // the kernel needed 116 VGPRs, and 16 waves of more than 104 cannot share a CU with a scan workgroup (96
            // VGPRs on every SIMD) -- K3 then ran BETWEEN the scans instead of under them (125k-row shard: 61 -> 104
            // us per query).
#pragma unroll
            for (int h = 0; h < kRoundSlots; h += 8) {
                double g_w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g_w[u] = s_w[h + u + lane_zero];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t d = g_doc[h + u];
                    // ds_add_f64: one LDS instruction instead of read -> wait -> add -> write (the slice is touched
                    // once per document and term, so this is the same IEEE addition, without the read's latency in
                    // front of every term's barrier; scores stay bit-identical: tests/test_gpu_bm25.py)
                    if (d >= lo && d < hi)
                        __builtin_amdgcn_ds_atomic_fadd_f64(
                            (__attribute__((address_space(3))) double *)(slice + (d - lo)), g_w[u] * g_val[h + u]);
                    if ((last_of_term >> (h + u)) & 1u) __syncthreads();  // term complete before the next touches the same document
                }
            }
            __syncthreads();  // the slot table is rewritten by the next round
            j0 = j1;
        }
    }
    if (n_terms == 0) __syncthreads();
    K3_STAMP(3);

    if constexpr (SCORES) {
        for (int i = tid; i < len; i += THREADS) {
            bool ok = true;
            if constexpr (FILTER) ok = source_ok(lds_allow, src[lo + i]);
            scores_out[lo + i] = ok ? slice[i] : neg_inf<double>();
        }
    } else {
        // Selection over the LDS slice: the partition's k best under (score desc, row asc), zero-score documents
        // included.  Sorting networks are what a selection costs (a 64-wide fp64 sort is ~400 dependent VALU
        // instructions; 16 of them + a 4-level merge took 8 us), so nothing is sorted:
        //   1. a bound: the documents are dealt to 64 groups (THREADS / 64 lanes each) so that group g holds rows
        //      g, 64 + (g+ROT)%64, 128 + (g+2 ROT)%64, ... -- every group spans the whole partition and the groups' lowest
        //      rows are rows 0..63, which keeps the bound tight when scores tie (all-zero partitions are the common
        //      case); a lane finds the best of its 4 documents, DPP steps the group's; every group best is
        //      RANKED among the 64 by counting (each of the 16 waves compares against 4 of them): the one of rank
        //      k-1, tau, is a document with at least k-1 documents ahead of it;
        //   2. survivors = documents not behind tau (typically 30-45 of 4,096), compacted into LDS;
        //   3. <= 64 survivors: ranked by counting again, each written to its place of the partition's list.  More
        //      (heavy ties): wave 0 takes them 64 at a time through the insertion / merge network; more than kSurvCap:
        //      the general path below (every wave selects, tree merge).
        double *surv_s = st_val;
        uint32_t *surv_r = reinterpret_cast<uint32_t *>(st_val + kSurvCap);
        double *cand_s = st_val + kSurvCap + kSurvCap / 2;
        uint32_t *cand_r = reinterpret_cast<uint32_t *>(cand_s + kWave);
        double *tau_s = cand_s + kWave + kWave / 2;
        uint32_t *tau_r = reinterpret_cast<uint32_t *>(tau_s + 1);
        int32_t *surv_n = reinterpret_cast<int32_t *>(tau_s + 2);
        int32_t *rank_cnt = reinterpret_cast<int32_t *>(tau_s + 3);  // [64] ranks, summed with LDS atomics
        const int lane = tid & (kWave - 1), wave = tid / kWave;
        constexpr int LPG = THREADS / 64;  // lanes per group
        constexpr int ROT = LPG == 16 ? 4 : 16;
        const int g = tid / LPG, gl = tid % LPG;
        constexpr int kCmp = kWave / WAVES;  // comparisons per wave and ranked entry
        double sc[DPT];
        uint32_t rw[DPT];
        uint32_t okbits = 0;  // bit u: the lane's u-th document counts (inside the partition, allowed source)
        double bs = neg_inf<double>();
        uint32_t br = kNoRow;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int j = gl + LPG * u;
            const int i = 64 * j + ((g + ROT * j) & 63);
            bool ok_u = i < len;
            sc[u] = neg_inf<double>();
            rw[u] = kNoRow;
            if (ok_u) {
                sc[u] = slice[i];
                if constexpr (FILTER) ok_u = source_ok(lds_allow, src[lo + i]);
            }
            okbits |= (ok_u ? 1u : 0u) << u;
            if (ok_u) {
                rw[u] = (uint32_t)(lo + i);
                if (br == kNoRow || sc[u] > bs) {  // rows ascend with u: a strict > keeps the lower row among equals
                    bs = sc[u];
                    br = rw[u];
                }
            }
        }
        if (tid == 0) {
            *surv_n = 0;
            *tau_s = neg_inf<double>();
            *tau_r = kNoRow;  // stays so when fewer than k groups hold a document: then everything survives
        }
        if (tid < kWave) rank_cnt[tid] = 0;
        // best of the group: butterfly over its 16 lanes (every lane ends with it)
        cmp_exchange<1>(bs, br, true);
        cmp_exchange<2>(bs, br, true);
        if constexpr (LPG == 16) {
            cmp_exchange<4>(bs, br, true);
            cmp_exchange<8>(bs, br, true);
        }
        if (gl == 0) {
            cand_s[g] = bs;
            cand_r[g] = br;
        }
        __syncthreads();
        {  // rank of group best `lane` among the 64: this wave's share of the comparisons
            const double ms = cand_s[lane];
            const uint32_t mr = cand_r[lane];
            int c = 0;
#pragma unroll
            for (int t = 0; t < kCmp; ++t) c += beats(cand_s[wave * kCmp + t], cand_r[wave * kCmp + t], ms, mr) ? 1 : 0;
            if (c) atomicAdd(&rank_cnt[lane], c);
        }
        __syncthreads();
        if (wave == 0 && rank_cnt[lane] == k - 1 && cand_r[lane] != kNoRow) {  // rows are unique: at most one lane
            *tau_s = cand_s[lane];
            *tau_r = cand_r[lane];
        }
        __syncthreads();
        K3_STAMP(4);
        if (tid < kWave) rank_cnt[tid] = 0;  // for the survivors' ranks (read again only behind the next barrier)
        {
            const double ts = *tau_s;
            const uint32_t tr = *tau_r;
            // the lane's survivors as bits of one register; the wave-wide masks are formed again where they are needed
            // (DPT of them held at once were 32 SGPRs: the 16-document form spilled 27)
            uint32_t keep = 0;
            int total = 0;
#pragma unroll
            for (int u = 0; u < DPT; ++u) {
                const bool kp = ((okbits >> u) & 1u) && !beats(ts, tr, sc[u], rw[u]);
                keep |= (kp ? 1u : 0u) << u;
                total += __builtin_popcountll(__ballot(kp));
            }
            int base = 0;
            if (total > 0) {
                if (lane == 0) base = atomicAdd(surv_n, total);
                base = __builtin_amdgcn_readfirstlane(base);
            }
#pragma unroll
            for (int u = 0; u < DPT; ++u) {
                const unsigned long long mu = __ballot((keep >> u) & 1u);
                if ((mu >> lane) & 1ull) {
                    const int pos = base + __builtin_popcountll(mu & ((1ull << lane) - 1ull));
                    if (pos < kSurvCap) {
                        surv_s[pos] = sc[u];
                        surv_r[pos] = rw[u];
                    }
                }
                base += __builtin_popcountll(mu);
            }
        }
        __syncthreads();
        K3_STAMP(5);
        const int n_surv = *surv_n;
        if (n_surv <= kWave) {
            {
                const bool in = lane < n_surv;
                const double ms = in ? surv_s[lane] : neg_inf<double>();
                const uint32_t mr = in ? surv_r[lane] : kNoRow;
                int c = 0;
#pragma unroll
                for (int t = 0; t < kCmp; ++t) {
                    const int o = wave * kCmp + t;
                    c += (o < n_surv && beats(surv_s[o], surv_r[o], ms, mr)) ? 1 : 0;
                }
                if (c) atomicAdd(&rank_cnt[lane], c);
            }
            __syncthreads();
            if (wave == 0) {
                const bool in = lane < n_surv;
                const int rank = in ? rank_cnt[lane] : lane;  // padding keeps its place behind the survivors
                blk_score[blockIdx.x * kListLen + rank] = in ? surv_s[lane] : neg_inf<double>();
                blk_row[blockIdx.x * kListLen + rank] = in ? surv_r[lane] : kNoRow;
            }
        } else {
            WaveTopK<double> top;
            top.init(k);
            if (n_surv <= kSurvCap) {
                if (wave == 0) {
                    for (int c = 0; c < n_surv; c += kWave) {
                        const bool in = c + lane < n_surv;
                        const double cs = in ? surv_s[c + lane] : neg_inf<double>();
                        const uint32_t cr = in ? surv_r[c + lane] : kNoRow;
                        top.offer_lanes(in && top.admits(cs, cr), cs, cr);
                    }
                }
            } else {
                // general path: every wave selects from its own documents (a thread's best first, so that one sort
                // sets a bound most of the rest fail), then the tree merge of the 16 wave lists
                __syncthreads();  // the survivor area is about to be reused by the merge lists
                double fs = neg_inf<double>();  // the thread's best document (rows ascend with u)
                uint32_t fr = kNoRow;
#pragma unroll
                for (int u = 0; u < DPT; ++u)
                    if (((okbits >> u) & 1u) && (fr == kNoRow || sc[u] > fs)) {
                        fs = sc[u];
                        fr = rw[u];
                    }
                top.offer_lanes(fr != kNoRow && top.admits(fs, fr), fs, fr);
#pragma unroll
                for (int u = 0; u < DPT; ++u)
                    top.offer_lanes(((okbits >> u) & 1u) && rw[u] != fr && top.admits(sc[u], rw[u]), sc[u], rw[u]);
                block_merge(top, lds_s, lds_r, WAVES);
            }
            if (threadIdx.x < kWave) {
                blk_score[blockIdx.x * kListLen + threadIdx.x] = top.s;
                blk_row[blockIdx.x * kListLen + threadIdx.x] = top.r;
            }
        }
        K3_STAMP(6);
#ifdef ANRAG_K3_STAMPS
        const unsigned k3_wg = blockIdx.y * gridDim.x + blockIdx.x;  // (query of the group, partition)
        if (threadIdx.x == 0 && k3_wg < 4096) {
            k3_lds_stamps[7] = __builtin_amdgcn_s_memtime() - k3_cycles0;
            for (int i = 0; i < 12; ++i) g_k3_stamps[k3_wg * 12 + i] = k3_lds_stamps[i];
        }
#endif
    }
}

// ------------------------------------------------------------------ host side
void free_bm25(anrag_index *idx) {
    void *ptrs[] = {idx->d_indptr,   idx->d_post_doc,  idx->d_post_impact, idx->d_idf,
                    idx->d_part_ptr, idx->d_part_slot, idx->d_bm25_src,    idx->d_bm25_doc};
    for (void *p : ptrs)
        if (p) (void)counted_free(p);
    idx->d_indptr = nullptr;
    idx->d_post_doc = nullptr;
    idx->d_post_impact = nullptr;
    idx->d_idf = nullptr;
    idx->d_part_ptr = nullptr;
    idx->d_part_slot = nullptr;
    idx->d_bm25_src = nullptr;
    idx->d_bm25_doc = nullptr;
    if (idx->d_scores_f64) (void)counted_free(idx->d_scores_f64);
    idx->d_scores_f64 = nullptr;
    idx->hbm_bytes -= idx->bm25_hbm_bytes;
    idx->bm25_hbm_bytes = 0;
    idx->n_docs = idx->n_terms = idx->n_postings = 0;
}

template <typename T>
static int bm25_alloc(anrag_index *idx, T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = counted_malloc(reinterpret_cast<void **>(p), (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%lld bytes) failed: %s", (long long)(count * (int64_t)sizeof(T)), hipGetErrorString(e));
        return ANRAG_ERR_NOMEM;
    }
    idx->hbm_bytes += count * (int64_t)sizeof(T);
    idx->bm25_hbm_bytes += count * (int64_t)sizeof(T);
    return ANRAG_OK;
}

int bm25_load(anrag_index *idx, const int64_t *indptr, int64_t n_terms, const int32_t *post_doc,
              const int32_t *post_tf, const double *idf, const int32_t *doc_len, int64_t n_docs, double avgdl,
              double k1, double b, const uint16_t *source_id, const int64_t *doc_id, int64_t doc_id_base) {
    ANRAG_REQUIRE(indptr && post_doc && post_tf && idf && doc_len, "NULL operand");
    ANRAG_REQUIRE(n_terms > 0 && n_terms < 0x7FFFFFFF, "n_terms %lld out of range", (long long)n_terms);
    ANRAG_REQUIRE(n_docs > 0 && n_docs < 0x7FFFFFFF, "n_docs %lld out of range (1 .. 2^31-2 per shard)",
                  (long long)n_docs);
    ANRAG_REQUIRE(indptr[0] == 0, "indptr[0] must be 0");
    ANRAG_REQUIRE(avgdl > 0.0 && avgdl < __builtin_huge_val(), "avgdl must be positive and finite");
    ANRAG_REQUIRE(k1 == k1 && b == b, "k1 / b must be numbers");
    for (int64_t t = 0; t < n_terms; ++t)  // a non-finite idf would make scores NaN / inf, which K3's order excludes
        ANRAG_REQUIRE(idf[t] - idf[t] == 0.0, "idf of term %lld is not finite", (long long)t);
    const int64_t n_postings = indptr[n_terms];
    for (int64_t t = 0; t < n_terms; ++t) {
        ANRAG_REQUIRE(indptr[t + 1] >= indptr[t], "indptr not monotone at term %lld", (long long)t);
        ANRAG_REQUIRE(indptr[t + 1] - indptr[t] <= n_docs, "term %lld has more postings than documents", (long long)t);
    }
    ANRAG_HIP(hipStreamSynchronize(idx->primary));
    ANRAG_HIP(hipStreamSynchronize(idx->secondary));
    free_bm25(idx);
    int rc;
    idx->n_docs = n_docs;
    idx->n_terms = n_terms;
    idx->n_postings = n_postings;
    idx->bm25_k1 = k1;
    idx->bm25_b = b;
    idx->bm25_avgdl = avgdl;
    idx->bm25_doc_base = doc_id_base;
    // partitioning: about `per_cu` workgroups per CU, slices of at most 32 KB
    int per_cu = 1;
    if (const char *env = getenv("ANRAG_BM25_PARTS_PER_CU")) per_cu = std::max(1, std::min(16, atoi(env)));
    int64_t pd = (n_docs + (int64_t)idx->n_cus * per_cu - 1) / ((int64_t)idx->n_cus * per_cu);
    pd = ((pd + 255) / 256) * 256;
    pd = std::max<int64_t>(256, std::min<int64_t>(kMaxPartDocs, pd));
    idx->part_docs = (int32_t)pd;
    idx->n_parts = (int32_t)((n_docs + pd - 1) / pd);

    if ((rc = bm25_alloc(idx, &idx->d_indptr, n_terms + 1))) return rc;
    // + 1: the sentinel posting (document INT32_MAX, impact 0) idle lanes of the query kernel load
    if ((rc = bm25_alloc(idx, &idx->d_post_doc, n_postings + 1))) return rc;
    if ((rc = bm25_alloc(idx, &idx->d_post_impact, n_postings + 1))) return rc;
    {
        const int32_t no_doc = 0x7FFFFFFF;
        const double zero = 0.0;
        ANRAG_HIP(hipMemcpy(idx->d_post_doc + n_postings, &no_doc, sizeof(no_doc), hipMemcpyHostToDevice));
        ANRAG_HIP(hipMemcpy(idx->d_post_impact + n_postings, &zero, sizeof(zero), hipMemcpyHostToDevice));
    }
    if ((rc = bm25_alloc(idx, &idx->d_idf, n_terms))) return rc;
    if ((rc = bm25_alloc(idx, &idx->d_part_slot, n_terms))) return rc;
    ANRAG_HIP(hipMemcpy(idx->d_indptr, indptr, (size_t)(n_terms + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    ANRAG_HIP(hipMemcpy(idx->d_idf, idf, (size_t)n_terms * sizeof(double), hipMemcpyHostToDevice));
    if (n_postings > 0)
        ANRAG_HIP(copy_in(idx, idx->d_post_doc, post_doc, (size_t)n_postings * sizeof(int32_t)));

    // impacts: tf and doc_len are only needed here
    {
        int32_t *d_tf = nullptr, *d_dl = nullptr;
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&d_tf), (size_t)std::max<int64_t>(n_postings, 1) * 4));
        hipError_t e = counted_malloc(reinterpret_cast<void **>(&d_dl), (size_t)n_docs * 4);
        if (e != hipSuccess) {
            (void)counted_free(d_tf);
            set_error("hipMalloc failed: %s", hipGetErrorString(e));
            return ANRAG_ERR_NOMEM;
        }
        e = hipMemcpy(d_dl, doc_len, (size_t)n_docs * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && n_postings > 0) e = copy_in(idx, d_tf, post_tf, (size_t)n_postings * 4);
        if (e == hipSuccess && n_postings > 0) {
            const int64_t blocks = (n_postings + 255) / 256;
            bm25_impact_kernel<<<(unsigned)blocks, 256, 0, idx->primary>>>(idx->d_post_doc, d_tf, d_dl, n_postings, k1,
                                                                           b, avgdl, idx->d_post_impact);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(idx->primary);
        }
        (void)counted_free(d_tf);
        (void)counted_free(d_dl);
        if (e != hipSuccess) {
            set_error("BM25 impact build failed: %s", hipGetErrorString(e));
            return ANRAG_ERR_HIP;
        }
    }
    // per-partition offsets of the frequent terms
    {
        std::vector<int32_t> slot(n_terms, -1), slot_term;
        for (int64_t t = 0; t < n_terms; ++t)
            if (indptr[t + 1] - indptr[t] >= kFrequentDf) {
                slot[t] = (int32_t)slot_term.size();
                slot_term.push_back((int32_t)t);
            }
        ANRAG_HIP(hipMemcpy(idx->d_part_slot, slot.data(), (size_t)n_terms * 4, hipMemcpyHostToDevice));
        const int64_t n_slots = (int64_t)slot_term.size();
        if ((rc = bm25_alloc(idx, &idx->d_part_ptr, n_slots * (idx->n_parts + 1)))) return rc;
        if (n_slots > 0) {
            int32_t *d_slot_term = nullptr;
            ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&d_slot_term), (size_t)n_slots * 4));
            hipError_t e = hipMemcpy(d_slot_term, slot_term.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                const int64_t total = n_slots * (idx->n_parts + 1);
                bm25_part_ptr_kernel<<<(unsigned)((total + 255) / 256), 256, 0, idx->primary>>>(
                    idx->d_indptr, idx->d_post_doc, d_slot_term, (int32_t)n_slots, idx->n_parts, idx->part_docs,
                    idx->d_part_ptr);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(idx->primary);
            }
            (void)counted_free(d_slot_term);
            if (e != hipSuccess) {
                set_error("BM25 partition table build failed: %s", hipGetErrorString(e));
                return ANRAG_ERR_HIP;
            }
        }
    }
    if (source_id) {
        if ((rc = bm25_alloc(idx, &idx->d_bm25_src, n_docs))) return rc;
        ANRAG_HIP(copy_in(idx, idx->d_bm25_src, source_id, (size_t)n_docs * sizeof(uint16_t)));
    }
    if (doc_id) {
        if ((rc = bm25_alloc(idx, &idx->d_bm25_doc, n_docs))) return rc;
        ANRAG_HIP(copy_in(idx, idx->d_bm25_doc, doc_id, (size_t)n_docs * sizeof(int64_t)));
    }
    // per-partition candidate lists
    if (idx->d_blk_score_f64) (void)counted_free(idx->d_blk_score_f64);
    if (idx->d_blk_row_b) (void)counted_free(idx->d_blk_row_b);
    idx->d_blk_score_f64 = nullptr;
    idx->d_blk_row_b = nullptr;
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_blk_score_f64), (size_t)kPipeSlots * idx->n_parts * kListLen * 8));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_blk_row_b), (size_t)kPipeSlots * idx->n_parts * kListLen * 4));
    return ANRAG_OK;
}

// K3 only: per-partition lists (or every score) are left in HBM; the tail kernel finishes the top-k.
// n <= kScanGroupMax queries in ONE launch: query i's lists into list set sets[i], or (d_scores_out given) its n_docs
// scores to d_scores_out[i].
int launch_bm25_lists_group(anrag_index *idx, hipStream_t st, const int32_t *const *d_terms, const int32_t *n_terms,
                            int32_t n_queries, int32_t k, const uint32_t *d_allow_bits, double *const *d_scores_out,
                            const int *sets) {
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= kScanGroupMax, "BM25 group of %d queries", n_queries);
    const uint32_t *allow = idx->d_bm25_src ? d_allow_bits : nullptr;
    int rc;
    // dynamic LDS is asked for explicitly (up to 60 KB per workgroup), per device
#define ANRAG_BM25_ATTR(F, S, T, D, G)                                                                               \
    if ((rc = ensure_dynamic_lds(idx->device, reinterpret_cast<const void *>(&bm25_kernel<F, S, T, D, G>),          \
                                 bm25_lds_bytes(T, D, F))))                                                         \
        return rc
    Bm25Queries Q{};
    for (int i = 0; i < kScanGroupMax; ++i) {
        const int j = i < n_queries ? i : 0;
        Q.terms[i] = d_terms[j];
        Q.n_terms[i] = n_terms[j];
        Q.blk_score[i] = idx->d_blk_score_f64 + (int64_t)(sets ? sets[j] : 0) * idx->n_parts * kListLen;
        Q.blk_row[i] = idx->d_blk_row_b + (int64_t)(sets ? sets[j] : 0) * idx->n_parts * kListLen;
        Q.scores[i] = d_scores_out ? d_scores_out[j] : nullptr;
    }
    {
        LaunchTimer t(idx, ANRAG_KERNEL_BM25, st, n_queries);
        // partitions of <= 1,024 documents: the 256-thread form (one wave per SIMD, cheaper barriers).  Larger ones: a
        // query alone takes the 1,024-thread form (its 16 waves finish a partition soonest: 17.6 against 23.8 us per
        // launch at 1M documents), a GROUP the 256-thread x 16-document form (two workgroups share a CU and a quarter
        // of the waves pay the per-wave overhead: 10.6 against 11.2 us per query at 8 queries per launch, 6.4 against
        // 8.6 for 2-term queries).  ANRAG_BM25_FORM=wide / tall forces one of them (measurements).
        const bool small = idx->part_docs <= kPostPerThread * kBm25ThreadsSmall;
        static const int forced = [] { const char *e = getenv("ANRAG_BM25_FORM"); return !e ? 0 : (e[0] == 'w' ? 1 : (e[0] == 't' ? 2 : 0)); }();
        const bool wide = forced == 1 || (forced == 0 && n_queries == 1);
        const dim3 grid((unsigned)idx->n_parts, (unsigned)n_queries);
#define ANRAG_BM25_TG(F, S, T, D, G)                                                                              \
    do {                                                                                                          \
        ANRAG_BM25_ATTR(F, S, T, D, G);                                                                           \
        bm25_kernel<F, S, T, D, G><<<grid, T, bm25_lds_bytes(T, D, F), st>>>(                                         \
            idx->d_indptr, idx->d_post_doc, idx->d_post_impact, idx->d_idf, idx->d_part_slot, idx->d_part_ptr,    \
            idx->n_parts, idx->part_docs, idx->n_docs, idx->n_terms, Q, k, idx->d_bm25_src, allow,                \
            idx->n_postings);                                                                                     \
    } while (0)
#define ANRAG_BM25_T(F, S, T, D)                                           \
    do {                                                                   \
        if (n_queries == 1) ANRAG_BM25_TG(F, S, T, D, false);              \
        else ANRAG_BM25_TG(F, S, T, D, true);                              \
    } while (0)
#define ANRAG_BM25(F, S)                                                                      \
    do {                                                                                      \
        if (small) ANRAG_BM25_T(F, S, kBm25ThreadsSmall, kPostPerThread);                     \
        else if (wide) ANRAG_BM25_T(F, S, kBm25Threads, kPostPerThread);                      \
        else ANRAG_BM25_T(F, S, kBm25ThreadsSmall, kDocsPerThreadTall);                       \
    } while (0)
        if (d_scores_out) {
            if (allow) ANRAG_BM25(true, true); else ANRAG_BM25(false, true);
        } else {
            if (allow) ANRAG_BM25(true, false); else ANRAG_BM25(false, false);
        }
#undef ANRAG_BM25
#undef ANRAG_BM25_T
#undef ANRAG_BM25_TG
#undef ANRAG_BM25_ATTR
        ANRAG_HIP(hipGetLastError());
    }
    return ANRAG_OK;
}

// Every score of n_queries queries in ONE launch (the full-ranking tiles of rank_batch.hip): query y's terms are
// d_terms_base[d_term_off[y] .. d_term_off[y + 1]) (offsets in HBM, n_queries + 1 of them), its n_docs scores go to
// d_scores_base + y * scores_stride.  A launch per 8 queries was 7 us of latency each on the evaluation corpus (3
// partitions: 24 workgroups per launch on 256 CUs); one launch keeps every CU's three workgroup slots full.
int launch_bm25_scores_table(anrag_index *idx, hipStream_t st, const int32_t *d_terms_base, const int64_t *d_term_off,
                             int32_t n_queries, const uint32_t *d_allow_bits, double *d_scores_base, int64_t scores_stride) {
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= 65535, "BM25 score table of %d queries", n_queries);
    ANRAG_REQUIRE(scores_stride >= idx->n_docs, "score tile rows overlap");
    const uint32_t *allow = idx->d_bm25_src ? d_allow_bits : nullptr;
    Bm25Queries Q{};
    Q.term_off = d_term_off;
    Q.terms_base = d_terms_base;
    Q.scores_base = d_scores_base;
    Q.scores_stride = scores_stride;
    const bool small = idx->part_docs <= kPostPerThread * kBm25ThreadsSmall;
    const dim3 grid((unsigned)idx->n_parts, (unsigned)n_queries);
    int rc;
    LaunchTimer t(idx, ANRAG_KERNEL_BM25, st, n_queries);
#define ANRAG_BM25_TABLE(F, T, D)                                                                                       \
    do {                                                                                                                \
        if ((rc = ensure_dynamic_lds(idx->device, reinterpret_cast<const void *>(&bm25_kernel<F, true, T, D, true>),    \
                                     bm25_lds_bytes(T, D, F))))                                                         \
            return rc;                                                                                                  \
        bm25_kernel<F, true, T, D, true><<<grid, T, bm25_lds_bytes(T, D, F), st>>>(                                      \
            idx->d_indptr, idx->d_post_doc, idx->d_post_impact, idx->d_idf, idx->d_part_slot, idx->d_part_ptr,          \
            idx->n_parts, idx->part_docs, idx->n_docs, idx->n_terms, Q, 0, idx->d_bm25_src, allow, idx->n_postings);    \
    } while (0)
    if (small) {
        if (allow) ANRAG_BM25_TABLE(true, kBm25ThreadsSmall, kPostPerThread); else ANRAG_BM25_TABLE(false, kBm25ThreadsSmall, kPostPerThread);
    } else {
        if (allow) ANRAG_BM25_TABLE(true, kBm25ThreadsSmall, kDocsPerThreadTall); else ANRAG_BM25_TABLE(false, kBm25ThreadsSmall, kDocsPerThreadTall);
    }
#undef ANRAG_BM25_TABLE
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

int launch_bm25_lists(anrag_index *idx, hipStream_t st, const int32_t *d_terms, int32_t n_terms, int32_t k,
                      const uint32_t *d_allow_bits, double *d_scores_out, int set) {
    return launch_bm25_lists_group(idx, st, &d_terms, &n_terms, 1, k, d_allow_bits, d_scores_out ? &d_scores_out : nullptr,
                                   &set);
}

int launch_bm25(anrag_index *idx, hipStream_t st, const int32_t *d_terms, int32_t n_terms, int32_t k,
                const uint32_t *d_allow_bits, anrag_candidate *d_out, double *d_scores_out) {
    int rc = launch_bm25_lists(idx, st, d_terms, n_terms, k, d_allow_bits, d_scores_out, 0);
    if (rc || d_scores_out) return rc;
    return launch_tail(idx, st, 0, /*dense*/ false, /*bm25*/ true, k, kTailCandidates, 0, 0, 0, 0, d_out, nullptr);
}

}  // namespace anrag

#ifdef ANRAG_K3_STAMPS
extern "C" int anrag_debug_k3_stamps(unsigned long long *out, int n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(anrag::g_k3_stamps), (size_t)n_words * 8);
}
#endif

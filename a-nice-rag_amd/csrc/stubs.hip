// Entry points declared in include/anrag.h whose kernels are not written yet.
#include "common.hpp"
namespace anrag {
int dense_search_large_k(anrag_index *, hipStream_t, const float *, int32_t, int32_t, const uint32_t *, int64_t *,
                         float *, int32_t *) {
    set_error("k > %d: large-k path not built yet", ANRAG_FUSED_K_MAX);
    return ANRAG_ERR_STATE;
}
void free_bm25(anrag_index *) {}
}  // namespace anrag
#define NOT_YET(name) \
    { ::anrag::set_error(name ": not built yet"); return ANRAG_ERR_STATE; }
extern "C" {
int anrag_bm25_load(anrag_index *, const int64_t *, int64_t, const int32_t *, const int32_t *, const double *,
                    const int32_t *, int64_t, double, double, double, const uint16_t *, const int64_t *, int64_t)
    NOT_YET("anrag_bm25_load")
int anrag_bm25_search(anrag_index *, const int32_t *, int32_t, int32_t, const uint8_t *, int32_t, int64_t *, double *,
                      int32_t *) NOT_YET("anrag_bm25_search")
int anrag_bm25_search_device(anrag_index *, const int32_t *, int32_t, int32_t, const uint32_t *, anrag_candidate *)
    NOT_YET("anrag_bm25_search_device")
int anrag_bm25_scores(anrag_index *, const int32_t *, int32_t, double *) NOT_YET("anrag_bm25_scores")
int anrag_wrrf(anrag_index *, const int64_t *, const int32_t *, const double *, int32_t, double, int32_t, int64_t *,
               double *, int32_t *) NOT_YET("anrag_wrrf")
int anrag_hybrid_search(anrag_index *, const float *, const int32_t *, int32_t, int32_t, double, double, double,
                        int32_t, const uint8_t *, int32_t, const uint8_t *, int32_t, int64_t *, double *, int32_t *)
    NOT_YET("anrag_hybrid_search")
int anrag_wrrf_device(anrag_index *, const anrag_candidate *, int32_t, const anrag_candidate *, int32_t, double,
                      double, double, int32_t, anrag_candidate *) NOT_YET("anrag_wrrf_device")
}

"""`SearchEngine`: the reference's src/search_engine.py:14-293, method for method, with the
arithmetic on the GPU through libanrag.so.

Same names, arguments, defaults, return types and error behaviour ("log and return empty",
search_engine.py:94-98, :144-146, :267-269, :291-293) so that query_rag_retrieval.py /
retrieval_eval.py-shaped callers run unchanged.  What differs, by design:
  * ties are ordered (score desc, row asc); the reference leaves them to numpy's unstable
    argpartition/argsort (its filtered BM25 path already is row-ascending: a stable sort);
  * dense scores of an fp32 query are fp32 products accumulated in fp32 (within 1e-4 of numpy's BLAS result); an
    fp64 query (the text path, :157) is scored in fp64 like numpy scores it (`anrag_dense_search_f64`);
  * a NaN dense score ranks first, as numpy ranks it, and is reported as +inf;
  * a missing HIP library or GPU raises `AnragError` instead of being swallowed: there is no CPU path.
"""
from __future__ import annotations

import logging
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from ._native import AnragError, FUSED_K_MAX
from .database_manager import ATTR, Bm25Proxy, DenseHandle, FusedPair
from .index import Index
from .preprocess_bm25 import preprocess_text

_FATAL = (-100, -5)  # library not built / no gfx950 device: never "log and return empty"


def _prefixes(filename_type_filter: str) -> Tuple[str, ...]:
    return tuple(p.strip().upper() for p in filename_type_filter.split(","))


def dense_allow(distinct_sources: Sequence[Optional[str]], filename_type_filter: str) -> np.ndarray:
    """search_engine.py:36-48 evaluated once per DISTINCT source string: one prefix -> startswith,
    several -> the reference's un-escaped regex `^(?:A|B)`; a missing source never matches (na=False)."""
    prefixes = _prefixes(filename_type_filter)
    if len(prefixes) == 1:
        test = lambda u: u.startswith(prefixes[0])
    else:
        rx = re.compile("^(?:" + "|".join(prefixes) + ")")
        test = lambda u: rx.search(u) is not None
    return np.array([isinstance(s, str) and test(s.upper()) for s in distinct_sources], dtype=np.uint8)


def bm25_allow(distinct_sources: Sequence[Optional[str]], filename_type_filter: str) -> np.ndarray:
    """search_engine.py:227-230: any(source.upper().startswith(prefix))."""
    prefixes = _prefixes(filename_type_filter)
    return np.array([isinstance(s, str) and any(s.upper().startswith(p) for p in prefixes)
                     for s in distinct_sources], dtype=np.uint8)


def _allow_of(owner, kind: str, filename_type_filter: str) -> np.ndarray:
    """Allow list of `owner` (a DenseHandle or Bm25Proxy) for one filter string, evaluated once: the reference
    re-runs its string tests over every row on every query (search_engine.py:39-48, :227-230); here they run
    over the DISTINCT sources, and only the first time a filter string is seen."""
    cache = owner.__dict__.setdefault("_allow_cache", {})
    key = (kind, filename_type_filter)
    hit = cache.get(key)
    if hit is None:
        if len(cache) >= 256:
            cache.clear()
        fn = dense_allow if kind == "dense" else bm25_allow
        hit = cache[key] = fn(owner.distinct_sources, filename_type_filter)
    return hit


class SearchEngine:
    def __init__(self, voyage_client=None, openai_client=None, encoder=None):
        self.vo = voyage_client
        self.openai_client = openai_client
        self.encoder = encoder  # local query encoder (anrag.encoder.LocalEncoder) or None
        self.logger = logging.getLogger(__name__)
        self._util: Optional[Index] = None

    # ------------------------------------------------------------------ fusion
    def _utility_index(self) -> Index:
        if self._util is None:
            self._util = Index()
        return self._util

    def weighted_reciprocal_rank_fusion(self, ranked_lists: List[Tuple], model_weights: Dict[str, float],
                                        k: int = 50) -> List[Tuple]:
        """search_engine.py:21-34 on the GPU (`anrag_wrrf`): fp64, same additions in the same order,
        equal scores keep first-insertion order."""
        code: Dict[object, int] = {}
        lists, weights = [], []
        for ranked, name in ranked_lists:
            lists.append([code.setdefault(doc_id, len(code)) for doc_id in ranked])
            weights.append(model_weights.get(name, 1.0))
        if not code:
            return []
        back = list(code)
        ids, scores = self._utility_index().wrrf(lists, weights, k, len(code))
        return [(back[i], float(s)) for i, s in zip(ids.tolist(), scores.tolist())]

    # ------------------------------------------------------------------ filter
    def _filter_by_filename_type(self, df: pd.DataFrame, filename_type_filter: str) -> pd.DataFrame:
        """search_engine.py:36-55.  Kept for callers that want the filtered frame; the search methods below
        never materialise it (the filter is an allow-bitmap over interned sources on the device)."""
        h = DenseHandle.of(df)
        mask = _allow_of(h, "dense", filename_type_filter)[h.source_id].astype(bool)
        filtered = df[mask].copy()
        filtered.attrs.pop(ATTR, None)
        self.logger.info(f"Filtered by filename type(s) '{', '.join(_prefixes(filename_type_filter))}': "
                         f"{len(filtered)} documents remaining from {len(df)} total")
        return filtered

    # ------------------------------------------------------------------ dense
    def _dense_topk(self, query_embedding: np.ndarray, df: pd.DataFrame, similarity_k: int,
                    filename_type_filter: Optional[str]):
        h = DenseHandle.of(df)
        allow = None
        if filename_type_filter:
            allow = _allow_of(h, "dense", filename_type_filter)
            if not allow.any():
                return h, None, None
        q = np.asarray(query_embedding)
        if q.ndim == 2 and q.shape[0] != 1:
            # the reference flattens a B x N score matrix and then fails on iloc (:81, :89): same outcome
            raise ValueError("batched query_embedding: use similarity_search_batch")
        if q.dtype == np.float64:  # the text path (:157): numpy promotes and scores in fp64 (:129) -- so does the device
            doc, score, n = h.index.dense_search_f64(q.reshape(-1), int(similarity_k), allow)
            return h, doc[:n], score[:n]
        doc, score, count = h.index.dense_search(q.reshape(-1), int(similarity_k), allow)
        n = int(count[0])
        return h, doc[0, :n], score[0, :n]

    def similarity_search_with_embedding(self, query_embedding: np.ndarray, df: pd.DataFrame,
                                         model_name: str = "voyage-3-large", similarity_k: int = 25,
                                         filename_type_filter: Optional[str] = None) -> pd.DataFrame:
        """search_engine.py:57-98."""
        try:
            if df.empty:
                return df
            h, rows, sims = self._dense_topk(query_embedding, df, similarity_k, filename_type_filter)
            if rows is None:
                self.logger.warning(f"No documents found after filtering by filename type: {filename_type_filter}")
                return df.iloc[0:0]
            result_df = df.iloc[rows].copy()
            result_df.attrs.pop(ATTR, None)
            result_df["similarity"] = sims
            return result_df
        except AnragError as e:
            if e.code in _FATAL:
                raise
            self.logger.error(f"Error in {model_name} similarity search with precalculated embedding: {e}")
            return pd.DataFrame()
        except Exception as e:
            self.logger.error(f"Error in {model_name} similarity search with precalculated embedding: {e}")
            return pd.DataFrame()

    def similarity_search(self, query_text: str, df: pd.DataFrame, model_name: str = "voyage-3-large",
                          similarity_k: int = 25, filename_type_filter: Optional[str] = None,
                          query_embedding: Optional[np.ndarray] = None) -> pd.DataFrame:
        """search_engine.py:100-146."""
        try:
            if df.empty:
                return df
            if filename_type_filter:
                h = DenseHandle.of(df)
                if not _allow_of(h, "dense", filename_type_filter).any():
                    self.logger.warning(f"No documents found after filtering by filename type: {filename_type_filter}")
                    return df.iloc[0:0]
            if query_embedding is not None:
                query_embedding = query_embedding.reshape(1, -1)
            else:
                query_embedding = self._generate_query_embedding(query_text, model_name)
            h, rows, sims = self._dense_topk(query_embedding, df, similarity_k, filename_type_filter)
            result_df = df.iloc[rows].copy()
            result_df.attrs.pop(ATTR, None)
            result_df["similarity"] = sims
            self.logger.info(f"{model_name} similarity search found {len(result_df)} results")
            return result_df
        except AnragError as e:
            if e.code in _FATAL:
                raise
            self.logger.error(f"Error in {model_name} similarity search: {e}")
            return pd.DataFrame()
        except Exception as e:
            self.logger.error(f"Error in {model_name} similarity search: {e}")
            return pd.DataFrame()

    def similarity_search_batch(self, query_embeddings: np.ndarray, df: pd.DataFrame, similarity_k: int = 25,
                                filename_type_filter: Optional[str] = None):
        """Many queries in one call (no reference counterpart: SURVEY.md 8a-2; the oracle is a loop of the
        single-query method).  -> (rows [nq, k] int64 (-1 padded), similarities [nq, k] float32)."""
        h = DenseHandle.of(df)
        allow = _allow_of(h, "dense", filename_type_filter) if filename_type_filter else None
        doc, score, _ = h.index.dense_search(np.asarray(query_embeddings, dtype=np.float32), int(similarity_k), allow)
        return doc, score

    def hybrid_search_ids_batch(self, query_embeddings: np.ndarray, df: pd.DataFrame, dense_weight: float,
                                query_token_lists: Sequence[List[str]], bm25, bm25_sections, bm25_section_ids,
                                bm25_weight: float, similarity_k: int, common_sections_n: int, wrrf_k,
                                filename_type_filter: Optional[str]) -> Optional[List[List[str]]]:
        """`hybrid_search_ids` for a list of queries in ONE library call (no reference counterpart; element i is
        what `hybrid_search_ids` returns for query i).  None when outside the fused path's envelope."""
        if similarity_k > FUSED_K_MAX or common_sections_n > 2 * FUSED_K_MAX:
            return None
        if any(not t for t in query_token_lists):
            return None
        proxy = self._proxy(bm25, bm25_sections)
        pair = FusedPair.of(df, proxy, bm25_section_ids)
        ad = ab = None
        if filename_type_filter:
            ad = _allow_of(pair.dense, "dense", filename_type_filter)
            ab = _allow_of(proxy, "bm25", filename_type_filter)
        q = np.asarray(query_embeddings, dtype=np.float32).reshape(len(query_token_lists), -1)
        ids, _, counts = pair.dense.index.hybrid_search_batch(
            q, [proxy.term_ids(t) for t in query_token_lists], int(similarity_k), float(dense_weight),
            float(bm25_weight), float(wrrf_k), int(common_sections_n), ad, ab)
        name = pair.id_of_doc
        return [[name[i] for i in row[:c].tolist()] for row, c in zip(ids, counts.tolist())]

    def _generate_query_embedding(self, query_text: str, model_name: str) -> np.ndarray:
        """search_engine.py:148-159, plus the local-encoder branch the north star adds in place of the
        `Unsupported model` error for the configured encoder."""
        if self.encoder is not None and model_name == self.encoder.model_name:
            return self.encoder.encode_query(query_text).reshape(1, -1)
        if model_name == "voyage-3-large":
            if not self.vo:
                raise ValueError("Voyage client not available")
            response = self.vo.embed(query_text, input_type="query", model="voyage-3-large",
                                     output_dimension=2048).embeddings
            return np.array(response).reshape(1, -1)
        raise ValueError(f"Unsupported model: {model_name}")

    # ------------------------------------------------------------------ rerank (remote; unchanged contract)
    def rerank_documents(self, query_text: str, documents: List, reranker_model: str = "rerank-2",
                         reranker_top_k: Optional[int] = None) -> List:
        """search_engine.py:161-203: Voyage's hosted cross-encoder; not part of the accelerated path.
        Any failure returns the documents in their original order (:201-203)."""
        try:
            if not documents:
                return documents
            texts = [doc.get("document", "") for doc in documents]
            result = self.vo.rerank(query=query_text, documents=texts, model=reranker_model,
                                    top_k=reranker_top_k or len(texts), truncation=True)
            return [{**documents[r.index], "rerank_score": r.relevance_score}
                    for r in result.results if r.index < len(documents)]
        except Exception as e:
            self.logger.warning(f"Reranking failed, returning original order: {e}")
            return documents

    # ------------------------------------------------------------------ BM25
    @staticmethod
    def _proxy(bm25, bm25_sections) -> Bm25Proxy:
        if isinstance(bm25, Bm25Proxy):
            return bm25
        proxy = getattr(bm25, "_anrag_proxy", None)
        if proxy is None:  # a rank_bm25-style object somebody built themselves: upload it once
            proxy = Bm25Proxy.from_rank_bm25(bm25, [s.metadata.get("source", "") for s in bm25_sections])
            bm25._anrag_proxy = proxy
        return proxy

    def _core_bm25_search(self, query_tokens: List[str], bm25, bm25_sections, bm25_section_ids, similarity_k: int,
                          filename_type_filter: Optional[str]) -> List[str]:
        """search_engine.py:205-243: get_scores + selection, zero-score sections ranked, not dropped."""
        if not query_tokens:
            return []
        proxy = self._proxy(bm25, bm25_sections)
        allow = _allow_of(proxy, "bm25", filename_type_filter) if filename_type_filter else None
        doc, _, count = proxy.index.bm25_search(proxy.term_ids(query_tokens), int(similarity_k), allow)
        return [bm25_section_ids[i] for i in doc[:count].tolist()]

    # ------------------------------------------------------------------ row-level legs (ids-only routes)
    def dense_rows(self, query_embedding: np.ndarray, df: pd.DataFrame, model_name: str, similarity_k: int,
                   filename_type_filter: Optional[str]) -> Optional[np.ndarray]:
        """The ranking `similarity_search_with_embedding` returns, as DataFrame POSITIONS and without building the
        result frame (a 9,609-row frame + `to_dict("records")` per query is ~20 ms of pandas; the search is < 1 ms).
        None = what the method would have answered with an empty frame (nothing left after the filter, or an
        error, logged as there)."""
        try:
            if df.empty:
                return None
            _, rows, _ = self._dense_topk(query_embedding, df, similarity_k, filename_type_filter)
            return rows
        except AnragError as e:
            if e.code in _FATAL:
                raise
            self.logger.error(f"Error in {model_name} similarity search with precalculated embedding: {e}")
            return None
        except Exception as e:
            self.logger.error(f"Error in {model_name} similarity search with precalculated embedding: {e}")
            return None

    def bm25_rows(self, query_tokens: Optional[List[str]], query_text: Optional[str], bm25, bm25_sections,
                  similarity_k: int, filename_type_filter: Optional[str]) -> Optional[np.ndarray]:
        """`bm25_search_preprocessed` (tokens given) / `bm25_search` (text, lemmatised) as section POSITIONS."""
        try:
            if not query_tokens:
                query_tokens = preprocess_text(query_text, use_lemmatization=True)
            if not query_tokens:
                return None
            proxy = self._proxy(bm25, bm25_sections)
            allow = _allow_of(proxy, "bm25", filename_type_filter) if filename_type_filter else None
            doc, _, count = proxy.index.bm25_search(proxy.term_ids(query_tokens), int(similarity_k), allow)
            return doc[:count]
        except AnragError as e:
            if e.code in _FATAL:
                raise
            self.logger.error(f"Error in BM25 search: {e}")
            return None
        except Exception as e:
            self.logger.error(f"Error in BM25 search: {e}")
            return None

    def fuse_rows(self, lists: Sequence[np.ndarray], weights: Sequence[float], k, top_n: int) -> np.ndarray:
        """Weighted RRF over integer id lists (`anrag_wrrf`), first `top_n` fused ids."""
        ids, _ = self._utility_index().wrrf([np.asarray(l, dtype=np.int64) for l in lists], list(weights), k, int(top_n))
        return ids

    def bm25_search(self, query_text: str, bm25, bm25_sections, bm25_section_ids, similarity_k: int = 25,
                    filename_type_filter: Optional[str] = None, use_lemmatized: bool = True) -> List[str]:
        """search_engine.py:245-269."""
        try:
            query_tokens = preprocess_text(query_text, use_lemmatization=use_lemmatized)
            return self._core_bm25_search(query_tokens, bm25, bm25_sections, bm25_section_ids, similarity_k,
                                          filename_type_filter)
        except AnragError as e:
            if e.code in _FATAL:
                raise
            self.logger.error(f"Error in BM25 search: {e}")
            return []
        except Exception as e:
            self.logger.error(f"Error in BM25 search: {e}")
            return []

    def bm25_search_preprocessed(self, query_tokens: List[str], bm25, bm25_sections, bm25_section_ids,
                                 similarity_k: int = 25, filename_type_filter: Optional[str] = None) -> List[str]:
        """search_engine.py:271-293."""
        try:
            return self._core_bm25_search(query_tokens, bm25, bm25_sections, bm25_section_ids, similarity_k,
                                          filename_type_filter)
        except AnragError as e:
            if e.code in _FATAL:
                raise
            self.logger.error(f"Error in preprocessed BM25 search: {e}")
            return []
        except Exception as e:
            self.logger.error(f"Error in preprocessed BM25 search: {e}")
            return []

    # ------------------------------------------------------------------ fused hybrid (one ABI call)
    def hybrid_search_ids(self, query_embedding: np.ndarray, df: pd.DataFrame, dense_weight: float,
                          query_tokens: List[str], bm25, bm25_sections, bm25_section_ids, bm25_weight: float,
                          similarity_k: int, common_sections_n: int, wrrf_k,
                          filename_type_filter: Optional[str]) -> Optional[List[str]]:
        """One dense model + BM25 -> fused chunk ids, as a single `anrag_hybrid_search` (dense scan, BM25,
        list merges and WRRF overlapped on three streams).  Returns None when the request is outside the
        fused path's envelope (k > 64, ...): the caller then takes the method-by-method route above."""
        if similarity_k > FUSED_K_MAX or common_sections_n > 2 * FUSED_K_MAX or not query_tokens:
            return None
        q = np.asarray(query_embedding)
        if q.ndim == 2 and q.shape[0] != 1:
            return None
        if q.dtype == np.float64:
            # the text path's embedding (:157): the reference scores it in fp64 (:129) and so does the method-by-method
            # route (`anrag_dense_search_f64`); the fused call scores fp32 -- inside 1e-4, but near-ties could order
            # differently between the two routes of ONE retrieve_documents call: declined, both routes score alike
            return None
        proxy = self._proxy(bm25, bm25_sections)
        pair = FusedPair.of(df, proxy, bm25_section_ids)
        ad = ab = None
        if filename_type_filter:
            ad = _allow_of(pair.dense, "dense", filename_type_filter)
            ab = _allow_of(proxy, "bm25", filename_type_filter)
        ids, _ = pair.dense.index.hybrid_search(q.reshape(-1), proxy.term_ids(query_tokens), int(similarity_k),
                                                float(dense_weight), float(bm25_weight), float(wrrf_k),
                                                int(common_sections_n), ad, ab)
        return [pair.id_of_doc[i] for i in ids.tolist()]

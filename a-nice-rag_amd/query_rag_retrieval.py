"""`RetrievalEvaluationSystem`: the reference's src/query_rag_retrieval.py:20-411 over the GPU
search engine.  Same constructor behaviour (load every configured DB once, any failure empties the
source, :105-108), same `retrieve_documents` keyword arguments, defaults, gating, fusion and return
types -- so a `retrieval_eval.py`-shaped harness is drop-in.

The reference's four cloned per-model blocks (:197-301) are one loop over `Config.DENSE_MODELS`
(same order), which also admits the local encoder's model key.  When exactly one dense model and
BM25 are active and k <= 64, the whole query is ONE `anrag_hybrid_search` call; results are
identical to the method-by-method route (tests/test_gpu_retrieval.py holds both to the reference's
golden vectors).
"""
from __future__ import annotations

import logging
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from .config import Config, InfoSource
from .database_manager import DatabaseManager
from .search_engine import SearchEngine

logger = logging.getLogger(__name__)


class RetrievalEvaluationSystem:
    def __init__(self, config: Optional[Config] = None, voyage_client=None, encoder=None, fused: bool = True):
        self.config = config or Config()
        self.db_manager = DatabaseManager()
        self.voyage_client = voyage_client  # reranking only (remote); None disables it, as :30-34
        self.search_engine = SearchEngine(voyage_client, None, encoder=encoder)
        self.fused = fused
        self._load_databases()

    def _load_databases(self):
        """query_rag_retrieval.py:38-111."""
        start = time.time()
        self.embeddings_data = {}
        self.bm25_data = {}
        for source in InfoSource:
            sc = self.config.SOURCE_CONFIGS[source]
            try:
                embeddings_dict = {}
                for key, attr, loader_name in self.config.DENSE_MODELS:
                    path = getattr(sc, attr, None)
                    if path:
                        embeddings_dict[key] = self.db_manager.load_embeddings_from_sql(path, loader_name)
                self.embeddings_data[source] = embeddings_dict
                self.bm25_data[source] = self.db_manager.load_bm25_from_pickle(sc.bm25_path)
            except Exception as e:
                from ._native import AnragError

                if isinstance(e, AnragError) and e.code in (-100, -5):
                    raise  # no library / no GPU is not a data problem
                logger.error(f"Failed to load {source.value}: {e}")
                self.embeddings_data[source] = {}
                self.bm25_data[source] = None
        logger.info(f"Database loading completed in {time.time() - start:.2f} seconds")

    def _validate_inputs(self, query_embeddings, similarity_k, common_sections_n, info_source):
        """query_rag_retrieval.py:113-139."""
        if not query_embeddings:
            raise ValueError("Query embeddings dictionary cannot be empty")
        for model_name, embedding in query_embeddings.items():
            if not isinstance(embedding, np.ndarray):
                raise ValueError(f"Embedding for {model_name} must be a numpy array")
            if embedding.size == 0:
                raise ValueError(f"Embedding for {model_name} cannot be empty")
        if similarity_k <= 0 or common_sections_n <= 0:
            raise ValueError("similarity_k and common_sections_n must be positive integers")
        try:
            InfoSource(info_source.lower())
        except ValueError:
            raise ValueError(f"Invalid info_source '{info_source}'. Must be one of: {[s.value for s in InfoSource]}")

    def get_sources(self, results: List[Tuple], info_source: str) -> List[str]:
        """query_rag_retrieval.py:141-147."""
        return [doc.get("id", "Unknown section") for doc, _ in results]

    def retrieve_documents(
        self,
        query_embeddings: Dict[str, np.ndarray],
        query_text: Optional[str] = None,
        query_tokens: Optional[List[str]] = None,
        similarity_k: int = 25,
        common_sections_n: int = 15,
        info_source: str = "NICE",
        model_weights: Optional[Dict[str, float]] = None,
        filename_type_filter: Optional[str] = None,
        use_hybrid_search: bool = False,
        wrrf_k: int = 60,
        use_reranker: bool = True,
        reranker_model: str = "rerank-2-lite",
        reranker_top_k: Optional[int] = 5,
        return_docs: bool = False,
    ):
        """query_rag_retrieval.py:149-411."""
        self._validate_inputs(query_embeddings, similarity_k, common_sections_n, info_source)
        if model_weights is None:
            model_weights = self.config.DEFAULT_MODEL_WEIGHTS.copy()
        source_enum = InfoSource(info_source.lower())
        try:
            embeddings_dict = self.embeddings_data.get(source_enum, {})
            bm25_tuple = self.bm25_data.get(source_enum)
            if not embeddings_dict:
                return []
            bm25, bm25_sections, bm25_section_ids = bm25_tuple if bm25_tuple else (None, [], [])

            active = [(key, embeddings_dict[key]) for key, _, _ in self.config.DENSE_MODELS
                      if embeddings_dict.get(key) is not None and not embeddings_dict[key].empty
                      and model_weights.get(key, 0) > 0 and key in query_embeddings]
            want_bm25 = use_hybrid_search and bm25 is not None and model_weights.get("BM25", 0) > 0
            will_rerank = use_reranker and bool(query_text)

            # ---- fused single-call route: one dense model + BM25, ids out, no rerank.  Tokens as the reference
            # picks them: pre-tokenised if given (:307), else the query text through the tokeniser (:317, use_lemmatized)
            fused_tokens = query_tokens
            if not fused_tokens and query_text and want_bm25:
                from .preprocess_bm25 import preprocess_text

                fused_tokens = preprocess_text(query_text, use_lemmatization=True)
            if (self.fused and not return_docs and not will_rerank and len(active) == 1 and want_bm25
                    and fused_tokens and similarity_k <= 64):
                key, df = active[0]
                ids = self.search_engine.hybrid_search_ids(
                    query_embeddings[key], df, model_weights.get(key, 1.0), fused_tokens, bm25, bm25_sections,
                    bm25_section_ids, model_weights.get("BM25", 1.0), similarity_k, common_sections_n, wrrf_k,
                    filename_type_filter)
                if ids is not None:
                    return ids

            # ---- ids-only route for everything else that returns ids and does not rerank (retrieval_eval's
            # full-ranking configurations: k = 12000, several dense models): the same legs and the same fusion as
            # below, but on row numbers -- no result frames, no per-row dicts, ids mapped to strings at the end
            if self.fused and not return_docs and not will_rerank:
                ids = self._rank_route(source_enum, active, query_embeddings, query_tokens, query_text, want_bm25,
                                       bm25, bm25_sections, bm25_section_ids, similarity_k, common_sections_n,
                                       model_weights, filename_type_filter, wrrf_k)
                if ids is not None:
                    return ids

            ranked_lists = []
            all_results: Dict[str, dict] = {}
            for key, df in active:
                results = self.search_engine.similarity_search_with_embedding(
                    query_embeddings[key], df, key, similarity_k, filename_type_filter)
                if not results.empty:
                    ranked_lists.append((results["id"].tolist(), key))
                    for rec in results.to_dict("records"):
                        all_results.setdefault(rec["id"], rec)  # an id a LATER model returns again keeps the first
                                                                # model's record (:242-245, :272-275, :293-294)

            if want_bm25:
                bm25_ranked = None
                if query_tokens:
                    bm25_ranked = self.search_engine.bm25_search_preprocessed(
                        query_tokens, bm25, bm25_sections, bm25_section_ids, similarity_k, filename_type_filter)
                elif query_text:
                    bm25_ranked = self.search_engine.bm25_search(
                        query_text, bm25, bm25_sections, bm25_section_ids, similarity_k, filename_type_filter)
                else:
                    logger.warning("BM25 search requested but no query_text or query_tokens provided - skipping BM25")
                if bm25_ranked:
                    ranked_lists.append((bm25_ranked, "BM25"))
                    missing = [sid for sid in bm25_ranked if sid not in all_results]
                    if missing:
                        section_of = self._section_dict(source_enum, bm25_sections)
                        for sid in missing:
                            section = section_of.get(sid)
                            if section:
                                all_results[sid] = {"id": sid, "document": section.page_content,
                                                    "source": section.metadata.get("source", "Unknown"),
                                                    "similarity": 0.0}

            if len(ranked_lists) > 1:
                fused = self.search_engine.weighted_reciprocal_rank_fusion(ranked_lists, model_weights, wrrf_k)
                most_common = [sid for sid, _ in fused[:common_sections_n]]
            elif len(ranked_lists) == 1:
                most_common = ranked_lists[0][0][:common_sections_n]
            else:
                most_common = []
            common_docs = [all_results[sid] for sid in most_common if sid in all_results][:common_sections_n]

            if use_reranker and common_docs and len(common_docs) > 1 and query_text:
                common_docs = self.search_engine.rerank_documents(query_text, common_docs, reranker_model,
                                                                  reranker_top_k)
            if return_docs:
                return common_docs
            final = [(doc, doc.get("rerank_score", doc.get("similarity", 0.0))) for doc in common_docs]
            return self.get_sources(final, info_source)
        except Exception as e:
            from ._native import AnragError

            if isinstance(e, AnragError) and e.code in (-100, -5):
                raise
            logger.error(f"Error in retrieval processing: {e}")
            return []

    # ------------------------------------------------------------------ ids-only route
    def _gids(self, source_enum, owner_key, id_strings) -> np.ndarray:
        """Chunk-id strings of one frame / of the BM25 sections as integers of one table per info source (built
        once per frame: the frames live as long as the system)."""
        tables = self.__dict__.setdefault("_gid_tables", {})
        t = tables.setdefault(source_enum, {"of": {}, "names": [], "arrays": {}})
        arr = t["arrays"].get(owner_key)
        if arr is not None:
            return arr
        import threading

        with self.__dict__.setdefault("_gid_lock", threading.Lock()):  # one writer: an id gets ONE integer
            arr = t["arrays"].get(owner_key)
            if arr is not None:
                return arr
            of, names = t["of"], t["names"]
            arr = np.empty(len(id_strings), dtype=np.int64)
            for i, cid in enumerate(id_strings):
                g = of.get(cid)
                if g is None:
                    g = of[cid] = len(names)
                    names.append(cid)
                arr[i] = g
            t["arrays"][owner_key] = arr
            return arr

    def _rank_route(self, source_enum, active, query_embeddings, query_tokens, query_text, want_bm25, bm25,
                    bm25_sections, bm25_section_ids, similarity_k, common_sections_n, model_weights,
                    filename_type_filter, wrrf_k) -> Optional[List[str]]:
        se = self.search_engine
        lists, weights = [], []
        for key, df in active:
            rows = se.dense_rows(query_embeddings[key], df, key, similarity_k, filename_type_filter)
            if rows is not None and len(rows):
                lists.append(self._gids(source_enum, ("dense", id(df)), df["id"].tolist())[rows])
                weights.append(model_weights.get(key, 1.0))
        if want_bm25:
            if query_tokens or query_text:
                rows = se.bm25_rows(query_tokens, query_text, bm25, bm25_sections, similarity_k, filename_type_filter)
                if rows is not None and len(rows):
                    lists.append(self._gids(source_enum, ("bm25", id(bm25_section_ids)), bm25_section_ids)[rows])
                    weights.append(model_weights.get("BM25", 1.0))
            else:
                logger.warning("BM25 search requested but no query_text or query_tokens provided - skipping BM25")
        if not lists:
            return []
        if len(lists) > 1:
            gids = se.fuse_rows(lists, weights, wrrf_k, common_sections_n)
        else:
            gids = lists[0][:common_sections_n]
        names = self._gid_tables[source_enum]["names"]
        return [names[g] for g in gids.tolist()]

    # ------------------------------------------------------------------ query lists
    def _rank_batch(self, queries: List[Dict], params: Dict, expected: Optional[List[str]] = None):
        """The ids-only route for a LIST of queries as ONE `anrag_rank_batch` (rank_batch.hip): every active dense model
        and BM25 scored into tiles, ranked and fused on the device -- what `retrieval_eval.py`'s k = 12,000 configurations
        need 8,000 times over.  -> per-query id arrays (a list of int64 arrays), or with `expected` (one chunk id per
        query) -> (ranks, totals): the 1-based rank of the expected id in each answer (-1: absent; retrieval_eval.py:75-82)
        and the answers' lengths, without materialising the lists.  None = outside this route (the caller answers
        query by query)."""
        from .index import rank_batch

        if not queries or not self.fused or params.get("return_docs"):
            return None
        similarity_k = params.get("similarity_k", 25)
        common_sections_n = params.get("common_sections_n", 15)
        info_source = params.get("info_source", "NICE")
        model_weights = params.get("model_weights") or self.config.DEFAULT_MODEL_WEIGHTS.copy()
        flt = params.get("filename_type_filter")
        try:  # the shared arguments once, the embeddings of every query (the per-query method raises where the reference raises)
            self._validate_inputs(queries[0].get("query_embeddings"), similarity_k, common_sections_n, info_source)
        except ValueError:
            return None
        for q in queries:
            qe = q.get("query_embeddings")
            if not qe or not all(isinstance(e, np.ndarray) and e.size for e in qe.values()):
                return None
        source_enum = InfoSource(info_source.lower())
        embeddings_dict = self.embeddings_data.get(source_enum, {})
        if not embeddings_dict:
            return None
        bm25_tuple = self.bm25_data.get(source_enum)
        bm25, bm25_sections, bm25_section_ids = bm25_tuple if bm25_tuple else (None, [], [])
        want_bm25 = bool(params.get("use_hybrid_search", False)) and bm25 is not None and model_weights.get("BM25", 0) > 0
        if params.get("use_reranker", True) and any(q.get("query_text") for q in queries):
            return None  # reranking materialises documents: per query
        from .database_manager import DenseHandle
        from .search_engine import _allow_of

        active = None
        dims = {key: DenseHandle.of(df).dim for key, df in embeddings_dict.items() if df is not None and not df.empty}
        # (which models can take part is a property of the call, not of a query: `DataFrame.empty` alone costs a microsecond,
        # and this loop runs thousands of times per evaluation call)
        usable = [key for key, _, _ in self.config.DENSE_MODELS if key in dims and model_weights.get(key, 0) > 0]
        for q in queries:
            qe = q["query_embeddings"]
            keys = [key for key in usable if key in qe]
            if active is not None and keys != active:
                return None  # the queries of one call must share their legs
            active = keys
            for key in keys:
                e = q["query_embeddings"][key]
                if e.dtype != np.float32 or e.size != dims[key]:
                    return None  # fp64 queries are scored in fp64 by the per-query path (search_engine.py:157)
        se = self.search_engine
        legs = []
        for key in active:
            df = embeddings_dict[key]
            h = DenseHandle.of(df)
            allow = _allow_of(h, "dense", flt) if flt else None
            legs.append(dict(index=h.index, weight=model_weights.get(key, 1.0), allow=allow,
                             queries=np.stack([q["query_embeddings"][key].reshape(-1) for q in queries]),
                             doc_of_row=self._gids(source_enum, ("dense", id(df)), df["id"].tolist())))
        if want_bm25:
            from .preprocess_bm25 import preprocess_text

            proxy = se._proxy(bm25, bm25_sections)
            term_lists = []
            for q in queries:
                tokens = q.get("query_tokens")
                if not tokens and q.get("query_text"):
                    tokens = preprocess_text(q["query_text"], use_lemmatization=True)
                term_lists.append(proxy.term_ids(tokens) if tokens else np.zeros(0, np.int32))
            allow = _allow_of(proxy, "bm25", flt) if flt else None
            legs.append(dict(index=proxy.index, weight=model_weights.get("BM25", 1.0), allow=allow, term_lists=term_lists,
                             doc_of_row=self._gids(source_enum, ("bm25", id(bm25_section_ids)), bm25_section_ids)))
        n = len(queries)
        if not legs:
            return ([np.zeros(0, np.int64)] * n) if expected is None else (np.full(n, -1, np.int32), np.zeros(n, np.int32))
        table = self._gid_tables[source_enum]
        id_space = len(table["names"])
        wrrf_k = params.get("wrrf_k", 60)
        if expected is not None:
            exp = np.array([table["of"].get(e, -2) for e in expected], dtype=np.int64)  # -2: an id no frame holds
            _, _, counts, ranks = rank_batch(legs, n, int(similarity_k), float(wrrf_k), int(common_sections_n),
                                             id_space=id_space, expect=exp, want_ids=False)
            return ranks, counts
        ids, _, counts = rank_batch(legs, n, int(similarity_k), float(wrrf_k), int(common_sections_n), id_space=id_space)
        return [ids[i, :c] for i, c in enumerate(counts.tolist())]

    def _names_of(self, source_enum, gid_rows: List[np.ndarray]) -> List[List[str]]:
        table = self._gid_tables[source_enum]
        arr = table.get("names_arr")
        if arr is None or len(arr) != len(table["names"]):
            arr = table["names_arr"] = np.array(table["names"], dtype=object)
        return [arr[g].tolist() for g in gid_rows]

    RANKED_LIST_MIN = 16  # queries from which a small-k list takes the ranking route (one score-tile group)

    def retrieve_documents_batch(self, queries: List[Dict], **params) -> List[List[str]]:
        """`retrieve_documents` for a LIST of queries (no reference counterpart: retrieval_eval.py:51-84 loops).
        Each element of `queries` holds the per-query arguments (`query_embeddings`, and `query_tokens` and/or
        `query_text`); `params` are the remaining keyword arguments of `retrieve_documents`, shared by all.
        Element i of the result is exactly `retrieve_documents(**queries[i], **params)`: requests inside the fused
        route's envelope (one dense model + BM25, ids out, no rerank, similarity_k <= 64) in lists of fewer than 16 go to
        the library as ONE `anrag_hybrid_search_batch` (the device pipeline, a single host sync); other ids-only requests
        -- longer lists, any number of dense models, with or without BM25, similarity_k up to retrieval_eval's 12,000 --
        as ONE `anrag_rank_batch`;
        anything else (reranking, `return_docs`, float64 embeddings) is answered by the per-query method."""
        def one_by_one():
            return [self.retrieve_documents(**q, **params) for q in queries]

        def ranked():
            try:
                rows = self._rank_batch(queries, params)
            except Exception as e:
                from ._native import AnragError

                if isinstance(e, AnragError) and e.code in (-100, -5):
                    raise
                logger.info(f"Batched ranking not applicable, answering one by one: {e}")
                rows = None
            if rows is None:
                return one_by_one()
            return self._names_of(InfoSource(params.get("info_source", "NICE").lower()), rows)

        if not queries or not self.fused or params.get("return_docs"):
            return one_by_one()
        if not params.get("use_hybrid_search", False):
            return ranked()
        similarity_k = params.get("similarity_k", 25)
        common_sections_n = params.get("common_sections_n", 15)
        info_source = params.get("info_source", "NICE")
        model_weights = params.get("model_weights") or self.config.DEFAULT_MODEL_WEIGHTS.copy()
        try:
            for q in queries:
                self._validate_inputs(q.get("query_embeddings"), similarity_k, common_sections_n, info_source)
        except ValueError:
            return one_by_one()  # let the per-query method raise where the reference raises
        source_enum = InfoSource(info_source.lower())
        embeddings_dict = self.embeddings_data.get(source_enum, {})
        bm25_tuple = self.bm25_data.get(source_enum)
        if not embeddings_dict or not bm25_tuple:
            return one_by_one()
        if similarity_k > 64:
            return ranked()
        if len(queries) >= self.RANKED_LIST_MIN:
            # A LIST of small-k queries is also served best by the ranking route: its score tiles read the corpus once
            # per 16 queries, the pipeline once per query (9,609 x 384, k = 25: 2.0 against 11.0 us per query; 100k x 768:
            # 7.9 against 45.5) -- same lists, same fused scores (the tiles are bit-identical to the single-query scan).
            try:
                rows = self._rank_batch(queries, params)
            except Exception as e:
                from ._native import AnragError

                if isinstance(e, AnragError) and e.code in (-100, -5):
                    raise
                rows = None
            if rows is not None:
                return self._names_of(source_enum, rows)
        bm25, bm25_sections, bm25_section_ids = bm25_tuple
        if bm25 is None or model_weights.get("BM25", 0) <= 0:
            return ranked()
        keys = None
        token_lists = []
        usable = [key for key, _, _ in self.config.DENSE_MODELS
                  if embeddings_dict.get(key) is not None and not embeddings_dict[key].empty and model_weights.get(key, 0) > 0]
        for q in queries:
            active = [key for key in usable if key in q["query_embeddings"]]
            if len(active) != 1 or (keys is not None and active[0] != keys):
                return ranked()
            keys = active[0]
            if params.get("use_reranker", True) and q.get("query_text"):
                return one_by_one()
            tokens = q.get("query_tokens")
            if not tokens and q.get("query_text"):
                from .preprocess_bm25 import preprocess_text

                tokens = preprocess_text(q["query_text"], use_lemmatization=True)
            if not tokens:
                return ranked()
            if np.asarray(q["query_embeddings"][keys]).dtype == np.float64:
                return one_by_one()  # scored in fp64 by the per-query route
            token_lists.append(tokens)
        try:
            emb = np.stack([np.asarray(q["query_embeddings"][keys], dtype=np.float32).reshape(-1) for q in queries])
            out = self.search_engine.hybrid_search_ids_batch(
                emb, embeddings_dict[keys], model_weights.get(keys, 1.0), token_lists, bm25, bm25_sections,
                bm25_section_ids, model_weights.get("BM25", 1.0), similarity_k, common_sections_n,
                params.get("wrrf_k", 60), params.get("filename_type_filter"))
        except Exception as e:
            from ._native import AnragError

            if isinstance(e, AnragError) and e.code in (-100, -5):
                raise
            logger.error(f"Batched retrieval failed, answering one by one: {e}")
            out = None
        return out if out is not None else one_by_one()

    def rank_of_expected_batch(self, queries: List[Dict], expected_ids: List[str], **params):
        """For each query the 1-based rank of `expected_ids[i]` in what `retrieve_documents(**queries[i], **params)`
        returns (-1 if it is not there) and that list's length -- `RetrievalEvaluator.evaluate_query`'s search of the
        returned list (retrieval_eval.py:75-82), computed where the list is (one `anrag_rank_batch`, nothing but two
        integers per query crosses PCIe).  None when the request is outside the batched ranking (the caller then asks
        for the lists)."""
        try:
            return self._rank_batch(queries, params, expected=list(expected_ids))
        except Exception as e:
            from ._native import AnragError

            if isinstance(e, AnragError) and e.code in (-100, -5):
                raise
            logger.info(f"Batched ranking not applicable: {e}")
            return None

    def _section_dict(self, source_enum, bm25_sections):
        # the reference rebuilds this dict over all sections on EVERY query (:192); once is enough
        cache = self.__dict__.setdefault("_section_dicts", {})
        d = cache.get(source_enum)
        if d is None:
            d = cache[source_enum] = {s.metadata["id"]: s for s in bm25_sections}
        return d

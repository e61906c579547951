"""GPU: BASELINE.json config 1 -- "10k chunks x 384-d, batch=1, via query_rag_retrieval.py (plumbing)": the whole
file-based route a user of the reference takes.  Stand-in corpus (9,609 shipped chunk ids, anrag/niceqa.py) ->
SQLite `chunks` DB written by index_io.create_embeddings_db + BM25 pickle written by index_io -> DatabaseManager
loaders -> RetrievalEvaluationSystem.retrieve_documents with the NICEQA QUESTION TEXT (tokenised by the package's
tokeniser, as search_engine.bm25_search does) -> Recall@10, which must equal the direct index path's and the
CPU reference path's; plus the reference's threading contract (one shared system, concurrent session threads)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class HashedBowEncoder:
    """Deterministic stand-in for the missing encoder weights: tokenise + hashed bag-of-words (anrag.niceqa)."""
    model_name = "bge-small-en-v1.5"

    def encode(self, texts):
        from anrag.niceqa import hashed_bow
        from anrag.preprocess_bm25 import preprocess_text

        return np.stack([hashed_bow(preprocess_text(t, use_lemmatization=True)) for t in texts])

    def encode_query(self, text):
        return self.encode([text])[0]


def test_c1_files_to_recall(tmp_path):
    from anrag import index_io, niceqa
    from anrag.config import Config, InfoSource, LOCAL_ENCODER_KEY
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem
    from oracle.niceqa_ref import cpu_ranked_ids

    data = niceqa.load_standin(os.path.join(GOLD, "suggested_queries_bm25_preprocessed.json.gz"),
                               os.path.join(GOLD, "NICEQA.csv"))
    enc = HashedBowEncoder()
    # the chunk "content" whose tokens are exactly data["tokens"]: join them (already lower-case, punctuation-free)
    chunks = [{"title": cid, "content": " ".join(toks) or "empty", "source": src}
              for cid, toks, src in zip(data["ids"], data["tokens"], data["sources"])]
    db = str(tmp_path / "local.db")
    pkl = str(tmp_path / "bm25.pkl")
    assert index_io.create_embeddings_db(chunks, _TokenJoinEncoder(data), db) == len(chunks)
    bi, sections, section_ids = index_io.index_with_bm25(data["ids"], data["sources"], [c["content"] for c in chunks],
                                                         data["tokens"])
    index_io.export_bm25_to_file(bi, sections, section_ids, pkl)

    cfg = Config()
    sc = cfg.SOURCE_CONFIGS[InfoSource.NICE]
    saved = dict(vars(sc))
    try:
        sc.db_path = sc.voyage_db_path = None
        sc.voyage_3_5_db_path = sc.openai_db_path = sc.qwen_db_path = None
        sc.local_db_path, sc.bm25_path = db, pkl
        system = RetrievalEvaluationSystem(cfg, encoder=enc)
        weights = {LOCAL_ENCODER_KEY: 5.0, "BM25": 1.0}
        qv, qt = niceqa.encode_questions(data)

        def ask(i):
            _, _, question = data["questions"][i]
            return system.retrieve_documents(
                query_embeddings={LOCAL_ENCODER_KEY: qv[i]}, query_text=question, similarity_k=25, common_sections_n=10,
                model_weights=weights, use_hybrid_search=True, wrrf_k=40, use_reranker=False)

        serial = [ask(i) for i in range(len(data["questions"]))]
        direct = niceqa.gpu_ranked_ids(data)
        assert serial == direct                                   # files + loaders + text tokenisation == direct index
        # the whole question list in ONE call (anrag_hybrid_search_batch underneath) == the per-query answers
        shared = dict(similarity_k=25, common_sections_n=10, model_weights=weights, use_hybrid_search=True, wrrf_k=40,
                      use_reranker=False)
        batch_in = [{"query_embeddings": {LOCAL_ENCODER_KEY: qv[i]}, "query_text": data["questions"][i][2]}
                    for i in range(len(data["questions"]))]
        assert system.retrieve_documents_batch(batch_in, **shared) == serial
        filtered = [system.retrieve_documents(**q, filename_type_filter="NG", **shared) for q in batch_in]
        assert system.retrieve_documents_batch(batch_in, filename_type_filter="NG", **shared) == filtered
        assert filtered != serial
        system.fused = False                                      # method-by-method route (three ABI calls per query)
        assert [ask(i) for i in range(len(data["questions"]))] == serial
        system.fused = True
        r = niceqa.recall_at_10(data, serial)
        assert r == niceqa.recall_at_10(data, cpu_ranked_ids(data, qv, qt))   # == CPU reference path
        assert r["with_gold_chunk"] == 68
        # one shared system, many session threads (src/app.py:17-27): same answers
        with ThreadPoolExecutor(8) as pool:
            threaded = list(pool.map(ask, range(len(data["questions"]))))
        assert threaded == serial
    finally:
        for k, v in saved.items():
            setattr(sc, k, v)


class _TokenJoinEncoder:
    """Embeds a chunk exactly as anrag.niceqa does (from its token list), so DB rows == stand-in embeddings."""

    def __init__(self, data):
        self.by_text = {" ".join(t) or "empty": e for t, e in zip(data["tokens"], data["embeddings"])}

    def encode(self, texts):
        return np.stack([self.by_text[t] for t in texts])

"""CPU: the oracle restatement must reproduce what the reference's own code returned
(tests/golden/ref_*.json, produced by oracle/make_golden.py)."""
import numpy as np
import pytest

from oracle import ref_retrieval, ref_search
from oracle.make_golden import synth_dense, synth_query, synth_sources, synth_chunks
from oracle.ref_bm25 import BM25Okapi
from helpers import assert_ranking_matches, load_golden


def _dense_case_inputs(c, cache={}):
    key = (c["n"], c["d"], c["corpus_seed"])
    if key not in cache:
        e = synth_dense(c["n"], c["d"], c["corpus_seed"])
        if "dups" in c:
            for j in c["dups"]:
                e[j] = e[c["dup_of"]]
        cache[key] = (e, synth_sources(c["n"], c["source_seed"]))
    e, sources = cache[key]
    if "dups" in c:
        q = e[c["dup_of"]].copy()
    else:
        q = synth_query(e, c["query_seed"], c["query_row"]).astype(c["qdtype"])
    return e, sources, q


def test_dense_matches_reference():
    cases = load_golden("ref_dense.json")
    checked = 0
    for c in cases:
        if "batched_query_returns_empty" in c:
            assert c["batched_query_returns_empty"] is True
            continue
        e, sources, q = _dense_case_inputs(c)
        full = ref_search.dense_scores(q, e)
        for canonical in (False, True):
            rows, sims = ref_search.similarity_search_with_embedding(q, e, sources, c["k"], c["filter"], canonical)
            # BLAS may pick another kernel for the filtered sub-matrix: allow 1 ulp-ish noise
            assert_ranking_matches(c["rows"], c["sims"], rows, sims, tol=2e-6, full_scores=full,
                                   what=f"dense n={c['n']} k={c['k']} f={c['filter']} canon={canonical}")
            if len(rows):
                assert str(sims.dtype) == c["sim_dtype"]
        checked += 1
    assert checked > 100


def test_dense_filter_masks():
    src = ["CG100", "ng12", "NGX1", "QS15", None, "TA210", "cg3"]
    assert ref_search.dense_filter_mask(src, "CG").tolist() == [True, False, False, False, False, False, True]
    assert ref_search.dense_filter_mask(src, "cg, ng").tolist() == [True, True, True, False, False, False, True]
    # an empty prefix ("CG,") matches every non-null source on both paths
    assert ref_search.dense_filter_mask(src, "CG,").tolist() == [True, True, True, True, False, True, True]
    s2 = [s for s in src if s is not None]
    assert ref_search.bm25_filter_mask(s2, "CG,").tolist() == [True] * len(s2)
    assert ref_search.bm25_filter_mask(s2, "cg, ng").tolist() == [True, True, True, False, False, True]


def test_wrrf_matches_reference_bitwise():
    for c in load_golden("ref_wrrf.json"):
        lists = [(l, n) for l, n in c["lists"]]
        out = ref_search.weighted_reciprocal_rank_fusion(lists, c["weights"], c["k"])
        assert [[i, s] for i, s in out] == c["fused"]  # ids, order (incl. ties) and fp64 bits


def test_bm25_selection_matches_reference():
    cases = load_golden("ref_bm25_selection.json")
    for c in cases:
        if "empty_tokens" in c:
            assert c["empty_tokens"] == []
            continue
        scores = np.array(c["scores"])
        ref_rows = [int(i[3:]) for i in c["ids"]]
        rows = ref_search.core_bm25_search(scores, c["sources"], c["k"], c["filter"])
        assert rows.tolist() == ref_rows  # same numpy -> same order even inside ties
        rows_c = ref_search.core_bm25_search(scores, c["sources"], c["k"], c["filter"], canonical=True)
        if c["filter"]:
            # the reference's filter path IS the canonical rule (stable sort)
            assert rows_c.tolist() == ref_rows
        else:
            assert_ranking_matches(ref_rows, scores[ref_rows], rows_c, scores[rows_c], 0.0, scores, "bm25sel")


@pytest.fixture(scope="module")
def e2e():
    g = load_golden("ref_end_to_end.json")
    co = g["corpus"]
    chunks = co["chunks"]
    regenerated = synth_chunks(co["n"], co["chunk_seed"])
    assert [c["tokens"] for c in regenerated] == [c["tokens"] for c in chunks]
    e1 = synth_dense(co["n"], co["d"], co["e1_seed"])
    e2 = synth_dense(co["n"], co["d"], co["e2_seed"])
    ids = [c["id"] for c in chunks]
    sources = [c["source"] for c in chunks]
    kept = [c for c in chunks if c["tokens"]]
    bm25 = BM25Okapi([c["tokens"] for c in kept], k1=1.7, b=0.83, epsilon=0.05)
    dense = {
        "voyage-3-large": ref_retrieval.DenseCorpus(ids, sources, e1),
        "text-embedding-3-large": ref_retrieval.DenseCorpus(ids, sources, e2),
    }
    bm = ref_retrieval.Bm25Corpus(bm25, [c["id"] for c in kept], [c["source"] for c in kept])
    return g, dense, bm, e1, e2


def test_end_to_end_matches_reference(e2e):
    g, dense, bm, e1, e2 = e2e
    for c in g["cases"]:
        q = {"voyage-3-large": synth_query(e1, c["q1_seed"], c["target"]),
             "text-embedding-3-large": synth_query(e2, c["q2_seed"], c["target"])}
        out = ref_retrieval.retrieve_ids(dense, bm, q, c["tokens"], **c["cfg"])
        assert out == c["ids"], (c["cfg"], c["tokens"])


def test_documents_and_rerank_handoff_match_reference(e2e):
    """`return_docs=True` and the reranker hand-off (query_rag_retrieval.py:372-407): the reference's own outputs with a
    deterministic stand-in for the hosted cross-encoder (oracle/make_golden.py: StubVoyageClient)."""
    from oracle.make_golden import StubVoyageClient

    g, dense, bm, e1, e2 = e2e
    chunks = g["corpus"]["chunks"]
    contents = [" ".join(c["tokens"]) or "-" for c in chunks]
    kept = [c for c in chunks if c["tokens"]]
    dense = {k: ref_retrieval.DenseCorpus(v.ids, v.sources, v.embeddings, contents) for k, v in dense.items()}
    bm = ref_retrieval.Bm25Corpus(bm.bm25, bm.section_ids, bm.section_sources, [" ".join(c["tokens"]) for c in kept])
    reranked = 0
    for c in load_golden("ref_end_to_end_docs.json")["cases"]:
        q = {"voyage-3-large": synth_query(e1, c["q1_seed"], c["target"]),
             "text-embedding-3-large": synth_query(e2, c["q2_seed"], c["target"])}
        client = StubVoyageClient()
        out = ref_retrieval.retrieve_docs(dense, bm, q, c["text"], c["tokens"], rerank_client=client, **c["cfg"])
        assert client.calls == c["rerank_calls"], c["cfg"]
        reranked += len(client.calls)
        if not c["cfg"]["return_docs"]:
            assert out == c["out"], c["cfg"]
            continue
        assert [d["id"] for d in out] == [d["id"] for d in c["out"]], c["cfg"]
        for got, want in zip(out, c["out"]):
            assert got["document"] == want["document"] and got["source"] == want["source"]
            assert abs(float(got["similarity"]) - want["similarity"]) <= 1e-6
            assert got.get("rerank_score") == want.get("rerank_score")
            if "embedding0" in want:  # which model's record stands (the first model's: :272-275)
                assert got["embedding0"] == want["embedding0"]
            else:
                assert "embedding0" not in got
    assert reranked >= 12


def test_metrics_match_reference():
    for c in load_golden("ref_metrics.json"):
        m = ref_retrieval.calculate_metrics(c["results"])
        got = {k: (None if v is None else float(v)) for k, v in m.items()}
        assert got == c["metrics"]

"""GPU: indexes that went through the on-disk formats answer exactly like the in-memory ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pickle_and_flat_index_search(tmp_path):
    from oracle.make_golden import synth_chunks, synth_dense, synth_query
    from oracle.ref_bm25 import BM25Okapi
    from oracle import ref_search
    from anrag import index_io
    from anrag.database_manager import DatabaseManager
    from anrag.search_engine import SearchEngine

    chunks = synth_chunks(300, 4)
    ids = [c["id"] for c in chunks]
    sources = [c["source"] for c in chunks]
    toks = [c["tokens"] for c in chunks]
    e = synth_dense(300, 32, 5)
    bi, sections, section_ids = index_io.index_with_bm25(ids, sources, [" ".join(t) for t in toks], toks)
    pkl = str(tmp_path / "bm25.pkl")
    index_io.export_bm25_to_file(bi, sections, section_ids, pkl)
    flat = str(tmp_path / "flat")
    index_io.save_flat_index(flat, ids, sources, e, bi, section_ids, [s.metadata["source"] for s in sections])

    dm, se = DatabaseManager(), SearchEngine(None, None)
    bm25_a, sec_a, sid_a = dm.load_bm25_from_pickle(pkl)
    df, (bm25_b, sec_b, sid_b) = dm.load_flat_index(flat)
    ref = BM25Okapi([t for t in toks if t], 1.7, 0.83, 0.05)
    q = synth_query(e, 9, 33)
    for query in (["asthma", "dose"], ["renal", "renal", "unknown"]):
        want = ref.get_scores(query)
        assert np.array_equal(bm25_a.get_scores(query), want) and np.array_equal(bm25_b.get_scores(query), want)
        for flt in (None, "CG,NG"):
            a = se.bm25_search_preprocessed(query, bm25_a, sec_a, sid_a, 10, flt)
            b = se.bm25_search_preprocessed(query, bm25_b, sec_b, sid_b, 10, flt)
            rows = ref_search.core_bm25_search(want, [s.metadata["source"] for s in sections], 10, flt, canonical=True)
            assert a == b == [section_ids[r] for r in rows]
    r = se.similarity_search_with_embedding(q, df, "m", 5)
    rows, sims = ref_search.similarity_search_with_embedding(q, e, sources, 5, None, canonical=True)
    assert r["id"].tolist() == [ids[i] for i in rows]

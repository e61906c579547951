"""CPU: on-disk formats either side of the hot path (no GPU: arrays and files only)."""
import pickle
import sqlite3

import numpy as np

from anrag import index_io
from anrag.bm25_index import Bm25Index


class FakeEncoder:
    def encode(self, texts):
        return np.stack([np.full(8, len(t), dtype=np.float32) for t in texts])


def test_create_embeddings_db_schema_and_incremental(tmp_path):
    db = str(tmp_path / "x" / "chunks.db")
    chunks = [{"title": f"NG1_sec{i}", "content": "text " * (i + 1), "source": "NG1"} for i in range(5)]
    chunks.append({"title": "", "content": "skipped"})
    assert index_io.create_embeddings_db(chunks, FakeEncoder(), db, batch_size=2) == 5
    assert index_io.create_embeddings_db(chunks, FakeEncoder(), db) == 0           # incremental: nothing new
    rows = sqlite3.connect(db).execute("SELECT id, content, source, embedding, url FROM chunks").fetchall()
    assert len(rows) == 5
    v = np.frombuffer(rows[2][3], dtype=np.float32)                                  # database_manager.py:49
    assert v.shape == (8,) and v[0] == len("text " * 3)


def test_bm25_build_skips_empty_and_pickles(tmp_path):
    ids = ["a", "b", "c", "d"]
    toks = [["asthma", "dose"], [], ["dose", "dose", "child"], None]
    idx, sections, section_ids = index_io.index_with_bm25(ids, ["CG1"] * 4, ["t"] * 4, toks)
    assert section_ids == ["a", "c"] and idx.n_docs == 2 and sections[1].metadata == {"id": "c", "source": "CG1"}
    p = str(tmp_path / "bm25.pkl")
    index_io.export_bm25_to_file(idx, sections, section_ids, p)
    data = pickle.load(open(p, "rb"))
    assert set(data) == {"bm25", "sections", "section_ids", "config"}
    st = data["bm25"]
    assert st.doc_freqs == [{"asthma": 1, "dose": 1}, {"dose": 2, "child": 1}] and st.doc_len == [2, 3]
    assert list(st.idf) == ["asthma", "dose", "child"] and st.k1 == 1.7


def test_flat_index_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    e = rng.standard_normal((50, 16), dtype=np.float32)
    corpus = [[f"t{j}" for j in rng.integers(0, 9, size=int(rng.integers(1, 8)))] for _ in range(50)]
    bi = Bm25Index(corpus, 1.7, 0.83, 0.05)
    ids = [f"CG{i}_x" for i in range(50)]
    index_io.save_flat_index(str(tmp_path / "flat"), ids, ["CG1"] * 50, e, bi, ids, ["CG1"] * 50)
    meta, emb, b2 = index_io.load_flat_index(str(tmp_path / "flat"))
    assert isinstance(emb, np.memmap) and np.array_equal(np.asarray(emb), e) and meta["ids"] == ids
    for name in ("indptr", "post_doc", "post_tf", "idf", "doc_len"):
        assert np.array_equal(np.asarray(getattr(b2, name)), getattr(bi, name))
    assert b2.vocab == bi.vocab and b2.avgdl == bi.avgdl and np.array_equal(b2.term_ids(["t1", "zz"]), bi.term_ids(["t1", "zz"]))

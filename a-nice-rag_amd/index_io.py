"""On-disk formats either side of the hot path.

  * `create_embeddings_db`  -- the reference's src/processing/create_database.py:51-69, :100-123 with the LOCAL
    encoder in place of the Voyage API: SQLite `chunks(id TEXT PK, content, source, embedding BLOB float32, url)`,
    `INSERT OR REPLACE`, incremental (existing ids are skipped, :87-97, :147-158).  `DatabaseManager` reads it.
  * `index_with_bm25` / `export_bm25_to_file` -- src/processing/bm25_search.py:45-93: chunks whose token list is
    empty are skipped; the pickle holds {"bm25", "sections", "section_ids", "config"} with a `Bm25Stats` object
    (plain arrays + vocabulary, no GPU handle) where the reference pickles a rank_bm25 object.
  * `save_flat_index` / `load_flat_index` -- a directory of .npy files (row-major fp32 matrix, CSR postings, idf,
    doc_len) + one JSON of ids/sources/vocabulary: `np.load(mmap_mode="r")` maps them and the upload reads
    straight from the page cache -- no per-row Python at load time (database_manager.py:47-61 loops over rows).
"""
from __future__ import annotations

import json
import os
import pickle
import sqlite3
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .bm25_index import Bm25Index


# ------------------------------------------------------------------ dense side: SQLite `chunks`
def init_database(db_path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(db_path)) or ".", exist_ok=True)
    conn = sqlite3.connect(db_path)
    conn.execute("CREATE TABLE IF NOT EXISTS chunks (id TEXT PRIMARY KEY, content TEXT NOT NULL, source TEXT NOT NULL,"
                 " embedding BLOB NOT NULL, created_at TIMESTAMP DEFAULT CURRENT_TIMESTAMP, url TEXT)")
    conn.commit()
    conn.close()


def create_embeddings_db(chunks: Sequence[Dict], encoder, db_path: str, batch_size: int = 100) -> int:
    """chunks: dicts with `title` (becomes the id, chunk_mds.py:464), `content`, `source`, optional `url`.
    Returns the number of rows written."""
    init_database(db_path)
    conn = sqlite3.connect(db_path)
    existing = {r[0] for r in conn.execute("SELECT id FROM chunks")}
    todo = [c for c in chunks if c.get("title") and c.get("content") and c["title"] not in existing]
    written = 0
    for lo in range(0, len(todo), batch_size):
        batch = todo[lo: lo + batch_size]
        emb = encoder.encode([c["content"] for c in batch])
        for c, v in zip(batch, emb):
            conn.execute("INSERT OR REPLACE INTO chunks (id, content, source, embedding, url) VALUES (?,?,?,?,?)",
                         (c["title"], c["content"], c.get("source", "unknown"),
                          np.asarray(v, dtype=np.float32).tobytes(), c.get("url")))
            written += 1
        conn.commit()
    conn.close()
    return written


# ------------------------------------------------------------------ BM25 side
class Section:
    """What the reference keeps per indexed chunk (a langchain `Document`, bm25_search.py:70): the callers read
    `.page_content` and `.metadata["id"|"source"]` only."""

    def __init__(self, page_content: str, metadata: Dict[str, str]):
        self.page_content = page_content
        self.metadata = metadata


@dataclass
class Bm25Stats:
    """rank_bm25-shaped view of a `Bm25Index` for pickling (doc_freqs / idf / doc_len / avgdl / k1 / b / epsilon)."""
    doc_freqs: List[Dict[str, int]]
    idf: Dict[str, float]
    doc_len: List[int]
    avgdl: float
    k1: float
    b: float
    epsilon: float
    average_idf: float


def index_with_bm25(ids: Sequence[str], sources: Sequence[str], contents: Sequence[str],
                    tokens: Sequence[Sequence[str]], k1: float = 1.7, b: float = 0.83, epsilon: float = 0.05
                    ) -> Tuple[Bm25Index, List[Section], List[str]]:
    """bm25_search.py:45-79."""
    sections, section_ids, corpus = [], [], []
    for cid, src, text, toks in zip(ids, sources, contents, tokens):
        if not toks or len(toks) == 0:
            continue
        sections.append(Section(text, {"id": cid, "source": src}))
        section_ids.append(cid)
        corpus.append(list(toks))
    return Bm25Index(corpus, k1=k1, b=b, epsilon=epsilon), sections, section_ids


def stats_of(index: Bm25Index, corpus_tokens: Optional[Sequence[Sequence[str]]] = None) -> Bm25Stats:
    words = list(index.vocab)
    doc_freqs: List[Dict[str, int]] = [dict() for _ in range(index.n_docs)]
    for t, w in enumerate(words):
        lo, hi = index.indptr[t], index.indptr[t + 1]
        for d, c in zip(index.post_doc[lo:hi].tolist(), index.post_tf[lo:hi].tolist()):
            doc_freqs[d][w] = c
    return Bm25Stats(doc_freqs, {w: float(index.idf[t]) for t, w in enumerate(words)}, index.doc_len.tolist(),
                     index.avgdl, index.k1, index.b, index.epsilon, index.average_idf)


def export_bm25_to_file(index: Bm25Index, sections, section_ids, filepath: str, config_info=None) -> None:
    """bm25_search.py:82-93."""
    os.makedirs(os.path.dirname(os.path.abspath(filepath)) or ".", exist_ok=True)
    with open(filepath, "wb") as f:
        pickle.dump({"bm25": stats_of(index), "sections": sections, "section_ids": section_ids,
                     "config": config_info or {"k1": index.k1, "b": index.b, "epsilon": index.epsilon}}, f)


# ------------------------------------------------------------------ flat, mmap-able index
def save_flat_index(path: str, ids: Sequence[str], sources: Sequence[str], embeddings: np.ndarray,
                    bm25: Optional[Bm25Index] = None, section_ids: Optional[Sequence[str]] = None,
                    section_sources: Optional[Sequence[str]] = None) -> None:
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, "embeddings.npy"), np.ascontiguousarray(embeddings, dtype=np.float32))
    meta = {"ids": list(ids), "sources": list(sources), "dim": int(embeddings.shape[1]), "format": 1}
    if bm25 is not None:
        for name in ("indptr", "post_doc", "post_tf", "idf", "doc_len"):
            np.save(os.path.join(path, f"bm25_{name}.npy"), getattr(bm25, name))
        meta["bm25"] = {"k1": bm25.k1, "b": bm25.b, "epsilon": bm25.epsilon, "avgdl": bm25.avgdl,
                        "average_idf": bm25.average_idf, "vocab": list(bm25.vocab),
                        "section_ids": list(section_ids), "section_sources": list(section_sources)}
    with open(os.path.join(path, "meta.json"), "w", encoding="utf-8") as f:
        json.dump(meta, f, ensure_ascii=False)


def load_flat_index(path: str):
    """-> (meta dict, embeddings memmap, Bm25Index or None)."""
    with open(os.path.join(path, "meta.json"), encoding="utf-8") as f:
        meta = json.load(f)
    emb = np.load(os.path.join(path, "embeddings.npy"), mmap_mode="r")
    bm25 = None
    if "bm25" in meta:
        m = meta["bm25"]
        bm25 = Bm25Index.__new__(Bm25Index)
        bm25.k1, bm25.b, bm25.epsilon = m["k1"], m["b"], m["epsilon"]
        bm25.avgdl, bm25.average_idf = m["avgdl"], m["average_idf"]
        bm25.vocab = {w: t for t, w in enumerate(m["vocab"])}
        for name in ("indptr", "post_doc", "post_tf", "idf", "doc_len"):
            setattr(bm25, name, np.load(os.path.join(path, f"bm25_{name}.npy"), mmap_mode="r"))
        bm25.n_docs = int(bm25.doc_len.shape[0])
    return meta, emb, bm25

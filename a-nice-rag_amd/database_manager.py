"""`DatabaseManager`: same surface as the reference's src/database_manager.py:14-99, but loading
is where the corpus goes to HBM.

    load_embeddings_from_sql(db_path, model_name=None) -> DataFrame[id, document, source, embedding, url]
    load_bm25_from_pickle(filepath) -> (bm25, sections, section_ids)

The DataFrame is what the reference returns (callers read its columns, :63); in addition its
`attrs["_anrag"]` carries a `DenseHandle`: the row-major fp32 matrix uploaded once
(`anrag_dense_load`) plus the interned `source` column, so that no query ever re-stacks the matrix
(the reference's np.stack per query, search_engine.py:80).  The `bm25` object is a `Bm25Proxy` over
CSR postings in HBM with the `get_scores` method callers of rank_bm25 expect.
"""
from __future__ import annotations

import logging
import os
import pickle
import sqlite3
import threading
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from .bm25_index import Bm25Index
from .index import Index

ATTR = "_anrag"


def _default_device() -> int:
    return int(os.environ.get("ANRAG_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def intern_sources(sources: Sequence[Optional[str]]) -> Tuple[np.ndarray, List[Optional[str]]]:
    """source strings -> (uint16 id per row, distinct strings).  None/NaN keeps its own id."""
    table: Dict[object, int] = {}
    ids = np.empty(len(sources), dtype=np.uint16)
    for i, s in enumerate(sources):
        key = s if isinstance(s, str) else None
        j = table.get(key)
        if j is None:
            j = table[key] = len(table)
            if j >= 65536:
                raise ValueError("more than 65536 distinct sources")
        ids[i] = j
    return ids, list(table)


class DenseHandle:
    """A DataFrame's corpus matrix resident in HBM (rows == DataFrame positions)."""

    def __init__(self, embeddings: np.ndarray, sources: Sequence[Optional[str]], device: Optional[int] = None):
        self.index = Index(_default_device() if device is None else device)
        self.source_id, self.distinct_sources = intern_sources(sources)
        self.index.dense_load(embeddings, source_id=self.source_id)
        self.n_rows, self.dim = embeddings.shape
        self._allow_cache: Dict[Tuple[str, str], np.ndarray] = {}
        self._frame = None  # weakref to the DataFrame whose rows these are

    def __deepcopy__(self, memo):
        # pandas deep-copies `attrs` into every derived frame; the HBM copy is not something to clone.
        return self

    def bind(self, df: pd.DataFrame) -> "DenseHandle":
        self._frame = weakref.ref(df)
        df.attrs[ATTR] = self
        return self

    @classmethod
    def of(cls, df: pd.DataFrame) -> "DenseHandle":
        """The handle of exactly this DataFrame.  A frame that did not come from `DatabaseManager` (or was
        derived from one: filtered, re-ordered) gets its own upload -- still the HIP path, there is no CPU
        search in this package."""
        h = df.attrs.get(ATTR)
        if h is None or h._frame is None or h._frame() is not df:
            emb = np.stack(df["embedding"].values).astype(np.float32, copy=False)  # once per frame, not per query
            h = cls(emb, df["source"].tolist()).bind(df)
        return h


class Bm25Proxy:
    """Stands where the pickled `rank_bm25.BM25Okapi` stood: same statistics attributes and a
    `get_scores(tokens)` (fp64, bit-identical arithmetic, computed on the GPU)."""

    def __init__(self, stats: Bm25Index, section_sources: Sequence[str], device: Optional[int] = None):
        self.stats = stats
        self.k1, self.b, self.epsilon = stats.k1, stats.b, stats.epsilon
        self.corpus_size = stats.n_docs
        self.avgdl = stats.avgdl
        self.doc_len = stats.doc_len
        self.average_idf = stats.average_idf
        self.source_id, self.distinct_sources = intern_sources(section_sources)
        self.index = Index(_default_device() if device is None else device)
        self.index.bm25_load(stats.indptr, stats.post_doc, stats.post_tf, stats.idf, stats.doc_len, stats.avgdl,
                             stats.k1, stats.b, source_id=self.source_id)
        self.fused: Dict[int, "FusedPair"] = {}

    @property
    def idf(self) -> Dict[str, float]:
        return {w: float(self.stats.idf[t]) for w, t in self.stats.vocab.items()}

    def term_ids(self, tokens: Sequence[str]) -> np.ndarray:
        return self.stats.term_ids(tokens)

    def get_scores(self, query: Sequence[str]) -> np.ndarray:
        return self.index.bm25_scores(self.term_ids(query))

    @classmethod
    def from_rank_bm25(cls, bm25, section_sources: Sequence[str]) -> "Bm25Proxy":
        """Convert an unpickled rank_bm25-style object (doc_freqs list of {term: tf}, idf dict, doc_len,
        avgdl, k1, b, epsilon) without re-deriving anything: its idf values are taken as they are."""
        stats = Bm25Index.__new__(Bm25Index)
        stats.k1, stats.b = float(bm25.k1), float(bm25.b)
        stats.epsilon = float(getattr(bm25, "epsilon", 0.25))
        vocab = {w: t for t, w in enumerate(bm25.idf)}
        term_of, doc_of, tf_of = [], [], []
        for d, freqs in enumerate(bm25.doc_freqs):
            for w, c in freqs.items():
                term_of.append(vocab[w])
                doc_of.append(d)
                tf_of.append(c)
        term_arr = np.asarray(term_of, dtype=np.int64)
        order = np.argsort(term_arr, kind="stable")
        stats.vocab = vocab
        stats.n_docs = len(bm25.doc_freqs)
        stats.doc_len = np.asarray(bm25.doc_len, dtype=np.int32)
        stats.avgdl = float(bm25.avgdl)
        stats.post_doc = np.asarray(doc_of, dtype=np.int32)[order]
        stats.post_tf = np.asarray(tf_of, dtype=np.int32)[order]
        df = np.bincount(term_arr, minlength=len(vocab)).astype(np.int64)
        stats.indptr = np.zeros(len(vocab) + 1, dtype=np.int64)
        np.cumsum(df, out=stats.indptr[1:])
        stats.idf = np.asarray([bm25.idf[w] for w in vocab], dtype=np.float64)
        stats.average_idf = float(getattr(bm25, "average_idf", 0.0))
        return cls(stats, section_sources)


class FusedPair:
    """One dense DataFrame + the BM25 sections joined on chunk id in ONE index, so that a hybrid query
    is a single `anrag_hybrid_search` (doc id = dense row; BM25-only sections get ids past the last row)."""

    def __init__(self, dense: DenseHandle, dense_ids: Sequence[str], proxy: Bm25Proxy, section_ids: Sequence[str]):
        pos = {cid: i for i, cid in enumerate(dense_ids)}
        n = len(dense_ids)
        extra: List[str] = []
        doc = np.empty(len(section_ids), dtype=np.int64)
        for j, cid in enumerate(section_ids):
            i = pos.get(cid)
            if i is None:
                i = n + len(extra)
                extra.append(cid)
            doc[j] = i
        self.id_of_doc = list(dense_ids) + extra
        self.dense, self.proxy = dense, proxy
        st = proxy.stats
        dense.index.bm25_load(st.indptr, st.post_doc, st.post_tf, st.idf, st.doc_len, st.avgdl, st.k1, st.b,
                              source_id=proxy.source_id, doc_id=doc)

    @classmethod
    def of(cls, df: pd.DataFrame, proxy: Bm25Proxy, section_ids: Sequence[str]) -> "FusedPair":
        dense = DenseHandle.of(df)
        pair = proxy.fused.get(id(dense))
        if pair is None:
            pair = proxy.fused[id(dense)] = cls(dense, df["id"].tolist(), proxy, section_ids)
        return pair


class DatabaseManager:
    def __init__(self):
        self._embeddings_cache: Dict[str, pd.DataFrame] = {}
        self._bm25_cache: Dict[str, Tuple] = {}
        self._lock = threading.Lock()
        self.logger = logging.getLogger(__name__)

    def load_embeddings_from_sql(self, db_path: str, model_name: str = None) -> pd.DataFrame:
        """database_manager.py:22-75: `chunks(id, content, source, embedding BLOB, url)` -> DataFrame,
        cached per (path, model); errors are logged and re-raised (:70-72)."""
        cache_key = f"{db_path}_{model_name}" if model_name else db_path
        with self._lock:
            if cache_key in self._embeddings_cache:
                return self._embeddings_cache[cache_key]
        conn = None
        try:
            if not os.path.exists(db_path):
                raise FileNotFoundError(f"Database not found: {db_path}")
            conn = sqlite3.connect(db_path)
            rows = conn.execute("SELECT id, content, source, embedding, url FROM chunks").fetchall()
            if not rows:
                self.logger.warning(f"No chunks found in {db_path}")
                return pd.DataFrame()
            ids, docs, sources, embs, urls = [], [], [], [], []
            for cid, content, source, blob, url in rows:
                try:
                    emb = np.frombuffer(blob, dtype=np.float32)
                except (ValueError, TypeError) as e:
                    self.logger.warning(f"Skipping invalid row {cid}: {e}")
                    continue
                ids.append(cid)
                docs.append(content)
                sources.append(source)
                embs.append(emb)
                urls.append(url)
            df = pd.DataFrame({"id": ids, "document": docs, "source": sources, "embedding": embs, "url": urls})
            matrix = np.stack(embs)  # the one and only stack: straight to HBM
            DenseHandle(matrix, sources).bind(df)
            with self._lock:
                self._embeddings_cache[cache_key] = df
            return df
        except Exception as e:
            self.logger.error(f"Error loading embeddings from {db_path}: {e}")
            raise
        finally:
            if conn is not None:
                conn.close()

    def load_flat_index(self, path: str, model_name: str = None):
        """A flat, mmap-able index directory (anrag.index_io.save_flat_index) -> the same objects the two loaders
        above return: (DataFrame with HBM handle, (bm25 proxy, sections, section_ids) or None)."""
        from . import index_io

        meta, emb, stats = index_io.load_flat_index(path)
        df = pd.DataFrame({"id": meta["ids"], "document": [""] * len(meta["ids"]), "source": meta["sources"],
                           "embedding": list(emb), "url": [None] * len(meta["ids"])})
        DenseHandle(np.asarray(emb), meta["sources"]).bind(df)
        bm25_tuple = None
        if stats is not None:
            m = meta["bm25"]
            sections = [index_io.Section("", {"id": i, "source": s}) for i, s in zip(m["section_ids"], m["section_sources"])]
            bm25_tuple = (Bm25Proxy(stats, m["section_sources"]), sections, m["section_ids"])
        return df, bm25_tuple

    def load_bm25_from_pickle(self, filepath: str) -> Tuple:
        """database_manager.py:77-99: `{"bm25", "sections", "section_ids"}` pickle (written by
        processing/bm25_search.py:82-93) -> (bm25, sections, section_ids); the bm25 object is replaced by
        a `Bm25Proxy` whose postings live in HBM."""
        with self._lock:
            if filepath in self._bm25_cache:
                return self._bm25_cache[filepath]
        try:
            if not os.path.exists(filepath):
                raise FileNotFoundError(f"BM25 index not found: {filepath}")
            with open(filepath, "rb") as f:
                data = pickle.load(f)
            sections, section_ids = data["sections"], data["section_ids"]
            bm25 = data["bm25"]
            if not isinstance(bm25, Bm25Proxy):
                sources = [s.metadata.get("source", "") for s in sections]
                bm25 = Bm25Proxy.from_rank_bm25(bm25, sources)
            result = (bm25, sections, section_ids)
            with self._lock:
                self._bm25_cache[filepath] = result
            return result
        except Exception as e:
            self.logger.error(f"Error loading BM25 index from {filepath}: {e}")
            raise

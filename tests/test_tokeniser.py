"""CPU: anrag.preprocess_bm25 against the reference's shipped (query -> tokens) pairs."""
import gzip
import json
import os

import pytest

from anrag.preprocess_bm25 import preprocess_text

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["suggested_queries_bm25_preprocessed", "test_queries_bm25"])
def test_shipped_pairs(name):
    with gzip.open(os.path.join(GOLD, name + ".json.gz"), "rt", encoding="utf-8") as f:
        rows = json.load(f)
    assert len(rows) > 8000
    for r in rows:
        assert preprocess_text(r["query"]) == r["tokens_regular"], r["query"]
        # lemmas come from a lexicon extracted from these same files: this half only proves the plumbing
        assert preprocess_text(r["query"], use_lemmatization=True) == r["tokens_lemmatized"], r["query"]


def test_edge_cases():
    assert preprocess_text("") == [] and preprocess_text(None) == []
    assert preprocess_text("The 12 mg dose, for a child's asthma!") == ["mg", "dose", "childs", "asthma"]
    assert preprocess_text("You cannot re-use it") == ["reuse"]  # "cannot" -> can + not, both stopwords
    assert preprocess_text("mother–baby “bonding”") == ["mother–baby", "bonding"]

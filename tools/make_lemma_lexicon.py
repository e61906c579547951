#!/usr/bin/env python3
"""tests/golden/*_bm25*.json.gz (the reference's shipped query DATA, re-encoded by tools/make_data_fixtures.py)
-> a-nice-rag_amd/data/lemma_lexicon.json.gz: every distinct (tokens_regular -> tokens_lemmatized) pair, as
{"changed": {token: lemma}, "unchanged": [token, ...]} -- what `anrag.preprocess_bm25.NounLemmatizer` stands on
in place of WordNet's lemma index (not available offline)."""
import gzip
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def observed_pairs():
    pairs = {}
    for name in ("suggested_queries_bm25_preprocessed", "test_queries_bm25"):
        with gzip.open(os.path.join(GOLD, name + ".json.gz"), "rt", encoding="utf-8") as f:
            for r in json.load(f):
                assert len(r["tokens_regular"]) == len(r["tokens_lemmatized"])
                for a, b in zip(r["tokens_regular"], r["tokens_lemmatized"]):
                    assert pairs.setdefault(a, b) == b, (a, b, pairs[a])  # the lemmatiser is a function of the token
    return pairs


def main():
    pairs = observed_pairs()
    data = {"changed": {a: b for a, b in sorted(pairs.items()) if a != b},
            "unchanged": sorted(a for a, b in pairs.items() if a == b)}
    out = os.path.join(REPO, "a-nice-rag_amd", "data", "lemma_lexicon.json.gz")
    with gzip.open(out, "wt", encoding="utf-8", compresslevel=9) as f:
        json.dump(data, f, ensure_ascii=False, separators=(",", ":"))
    print("changed", len(data["changed"]), "unchanged", len(data["unchanged"]), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

"""CPU: anrag.preprocess_bm25 against the reference's shipped (query -> tokens) pairs.

`tokens_regular` pins lower / punctuation / word_tokenize / stopwords / numeric / length: exact on every pair.
The lemmatised half is pinned two ways: (1) the shipped pairs themselves (exact -- but the product's table was
read off these files, so that alone only proves plumbing); (2) HELD OUT: 5-fold cross-validation over word TYPES --
the lemmatiser is rebuilt without the fold's words, which then are unseen words to it, and its answers are scored
against what WordNet said in the shipped data.  (Splitting by FILE would hold nothing out: 8,134 of the 8,168
queries of test_queries_bm25 are also in suggested_queries_bm25_preprocessed.)"""
import gzip
import json
import os
import random
import sys
from collections import Counter

import pytest

from anrag.preprocess_bm25 import NounLemmatizer, lemmatizer, preprocess_text

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rows(name):
    with gzip.open(os.path.join(GOLD, name + ".json.gz"), "rt", encoding="utf-8") as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["suggested_queries_bm25_preprocessed", "test_queries_bm25"])
def test_shipped_pairs(name):
    rows = _rows(name)
    assert len(rows) > 8000
    for r in rows:
        assert preprocess_text(r["query"]) == r["tokens_regular"], r["query"]
        assert preprocess_text(r["query"], use_lemmatization=True) == r["tokens_lemmatized"], r["query"]


def test_lemmatiser_on_held_out_words():
    freq = Counter()
    seen_queries = set()
    for name in ("suggested_queries_bm25_preprocessed", "test_queries_bm25"):
        for r in _rows(name):
            if r["query"] in seen_queries:
                continue
            seen_queries.add(r["query"])
            freq.update(zip(r["tokens_regular"], r["tokens_lemmatized"]))
    truth = {a: b for a, b in freq}
    types = sorted(truth)
    random.Random(0).shuffle(types)
    folds = 5
    tok = tok_ok = typ_ok = base_ok = 0
    routes = Counter()
    for k in range(folds):
        held = set(types[k::folds])
        lem = NounLemmatizer((a, b) for a, b in truth.items() if a not in held)
        for a in held:
            n = freq[(a, truth[a])]
            got = lem.lemmatize(a)
            tok += n
            tok_ok += n * (got == truth[a])
            typ_ok += got == truth[a]
            base_ok += n * (a == truth[a])
        routes.update(lem.counts)
        assert lem.counts["observed"] == 0  # every scored word was unseen
    token_acc, type_acc, unchanged_acc = tok_ok / tok, typ_ok / len(types), base_ok / tok
    print(f"\nheld-out lemmatiser accuracy over {len(types)} word types / {tok} tokens: "
          f"tokens {token_acc:.4f}, types {type_acc:.4f} (unseen words left unchanged: {unchanged_acc:.4f}); "
          f"routes {dict(routes)}", file=sys.stderr)
    assert token_acc >= 0.96 and type_acc >= 0.97
    assert token_acc > unchanged_acc + 0.15


def test_morphy_rules_and_exceptions():
    lem = NounLemmatizer([("child", "child"), ("children", "child"), ("glass", "glass"), ("ga", "ga"), ("gas", "ga"),
                          ("study", "study"), ("woman", "woman"), ("knife", "knife"), ("box", "box"),
                          ("criteria", "criterion")])
    assert lem.exceptions == {"children": "child", "criteria": "criterion"}
    assert lem.lemmatize("glasses") == "glass"        # ses -> s, accepted by the dictionary
    assert lem.lemmatize("studies") == "study"        # ies -> y
    assert lem.lemmatize("women") == "woman"          # men -> man
    assert lem.lemmatize("knives") == "knife" or lem.lemmatize("knives") == "knif"  # ves -> f ("knif" unless known)
    assert lem.lemmatize("boxes") == "box"            # xes -> x
    assert lem.lemmatize("gas") == "ga"               # observed: WordNet's shortest-candidate quirk is kept
    # unseen word, no dictionary candidate: the plain plural rule is taken anyway.  PARITY UNPINNED: WordNet accepts a
    # candidate only if it is one of its nouns ("tablets" -> "tablet", but "diabetes" stays "diabetes"); without its
    # index (not available offline) rule 3 is a guess, so nothing here asserts WHICH lemma such a word gets -- only
    # that the answer is the word itself or one of morphy's detachment candidates, and that the route is counted.
    assert lem.lemmatize("tablets") in ("tablet", "tablets")
    assert lem.lemmatize("diabetes") in ("diabetes", "diabete")
    assert lem.lemmatize("class") == "class" and lem.lemmatize("virus") == "virus" and lem.lemmatize("25s") == "25s"
    assert lem.counts["observed"] == 1 and lem.counts["rule"] >= 2 and lem.counts["unchanged"] >= 3


def test_product_lemmatiser_counts_routes():
    lem = lemmatizer()
    before = Counter(lem.counts)
    assert preprocess_text("children with unseenwordzzs", use_lemmatization=True) == ["child", "unseenwordzz"]
    delta = lem.counts - before
    assert delta["observed"] == 1 and delta["rule"] == 1


def test_edge_cases():
    assert preprocess_text("") == [] and preprocess_text(None) == []
    assert preprocess_text("The 12 mg dose, for a child's asthma!") == ["mg", "dose", "childs", "asthma"]
    assert preprocess_text("You cannot re-use it") == ["reuse"]  # "cannot" -> can + not, both stopwords
    assert preprocess_text("mother–baby “bonding”") == ["mother–baby", "bonding"]

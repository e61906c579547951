#!/usr/bin/env python3
"""Turn the reference's shipped DATA files (not source) into compact fixtures.  Build-container only.

  /root/reference/data/suggested_queries_bm25_preprocessed.csv   (9,609: id, query, tokens_regular, tokens_lemmatized)
  /root/reference/data/test_queries_bm25.csv                     (8,168: same schema)
  /root/reference/data/NICEQA.csv                                (70: Guideline ID, Section, Question)
->
  tests/golden/suggested_queries_bm25_preprocessed.json.gz, tests/golden/test_queries_bm25.json.gz
        the tokeniser's golden vectors and the id-space / stand-in corpus of the
        NICEQA Recall@10 measurement
  tests/golden/NICEQA.csv
(the lemmatiser's observed-pair table is derived from these fixtures by tools/make_lemma_lexicon.py)
"""
import ast
import csv
import gzip
import json
import os
import shutil

REF = "/root/reference/data"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def main():
    for name in ("suggested_queries_bm25_preprocessed", "test_queries_bm25"):
        rows = []
        with open(os.path.join(REF, name + ".csv"), encoding="utf-8") as f:
            for r in csv.DictReader(f):
                reg = ast.literal_eval(r["tokens_regular"])
                lem = ast.literal_eval(r["tokens_lemmatized"])
                assert len(reg) == len(lem)
                rows.append({"id": r["id"], "query": r["query"], "tokens_regular": reg, "tokens_lemmatized": lem})
        with gzip.open(os.path.join(GOLD, name + ".json.gz"), "wt", encoding="utf-8", compresslevel=9) as f:
            json.dump(rows, f, ensure_ascii=False, separators=(",", ":"))
        print(name, len(rows))
    shutil.copyfile(os.path.join(REF, "NICEQA.csv"), os.path.join(GOLD, "NICEQA.csv"))


if __name__ == "__main__":
    main()

"""GPU: the reference-shaped host layer (DatabaseManager / SearchEngine / RetrievalEvaluationSystem) end to end
over real SQLite `chunks` DBs and a BM25 pickle, against the ids the REFERENCE's own retrieve_documents returned
(tests/golden/ref_end_to_end.json) -- through both routes: method-by-method and the fused single ABI call."""
import pickle
import sqlite3

import numpy as np
import pandas as pd
import pytest

from helpers import load_golden

pytestmark = pytest.mark.gpu


def _write_db(path, chunks, e):
    conn = sqlite3.connect(path)
    conn.execute("CREATE TABLE chunks (id TEXT PRIMARY KEY, content TEXT NOT NULL, source TEXT NOT NULL,"
                 " embedding BLOB NOT NULL, created_at TIMESTAMP DEFAULT CURRENT_TIMESTAMP, url TEXT)")
    for c, v in zip(chunks, e):
        conn.execute("INSERT INTO chunks (id, content, source, embedding, url) VALUES (?,?,?,?,?)",
                     (c["id"], " ".join(c["tokens"]) or "-", c["source"], np.asarray(v, np.float32).tobytes(), None))
    conn.commit()
    conn.close()


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from oracle.make_golden import Document, synth_dense
    from oracle.ref_bm25 import BM25Okapi
    from anrag.config import Config, InfoSource

    g = load_golden("ref_end_to_end.json")
    co = g["corpus"]
    chunks = co["chunks"]
    tmp = tmp_path_factory.mktemp("anrag_e2e")
    e1 = synth_dense(co["n"], co["d"], co["e1_seed"])
    e2 = synth_dense(co["n"], co["d"], co["e2_seed"])
    db1, db2, pkl = str(tmp / "m1.db"), str(tmp / "m2.db"), str(tmp / "bm25.pkl")
    _write_db(db1, chunks, e1)
    _write_db(db2, chunks, e2)
    kept = [c for c in chunks if c["tokens"]]
    # the pickle holds a rank_bm25-SHAPED object (the oracle's restatement: rank_bm25 itself is not installed)
    bm25 = BM25Okapi([c["tokens"] for c in kept], k1=1.7, b=0.83, epsilon=0.05)
    sections = [Document(" ".join(c["tokens"]), {"id": c["id"], "source": c["source"]}) for c in kept]
    with open(pkl, "wb") as f:
        pickle.dump({"bm25": bm25, "sections": sections, "section_ids": [c["id"] for c in kept], "config": {}}, f)
    cfg = Config()
    sc = cfg.SOURCE_CONFIGS[InfoSource.NICE]
    saved = dict(vars(sc))
    sc.db_path = sc.voyage_db_path = db1
    sc.openai_db_path = db2
    sc.voyage_3_5_db_path = sc.qwen_db_path = sc.local_db_path = None
    sc.bm25_path = pkl
    yield g, cfg, e1, e2, kept
    for k, v in saved.items():
        setattr(sc, k, v)


@pytest.mark.parametrize("fused", [False, True])
def test_retrieve_documents_matches_reference(world, fused):
    from oracle.make_golden import synth_query
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem

    g, cfg, e1, e2, kept = world
    system = RetrievalEvaluationSystem(cfg, fused=fused)
    for c in g["cases"]:
        q = {"voyage-3-large": synth_query(e1, c["q1_seed"], c["target"]),
             "text-embedding-3-large": synth_query(e2, c["q2_seed"], c["target"])}
        out = system.retrieve_documents(query_embeddings=q, query_tokens=c["tokens"], use_reranker=False, **c["cfg"])
        assert out == c["ids"], (fused, c["cfg"], c["tokens"])


@pytest.mark.parametrize("ranked_from", [1, 10 ** 9])
def test_retrieve_documents_batch_matches_reference(world, ranked_from):
    """The reference's own `retrieve_documents` outputs (72 cases: 1-2 dense models, with / without BM25 and filter,
    k 10 ... 300) through the LIST entry point: cases that share their keyword arguments are asked together.  The
    k <= 64 hybrid cases take `anrag_hybrid_search_batch` (short lists: `ranked_from` = never) or, like everything else,
    `anrag_rank_batch` (lists of 16 and more in production: `ranked_from` = 1 sends every list there); and the
    expected-id rank the evaluation harness asks for equals the position in the reference's list."""
    import json

    from oracle.make_golden import synth_query
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem

    g, cfg, e1, e2, kept = world
    system = RetrievalEvaluationSystem(cfg)
    system.RANKED_LIST_MIN = ranked_from
    groups = {}
    for c in g["cases"]:
        groups.setdefault(json.dumps(c["cfg"], sort_keys=True), []).append(c)
    routed = 0
    for key, cases in groups.items():
        params = json.loads(key)
        asks = [{"query_embeddings": {"voyage-3-large": synth_query(e1, c["q1_seed"], c["target"]),
                                      "text-embedding-3-large": synth_query(e2, c["q2_seed"], c["target"])},
                 "query_tokens": c["tokens"]} for c in cases]
        out = system.retrieve_documents_batch(asks, use_reranker=False, **params)
        assert out == [c["ids"] for c in cases], params
        expected = [c["ids"][len(c["ids"]) // 2] if c["ids"] else "no-such-id" for c in cases]
        expected[0] = "no-such-id"
        got = system.rank_of_expected_batch(asks, expected, use_reranker=False, **params)
        if got is not None:
            routed += 1
            ranks, totals = got
            for c, e, r, t in zip(cases, expected, ranks.tolist(), totals.tolist()):
                assert t == len(c["ids"]) and r == (c["ids"].index(e) + 1 if e in c["ids"] else -1), (params, e)
    assert routed == len(groups)  # every golden configuration is inside the batched ranking's envelope


def test_return_docs_and_rerank_handoff_match_reference(world):
    """query_rag_retrieval.py:372-407: `return_docs=True` hands back the records themselves (for an id two dense models
    return, the FIRST model's record: later models' rows are filtered by `existing_ids`, :272-275; BM25-only sections
    carry four keys and similarity 0.0, :336-350), and with `use_reranker` the fused documents go through `vo.rerank`
    -- here the deterministic stand-in client the golden vectors were generated with."""
    from oracle.make_golden import StubVoyageClient, synth_query
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem

    g, cfg, e1, e2, kept = world
    from oracle import ref_retrieval
    from oracle.ref_bm25 import BM25Okapi

    client = StubVoyageClient()
    system = RetrievalEvaluationSystem(cfg, voyage_client=client)
    chunks = g["corpus"]["chunks"]
    ids, sources = [c["id"] for c in chunks], [c["source"] for c in chunks]
    contents = [" ".join(c["tokens"]) or "-" for c in chunks]
    o_dense = {"voyage-3-large": ref_retrieval.DenseCorpus(ids, sources, e1, contents),
               "text-embedding-3-large": ref_retrieval.DenseCorpus(ids, sources, e2, contents)}
    o_bm = ref_retrieval.Bm25Corpus(BM25Okapi([c["tokens"] for c in kept], k1=1.7, b=0.83, epsilon=0.05),
                                    [c["id"] for c in kept], [c["source"] for c in kept],
                                    [" ".join(c["tokens"]) for c in kept])
    seen_rerank = tied = 0
    for c in load_golden("ref_end_to_end_docs.json")["cases"]:
        q = {"voyage-3-large": synth_query(e1, c["q1_seed"], c["target"]),
             "text-embedding-3-large": synth_query(e2, c["q2_seed"], c["target"])}
        client.calls.clear()
        out = system.retrieve_documents(query_embeddings=q, query_text=c["text"], query_tokens=c["tokens"], **c["cfg"])
        assert client.calls == c["rerank_calls"], c["cfg"]
        seen_rerank += len(client.calls)
        if not c["tie_free"]:
            # the reference's answer here depends on how numpy ordered EQUAL scores at a cut (unspecified); the build's
            # rule is (score desc, row asc): the device must equal the oracle's canonical reading ("Tie rule")
            tied += 1
            want = ref_retrieval.retrieve_docs(o_dense, o_bm, q, c["text"], c["tokens"], rerank_client=StubVoyageClient(),
                                               canonical=True, **c["cfg"])
            got_ids = [d["id"] for d in out] if c["cfg"]["return_docs"] else out
            assert got_ids == ([d["id"] for d in want] if c["cfg"]["return_docs"] else want), c["cfg"]
            continue
        if not c["cfg"]["return_docs"]:
            assert out == c["out"], c["cfg"]
            continue
        assert [d["id"] for d in out] == [d["id"] for d in c["out"]], c["cfg"]
        for got, want in zip(out, c["out"]):
            assert sorted(got.keys()) == want["keys"], (got.keys(), want["keys"])
            assert got["document"] == want["document"] and got["source"] == want["source"]
            assert abs(float(got["similarity"]) - want["similarity"]) <= 1e-4
            assert got.get("rerank_score") == want.get("rerank_score")
            if "embedding0" in want:
                assert float(np.asarray(got["embedding"]).reshape(-1)[0]) == want["embedding0"]
    assert seen_rerank >= 12 and tied <= 4  # nearly every golden case is free of ties at a cut


def test_fp64_query_scores_alike_on_both_routes(world):
    """The text path hands `retrieve_documents` a float64 embedding (search_engine.py:157: the API's list through
    np.array); the reference scores it in fp64 (:129).  Both routes of the host layer must do the same -- the fused
    single-call route declines fp64 queries -- and agree with the oracle's fp64 ranking."""
    from oracle import ref_retrieval
    from oracle.make_golden import synth_query
    from oracle.ref_bm25 import BM25Okapi
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem

    g, cfg, e1, e2, kept = world
    chunks = g["corpus"]["chunks"]
    ids, sources = [c["id"] for c in chunks], [c["source"] for c in chunks]
    bm = ref_retrieval.Bm25Corpus(BM25Okapi([c["tokens"] for c in kept], k1=1.7, b=0.83, epsilon=0.05),
                                  [c["id"] for c in kept], [c["source"] for c in kept])
    dense = {"voyage-3-large": ref_retrieval.DenseCorpus(ids, sources, e1)}
    fused, plain = RetrievalEvaluationSystem(cfg, fused=True), RetrievalEvaluationSystem(cfg, fused=False)
    rng = np.random.default_rng(31)
    for i in range(12):
        target = int(rng.integers(0, len(chunks)))
        q64 = synth_query(e1, 900 + i, target).astype(np.float64) + rng.standard_normal(e1.shape[1]) * 1e-9
        toks = chunks[target]["tokens"][:3] or ["asthma"]
        kw = dict(query_tokens=toks, similarity_k=25, common_sections_n=15, use_hybrid_search=True, wrrf_k=40,
                  model_weights={"voyage-3-large": 5.0, "BM25": 1.0}, filename_type_filter=None, use_reranker=False)
        a = fused.retrieve_documents(query_embeddings={"voyage-3-large": q64}, **kw)
        b = plain.retrieve_documents(query_embeddings={"voyage-3-large": q64}, **kw)
        want = ref_retrieval.retrieve_ids(dense, bm, {"voyage-3-large": q64}, toks, 25, 15,
                                          {"voyage-3-large": 5.0, "BM25": 1.0}, None, True, 40, canonical=True)
        assert a == b == want, i
        lists = fused.retrieve_documents_batch([{"query_embeddings": {"voyage-3-large": q64}, "query_tokens": toks}],
                                               **{k: v for k, v in kw.items() if k != "query_tokens"})
        assert lists == [want]


def test_database_manager_contract(world):
    from anrag.database_manager import ATTR, Bm25Proxy, DatabaseManager
    from anrag.config import InfoSource

    g, cfg, e1, e2, kept = world
    sc = cfg.SOURCE_CONFIGS[InfoSource.NICE]
    dm = DatabaseManager()
    df = dm.load_embeddings_from_sql(sc.db_path, "voyage-3-large")
    assert list(df.columns) == ["id", "document", "source", "embedding", "url"]
    assert len(df) == g["corpus"]["n"] and df["embedding"].iloc[0].dtype == np.float32
    assert dm.load_embeddings_from_sql(sc.db_path, "voyage-3-large") is df  # cached (database_manager.py:27-29)
    assert ATTR in df.attrs
    bm25, sections, section_ids = dm.load_bm25_from_pickle(sc.bm25_path)
    assert isinstance(bm25, Bm25Proxy) and len(sections) == len(section_ids) == len(kept)
    # the proxy answers like the pickled object did
    from oracle.ref_bm25 import BM25Okapi

    ref = BM25Okapi([c["tokens"] for c in kept], k1=1.7, b=0.83, epsilon=0.05)
    for toks in (["asthma", "dose"], ["insulin", "insulin", "nothere"], []):
        assert np.array_equal(bm25.get_scores(toks), ref.get_scores(toks))
    with pytest.raises(FileNotFoundError):
        dm.load_embeddings_from_sql("/nonexistent.db")
    with pytest.raises(FileNotFoundError):
        dm.load_bm25_from_pickle("/nonexistent.pkl")


def test_search_engine_surface(world):
    from oracle import ref_search
    from oracle.make_golden import synth_query
    from anrag.database_manager import DatabaseManager
    from anrag.config import InfoSource
    from anrag.search_engine import SearchEngine

    g, cfg, e1, e2, kept = world
    sc = cfg.SOURCE_CONFIGS[InfoSource.NICE]
    dm = DatabaseManager()
    df = dm.load_embeddings_from_sql(sc.db_path)
    bm25, sections, section_ids = dm.load_bm25_from_pickle(sc.bm25_path)
    se = SearchEngine(None, None)
    q = synth_query(e1, 1, 17)
    r = se.similarity_search_with_embedding(q, df, "m", 10, "CG,NG")
    rows, sims = ref_search.similarity_search_with_embedding(q, e1, df["source"].tolist(), 10, "CG,NG", canonical=True)
    assert r.index.tolist() == rows.tolist() and list(r.columns)[-1] == "similarity"
    assert np.max(np.abs(r["similarity"].to_numpy() - sims)) <= 1e-4 and r["similarity"].dtype == np.float32
    assert se.similarity_search_with_embedding(q.astype(np.float64), df, "m", 3)["similarity"].dtype == np.float64
    assert se.similarity_search_with_embedding(q, df, "m", 10, "ZZ").empty          # filter leaves nothing (:71-75)
    assert se.similarity_search_with_embedding(np.stack([q, q]), df, "m", 3).empty  # batched query: reference errors -> empty
    assert se.similarity_search("text", df, "voyage-3-large", 5).empty               # no Voyage client: logged, empty (:144-146)
    assert se.similarity_search("text", df, "m", 5, None, q).index.tolist() == r.index.tolist()[:0] + \
        se.similarity_search_with_embedding(q, df, "m", 5).index.tolist()
    # a frame that never saw DatabaseManager still searches on the GPU
    plain = pd.DataFrame({"id": df["id"], "source": df["source"], "embedding": list(e1)})
    assert se.similarity_search_with_embedding(q, plain, "m", 5).index.tolist() == \
        se.similarity_search_with_embedding(q, df, "m", 5).index.tolist()
    assert se.bm25_search_preprocessed([], bm25, sections, section_ids) == []
    ids = se.bm25_search_preprocessed(["asthma", "dose"], bm25, sections, section_ids, 5, "NG")
    assert len(ids) == 5 and all(i.upper().startswith("NG") for i in ids)
    assert se.bm25_search("What dose of asthma inhalers?", bm25, sections, section_ids, 5) == \
        se.bm25_search_preprocessed(["dose", "asthma", "inhaler"], bm25, sections, section_ids, 5)
    fused = se.weighted_reciprocal_rank_fusion([(["a", "b", "c"], "m1"), (["c", "a", "d"], "BM25")],
                                               {"m1": 5.0, "BM25": 1.0}, 40)
    assert fused == ref_search.weighted_reciprocal_rank_fusion([(["a", "b", "c"], "m1"), (["c", "a", "d"], "BM25")],
                                                               {"m1": 5.0, "BM25": 1.0}, 40)


def test_evaluator_and_metrics(world):
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem
    from anrag.retrieval_eval import RetrievalEvaluator, calculate_metrics
    from oracle.make_golden import synth_query

    g, cfg, e1, e2, kept = world
    ev = RetrievalEvaluator(retrieval_system=RetrievalEvaluationSystem(cfg))
    params = dict(similarity_k=25, common_sections_n=15, model_weights={"voyage-3-large": 5.0, "BM25": 1.0},
                  use_hybrid_search=True, wrrf_k=40, filename_type_filter=None)
    results = []
    for i in range(20):
        target = (i * 7) % len(g["corpus"]["chunks"])
        q = synth_query(e1, 2000 + i, target)
        results.append(ev.evaluate_query("q", g["corpus"]["chunks"][target]["id"], {"voyage-3-large": q}, params,
                                         g["corpus"]["chunks"][target]["tokens"][:4]))
    m = calculate_metrics(results)
    assert m["recall@1"] >= 0.9 and m["total"] == 20
    for c in load_golden("ref_metrics.json"):
        got = {k: (None if v is None else float(v)) for k, v in calculate_metrics(c["results"]).items()}
        assert got == c["metrics"]


def test_evaluation_run_lists_equal_single_queries(world, tmp_path):
    """The evaluation run (reference retrieval_eval.py:270-420: configuration table, query selection, 85 % split,
    CSV rows) asks its queries as LISTS; every configuration's metrics must equal those of the per-query
    `evaluate_query` loop -- full-ranking configurations (k = 12000, radix-sort path), BM25 only, the fused
    hybrid route and a two-model fusion."""
    import sqlite3

    from oracle.make_golden import synth_query
    from anrag.query_rag_retrieval import RetrievalEvaluationSystem
    from anrag import retrieval_eval as re_

    g, cfg, e1, e2, kept = world
    chunks = g["corpus"]["chunks"]
    targets = [i for i in range(len(chunks)) if chunks[i]["tokens"]][:120]

    def write_queries(path, emb, seed0, skip=()):
        conn = sqlite3.connect(path)
        conn.execute("CREATE TABLE queries (id TEXT PRIMARY KEY, query TEXT, query_embedding BLOB)")
        for j, t in enumerate(targets):
            if j in skip:
                continue
            conn.execute("INSERT INTO queries VALUES (?,?,?)",
                         (chunks[t]["id"], "q about " + " ".join(chunks[t]["tokens"][:3]),
                          synth_query(emb, seed0 + j, t).astype(np.float32).tobytes()))
        conn.commit()
        conn.close()

    p1, p2 = str(tmp_path / "q1.db"), str(tmp_path / "q2.db")
    write_queries(p1, e1, 7000)
    write_queries(p2, e2, 8000)
    cached = {"voyage-3-large": re_.load_queries_from_db(p1), "text-embedding-3-large": re_.load_queries_from_db(p2)}
    pre = pd.DataFrame({"id": [chunks[t]["id"] for t in targets],
                        "tokens_lemmatized": [chunks[t]["tokens"][:4] for t in targets]})
    configs = [
        re_._configuration("dense full ranking", {"voyage-3-large": 1.0}, False),
        re_._configuration("bm25 full ranking", {"BM25": 1.0}, True),
        re_._configuration("hybrid fused", {"voyage-3-large": 5.0, "BM25": 1.0}, True, 25, 15),
        re_._configuration("two models", {"voyage-3-large": 2.0, "text-embedding-3-large": 1.0}, False, 25, 15),
    ]
    ev = re_.RetrievalEvaluator(retrieval_system=RetrievalEvaluationSystem(cfg))
    out_csv = str(tmp_path / "results" / "eval.csv")
    metrics = re_.run_evaluation(ev, cached, pre, out_csv, configs, chunk=50)
    lines = open(out_csv).read().splitlines()
    assert lines[0] == ",".join(re_.CSV_HEADER) and len(lines) == 1 + len(configs)

    from sklearn.model_selection import train_test_split
    train_idx, _ = train_test_split(range(len(targets)), test_size=0.15, random_state=42, shuffle=True)
    tokens_of = dict(zip(pre["id"], pre["tokens_lemmatized"]))
    for config, m, line in zip(configs, metrics, lines[1:]):
        params = dict(re_.BASE_PARAMS)
        params.update({k: v for k, v in config.items() if k != "name"})
        queries, embeddings = re_.select_queries(cached, config["model_weights"])
        need = config["model_weights"]["BM25"] > 0 and config["use_hybrid_search"]
        single = []
        for i in train_idx:
            row = queries.iloc[i]
            single.append(ev.evaluate_query(row["query"], row["id"], {k: v[i] for k, v in embeddings.items()}, params,
                                            tokens_of.get(row["id"]) if need else None))
        want = re_.calculate_metrics(single)
        assert m == want, config["name"]
        assert m["total"] == len(train_idx) and m["recall@10"] > 0.5, (config["name"], m)
        assert line == re_.format_csv_row(config["name"], want).rstrip("\n")

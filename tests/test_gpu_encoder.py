"""GPU: text in -> local encoder on PyTorch-ROCm -> dense scan, i.e. `similarity_search(query_text, ...)` with no
remote API (the north star's replacement for search_engine.py:148-159)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def test_text_query_end_to_end():
    from anrag.config import LOCAL_ENCODER_KEY
    from anrag.encoder import LocalEncoder
    from anrag.search_engine import SearchEngine

    enc = LocalEncoder()  # cuda, fp16, random-init bge-small architecture
    assert enc.device.type == "cuda"
    cpu = LocalEncoder(device="cpu")
    docs = [f"guideline section {i} about {w}" for i, w in enumerate(
        ["asthma inhaler dose", "stroke rehabilitation", "diabetes insulin titration", "sepsis antibiotics",
         "antenatal screening", "hypertension treatment", "chronic kidney disease", "opioid prescribing"] * 40)]
    emb = enc.encode(docs)
    ref = cpu.encode(docs[:16])
    assert np.max(np.abs(emb[:16] - ref)) < 2e-2  # fp16 on the GPU vs fp32 on the host
    df = pd.DataFrame({"id": [f"NG{i}_sec" for i in range(len(docs))], "document": docs,
                       "source": [f"NG{i % 9}" for i in range(len(docs))], "embedding": list(emb)})
    se = SearchEngine(None, None, encoder=enc)
    # the query is one of the documents verbatim, minus the instruction prefix difference: it must rank near the top
    r = se.similarity_search(docs[17], df, LOCAL_ENCODER_KEY, similarity_k=5)
    assert len(r) == 5 and "similarity" in r.columns
    q = enc.encode_query(docs[17])
    full = emb @ q
    assert r.index.tolist() == np.lexsort((np.arange(len(docs)), -full))[:5].tolist() or \
        np.max(np.abs(np.sort(full)[::-1][:5] - r["similarity"].to_numpy())) < 1e-4


def test_query_graph_replay_matches_eager():
    """`encode_query` replays a captured hipGraph per token-length bucket; the result must be the eager forward's up
    to fp16 noise (the bucket pads the sequence; padded positions are masked), for queries of several lengths and
    when the same bucket is reused."""
    from anrag.encoder import LocalEncoder

    enc = LocalEncoder()
    texts = ["asthma", "what dose of inhaled corticosteroid for adults with asthma and chronic kidney disease",
             " ".join(["hypertension treatment threshold"] * 20), "stroke rehabilitation at home"]
    enc.use_graphs = False
    eager = np.stack([enc.encode_query(t) for t in texts])
    enc.use_graphs = True
    replay = np.stack([enc.encode_query(t) for t in texts])
    again = np.stack([enc.encode_query(t) for t in texts])
    assert any(v is not None for v in enc._graphs.values()), "graph capture did not happen on this build"
    assert np.max(np.abs(replay - eager)) < 3e-3 and np.array_equal(replay, again)
    assert np.allclose(np.linalg.norm(replay, axis=1), 1.0, atol=1e-3)

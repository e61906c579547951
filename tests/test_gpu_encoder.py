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

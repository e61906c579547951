"""CPU: the local encoder that stands where the Voyage API call stood (random-init bge-small architecture:
no weights exist offline) and its wiring into SearchEngine._generate_query_embedding."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def enc():
    from anrag.encoder import LocalEncoder

    return LocalEncoder(device="cpu")


def test_shapes_norm_determinism(enc):
    from anrag.encoder import LocalEncoder

    texts = ["What dose of inhaled corticosteroid for a child with asthma?", "stroke", ""]
    a = enc.encode(texts)
    assert a.shape == (3, 384) and a.dtype == np.float32
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5)
    b = LocalEncoder(device="cpu").encode(texts)  # same seed -> same weights -> same vectors
    assert np.array_equal(a, b)
    single = enc.encode(texts[:1])  # padding in a batch must not change a row
    assert np.max(np.abs(single[0] - a[0])) < 1e-5
    assert enc.model.config.num_hidden_layers == 12 and enc.model.config.intermediate_size == 1536


def test_search_engine_uses_it_for_its_model_key(enc):
    from anrag.config import LOCAL_ENCODER_KEY
    from anrag.search_engine import SearchEngine

    se = SearchEngine(None, None, encoder=enc)
    v = se._generate_query_embedding("asthma in children", LOCAL_ENCODER_KEY)
    assert v.shape == (1, 384)
    assert np.allclose(v[0], enc.encode_query("asthma in children"))
    with pytest.raises(ValueError):
        se._generate_query_embedding("x", "some-other-model")      # search_engine.py:158-159
    with pytest.raises(ValueError):
        se._generate_query_embedding("x", "voyage-3-large")        # no Voyage client: :151-152

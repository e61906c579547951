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


def test_local_model_path_route(tmp_path):
    """`LocalEncoder(model_path=...)`: tokenizer + weights from a LOCAL directory (what a deployment with
    bge-small-en-v1.5 on disk uses; nothing is fetched).  A small random BERT + a WordPiece vocabulary are saved with
    transformers' own `save_pretrained`, loaded back through the encoder, and its vectors compared with the model run
    directly: CLS pooling, L2 normalisation, the query instruction prefix, padding-invariance."""
    import torch
    from transformers import BertConfig, BertModel, BertTokenizerFast
    from anrag.encoder import BGE_QUERY_PREFIX, LocalEncoder

    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "represent", "this", "sentence", "for", "searching", "relevant",
             "passages", ":", "asthma", "in", "children", "stroke", "dose", "of", "what", "?", "##s", "child", "with", "a"]
    (tmp_path / "vocab.txt").write_text("\n".join(words) + "\n")
    tok = BertTokenizerFast(vocab_file=str(tmp_path / "vocab.txt"), do_lower_case=True)
    tok.save_pretrained(str(tmp_path))
    torch.manual_seed(3)
    cfg = BertConfig(vocab_size=len(words), hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=128, max_position_embeddings=64)
    BertModel(cfg, add_pooling_layer=False).save_pretrained(str(tmp_path))

    enc = LocalEncoder(model_path=str(tmp_path), device="cpu")
    assert enc.pretrained and enc.dim == 64
    texts = ["What dose for a child with asthma?", "stroke"]
    got = enc.encode(texts)
    model = BertModel.from_pretrained(str(tmp_path), local_files_only=True).eval()
    with torch.no_grad():
        batch = tok(texts, padding=True, truncation=True, max_length=512, return_tensors="pt")
        want = torch.nn.functional.normalize(model(**batch).last_hidden_state[:, 0], dim=1).numpy()
    assert got.shape == (2, 64) and np.max(np.abs(got - want)) < 1e-5
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    q = enc.encode_query("asthma in children")
    with torch.no_grad():
        b1 = tok([BGE_QUERY_PREFIX + "asthma in children"], return_tensors="pt")
        w1 = torch.nn.functional.normalize(model(**b1).last_hidden_state[:, 0], dim=1).numpy()[0]
    assert np.max(np.abs(q - w1)) < 1e-5

"""Local sentence-transformer query/document encoder on PyTorch-ROCm -- what replaces the Voyage API call of
`SearchEngine._generate_query_embedding` (src/search_engine.py:148-159; document side:
src/processing/create_database.py:27-48).

Default architecture: bge-small-en-v1.5 (BERT, 12 layers, hidden 384, 12 heads, FFN 1536, CLS pooling, L2
normalisation, query instruction prefix).  Weights cannot be fetched offline:
  * `LocalEncoder(model_path=...)` loads tokenizer + weights from a LOCAL directory (`local_files_only=True`);
  * without a path the same architecture is instantiated with seeded random weights and a hashed word-level
    tokenizer -- right shapes, right cost, meaningless semantics; good for plumbing, smoke tests and timing.
PyTorch is the engine here (the encoder is a library GEMM workload, not the hand-written hot path).
"""
from __future__ import annotations

import re
import zlib
from typing import List, Optional, Sequence

import numpy as np
import torch

from .config import LOCAL_ENCODER_KEY

BGE_QUERY_PREFIX = "Represent this sentence for searching relevant passages: "
_WORD = re.compile(r"[a-z0-9]+")


class HashTokenizer:
    """Stand-in for a WordPiece vocabulary: lower-case alphanumeric words hashed into the id range."""

    def __init__(self, vocab_size: int = 30522, max_length: int = 512):
        self.vocab_size, self.max_length = vocab_size, max_length
        self.cls, self.sep, self.pad = 101, 102, 0

    def __call__(self, texts: Sequence[str]):
        rows = []
        for t in texts:
            ids = [1000 + zlib.crc32(w.encode()) % (self.vocab_size - 1000) for w in _WORD.findall(t.lower())]
            rows.append([self.cls] + ids[: self.max_length - 2] + [self.sep])
        width = max(len(r) for r in rows)
        input_ids = torch.full((len(rows), width), self.pad, dtype=torch.long)
        mask = torch.zeros((len(rows), width), dtype=torch.long)
        for i, r in enumerate(rows):
            input_ids[i, : len(r)] = torch.tensor(r)
            mask[i, : len(r)] = 1
        return {"input_ids": input_ids, "attention_mask": mask}


class LocalEncoder:
    def __init__(self, model_path: Optional[str] = None, device: Optional[str] = None,
                 dtype: Optional[torch.dtype] = None, max_length: int = 512, seed: int = 0,
                 model_name: str = LOCAL_ENCODER_KEY):
        from transformers import BertConfig, BertModel

        self.model_name = model_name
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.dtype = dtype or (torch.float16 if self.device.type == "cuda" else torch.float32)
        self.max_length = max_length
        if model_path:
            from transformers import AutoModel, AutoTokenizer

            tok = AutoTokenizer.from_pretrained(model_path, local_files_only=True)
            self.tokenizer = lambda texts: tok(list(texts), padding=True, truncation=True, max_length=max_length,
                                               return_tensors="pt")
            self.model = AutoModel.from_pretrained(model_path, local_files_only=True)
            self.pretrained = True
        else:
            cfg = BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=12, num_attention_heads=12,
                             intermediate_size=1536, max_position_embeddings=512)
            g = torch.random.get_rng_state()
            torch.manual_seed(seed)
            self.model = BertModel(cfg, add_pooling_layer=False)
            torch.random.set_rng_state(g)
            self.tokenizer = HashTokenizer(cfg.vocab_size, max_length)
            self.pretrained = False
        self.model.eval().to(self.device, self.dtype)
        self.dim = self.model.config.hidden_size

    @torch.no_grad()
    def encode(self, texts: Sequence[str], batch_size: int = 64) -> np.ndarray:
        """-> [n, dim] float32, CLS-pooled, L2-normalised (what the `chunks.embedding` BLOBs hold)."""
        out: List[np.ndarray] = []
        for lo in range(0, len(texts), batch_size):
            batch = self.tokenizer(texts[lo: lo + batch_size])
            batch = {k: v.to(self.device) for k, v in batch.items() if k in ("input_ids", "attention_mask", "token_type_ids")}
            hidden = self.model(**batch).last_hidden_state[:, 0].float()
            hidden = torch.nn.functional.normalize(hidden, dim=1)
            out.append(hidden.cpu().numpy().astype(np.float32))
        return np.concatenate(out) if out else np.zeros((0, self.dim), np.float32)

    def encode_query(self, text: str) -> np.ndarray:
        return self.encode([BGE_QUERY_PREFIX + text])[0]

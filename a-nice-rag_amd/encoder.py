"""Local sentence-transformer query/document encoder on PyTorch-ROCm -- what replaces the Voyage API call of
`SearchEngine._generate_query_embedding` (src/search_engine.py:148-159; document side:
src/processing/create_database.py:27-48).

Default architecture: bge-small-en-v1.5 (BERT, 12 layers, hidden 384, 12 heads, FFN 1536, CLS pooling, L2
normalisation, query instruction prefix).  Weights cannot be fetched offline:
  * `LocalEncoder(model_path=...)` loads tokenizer + weights from a LOCAL directory (`local_files_only=True`);
  * without a path the same architecture is instantiated with seeded random weights and a hashed word-level
    tokenizer -- right shapes, right cost, meaningless semantics; good for plumbing, smoke tests and timing.
PyTorch is the engine here (the encoder is a library GEMM workload, not the hand-written hot path); the
single-query path replays a captured hipGraph per token-length bucket (3.1 -> 0.87 ms per query on MI355X).
"""
from __future__ import annotations

import re
import threading
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
        self.use_graphs = True          # single-query path: replay a captured graph per token-length bucket
        self._graphs: dict = {}
        self._graph_lock = threading.Lock()

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

    # ------------------------------------------------------------------ single query: hipGraph replay
    # A query is ~20 tokens and the eager forward is ~200 kernel launches (3.3 ms on MI355X, seven times the
    # 1M-row retrieval behind it).  Per token-length bucket the forward is captured ONCE into a graph
    # (torch.cuda.CUDAGraph = hipGraph on ROCm) over static input buffers and replayed per query.
    _BUCKETS = (16, 32, 64, 128, 256, 512)

    def _forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        hidden = self.model(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state[:, 0].float()
        return torch.nn.functional.normalize(hidden, dim=1)

    def _graph_for(self, length: int):
        graphs = self._graphs
        if length in graphs:
            return graphs[length]
        entry = None
        try:
            ids = torch.zeros((1, length), dtype=torch.long, device=self.device)
            mask = torch.zeros((1, length), dtype=torch.long, device=self.device)
            mask[0, 0] = 1
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(3):
                    self._forward(ids, mask)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                out = self._forward(ids, mask)
            entry = (graph, ids, mask, out)
        except Exception:  # capture not possible with this model / build: the eager path stays
            torch.cuda.synchronize(self.device)
            entry = None
        graphs[length] = entry
        return entry

    @torch.no_grad()
    def encode_query(self, text: str) -> np.ndarray:
        batch = self.tokenizer([BGE_QUERY_PREFIX + text])
        ids, mask = batch["input_ids"], batch["attention_mask"]
        n = int(ids.shape[1])
        if self.device.type == "cuda" and n <= self._BUCKETS[-1] and self.use_graphs:
            length = next(b for b in self._BUCKETS if b >= n)
            with self._graph_lock:  # one set of static buffers per bucket: callers on several threads take turns
                entry = self._graph_for(length)
                if entry is not None:
                    graph, s_ids, s_mask, out = entry
                    s_ids.zero_()
                    s_mask.zero_()
                    s_ids[0, :n].copy_(ids[0], non_blocking=True)
                    s_mask[0, :n].copy_(mask[0], non_blocking=True)
                    graph.replay()
                    return out[0].cpu().numpy().astype(np.float32)
        return self.encode([BGE_QUERY_PREFIX + text])[0]

"""Paths and default fusion weights -- mirror of the reference's src/config.py:12-59, plus one
model key for the local encoder that replaces the Voyage API call (search_engine.py:148-159)."""
from dataclasses import dataclass
from enum import Enum
from typing import Optional


class InfoSource(Enum):
    NICE = "nice"


LOCAL_ENCODER_KEY = "bge-small-en-v1.5"


@dataclass
class SourceConfig:
    db_path: str
    bm25_path: str
    context_description: str
    not_found_message: str
    voyage_db_path: Optional[str] = None
    voyage_3_5_db_path: Optional[str] = None
    openai_db_path: Optional[str] = None
    qwen_db_path: Optional[str] = None
    local_db_path: Optional[str] = None  # chunks embedded by the local encoder (LOCAL_ENCODER_KEY)

    def __post_init__(self):
        if self.voyage_db_path is None:
            self.voyage_db_path = self.db_path


class Config:
    # src/config.py:30-36 + the local encoder (weight 0 until a local_db_path is configured)
    DEFAULT_MODEL_WEIGHTS = {
        "voyage-3-large": 5.0,
        "text-embedding-3-large": 0.0,
        "voyage-3.5": 0.0,
        "Qwen3": 0.0,
        LOCAL_ENCODER_KEY: 0.0,
        "BM25": 1.0,
    }

    SOURCE_CONFIGS = {
        InfoSource.NICE: SourceConfig(
            db_path="databases/voyage_3_large_nice_guidelines_2048.db",
            bm25_path="databases/bm25_index_nice_guidelines.pkl",
            context_description="NICE guidelines",
            not_found_message="no relevant NICE guidelines were found",
            voyage_db_path="databases/voyage_3_large_nice_guidelines_2048.db",
            voyage_3_5_db_path="databases/voyage_3.5_nice_guidelines_2048.db",
            openai_db_path="databases/text_embedding_3_large_nice_guidelines.db",
            qwen_db_path="databases/Qwen3-Embedding-0.6B_nice_guidelines.db",
        )
    }

    # (model key in query_embeddings / model_weights, SourceConfig attribute, name passed to the loader):
    # the order the reference walks its dense models in (query_rag_retrieval.py:197, :222, :253, :282)
    DENSE_MODELS = (
        ("voyage-3-large", "voyage_db_path", "voyage-3-large"),
        ("voyage-3.5", "voyage_3_5_db_path", "voyage-3.5"),
        ("text-embedding-3-large", "openai_db_path", "text-embedding-3-large"),
        ("Qwen3", "qwen_db_path", "Qwen3-Embedding-0.6B"),
        (LOCAL_ENCODER_KEY, "local_db_path", LOCAL_ENCODER_KEY),
    )

    @classmethod
    def get_source_config(cls, source: str) -> SourceConfig:
        try:
            return cls.SOURCE_CONFIGS[InfoSource(source.lower())]
        except ValueError:
            raise ValueError(f"Unknown source: {source}. Valid sources: {[s.value for s in InfoSource]}")

"""anrag -- MI355X-native hybrid retriever behind A-NICE-RAG's search API.

Host side (Python, as the reference is Python) of the hot path of
`src/search_engine.py`: `DatabaseManager` / `SearchEngine` /
`RetrievalEvaluationSystem` look-alikes over the C ABI of `libanrag.so`
(include/anrag.h).  There is no CPU fallback in this package: without the HIP
library and a gfx950 device every search raises.
"""
__version__ = "0.1.0"

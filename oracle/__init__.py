"""CPU oracle for the A-NICE-RAG retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / pure-Python restatement of what the reference
(`/root/reference/src/search_engine.py` and friends) computes on the hot path.
It exists so that the HIP kernels in `a-nice-rag_amd/csrc` have something to be
checked against.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product package
(`a-nice-rag_amd/`, importable as `anrag`) never does, and fails loudly when its
HIP library is missing instead of falling back to anything in here.

Pinning status (see DESIGN.md "Oracle"):
  * dense top-k, source filter, weighted RRF, BM25 *selection*, the
    `retrieve_documents` glue and the eval metrics are pinned against outputs of
    the reference's own unmodified code, captured by `oracle/make_golden.py`
    (imports `/root/reference/src` with two stub modules) and committed under
    `tests/golden/`.
  * BM25 *scoring* (`rank_bm25.BM25Okapi`, un-vendored and unpinned in the
    reference's requirements.txt:5, not installed here) is a restatement of the
    published rank-bm25 0.2.2 algorithm: PARITY UNPINNED for that one function.
    It is cross-checked by hand-computed known-answer cases and by an independent
    CSR restatement that must agree bit-for-bit (`tests/test_oracle_bm25.py`).
"""

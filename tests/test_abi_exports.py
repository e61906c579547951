"""CPU: libanrag.so loads, exports every entry point include/anrag.h declares, and refuses to work without a
gfx950 device instead of falling back to anything (no compute calls here: there is no GPU in this container)."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(REPO, "include", "anrag.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(anrag_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    from anrag import _native as nat

    lib = nat.load_library()
    names = declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/anrag.h but not exported"
    assert set(names) == set(nat.EXPORTS), set(names) ^ set(nat.EXPORTS)
    assert lib.anrag_abi_version() == 1


def test_no_gpu_means_loud_failure():
    import torch
    from anrag import _native as nat
    from anrag.index import Index

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = nat.load_library()
    h = C.c_void_p()
    assert lib.anrag_index_create(0, C.byref(h)) == -5  # ANRAG_ERR_NODEVICE
    assert b"no CPU fallback" in lib.anrag_last_error()
    with pytest.raises(nat.AnragError):
        Index(0)
    # the reference-shaped layer does not swallow it either (search_engine.py's catch-all would have)
    import numpy as np
    import pandas as pd
    from anrag.search_engine import SearchEngine

    df = pd.DataFrame({"id": ["a"], "source": ["CG1"], "embedding": [np.ones(4, np.float32)]})
    with pytest.raises(nat.AnragError):
        SearchEngine(None, None).similarity_search_with_embedding(np.ones(4, np.float32), df)
    with pytest.raises(nat.AnragError):
        SearchEngine(None, None).weighted_reciprocal_rank_fusion([(["a"], "m")], {"m": 1.0}, 40)


def test_oracle_is_not_imported_by_the_product():
    pkg = os.path.join(REPO, "a-nice-rag_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"

"""GPU: a short, seeded run of the randomised differential check (scripts/fuzz_parity.py): random corpus shapes
(1 .. 20,000 rows, 16 dimensions incl. the generic-kernel ones, exact duplicate rows), k below / at / above the
fused limit and above N, source filters, duplicate / unknown / empty term lists -- dense within 1e-4 of the
oracle, BM25 ids, scores and score vectors bit-exact, fused results bit-exact."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _fuzzer():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_seeded_fuzz_against_the_oracle():
    assert _fuzzer().run(budget=12.0, seed=20260) > 200


def test_seeded_fuzz_of_bm25_on_large_partitions():
    """262k .. 1.2M documents: the 1,024-thread form of K3 (partitions of 1,280 .. 4,096 documents)."""
    assert _fuzzer().run_big(budget=15.0, seed=20261) >= 10


def test_seeded_fuzz_of_the_batched_path():
    """65k .. 200k rows, 16 .. 300 queries per call, both K2 arithmetic modes, filters, ties, clustered corpora."""
    assert _fuzzer().run_batched(budget=25.0, seed=20262) >= 300


def test_seeded_fuzz_of_the_full_ranking_lists_and_scan_lanes():
    """`anrag_rank_batch` on corpora below / at / past the sort kernel's LDS capacity, ties at every cut, filters, 1-3 legs
    with permuted document maps, against the per-query entry points (ids and score bits) and the reference's fusion; and
    single dense queries on the scan lanes interleaved with group calls and syncs."""
    assert _fuzzer().run_rank(budget=25.0, seed=20263) >= 60

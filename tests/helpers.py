"""Shared comparison helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def assert_ranking_matches(ref_rows, ref_scores, got_rows, got_scores, tol=0.0, full_scores=None, what=""):
    """Rank-order parity modulo equal-score groups.

    * the score SEQUENCES must agree position by position (exactly when tol == 0,
      within `tol` otherwise);
    * inside a run of reference scores that are equal (or closer than 2*tol to
      their neighbour) the reference's order is unspecified (numpy
      argpartition/argsort) -- only the SET of rows must agree;
    * the last run may be cut by k: there the rows need only have the right score
      (checked against `full_scores` when given).
    """
    ref_rows = list(ref_rows)
    got_rows = list(got_rows)
    assert len(ref_rows) == len(got_rows), f"{what}: length {len(got_rows)} != {len(ref_rows)}"
    if not ref_rows:
        return
    rs = np.asarray(ref_scores, dtype=np.float64)
    gs = np.asarray(got_scores, dtype=np.float64)
    if tol == 0.0:
        assert np.array_equal(rs, gs), f"{what}: score sequences differ: {rs[:8]} vs {gs[:8]}"
    else:
        assert np.max(np.abs(rs - gs)) <= tol, f"{what}: max |dscore| {np.max(np.abs(rs - gs))} > {tol}"
    # split into runs of (near-)equal reference scores
    start = 0
    n = len(ref_rows)
    for i in range(1, n + 1):
        if i == n or abs(rs[i] - rs[i - 1]) > 2 * tol:
            a, b = set(ref_rows[start:i]), set(got_rows[start:i])
            if a != b:
                last = i == n
                assert last, f"{what}: rows differ in ranks [{start},{i}): {sorted(a)} vs {sorted(b)}"
                if full_scores is not None:
                    fs = np.asarray(full_scores, dtype=np.float64)
                    # a row the reference left out must tie with the reference's LAST entry (the cut), not with the
                    # start of the run: a run can be a long chain of scores each within 2*tol of its neighbour
                    for r in b - a:
                        assert abs(fs[r] - rs[i - 1]) <= 2 * tol + 1e-30, (
                            f"{what}: row {r} (score {fs[r]}) does not tie with the cut at {rs[i - 1]}")
            start = i
    assert len(set(got_rows)) == len(got_rows), f"{what}: duplicate rows returned"

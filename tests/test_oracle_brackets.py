"""Bounds the one guessed piece of the oracle: the bracketing phase in front of Brent's ``localmin``.

brent-search is absent from this image and its first bracketing step is not recoverable (oracle/brent.py); the
restatement starts at x = 0 (delta = 0.5) with a first step of 1 and growth 2.  Any other bracketing of the same basin
starts ``localmin`` from another triple, the search stops elsewhere inside its tolerance (rtol = atol = 1e-6 on logit
delta; reference call site cellregmap/_cellregmap.py:352) and Q / p move with it.  This test reruns the oracle's
interaction scan with nine other plausible phases (tools/bracket_variants.py: first step 0.5 / 2 / golden ratio / -1 /
tolerance-sized, start -0.5 / +0.5, growth golden ratio / 3) on the end-to-end goldens and on 20 problems of the fuzz
stream and asserts the envelope over all of them per variant scan:

  * rho* never changes (the grid argmax is decided by lml differences far above the search tolerance);
  * Q stays inside the oracle-vs-oracle envelope of tests/test_oracle_spread.py (2e-5), and beyond the north-star 1e-6
    only on a counted share;
  * p stays inside 1e-5 except on a counted share (< 1 %), and inside 5e-5 always.

The 200-problem run of the same code (2 292 variant scans; larger problems) is committed as
profiles/r04_oracle_bracket_variants_200_problems_seed2026.json: Q moves by up to 3.6e-6 (5.8 % of the scans beyond 1e-6
under at least one variant), p by up to 9.5e-6 -- no scan beyond 1e-5 -- and rho* never.
So "p within 1e-5 of the reference" is bounded against the unknown bracketing by measurement: the guess moves the
oracle by what another rounding of its own objective moves it (same envelope), not by more.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def envelope():
    import bracket_variants as bv

    res = bv.run(int(os.environ.get("CRM_BRACKET_PROBLEMS", "20")), 2026, max_cells=200, max_variants=6, max_contexts=20)
    dest = os.environ.get("CRM_BRACKET_JSON")
    if dest:
        with open(dest, "w") as fh:
            json.dump(res, fh, indent=1)
    return res


def test_the_goldens_and_the_fuzz_problems_were_scanned(envelope):
    assert envelope["problems"] >= 20 and envelope["envelope_over_variants"]["variant_scans"] >= 150, envelope["problems"]
    assert len(envelope["variants"]) == 9


def test_rho_star_does_not_depend_on_the_bracketing(envelope):
    for name, v in envelope["variants"].items():
        assert v["rho_star_differs"] == 0 or v["worst_rel_lml_where_rho_differs"] < 1e-11, (name, v)


def test_envelope_of_Q_and_p_over_the_bracketing_variants(envelope):
    e = envelope["envelope_over_variants"]
    assert e["worst_rel_Q"] < 2e-5, e                       # the oracle-vs-oracle envelope (tests/test_oracle_spread.py)
    assert e["worst_rel_p"] < 5e-5, e
    assert e["share_p_beyond_1e-5"] < 0.01, e               # p inside the north-star 1e-5 except on a counted share
    assert e["share_Q_beyond_1e-6"] < 0.25, e               # (measured ~8 %: one stopping tolerance is ~1e-6 on Q)
    for name, v in envelope["variants"].items():
        assert v["worst_rel_Q"] < 2e-5 and v["worst_rel_p"] < 5e-5, (name, v)

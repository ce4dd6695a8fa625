"""MI355X-native engine for CellRegMap's per-variant score-test path.

Exports the names of the reference package (cellregmap/__init__.py:1-20).
"""
from enum import Enum

from ._engine import (
    CellRegMap,
    GenotypePanel,
    candidate_groups,
    detect_groups,
    compute_maf,
    estimate_betas,
    get_L_values,
    lrt_pvalues,
    release_workspaces,
    run_association,
    run_association_fast,
    run_interaction,
    run_interaction_many,
    scan_interaction_many,
    scan_interaction_resumable,
)


class Term(Enum):
    """cellregmap/_types.py:1-8."""

    FIXED = 1
    RANDOM = 2


__version__ = "0.6.0"

__all__ = [
    "__version__",
    "CellRegMap",
    "GenotypePanel",
    "candidate_groups",
    "detect_groups",
    "run_association",
    "run_association_fast",
    "run_interaction",
    "run_interaction_many",
    "scan_interaction_many",
    "scan_interaction_resumable",
    "compute_maf",
    "estimate_betas",
    "get_L_values",
    "lrt_pvalues",
    "release_workspaces",
    "Term",
]

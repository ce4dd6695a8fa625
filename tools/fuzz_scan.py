"""Random problems, device (dense + collapsed) vs oracle, both null-fit procedures: prints the summaries that
tests/test_gpu_fuzz.py asserts on.  GPU only.   python tools/fuzz_scan.py [polished|verbatim|both] [count 150] [seed] [max_variants] [max_cells]
CRM_FUZZ_MANY_CONTEXTS=1: contexts up to 256, contexts + covariates + 2 up to 288, 30 / 70 covariate columns among the choices."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _run  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
seed = int(sys.argv[3]) if len(sys.argv) > 3 else None
limits = {}
if len(sys.argv) > 4:
    limits["max_variants"] = int(sys.argv[4])      # e.g. 1400: panels wide enough for the queue-based null fits
if len(sys.argv) > 5:
    limits["max_cells"] = int(sys.argv[5])
if os.environ.get("CRM_FUZZ_MANY_CONTEXTS"):   # the sizes of the slower kernel forms: up to 256 contexts, 288 Gram rows
    limits.update(max_contexts=256, max_rows=288, extra_covariates=(30, 70))
share = bool(os.environ.get("CRM_FUZZ_SHARE_DECOMPOSITION"))   # the oracle on the device's (Q0, S0): isolates the scan
docs = [{**_run(polish, count=count, seed=seed, share_decomposition=share, **limits)[0], **limits}
        for polish in ([True, False] if which == "both" else [which == "polished"])]
print(json.dumps(docs[0] if len(docs) == 1 else {"procedures": docs}, indent=1), flush=True)     # (one JSON document)

"""Random problems, device (dense + collapsed) vs oracle, both null-fit procedures: prints the summaries that
tests/test_gpu_fuzz.py asserts on.  GPU only.   python tools/fuzz_scan.py [polished|verbatim|both] [count 150] [seed]"""
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
for polish in ([True, False] if which == "both" else [which == "polished"]):
    print(json.dumps(_run(polish, count=count, seed=seed)[0], indent=1), flush=True)

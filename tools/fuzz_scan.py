"""Random problems, device (dense + collapsed) vs oracle, both null-fit procedures: prints the summaries that
tests/test_gpu_fuzz.py asserts on.  GPU only.   python tools/fuzz_scan.py [polished|verbatim]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _run  # noqa: E402

for polish in ([True, False] if len(sys.argv) < 2 else [sys.argv[1] == "polished"]):
    print(json.dumps(_run(polish)[0], indent=1))

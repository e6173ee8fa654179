"""DESIGN section 9, defect 1: are table images uploaded by the copy engine (hipMemcpyAsync + stream synchronise) into
freshly recycled addresses read stale by the first kernel launch?  Same recycling loop as
tests/test_gpu_full_size.py::test_table_recycling_first_launch, once per upload path.
usage: python scratch/upload_mode_experiment.py [seconds]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_full_size as T  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
for mode in (None, "memcpy"):
    out = T._run_recycling_subprocess(seconds, 777, upload_mode=mode)
    print(json.dumps({"upload": mode or "copy kernel (production)", "trials": out["trials"], "failures": len(out["failures"]),
                      "first_failures": out["failures"][:8]}), flush=True)

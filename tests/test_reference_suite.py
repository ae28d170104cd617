"""The reference's OWN test files, unmodified, against this package (build
container only: they are read from /root/reference and never copied).

``tests/refshim`` makes ``import triflow`` resolve to triflow_amd (every
``Model`` compiled by the HIP plugin, run through the host emulation here) and
provides the absent third-party ``path`` helpers the tests use.  In scope are
the files that exercise the hot path and its callers: test_model.py,
test_routines.py, test_simulation.py, test_fields.py (test_containers.py needs
xarray, test_displays.py a plotting stack: out of scope)."""
import os
import re
import subprocess
import sys

import pytest

REF_TESTS = "/root/reference/tests"
FILES = ["test_model.py", "test_routines.py", "test_simulation.py", "test_fields.py"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_TESTS, FILES[0])),
                                reason="reference tree not present")


def test_reference_tests_pass_against_this_package(tmp_path):
    env = dict(os.environ,
               PYTHONDONTWRITEBYTECODE="1",          # the reference tree is read-only territory
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "refshim"), ROOT]))
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
           *[os.path.join(REF_TESTS, f) for f in FILES]]
    res = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    tail = res.stdout[-3000:]
    summary = re.search(r"(\d+) passed", res.stdout)
    assert res.returncode == 0 and " failed" not in tail, tail
    assert summary and int(summary.group(1)) >= 120, tail

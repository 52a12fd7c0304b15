"""Run the reference's OWN unit tests for the hot path under the NumPy-backed stubs
(build container only).  This is the check that the stub executes the reference faithfully
before its outputs are trusted as golden vectors."""
import sys

import pytest

from ref_import import REFERENCE_ROOT, load_reference

if __name__ == "__main__":
    load_reference()
    tests = [
        f"{REFERENCE_ROOT}/tests/interpolation_test.py",
        f"{REFERENCE_ROOT}/tests/distributions_test.py",
        f"{REFERENCE_ROOT}/tests/models/bsplines/smoothing_test.py",
    ]
    sys.exit(pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir=/tmp", "-o", "python_files=*_test.py", *tests]))

"""The suite checks itself: no test file shadows one of its own tests (tests/conftest.py fails collection on it)."""
import glob
import os

import conftest


def test_no_test_file_defines_a_test_name_twice():
    here = os.path.dirname(os.path.abspath(__file__))
    dup = [d for f in sorted(glob.glob(os.path.join(here, "test_*.py"))) for d in conftest._duplicate_test_names(f)]
    assert dup == []


def test_the_guard_sees_a_shadowed_test(tmp_path):
    f = tmp_path / "test_x.py"
    f.write_text("def test_a():\n    pass\n\n\ndef helper():\n    pass\n\n\ndef test_a():\n    pass\n")
    assert conftest._duplicate_test_names(str(f)) == ["test_x.py:9 test_a"]

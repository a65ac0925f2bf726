"""Run pytest in this process with one of the library's diagnostic switches set first:
python scripts/run_test_with_option.py conv_mid 0 tests/test_gpu_api2.py -k accuracy -x -q -s"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402

L.check(L.lib().dvae_set_option(sys.argv[1].encode(), int(sys.argv[2])))
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[3:]))

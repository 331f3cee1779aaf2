#!/usr/bin/env python3
"""The head's golden-fixture parity tests with the forward GEMMs on the 3-product kernel (ops.linear.set_forward_precision("x3")) instead
of the exact-fp32 MFMA: which of tests/test_head_gpu.py still pass, and the bench step."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, pytest; sys.path.insert(0, %r); import combo_avs_amd; from combo_avs_amd.ops import linear as L; "
        "L.set_forward_precision('x3'); L.FORWARD_PRECISION_LOCK = True; "
        "sys.exit(pytest.main(['-q', '-m', 'gpu', '-x', '--no-header', '-p', 'no:cacheprovider', %r] + sys.argv[1:]))")
sys.exit(subprocess.call([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests", "test_head_gpu.py"))] + sys.argv[1:], cwd=ROOT))

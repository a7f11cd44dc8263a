"""A short slice of the randomised differential checkers of tests/fuzz/ inside the GPU suite (each in its own process, one after the
other; the full runs are recorded in DESIGN.md §2)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('script,args', [('conv_fuzz.py', ['7', '40']), ('wgrad_fuzz.py', ['7', '30']), ('net_fuzz.py', ['7', '5']),
                                         ('disc_fuzz.py', ['7', '4']), ('block_fuzz.py', ['7', '16'])])
def test_fuzz_slice(cuda, script, args):
    r = subprocess.run([sys.executable, os.path.join(HERE, 'fuzz', script)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert 'mismatches: 0' in r.stdout

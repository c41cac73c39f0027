"""GPU: race / state-dependence screen (scripts/soak.py in the suite).  Many generate() calls of mixed batch sizes and lengths on ONE
engine -- the pooled decode state is re-allocated and reused with different cache capacities, the hipGraph is re-captured, workspaces
change size -- and every repeat of a request must return bit-identical ids.  (Round 3: this caught the first MFMA decode attention
splitting its keys by cache capacity instead of by sequence length.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fmt,calls", [("native", 120), ("fp8", 60)])
def test_repeated_requests_are_bit_identical_whatever_ran_in_between(fmt, calls):
    env = dict(os.environ)
    env.pop("SRGPT_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak.py"), fmt, str(calls)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "soak ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])

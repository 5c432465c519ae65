"""CPU: the timed CPU baseline (oracle/cpu_baseline.c, bench.py's cpu_baseline leg) computes what the oracle computes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpu_baseline_matches_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(HERE, "cpu_baseline_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cpu_baseline OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

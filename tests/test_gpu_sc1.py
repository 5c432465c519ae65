"""GPU: the hand-written streaming store (`global_store_dwordx4 … sc1 nt` + a hand-counted `s_nop` for the store-data
hazard, csrc/common.hpp st_vec_sc1) at EVERY call site, bit for bit against (a) the oracle and (b) a second build of the
library with the switch back to `__builtin_nontemporal_store` (make nosc1 → AGPU_USE_SC1=0).  A compiler bump that changes
the schedule around the inline asm shows up here as a per-site hash mismatch, not as a silently corrupted column
(VERDICT r2 weak #8)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "tools", "sc1_sites.py")
NOSC1 = os.path.join(ROOT, "arrow_gpu_amd", "lib", "libarrow_gpu_hip_nosc1.so")


def run(env_extra, *args):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, SCRIPT, *args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_every_sc1_call_site_matches_the_oracle_and_the_build_without_it():
    assert os.path.exists(NOSC1), "make -C arrow_gpu_amd/csrc nosc1 (built by __graft_entry__.build())"
    with_sc1 = run({}, "--oracle")
    assert with_sc1["bad"] == [], with_sc1["bad"][:10]
    assert len(with_sc1["sites"]) == 3 * (12 + 4 + 2 + 8 + 13 + 12 + 5 + 6)
    without = run({"AGPU_LIB": NOSC1})
    assert without["lib"] == NOSC1 and without["bad"] == []
    diff = [k for k in with_sc1["sites"] if with_sc1["sites"][k] != without["sites"][k]]
    assert diff == [], diff[:10]

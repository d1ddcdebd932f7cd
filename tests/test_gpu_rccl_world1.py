"""RCCL on the real device at world size 1 (the only part of the N > 1 path a one-GPU box can run): bench.py's own gather_step /
max_over_ranks / fence on the "nccl" backend, sharing one non-null stream with the library's launches, in a fresh child process so
that the process group is created before anything else touches the GPU.  Catches stream-ordering bugs between the solver's kernels,
the *_dev getters and the collectives.  No scaling is measured by this."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def test_gather_over_rccl_on_the_shared_stream():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "rccl_world1.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out
    assert out["max_over_ranks"] == 0.25 and out["finite_frac"] > 0.3

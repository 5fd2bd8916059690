"""GPU: a short, seeded run of every fuzzer under tools/ (random sizes / contents / parameters against the oracle).  The long runs
quoted in DESIGN.md are made by hand; these few seconds per tool keep the fuzzers themselves working and add a few hundred random
cases to every run of the suite."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("tool,ok", [("fuzz_orb.py", "all bit-exact"), ("fuzz_orb_batch.py", "all bit-exact"), ("fuzz_match.py", "all bit-exact"),
                                     ("fuzz_match_batch.py", "all bit-exact"), ("fuzz_track_batch.py", "all bit-exact"),
                                     ("fuzz_ba_pose.py", "all within tolerance")])
def test_fuzzer_short_run(tool, ok):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / tool), "--seconds", "5", "--seed", "20260928"], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=600)
    assert r.returncode == 0 and ok in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])

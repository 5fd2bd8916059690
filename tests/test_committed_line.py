"""CPU: the bench line committed as the round's latest (profiles/LATEST_DEFAULT_LINE names it) is a complete contract line whose
three legs were verified against the oracle when it was measured -- so the numbers quoted in README / DESIGN point at evidence
that carries its own check."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_latest_committed_default_line_is_complete_and_verified():
    name = (ROOT / "profiles" / "LATEST_DEFAULT_LINE").read_text().strip()
    d = json.loads((ROOT / "profiles" / name).read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] is not None
    assert "valu" in r and 0.0 < r["valu"]["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["identical_to_gpu"] is True and c["frames_checked"] >= 256
    assert c["ba"]["identical_to_gpu"] is True and len(c["ba"]["windows_checked"]) == 3
    assert d["tracking"]["cpu_baseline"]["identical_to_gpu"] is True
    ba = d["ba"]
    assert ba["roofline"]["frac"] >= 0.35 and ba["value"] >= 4.95e5      # the round-2 review's bar for the BA leg
    assert d["value"] >= 1.0e4                                            # north_star: >= 10 000 frames/s
    assert abs(d["ms_per_step"] * d["steps"] / 1e3 - d["timed_region_s"]) < 0.02 * d["timed_region_s"] + 1e-3

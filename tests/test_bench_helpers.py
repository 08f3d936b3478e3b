"""bench.py's host-side helpers (no GPU): which PMC file `roofline.traffic` is taken from, and when it is withheld."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("cimg_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_traffic_comes_from_the_newest_round_whose_sources_match(tmp_path, monkeypatch):
    b = _bench()
    h = b.kernel_source_hash()
    assert len(h) == 16 and h == b.kernel_source_hash()
    prof = tmp_path / "profiles"
    for rnd, sh, fetch in (("r03", h, 100), ("r07", "0" * 16, 200), ("r05", h, 300)):
        (prof / rnd).mkdir(parents=True)
        (prof / rnd / "pmc_per_launch.json").write_text(json.dumps(
            {"_source_hash": sh, "cimg_encode_streams": {"FETCH_SIZE": fetch, "WRITE_SIZE": 50}}))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    monkeypatch.setattr(b, "kernel_source_hash", lambda: h)
    # the newest directory (r07) is from other sources: nothing is reported, and the reason is
    t, why = b.pmc_traffic("cimg_encode_streams")
    assert t is None and "other kernel sources" in why and "r07" in why
    (prof / "r07" / "pmc_per_launch.json").unlink()
    t, why = b.pmc_traffic("cimg_encode_streams")
    assert t == (300 * b.FETCH_FACTOR + 50) * 1024 and "r05" in why        # FETCH_SIZE doubled (the gfx950 rule), KiB -> bytes
    # an explicit file is taken as given
    t, _ = b.pmc_traffic("cimg_encode_streams", str(prof / "r03" / "pmc_per_launch.json"))
    assert t == (100 * b.FETCH_FACTOR + 50) * 1024


def test_committed_counters_name_their_sources():
    """profiles/rNN/pmc_per_launch.json carries the hash of the kernel sources it was collected on (bench.py reports the traffic
    only on a match -- a mismatch is a null in the line, never a stale number)."""
    prof = os.path.join(ROOT, "profiles")
    rounds = sorted(r for r in os.listdir(prof) if r[:1] == "r" and r[1:].isdigit())
    newest = os.path.join(prof, rounds[-1], "pmc_per_launch.json")
    with open(newest) as f:
        table = json.load(f)
    assert isinstance(table.get("_source_hash"), str) and len(table["_source_hash"]) >= 16
    assert "FETCH_SIZE" in table["cimg_encode_streams"] and "WRITE_SIZE" in table["cimg_decode_lean"]

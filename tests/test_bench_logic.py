"""CPU: the part of bench.py's N > 1 line that decides which transport's rate becomes `value`: an IPC attempt -- hand-made
coherence that has never met two devices -- counts only if its reduced totals are those of a transport whose exchange is a library call."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("dflo_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _att(transport, ok=True, totals=None, value=1.0, check="state finite and admissible"):
    base = [1.0, 2.0, 300.0, 700.0, 1.0, 2.0, 300.0, 700.0, -0.5, -0.4]
    return {"transport": transport, "ok": ok, "check": check, "value": value, "totals": list(base if totals is None else totals), "sec": 1.0}


def test_an_ipc_attempt_whose_totals_are_the_library_transports_counts():
    atts = [_att("rccl", value=10.0), _att("ipc_gloo", value=12.0), _att("ipc", value=13.0)]
    bench.cross_validate(atts, "c4")
    assert all(a["ok"] for a in atts)
    assert "equal the rccl run's" in atts[1]["validated"] and "equal the rccl run's" in atts[2]["validated"] and "validated" not in atts[0]
    assert max((a for a in atts if a["ok"]), key=lambda a: a["value"])["transport"] == "ipc"


def test_an_ipc_attempt_that_read_stale_halos_cannot_win_where_its_own_check_is_weak():
    t = _att("rccl")["totals"]
    off = list(t)
    off[6] *= 1.0 + 1e-9           # the density total after the run differs in the 9th digit: still "finite and admissible"
    atts = [_att("rccl", value=10.0), _att("ipc_gloo", value=20.0, totals=off)]
    bench.cross_validate(atts, "c3")
    assert atts[0]["ok"] and not atts[1]["ok"] and "NOT COUNTED" in atts[1]["check"] and "differ from the rccl run's" in atts[1]["check"]
    assert [a["transport"] for a in atts if a["ok"]] == ["rccl"]


def test_the_host_staged_run_is_the_reference_where_rccl_gave_no_line():
    atts = [_att("rccl", ok=False), _att("gloo", value=3.0), _att("ipc_gloo", value=9.0)]
    bench.cross_validate(atts, "c5")
    assert atts[2]["ok"] and "equal the gloo run's" in atts[2]["validated"]
    nan = _att("ipc_gloo", value=9.0)
    nan["totals"][5] = float("nan")
    atts = [_att("gloo", value=3.0), nan]
    bench.cross_validate(atts, "c2")
    assert not atts[1]["ok"]


def test_without_a_reference_only_the_periodic_boxs_own_conservation_check_lets_an_ipc_attempt_count():
    for config, kept in (("c2", True), ("c3", False), ("c4", False), ("c5", False)):
        atts = [_att("rccl", ok=False), _att("ipc_gloo", value=9.0)]
        bench.cross_validate(atts, config)
        assert atts[1]["ok"] == kept, config
        if kept:
            assert "own conservation check only" in atts[1]["validated"]
        else:
            assert "no reference transport" in atts[1]["check"]
    # an attempt that failed before it ran is left alone
    atts = [_att("rccl"), {"transport": "ipc_gloo", "ok": False, "check": "failed: could not be made", "value": 0.0}]
    bench.cross_validate(atts, "c4")
    assert atts[1]["check"] == "failed: could not be made"

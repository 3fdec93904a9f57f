"""Short runs of the randomised parity campaigns (tests/campaigns/*.py), in this process: every
kernel family against the oracle on a few dozen random batches per run."""
import importlib
import os
import sys

import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "campaigns")


@pytest.mark.parametrize("name,cases,extra", [("fuzz_parity", 150, []),
                                              ("fuzz_parity", 150, ["--extra", "0.1", "--agents", "256"]),
                                              ("fuzz_nominal", 60, []),
                                              ("fuzz_rollout", 60, []), ("fuzz_rollout", 60, ["--dtype", "f32"]),
                                              ("fuzz_ingest", 40, [])])
def test_campaign(oracle, monkeypatch, name, cases, extra):
    monkeypatch.syspath_prepend(HERE)
    monkeypatch.setattr(sys, "argv", [name, "--cases", str(cases), "--seed", "4242"] + extra)
    mod = importlib.import_module(name)
    with pytest.raises(SystemExit) as done:
        mod.main()
    assert done.value.code == 0

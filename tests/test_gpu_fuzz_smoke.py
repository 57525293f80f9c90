"""GPU: the randomised checkers under bench/ (run at thousands of cases outside the suite, profiles/r03_fuzz.txt) with a few dozen
cases each, so that they keep running against the library as it changes.  They compare the C-ABI entry points with fp64 torch /
numpy arithmetic on the same inputs (no oracle import: bench scripts never touch oracle/).  tests/lars_fuzz.py is the one checker that
holds the kernels against the oracle's restatement of lsa.py:90-212, which is why it lives here."""
import os
import runpy
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SCRIPTS = [("gram_fuzz.py", 40), ("pass_fuzz.py", 60), ("fit_fuzz.py", 40), ("linear_fuzz.py", 25), ("onehot_fuzz.py", 40),
           ("eval_fuzz.py", 40), ("reduce_fuzz.py", 40), ("frame_fuzz.py", 30), ("dummy_frame_fuzz.py", 30),
           ("lockstep_fuzz.py", 40), ("../tests/lars_fuzz.py", 30),
           # widths 400 .. 1020 (lars_c.hip by default, lars_q.hip forced, lars.hip) and 1021 .. 2000 (lars_c.hip at two workgroup counts, lars.hip)
           ("../tests/lars_fuzz.py", 16, "1"), ("../tests/lars_fuzz.py", 10, "2")]


@pytest.mark.parametrize("script,cases,extra", [(s[0], s[1], s[2:]) for s in SCRIPTS])
def test_bench_fuzzers_run_clean(script, cases, extra, monkeypatch, capsys):
    assert torch.cuda.is_available()
    monkeypatch.setattr(sys, "argv", [os.path.basename(script), str(cases), "20261003"] + list(extra))
    try:
        runpy.run_path(os.path.normpath(os.path.join(ROOT, "bench", script)), run_name="__main__")
    except SystemExit as e:                      # the scripts exit non-zero on a mismatch
        assert not e.code, capsys.readouterr().out[-2000:]
    out = capsys.readouterr().out
    assert " ok:" in out, out[-2000:]

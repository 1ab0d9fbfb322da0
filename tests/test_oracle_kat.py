"""The oracle is pinned by the reference's own unit-test vectors (oracle/kat.cpp restates them).
Each KAT case becomes one pytest case."""
import os
import subprocess

import pytest

import oracle_binding


def _run_kat():
    if not os.path.exists(oracle_binding.KAT):
        oracle_binding.build()
    p = subprocess.run([oracle_binding.KAT], capture_output=True, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith(("PASS ", "FAIL "))]
    return p.returncode, lines, p.stdout


_RC, _LINES, _OUT = _run_kat()


def test_kat_binary_exit_status():
    assert _RC == 0, _OUT
    assert len(_LINES) >= 45


@pytest.mark.parametrize("line", _LINES, ids=[ln.split(" ", 2)[1] for ln in _LINES])
def test_kat_case(line):
    assert line.startswith("PASS "), line

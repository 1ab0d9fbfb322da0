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


def test_kat_under_address_and_undefined_behaviour_sanitizers():
    """GPU sanitizers are unavailable on this pool, so the CPU restatement runs under
    -fsanitize=address,undefined (oracle/Makefile target _build/kat_asan)."""
    r = subprocess.run(["make", "-s", "-C", oracle_binding.ORACLE_DIR, "_build/kat_asan"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout):
        pytest.skip("sanitizer runtime not available in this toolchain")
    assert r.returncode == 0, r.stderr
    p = subprocess.run([os.path.join(oracle_binding.ORACLE_DIR, "_build", "kat_asan")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "fail=0" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr


@pytest.mark.parametrize("line", _LINES, ids=[ln.split(" ", 2)[1] for ln in _LINES])
def test_kat_case(line):
    assert line.startswith("PASS "), line

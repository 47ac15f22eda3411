"""Documentation that cannot drift: every C++ snippet of INTEGRATION.md is the text of a marked region of
tests/cpp/integration_snippets.cpp, and that file compiles against include/ (g++ -fsyntax-only; Ceres and MPI are
interface doubles, declarations only); names the documents cite must exist."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(t):
    return "\n".join(l.rstrip() for l in t.strip().splitlines())


def test_integration_snippets_are_the_compiled_ones():
    src = open(os.path.join(ROOT, "tests", "cpp", "integration_snippets.cpp")).read()
    snips = {_norm(b) for _, b in re.findall(r"// \[snippet:(\w+)\]\n(.*?)// \[/snippet\]", src, re.S)}
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", md, re.S)
    assert len(blocks) >= 4
    for b in blocks:
        assert _norm(b) in snips, "INTEGRATION.md has a C++ snippet that is not a marked region of integration_snippets.cpp:\n" + b


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_integration_snippets_compile():
    inc = [os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "ceres_double"),
           os.path.join(ROOT, "tests", "cpp", "mpi_double")]
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + [x for i in inc for x in ("-I", i)] + [
        os.path.join(ROOT, "tests", "cpp", "integration_snippets.cpp")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_cited_files_and_symbols_exist():
    """Files the headers / documents point a reader to (paths of THIS repository; include/Sim3BA.h etc. are the reference's)."""
    hdr = open(os.path.join(ROOT, "include", "bodyfit_ceres.h")).read()
    for path in re.findall(r"(tests/[\w/\.]+\.(?:py|cpp|h))", hdr):
        assert os.path.exists(os.path.join(ROOT, path)), path
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(re.findall(r"`((?:tests|tools|oracle|profiles|3dbodyanimation_amd|include/bodyfit)[\w/\.\-]*\.(?:py|cpp|h|hpp|hip|c|md|json|csv|sh))`", text)):
            assert os.path.exists(os.path.join(ROOT, path)), f"{doc} cites {path}"
    api_h = open(os.path.join(ROOT, "include", "bodyfit.h")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for sym in set(re.findall(r"\b(bodyfit_[a-z_0-9]+)\b", md)):
        if sym in ("bodyfit_ceres",):
            continue
        assert re.search(r"\b" + sym + r"\b", api_h) or sym.endswith("_h"), f"INTEGRATION.md names {sym}, not in bodyfit.h"

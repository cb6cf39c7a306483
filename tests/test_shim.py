"""The R `.Call` shim (shim/init_shim.cpp): R is not in this image, so the shim is type-checked against a declarations-only R
API header (tests/r_api_decl) and its interface is compared with the reference's, recorded in
tests/golden/reference_call_interface.json by tools/make_call_table_fixture.py (reference src/init.cpp:1215-1229,
src/stan_sampler.cpp:67-96)."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "init_shim.cpp")
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_call_interface.json")))


def test_shim_type_checks_against_the_r_api_declarations():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tests", "r_api_decl"), SHIM], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


def test_routine_table_matches_the_reference():
    src = open(SHIM).read()
    table = [[m.group(1), int(m.group(3))] for m in re.finditer(r'S4B_DEF\("(\w+)",\s*(\w+),\s*(\d+)\)', src)]
    assert table == REF["routines"] and len(table) == 12
    # every registered function is defined with exactly that many SEXP parameters
    for name, fn, nargs in re.findall(r'S4B_DEF\("(\w+)",\s*(\w+),\s*(\d+)\)', src):
        m = re.search(r"static SEXP %s\(([^)]*)\)" % fn, src)
        assert m, fn
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == int(nargs) and all(p.strip().startswith("SEXP") for p in params), (fn, params)


def test_argument_lists_cover_the_reference_names():
    src = open(SHIM).read()
    block = src[src.index("static const char* const names[44]"):]
    block = block[block.index("{") + 1:block.index("};")]
    assert re.findall(r'"(\w+)"', block) == REF["dataNames"] and len(REF["dataNames"]) == 44
    unpack = src[src.index("s4b_stan_control unpack_stan_control"):src.index("void unpack_bart")]
    assert sorted(re.findall(r'_element\(c, "(\w+)"', unpack) + []) == sorted(REF["controlNames"] + ["hmc_mode"]) and len(REF["controlNames"]) == 13
    create = src[src.index("static SEXP createSampler"):src.index("static SEXP run(")]
    used = set(re.findall(r'_element\(commonControlExpr, "(\w+)"', create))
    assert set(REF["commonControl"]) <= used, set(REF["commonControl"]) - used

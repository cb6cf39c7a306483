#!/usr/bin/env python3
"""Writes tests/golden/reference_call_interface.json: the names and arities of the reference's registered `.Call` routines
(reference src/init.cpp:1215-1229), the names of its stanData list (src/stan_sampler.cpp:67-80) and of its stanControl list
(src/stan_sampler.cpp:82-96), read from the reference sources in the build container.  The fixture is data (identifiers and
integers; the element names of the bart result list, src/bart_util.cpp:70-76), so that the shim can be checked against the reference's interface where the reference tree is absent."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
init = open("/root/reference/src/init.cpp").read()
stan = open("/root/reference/src/stan_sampler.cpp").read()
routines = [[m.group(1), int(m.group(2))] for m in re.finditer(r'DEF_FUNC\("(\w+)",\s*\w+,\s*(\d+)\)', init)]


def names(block_name):
    body = stan[stan.index("const char* const %s[]" % block_name):]
    body = body[body.index("{") + 1:body.index("};")]
    return re.findall(r'"(\w+)"', body)


common = sorted(set(re.findall(r'rc_getListElement\(commonControlExpr,\s*"(\w+)"\)', init)))
# names of the bart result list (reference src/bart_util.cpp:70-76): four elements, a fifth ("k") when k is a modeled parameter
util = open("/root/reference/src/bart_util.cpp").read()
res = re.findall(r'SET_STRING_ELT\(namesExpr,\s*(\d+),\s*Rf_mkChar\("(\w+)"\)\)', util)
bart_names = [nm for _, nm in sorted((int(i), nm) for i, nm in res)]
assert "results.kSamples == NULL ? 4 : 5" in util
out = {"routines": routines, "dataNames": names("dataNames"), "controlNames": names("controlNames"), "commonControl": common,
       "bartResultNames": bart_names[:4], "bartResultNamesWithModeledK": bart_names}
path = os.path.join(ROOT, "tests", "golden", "reference_call_interface.json")
json.dump(out, open(path, "w"), indent=1)
print(path, len(routines), "routines,", len(out["dataNames"]), "data names,", len(out["controlNames"]), "control names,", len(common), "common control fields")

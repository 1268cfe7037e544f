#!/usr/bin/env python3
"""Generates tests/golden/native_surface.json: the reference's Level-2 native surface as DATA -- for every pybind11
module the exported function names (`m.def("name", ...)` in <pkg>/src/bindings.cpp) and, from the declaration in <pkg>/src/<pkg>.h,
the kind of each argument in order ("tensor" for at::Tensor, else the C type).  Read as text in the authoring container only;
tests/test_native_modules.py compares the shims built from nerf-navigation_amd/bindings/ against it."""
import sys

sys.dont_write_bytecode = True      # nothing is written under /root/reference (no __pycache__ beside the files this script reads or imports)
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = {"_raymarching": ("raymarching", "raymarching.h"), "_gridencoder": ("gridencoder", "gridencoder.h"),
           "_shencoder": ("shencoder", "shencoder.h"), "_ffmlp": ("ffmlp", "ffmlp.h"), "_freqencoder": ("freqencoder", "freqencoder.h")}


def main():
    out = {}
    for mod, (pkg, header) in MODULES.items():
        bind = open(os.path.join(REF, pkg, "src", "bindings.cpp")).read()
        head = re.sub(r"//[^\n]*", "", open(os.path.join(REF, pkg, "src", header)).read())
        names = re.findall(r'm\.def\("(\w+)"\s*,\s*&(\w+)', bind)
        funcs = {}
        for exported, symbol in names:
            m = re.search(r"void\s+" + symbol + r"\s*\(([^)]*)\)\s*;", head)
            assert m, (mod, symbol)
            kinds = []
            for arg in [a.strip() for a in m.group(1).split(",") if a.strip()]:
                typ = re.sub(r"\bconst\b", "", arg).split()
                typ = " ".join(typ[:-1]) if len(typ) > 1 else typ[0]
                kinds.append("tensor" if "Tensor" in typ else typ.strip())
            funcs[exported] = kinds
        out[mod] = funcs
    with open(os.path.join(HERE, "native_surface.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Flag tables of the reference's three get_args() (run_stage1.py:53-247, run_stage2.py:54-324, run_stage3.py:62-309), read from the
SOURCE TEXT with `ast` (the scripts themselves cannot be imported: wandb / decord / timm.optim) -> tests/golden/cli_flags.json.

TEST INFRASTRUCTURE; runs only where /root/reference exists.  What is stored is the interface data of each parser -- per flag:
option strings, dest, default, type name, action, nargs, choices -- plus the set_defaults() calls, i.e. the inputs a drop-in CLI has
to accept and the values it has to produce with no flags given.  tests/test_cli.py checks unite_amd/cli.py against it."""
import ast
import json
import os
import sys

REF = os.environ.get("UNITE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cli_flags.json")


def lit(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return {"expr": ast.unparse(node)}


def extract(path):
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "get_args")
    flags, defaults = [], {}
    for call in [n for n in ast.walk(fn) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)]:
        if call.func.attr == "add_argument":
            opts = [lit(a) for a in call.args]
            kw = {k.arg: (ast.unparse(k.value) if k.arg == "type" else lit(k.value)) for k in call.keywords if k.arg not in ("help", "metavar")}
            flags.append({"opts": opts, **kw, "line": call.lineno})
        elif call.func.attr == "set_defaults":
            for k in call.keywords:
                defaults[k.arg] = lit(k.value)
    flags.sort(key=lambda f: f["line"])
    return {"flags": flags, "set_defaults": defaults}


def main():
    out = {s: extract(os.path.join(REF, f"run_{s}.py")) for s in ("stage1", "stage2", "stage3")}
    json.dump(out, open(OUT, "w"), indent=0, sort_keys=True)
    for s, d in out.items():
        print(s, len(d["flags"]), "flags", len(d["set_defaults"]), "set_defaults")


if __name__ == "__main__":
    main()

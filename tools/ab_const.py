#!/usr/bin/env python3
"""A/B of a module constant inside the bench step (the COMBO_* environment switches of round 3 are gone):
    python tools/ab_const.py combo_avs_amd.ops.convwrw.DX_OWN=3 [more assignments] -- [bench.py arguments]
sets the constants after importing the package, then runs bench.main() in this process.  `call:<c_abi_setter>=<int>` calls an
int-argument setter of the C ABI instead (e.g. call:combo_gemm_tn_tile256=0)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combo_avs_amd  # noqa: E402,F401

args = sys.argv[1:]
rest = []
if "--" in args:
    i = args.index("--")
    args, rest = args[:i], args[i + 1:]
for a in args:
    path, val = a.split("=", 1)
    if path.startswith("call:"):  # call:combo_gemm_tn_tile256=0 -> an int-argument setter of the C ABI (host-side kernel state)
        from combo_avs_amd import _lib
        prev = getattr(_lib.lib(), path[5:])(int(val))
        print(f"[ab_const] {path[5:]}({val}) (was {prev})", file=sys.stderr)
        continue
    mod, name = path.rsplit(".", 1)
    try:
        m = importlib.import_module(mod)
    except ModuleNotFoundError:  # module.Class.attribute
        mod2, cls = mod.rsplit(".", 1)
        m = getattr(importlib.import_module(mod2), cls)
    old = getattr(m, name)
    setattr(m, name, type(old)(eval(val)) if not isinstance(old, bool) else val in ("1", "True", "true"))
    print(f"[ab_const] {path}: {old!r} -> {getattr(m, name)!r}", file=sys.stderr)
import bench  # noqa: E402
sys.argv = ["bench.py"] + rest
bench.main()

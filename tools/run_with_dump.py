#!/usr/bin/env python3
"""python tools/run_with_dump.py <seconds> [module constant assignments ...] -- <bench.py arguments>: bench.main() with a watchdog that
dumps every thread's Python stack to stderr after <seconds> and exits (where does a run hang?)."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
secs = int(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = [os.path.join(ROOT, "tools", "ab_const.py")] + sys.argv[2:]
exec(compile(open(sys.argv[0]).read(), sys.argv[0], "exec"))

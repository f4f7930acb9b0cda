#!/usr/bin/env python3
"""scripts/fuzz_one.py <seed>: one scene of scripts/fuzz_parity.py with a traceback dump every 60 s (where does it spend its time)."""
import faulthandler, os, runpy, sys
faulthandler.dump_traceback_later(60, repeat=True)
seed = sys.argv[1]
sys.argv = ["fuzz_parity.py", "1", seed]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"), run_name="__main__")

#!/usr/bin/env python3
"""Run a script with a stack-dumping watchdog: after SECONDS without finishing, every Python thread's stack goes to stderr
and the process exits (faulthandler) -- for runs that stop making progress on the GPU box.
    python tools/run_with_watchdog.py SECONDS script.py [args ...]"""
import faulthandler
import runpy
import sys

faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

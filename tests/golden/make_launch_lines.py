#!/usr/bin/env python3
"""Writes tests/golden/reference_launch_lines.json: the argument vectors of the reference's nine `scripts/*/save_videos*.sh`
launch lines (everything after `helpers/generator.py`, ${GPU_IDS} -> 0).  Data for tests/test_reference_scripts_gpu.py, which
runs on a box without the reference tree.   python tests/golden/make_launch_lines.py"""
import glob
import json
import os
import re
import shlex

REF = os.environ.get("CCVS_REFERENCE_ROOT", "/root/reference")
out = {}
for path in sorted(glob.glob(os.path.join(REF, "scripts", "*", "save_videos*.sh"))):
    body = open(path).read().split("helpers/generator.py", 1)[1].replace("\\\n", " ")
    out["/".join(path.split(os.sep)[-2:])] = shlex.split(re.sub(r"\$\{GPU_IDS\}", "0", body))
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_launch_lines.json")
json.dump(out, open(dst, "w"), indent=0)
print(len(out), "launch lines ->", dst)

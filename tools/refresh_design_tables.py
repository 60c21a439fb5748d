#!/usr/bin/env python
"""Rewrites the generated block(s) of DESIGN.md from the built objects: the kernel-resources table of section 4
(tools/kernel_resources.py --design).  tests/test_host_mirror.py::test_design_kernel_figures_match_the_built_objects fails when
DESIGN.md and the built objects disagree; run this after a kernel change."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

path = os.path.join(ROOT, "DESIGN.md")
doc = open(path).read()
new = "<!-- kernel-resources:begin -->\n" + kr.design_table().strip() + "\n<!-- kernel-resources:end -->"
doc2, n = re.subn(r"<!-- kernel-resources:begin -->\n.*?\n<!-- kernel-resources:end -->", lambda m: new, doc, flags=re.S)
assert n == 1, "DESIGN.md has no kernel-resources block"
open(path, "w").write(doc2)
print("DESIGN.md: kernel-resources table %s" % ("unchanged" if doc2 == doc else "updated"))

#!/usr/bin/env python3
"""Ratio of the encoder model (bit-identical to the GPU frames) against libzstd 1.5.x per real-data item and level: the table of
DESIGN.md section 4.1.  usage: ratio_table.py ["[dict(far_step_log=4), ...]"] ["[3, 9]"]  (parameter overrides are for experiments)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import harness, realdata
z = next(z for z in harness.libzstds() if z.version.startswith("1.5"))
items = {k: v for k, v in realdata.items().items() if v}
c = harness.Corpus()
for k in range(3):
    items["corpus_k%d_1m" % k] = c.entry(k, 1 << 20, k)
if os.environ.get("ITEMS"):   # ITEMS=elf_mid_4m,pyc_2m: only these
    items = {k: v for k, v in items.items() if k in os.environ["ITEMS"].split(",")}
variants = eval(sys.argv[1]) if len(sys.argv) > 1 else [dict()]
levels = eval(sys.argv[2]) if len(sys.argv) > 2 else [3, 9]
o = harness.Oracle()
ref = {(n, l): len(z.compress(d, l, 1)) for n, d in items.items() for l in levels}
for var in variants:
    print("overrides", var)
    for l in levels:
        print("  L%d: " % l + " | ".join("%s %.3f" % (n, len(o.zge_encode(d, o.params(level=l, **var))) / ref[(n, l)]) for n, d in items.items()))

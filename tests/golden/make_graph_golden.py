"""Writes tests/golden/graph_<map>.npz: the bit-exact integer state (spanning-tree tables, edge list, capsule integer arrays) of the maps
listed in tests/_graphdump.py. Generated ONCE from the round-1 front-end (commit 9d3c575, the std::map / deque host layer that had been checked
against the reference's SpanTreeTests and mini-problems), before the flat-container rewrite of include/srba/; the rewrite must reproduce them.
Run from the repo root: python tests/golden/make_graph_golden.py [map ...]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import _graphdump  # noqa: E402

for name in (sys.argv[1:] or _graphdump.MAPS):
    d = _graphdump.run_map(name)
    path = os.path.join(HERE, "graph_%s.npz" % name)
    np.savez_compressed(path, **d)
    print("%-28s %5d edges %6d capsules, ST rows %d/%d, %d bytes" % (name, len(d["edges"]), len(d["capsule_digest"]), d["n_next_edge"], d["n_all_edges"], os.path.getsize(path)))

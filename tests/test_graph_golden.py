"""Bit-exactness of the host graph layer: spanning-tree tables, kf2kf edge list and every integer array of
every problem capsule of the maps in tests/_graphdump.py must equal the golden dumps tests/golden/graph_<map>.npz, which were written by the
round-1 std::map / deque front-end (itself checked against the reference's SpanTreeTests grid and mini-problems) before the host layer was
re-implemented on flat containers. `north_star`: "bit-exact on spanning-tree indices"."""
import os

import numpy as np
import pytest

import _graphdump

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", _graphdump.MAPS)
def test_graph_layer_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, "graph_%s.npz" % name))
    d = _graphdump.run_map(name)
    assert np.array_equal(d["edges"], g["edges"]), "kf2kf edge list differs"
    assert int(d["n_next_edge"]) == int(g["n_next_edge"]) and int(d["n_all_edges"]) == int(g["n_all_edges"])
    assert np.array_equal(d["next_edge_head"], g["next_edge_head"]), "next_edge tables differ (first key-frames, full rows)"
    assert np.array_equal(d["all_edges_head"], g["all_edges_head"]), "all_edges paths differ (first key-frames, full rows)"
    bad = np.flatnonzero(d["next_edge_digest"] != g["next_edge_digest"])
    assert bad.size == 0, "next_edge rows differ for source key-frames %s..." % bad[:8]
    bad = np.flatnonzero(d["all_edges_digest"] != g["all_edges_digest"])
    assert bad.size == 0, "all_edges rows differ for source key-frames %s..." % bad[:8]
    assert np.array_equal(d["capsule_kf"], g["capsule_kf"]), "capsules are produced at different key-frames"
    bad = np.flatnonzero((d["capsule_sizes"] != g["capsule_sizes"]).any(axis=1))
    assert bad.size == 0, "capsule sizes differ: first at capsule %d (kf %d): %s vs golden %s" % (bad[0], d["capsule_kf"][bad[0]], d["capsule_sizes"][bad[0]], g["capsule_sizes"][bad[0]])
    bad = np.flatnonzero(d["capsule_digest"] != g["capsule_digest"])
    assert bad.size == 0, "integer arrays of %d capsules differ, first: capsule %d (kf %d)" % (bad.size, bad[0], d["capsule_kf"][bad[0]])

"""Comparison helpers shared by the parity tests (they live next to the oracle so that
``__graft_entry__.smoke()`` can use them too)."""
from oracle.compare import *  # noqa: F401,F403
from oracle.compare import TIE_RTOL, canonical, compare_edge_sets, compare_knn, knn_rows, rdist64  # noqa: F401

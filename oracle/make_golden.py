"""TEST INFRASTRUCTURE ONLY -- writes ``tests/golden/*.npz`` from the UNMODIFIED reference.

Run in the build container only (needs ``/root/reference``):

    python -m oracle.make_golden

Each fixture holds inputs and the outputs of the reference's own classes
(``anemoi.graphs.edges.KNNEdges`` etc., loaded by ``oracle/ref_harness.py``) on those
inputs.  Fixtures are data only.  Cases that go through ``TriNodes``/``MultiScaleEdges``
use this repository's icosphere restatement in place of trimesh (absent here) and are
marked ``parity_unpinned_geometry=1``.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from anemoi_graphs_amd import grids  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = ROOT / "tests" / "golden"
NORMS = [None, "l1", "l2", "unit-max", "unit-range", "unit-std"]


def _graph(ref, **nodes):
    g = ref.HeteroData()
    for name, x in nodes.items():
        g[name].x = torch.as_tensor(x)
    return g


def _attrs(ref, g, name, out, prefix):
    for norm in NORMS:
        tag = "none" if norm is None else norm.replace("-", "_")
        out[f"{prefix}len_{tag}"] = ref.EdgeLength(norm=norm).compute(g, name).numpy()
        out[f"{prefix}dir_{tag}"] = ref.EdgeDirection(norm=norm).compute(g, name).numpy()
    out[f"{prefix}len_inv_unit_max"] = ref.EdgeLength(norm="unit-max", invert=True).compute(g, name).numpy()
    out[f"{prefix}dir_plain_none"] = ref.EdgeDirection(norm=None, luse_rotated_features=False).compute(g, name).numpy()


def case_conftest(ref, dtype=np.float64):
    """The 3x4 grid of /root/reference/tests/conftest.py:16-17,56-63 and the recipe of :99-112.
    The fixture's coordinates are float64; node builders always produce float32, so a float32 copy
    of the case is recorded too (the self edges of this self-query have a direction that is pure
    rounding noise of the coordinate dtype)."""
    lats, lons = [-0.15, 0, 0.15], [0, 0.25, 0.5, 0.75]
    coords = np.array([[lat, lon] for lat in lats for lon in lons])
    x = (2 * torch.pi * torch.tensor(coords)).numpy().astype(dtype)
    g = _graph(ref, test_nodes=x)
    g = ref.KNNEdges("test_nodes", "test_nodes", 3).update_graph(g)
    name = ("test_nodes", "to", "test_nodes")
    out = {"x": x, "knn3_edge_index": g[name].edge_index.numpy()}
    _attrs(ref, g, name, out, "knn3_")
    g2 = _graph(ref, test_nodes=x)
    c = ref.CutOffEdges("test_nodes", "test_nodes", 0.5)
    g2 = c.update_graph(g2)
    out["cutoff05_edge_index"] = g2[name].edge_index.numpy()
    out["cutoff05_radius"] = np.float64(c.radius)
    return out


def case_random(ref):
    src = grids.random_sphere(500, seed=1)
    tgt = grids.random_sphere(2000, seed=2)
    out = {"src": src, "tgt": tgt}
    for k in (1, 3, 8):
        g = _graph(ref, a=src, b=tgt)
        g = ref.KNNEdges("a", "b", k).update_graph(g)
        out[f"knn{k}_edge_index"] = g[("a", "to", "b")].edge_index.numpy()
        if k == 3:
            _attrs(ref, g, ("a", "to", "b"), out, "knn3_")
    for f in (0.6, 1.3):
        g = _graph(ref, a=src, b=tgt)
        c = ref.CutOffEdges("b", "a", f)
        g = c.update_graph(g)
        tag = str(f).replace(".", "")
        out[f"cutoff{tag}_edge_index"] = g[("b", "to", "a")].edge_index.numpy()
        out[f"cutoff{tag}_radius"] = np.float64(c.radius)
        if f == 0.6:
            _attrs(ref, g, ("b", "to", "a"), out, "cutoff06_")
    out["refdist_src"] = np.float64(ref.get_grid_reference_distance(torch.as_tensor(src)))
    out["refdist_tgt"] = np.float64(ref.get_grid_reference_distance(torch.as_tensor(tgt)))
    return out


def case_masks(ref):
    """source/target masks: edges/builder.py:162-193."""
    src = grids.random_sphere(300, seed=3)
    tgt = grids.random_sphere(700, seed=4)
    rng = np.random.default_rng(5)
    smask = rng.uniform(size=(300, 1)) < 0.6
    tmask = rng.uniform(size=(700, 1)) < 0.5
    out = {"src": src, "tgt": tgt, "src_mask": smask, "tgt_mask": tmask}
    g = _graph(ref, a=src, b=tgt)
    g["a"]["m"] = torch.as_tensor(smask)
    g["b"]["m"] = torch.as_tensor(tmask)
    g = ref.KNNEdges("a", "b", 4, source_mask_attr_name="m", target_mask_attr_name="m").update_graph(g)
    out["knn4_edge_index"] = g[("a", "to", "b")].edge_index.numpy()
    c = ref.CutOffEdges("b", "a", 0.8, source_mask_attr_name="m", target_mask_attr_name="m")
    g = c.update_graph(g)
    out["cutoff08_edge_index"] = g[("b", "to", "a")].edge_index.numpy()
    out["cutoff08_radius"] = np.float64(c.radius)
    return out


def case_regular(ref):
    """Regular 2-degree grid (poles included, many exact ties) against a 642-point spiral set."""
    data = grids.regular(2.0)
    hidden = grids.fibonacci_sphere(642)
    out = {"data": data, "hidden": hidden}
    g = _graph(ref, data=data, hidden=hidden)
    g = ref.KNNEdges("hidden", "data", 3).update_graph(g)
    out["dec_knn3_edge_index"] = g[("hidden", "to", "data")].edge_index.numpy()
    c = ref.CutOffEdges("data", "hidden", 0.6)
    g = c.update_graph(g)
    out["enc_cutoff06_edge_index"] = g[("data", "to", "hidden")].edge_index.numpy()
    out["enc_cutoff06_radius"] = np.float64(c.radius)
    out["refdist_data"] = np.float64(ref.get_grid_reference_distance(torch.as_tensor(data)))
    out["enc_len_none"] = ref.EdgeLength().compute(g, ("data", "to", "hidden")).numpy()
    out["enc_dir_none"] = ref.EdgeDirection().compute(g, ("data", "to", "hidden")).numpy()
    return out


def case_o96_tri4(ref):
    """The small plumbing configuration: O96 (40,320) -> TriNodes r=4, KNNEdges k=3 (decoder direction)
    plus the CutOff encoder, through the reference's own classes."""
    data = grids.octahedral(96)
    g = _graph(ref, data=data)
    g["data"].node_type = "LatLonNodes"
    g = ref.TriNodes(4, "hidden").update_graph(g, {})
    out = {"data": data, "hidden": g["hidden"].x.numpy(), "parity_unpinned_geometry": np.int32(1)}
    g = ref.KNNEdges("hidden", "data", 3).update_graph(g)
    out["dec_knn3_edge_index"] = g[("hidden", "to", "data")].edge_index.numpy()
    c = ref.CutOffEdges("data", "hidden", 0.6)
    g = c.update_graph(g)
    out["enc_cutoff06_edge_index"] = g[("data", "to", "hidden")].edge_index.numpy()
    out["enc_cutoff06_radius"] = np.float64(c.radius)
    g = ref.MultiScaleEdges("hidden", "hidden", 1).update_graph(g)
    out["proc_edge_index"] = g[("hidden", "to", "hidden")].edge_index.numpy()
    return out


def case_tri(ref):
    """TriNodes / MultiScaleEdges logic of the reference on this repository's icosphere."""
    out = {"parity_unpinned_geometry": np.int32(1)}
    for r, hops in ((0, 1), (1, 1), (2, 1), (2, 2), (3, 1), (3, 3)):
        g = ref.HeteroData()
        g = ref.TriNodes(r, "h").update_graph(g, {})
        g = ref.MultiScaleEdges("h", "h", hops).update_graph(g)
        out[f"r{r}_x"] = g["h"].x.numpy()
        out[f"r{r}_node_ordering"] = np.asarray(g["h"]["_node_ordering"], dtype=np.int64)
        out[f"r{r}_h{hops}_edge_index"] = g[("h", "to", "h")].edge_index.numpy()
    g = ref.HeteroData()
    g = ref.TriNodes([0, 2, 3], "h").update_graph(g, {})
    g = ref.MultiScaleEdges("h", "h", 1).update_graph(g)
    out["r023_x"] = g["h"].x.numpy()
    out["r023_h1_edge_index"] = g[("h", "to", "h")].edge_index.numpy()
    return out


def _area_nodes():
    """A regional 0.5-degree patch (lat 35..60, lon -10..30 degrees) as the reference nodes of the area masks."""
    lat = np.arange(60.0, 34.9, -0.5)
    lon = np.arange(-10.0, 30.1, 0.5)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    return np.deg2rad(np.stack([la.ravel(), lo.ravel()], axis=1)).astype(np.float32)


def case_lam(ref):
    """LimitedAreaTriNodes / StretchedTriNodes with MultiScaleEdges through the reference's classes
    (on this repository's icosphere)."""
    area = _area_nodes()
    out = {"area": area, "parity_unpinned_geometry": np.int32(1)}
    for r, hops, margin in ((4, 1, 100.0), (5, 1, 400.0), (4, 2, 300.0)):
        g = _graph(ref, area=area)
        g = ref.LimitedAreaTriNodes(r, "area", "h", margin_radius_km=margin).update_graph(g, {})
        g = ref.MultiScaleEdges("h", "h", hops).update_graph(g)
        tag = f"lam_r{r}_h{hops}"
        out[f"{tag}_margin"] = np.float64(margin)
        out[f"{tag}_x"] = g["h"].x.numpy()
        out[f"{tag}_edge_index"] = g[("h", "to", "h")].edge_index.numpy()
    for rb, rl, hops, margin in ((2, 4, 1, 100.0), (1, 4, 1, 500.0), (2, 4, 2, 200.0)):
        g = _graph(ref, area=area)
        g = ref.StretchedTriNodes(rb, rl, "h", "area", None, margin_radius_km=margin).update_graph(g, {})
        g = ref.MultiScaleEdges("h", "h", hops).update_graph(g)
        tag = f"stretched_r{rb}_{rl}_h{hops}"
        out[f"{tag}_margin"] = np.float64(margin)
        out[f"{tag}_x"] = g["h"].x.numpy()
        out[f"{tag}_edge_index"] = g[("h", "to", "h")].edge_index.numpy()
    return out


def case_area_weights(ref):
    """SphericalAreaWeights / UniformWeights of the reference on a reduced Gaussian grid, icosphere nodes,
    random points and an anisotropic regular grid (no pole rows: scipy refuses duplicate generators)."""
    from oracle import reference_path as rp

    lat = np.deg2rad(np.arange(87.0, -87.1, -3.0))
    lon = np.deg2rad(np.arange(0.0, 360.0, 1.0))
    aniso = np.stack(np.meshgrid(lat, lon, indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float32)
    sets = {
        "o32": grids.named("o32"),
        "tri3": rp.tri_nodes(3)[0].astype(np.float32),
        "random": grids.random_sphere(3000, seed=5),
        "aniso": aniso,
    }
    out = {}
    for name, x in sets.items():
        g = _graph(ref, nodes=x)
        out[f"{name}_x"] = x
        for norm in NORMS:
            tag = "none" if norm is None else norm.replace("-", "_")
            out[f"{name}_area_{tag}"] = ref.SphericalAreaWeights(norm=norm).compute(g, "nodes").numpy()
        # the float64 values before post-processing, for the precision of the cells themselves
        out[f"{name}_area_raw"] = ref.SphericalAreaWeights().get_raw_values(g["nodes"])
    g = _graph(ref, nodes=sets["o32"])
    out["o32_uniform_l1"] = ref.UniformWeights(norm="l1").compute(g, "nodes").numpy()
    out["o32_uniform_none"] = ref.UniformWeights().compute(g, "nodes").numpy()
    return out


def case_concat(ref):
    """/root/reference/tests/test_utils.py:7-18 plus a random case."""
    e1 = torch.tensor([[0, 1, 2, 3], [-1, -2, -3, -4]])
    e2 = torch.tensor([[0, 4], [-1, -5]])
    rng = np.random.default_rng(7)
    r1 = rng.integers(0, 50, size=(2, 400)).astype(np.int32)
    r2 = rng.integers(0, 50, size=(2, 300)).astype(np.int32)
    return {
        "e1": e1.numpy(), "e2": e2.numpy(), "e12": ref.concat_edges(e1, e2).numpy(),
        "r1": r1, "r2": r2, "r12": ref.concat_edges(torch.as_tensor(r1), torch.as_tensor(r2)).numpy(),
    }


def main() -> None:
    ref = ref_harness.reference()
    OUT.mkdir(parents=True, exist_ok=True)
    cases = {
        "conftest_grid": case_conftest,
        "conftest_grid_f32": lambda ref: case_conftest(ref, np.float32),
        "random_sphere": case_random,
        "masks": case_masks,
        "regular_2deg": case_regular,
        "o96_tri4": case_o96_tri4,
        "tri_multiscale": case_tri,
        "concat_edges": case_concat,
        "lam_stretched": case_lam,
        "area_weights": case_area_weights,
    }
    only = set(sys.argv[1:])
    for name, fn in cases.items():
        if only and name not in only:
            continue
        data = fn(ref)
        np.savez_compressed(OUT / f"{name}.npz", **data)
        print(name, {k: getattr(v, "shape", v) for k, v in data.items()})


if __name__ == "__main__":
    main()
